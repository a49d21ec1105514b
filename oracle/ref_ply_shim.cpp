// ORACLE/_ref - TEST INFRASTRUCTURE ONLY.
// C entry point over the reference's OWN PLY reader: src/loader/formats/ply.cpp (whole file, compiled in place against CPU libtorch by `make -C oracle refply`; sed:
// torch::kCUDA -> torch::kCPU) with SplatData's constructors / getters (splat_data.cpp:200-287, 386-434). Stand-ins under ref_stub/: the logger, <expected>,
// <format>, tbb/parallel_for.h (serial), glm, geometry/bounding_box.hpp. Used by tests/test_loader_reference.py to hold the product's loader.load_ply to what the
// reference's load_ply returns for the same files. Nothing here is product code.
#include "formats/ply.hpp"
#include <cstring>

#define REF_API extern "C" __attribute__((visibility("default")))

static std::string g_error;
static std::unique_ptr<gs::SplatData> g_model;
REF_API const char* refply_last_error() { return g_error.c_str(); }

// load_ply(path) -> N (< 0: error, text in refply_last_error); the model stays loaded for refply_get
REF_API int64_t refply_load(const char* path) {
    try {
        auto r = gs::loader::load_ply(path);
        if (!r) {
            g_error = r.error();
            return -1;
        }
        g_model = std::make_unique<gs::SplatData>(std::move(*r));
        return g_model->size();
    } catch (const std::exception& e) {
        g_error = e.what();
        return -1;
    }
}
// which: 0 means, 1 sh0, 2 shN, 3 scaling, 4 rotation, 5 opacity (raw tensors). Returns the element count and the shape (up to 3 dims, 0-padded); out may be null
REF_API int64_t refply_get(int which, int64_t* shape, float* out) {
    torch::Tensor t = which == 0 ? g_model->means() : which == 1 ? g_model->sh0() : which == 2 ? g_model->shN() : which == 3 ? g_model->scaling_raw()
                      : which == 4 ? g_model->rotation_raw() : g_model->opacity_raw();
    t = t.detach().to(torch::kCPU).to(torch::kFloat32).contiguous();
    for (int i = 0; i < 3; ++i) shape[i] = i < t.dim() ? t.size(i) : 0;
    if (out) std::memcpy(out, t.data_ptr<float>(), sizeof(float) * t.numel());
    return t.numel();
}
REF_API int refply_sh_degree() { return g_model->get_active_sh_degree(); }
REF_API float refply_scene_scale() { return g_model->get_scene_scale(); }
