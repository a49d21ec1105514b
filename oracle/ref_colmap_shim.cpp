// ORACLE/_ref - TEST INFRASTRUCTURE ONLY.
// C entry points over the reference's OWN COLMAP reader: /root/reference/src/loader/formats/colmap.cpp, compiled in place against libtorch by oracle/Makefile
// (`make refcolmap`) together with this file. Its two absent dependencies are stood in for under oracle/ref_stub/: the logger (macros that expand to nothing, a
// "{}" std::format) and image_io (only get_image_info, answered below from the file header). Used by tests/test_loader_reference.py to hold liblfs_io.so and
// oracle/colmap_io.py to what the reference's reader returns for the same files - SURVEY.md §8f row 4. Nothing here is product code.
#include "loader/formats/colmap.hpp" // resolved from $(REF)/src by the Makefile include path
#include "core/image_io.hpp"
#include <cstring>
#include <fstream>

#define REF_API extern "C" __attribute__((visibility("default")))

// PNG: IHDR at byte 16 (big-endian width, height; colour type at 25). PNM: "P5"/"P6" <w> <h> <max>.
std::tuple<int, int, int> get_image_info(std::filesystem::path p) {
    std::ifstream f(p, std::ios::binary);
    unsigned char h[32] = {0};
    f.read(reinterpret_cast<char*>(h), 32);
    if (f.gcount() >= 26 && h[0] == 0x89 && h[1] == 'P' && h[2] == 'N' && h[3] == 'G') {
        auto be = [&](int o) { return (int)((h[o] << 24) | (h[o + 1] << 16) | (h[o + 2] << 8) | h[o + 3]); };
        const int ct = h[25];
        return {be(16), be(20), ct == 0 ? 1 : ct == 4 ? 2 : ct == 6 ? 4 : 3};
    }
    if (h[0] == 'P' && (h[1] == '5' || h[1] == '6')) {
        int w = 0, hh = 0;
        std::sscanf(reinterpret_cast<const char*>(h) + 2, "%d %d", &w, &hh);
        return {w, hh, h[1] == '6' ? 3 : 1};
    }
    throw std::runtime_error("ref_colmap_shim: unsupported image header: " + p.string());
}

struct RefView { // the layout lichtfeld_studio_amd.loader._View declares
    uint32_t camera_id;
    int32_t colmap_model, camera_model_type;
    uint64_t width, height;
    float focal_x, focal_y, center_x, center_y, R[9], T[3];
    int32_t n_radial;
    float radial[6];
    int32_t n_tangential;
    float tangential[2];
    int32_t n_params;
    float params[12];
};

struct RefScene {
    std::vector<gs::loader::CameraData> cams;
    torch::Tensor center;
};

static thread_local std::string g_error;
REF_API const char* refcolmap_last_error() { return g_error.c_str(); }

template <class F> static int guarded(F&& f) {
    try {
        f();
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return 1;
    }
}

static int copy(const torch::Tensor& t, float* dst, int cap) {
    if (!t.defined()) return 0;
    auto c = t.to(torch::kFloat32).contiguous().reshape({-1});
    const int n = (int)c.numel();
    if (n > cap) throw std::runtime_error("ref_colmap_shim: tensor larger than the view slot");
    std::memcpy(dst, c.data_ptr<float>(), sizeof(float) * n);
    return n;
}

REF_API int refcolmap_open(const char* base, const char* images_folder, int text, RefScene** out) {
    return guarded([&] {
        auto s = std::make_unique<RefScene>();
        auto r = text ? gs::loader::read_colmap_cameras_and_images_text(base, images_folder) : gs::loader::read_colmap_cameras_and_images(base, images_folder);
        s->cams = std::move(std::get<0>(r));
        s->center = std::get<1>(r);
        *out = s.release();
    });
}
REF_API void refcolmap_close(RefScene* s) { delete s; }
REF_API uint64_t refcolmap_num_views(const RefScene* s) { return s->cams.size(); }
REF_API const char* refcolmap_image_name(const RefScene* s, uint64_t i) { return s->cams[i]._image_name.c_str(); }
REF_API const char* refcolmap_image_path(const RefScene* s, uint64_t i) { return s->cams[i]._image_path.c_str(); }
REF_API int refcolmap_scene_center(const RefScene* s, float c[3]) {
    return guarded([&] { copy(s->center, c, 3); });
}
REF_API int refcolmap_view_at(const RefScene* s, uint64_t i, RefView* v) {
    return guarded([&] {
        const auto& c = s->cams.at(i);
        std::memset(v, 0, sizeof(*v));
        v->camera_id = c._camera_ID;
        v->colmap_model = (int32_t)c._camera_model;
        v->camera_model_type = (int32_t)c._camera_model_type;
        v->width = c._width;
        v->height = c._height;
        v->focal_x = c._focal_x, v->focal_y = c._focal_y, v->center_x = c._center_x, v->center_y = c._center_y;
        copy(c._R, v->R, 9);
        copy(c._T, v->T, 3);
        v->n_radial = copy(c._radial_distortion, v->radial, 6);
        v->n_tangential = copy(c._tangential_distortion, v->tangential, 2);
        v->n_params = copy(c._params, v->params, 12);
    });
}

// points3D: returns N (< 0 on error); call with null pointers to size
REF_API int64_t refcolmap_points(const char* base, int text, float* positions, uint8_t* colors, int64_t cap) {
    int64_t n = -1;
    guarded([&] {
        auto pc = text ? gs::loader::read_colmap_point_cloud_text(base) : gs::loader::read_colmap_point_cloud(base);
        auto m = pc.means.to(torch::kFloat32).contiguous();
        auto col = pc.colors.to(torch::kUInt8).contiguous();
        if (positions && m.size(0) <= cap) {
            std::memcpy(positions, m.data_ptr<float>(), sizeof(float) * m.numel());
            std::memcpy(colors, col.data_ptr<uint8_t>(), col.numel());
        }
        n = m.size(0);
    });
    return n;
}
