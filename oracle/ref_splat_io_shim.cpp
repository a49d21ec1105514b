// ORACLE/_ref - TEST INFRASTRUCTURE ONLY.
// C entry points over two host functions of the reference's src/core/splat_data.cpp, compiled in place (oracle/Makefile, `make refsplatio`): the anonymous
// namespace's compute_mean_neighbor_distances (:64-111, nanoflann kd-tree: mean distance to the 3 nearest neighbours = the initial scale of every Gaussian) and
// write_ply_impl (:113-169, tinyply: the splat PLY the reference exports), and SplatData::init_model_from_pointcloud (:508-614, with the class's constructors and
// getters :200-287, 386-434). Only those line ranges are compiled - the rest of the file needs glm and the SOG writer; <expected> / <print> are stood in for (ref_stub/); the vendored include/external/nanoflann.hpp, tinyply.hpp and include/core/point_cloud.hpp are used as they are.
// What SplatData::to_point_cloud (:484-505) and get_attribute_names (:402-419) do around write_ply_impl is restated in refsplat_write_ply below, cited per line.
// Used by tests/test_loader_reference.py and tests/golden/ref_splat_io.npz - SURVEY.md §8f row 4. Nothing here is product code.
#include "core/logger.hpp" // ref_stub
#include "core/point_cloud.hpp"
#include "external/nanoflann.hpp"
#define TINYPLY_IMPLEMENTATION
#include "external/tinyply.hpp"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <sstream>
#include <string>
#include <torch/torch.h>
#include <vector>

#include "core/parameters.hpp"
#include "core/splat_data.hpp"
#include <iostream>

#define unexpected ref_unexpected // libstdc++ 11 has std::unexpected() the C++98 function; ref_stub/expected calls the C++23 class template std::ref_unexpected
#include "k_splat_data_io.inc" // splat_data.cpp:28-169 (anonymous namespace, closed), then namespace gs: 200-287, 386-434 (SplatData's constructors, getters, small
                               // methods), 508-614 (init_model_from_pointcloud, with torch::kCUDA -> torch::kCPU), closed
#undef unexpected

#define REF_API extern "C" __attribute__((visibility("default")))

static torch::Tensor f32(const float* p, std::vector<int64_t> shape) { return torch::from_blob(const_cast<float*>(p), shape, torch::kFloat32).clone(); }

REF_API void refsplat_mean_neighbor_distances(int64_t N, const float* points, float* out) {
    auto r = compute_mean_neighbor_distances(f32(points, {N, 3})).contiguous();
    std::memcpy(out, r.data_ptr<float>(), sizeof(float) * N);
}

// SplatData::init_model_from_pointcloud (the non-random branch): positions [N,3] f32, colors [N,3] u8, scene_center [3] -> the six parameter tensors
// (sh0 [N,1,3], shN [N,K,3] with K = (sh_degree+1)^2 - 1, scaling [N,3], rotation [N,4], opacity [N,1]) and the scene scale
REF_API int refsplat_init_model(int64_t N, const float* positions, const uint8_t* colors, const float* scene_center, int sh_degree, float init_scaling, float init_opacity,
                                float* means, float* sh0, float* shN, float* scaling, float* rotation, float* opacity, float* scene_scale) {
    try {
        gs::param::TrainingParameters params;
        params.optimization.random = false;
        params.optimization.sh_degree = sh_degree;
        params.optimization.init_scaling = init_scaling;
        params.optimization.init_opacity = init_opacity;
        gs::PointCloud pcd(f32(positions, {N, 3}), torch::from_blob(const_cast<uint8_t*>(colors), {N, 3}, torch::kUInt8).clone());
        auto r = gs::SplatData::init_model_from_pointcloud(params, f32(scene_center, {3}), pcd);
        if (!r) {
            std::fprintf(stderr, "refsplat_init_model: %s\n", r.error().c_str());
            return 1;
        }
        auto put = [](const torch::Tensor& t, float* dst) {
            auto c = t.detach().contiguous();
            std::memcpy(dst, c.data_ptr<float>(), sizeof(float) * c.numel());
        };
        put(r->means(), means), put(r->sh0(), sh0), put(r->shN(), shN), put(r->scaling_raw(), scaling), put(r->rotation_raw(), rotation), put(r->opacity_raw(), opacity);
        *scene_scale = r->get_scene_scale();
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "refsplat_init_model: %s\n", e.what());
        return 1;
    }
}

// sh0 [N,K0,3], shN [N,KN,3], opacity [N,1], scaling [N,3], rotation [N,4] (raw) -> <root>/<stem>.ply
REF_API int refsplat_write_ply(const char* root, const char* stem, int64_t N, int64_t K0, int64_t KN, const float* means, const float* sh0, const float* shN,
                               const float* opacity, const float* scaling, const float* rotation) {
    try {
        gs::PointCloud pc;
        pc.means = f32(means, {N, 3});                                                                   // to_point_cloud :488
        pc.normals = torch::zeros_like(pc.means);                                                        // :489
        pc.sh0 = f32(sh0, {N, K0, 3}).transpose(1, 2).flatten(1);                                        // :492
        pc.shN = f32(shN, {N, KN, 3}).transpose(1, 2).flatten(1);                                        // :493
        pc.opacity = f32(opacity, {N, 1});                                                               // :494
        pc.scaling = f32(scaling, {N, 3});                                                               // :495
        pc.rotation = torch::nn::functional::normalize(f32(rotation, {N, 4}), torch::nn::functional::NormalizeFuncOptions().dim(-1)).contiguous(); // :497-500
        std::vector<std::string> a{"x", "y", "z", "nx", "ny", "nz"};                                     // get_attribute_names :403
        for (int64_t i = 0; i < 3 * K0; ++i) a.emplace_back("f_dc_" + std::to_string(i));                // :405-406
        for (int64_t i = 0; i < 3 * KN; ++i) a.emplace_back("f_rest_" + std::to_string(i));              // :407-408
        a.emplace_back("opacity");                                                                       // :410
        for (int i = 0; i < 3; ++i) a.emplace_back("scale_" + std::to_string(i));                        // :412-413
        for (int i = 0; i < 4; ++i) a.emplace_back("rot_" + std::to_string(i));                          // :414-415
        pc.attribute_names = a;
        write_ply_impl(pc, root, 0, stem);
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "refsplat_write_ply: %s\n", e.what());
        return 1;
    }
}
