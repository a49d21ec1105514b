// ORACLE/_ref - TEST INFRASTRUCTURE ONLY.
// C entry points over two host functions of the reference's src/core/splat_data.cpp, compiled in place (oracle/Makefile, `make refsplatio`): the anonymous
// namespace's compute_mean_neighbor_distances (:64-111, nanoflann kd-tree: mean distance to the 3 nearest neighbours = the initial scale of every Gaussian) and
// write_ply_impl (:113-169, tinyply: the splat PLY the reference exports). Only that line range (28-169) is compiled - the rest of the file needs <expected>,
// <print>, glm and the SOG writer; the vendored include/external/nanoflann.hpp, tinyply.hpp and include/core/point_cloud.hpp are used as they are.
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

#include "k_splat_data_io.inc" // splat_data.cpp:28-169 + the closing brace of its anonymous namespace

#define REF_API extern "C" __attribute__((visibility("default")))

static torch::Tensor f32(const float* p, std::vector<int64_t> shape) { return torch::from_blob(const_cast<float*>(p), shape, torch::kFloat32).clone(); }

REF_API void refsplat_mean_neighbor_distances(int64_t N, const float* points, float* out) {
    auto r = compute_mean_neighbor_distances(f32(points, {N, 3})).contiguous();
    std::memcpy(out, r.data_ptr<float>(), sizeof(float) * N);
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
