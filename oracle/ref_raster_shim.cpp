// ORACLE/_ref - TEST INFRASTRUCTURE ONLY.
// The reference's OWN training-time render path, end to end, on the CPU (oracle/Makefile, `make refraster`):
//   src/training/rasterization/rasterizer.cpp        gs::training::rasterize(): camera -> projection -> SH colours -> tile intersection -> rasterization, the
//                                                    colour offset / clamps / masks / background handling around them   (whole file)
//   src/training/rasterization/rasterizer_autograd.cpp   the autograd Functions behind it (SH, projection wrapper, GUT rasterization)   (whole file)
//   src/core/camera.cpp                              Camera: world_view_transform, K() with the image / camera size scaling   (whole file)
//   src/core/splat_data.cpp:200-287, 386-434         SplatData's constructors, activations (get_opacity ...), SH degree
// compiled in place against CPU libtorch with their real headers (rasterizer.hpp, rasterizer_autograd.hpp, core/camera.hpp, core/splat_data.hpp, gsplat/Ops.h,
// Cameras.h, Common.h). sed edits on the way into the scratch directory (Makefile): torch::kCUDA -> torch::kCPU, and the single-line `TORCH_CHECK(x.is_cuda(), ...)`
// device assertions dropped. The gsplat operators the path calls are defined below as their host launch sequences over the reference's own code run on the CPU:
//   projection_ut_3dgs_fused, rasterize_to_pixels_from_world_3dgs_fwd / _bwd : the device kernels (oracle/_ref/libref_kernels.so, ref_emul/)
//   spherical_harmonics_fwd / _bwd, intersect_tile                           : the reference's own CPU implementation tests/torch_impl.cpp (libtorch_impl_ref.so),
//                                                                              which SURVEY.md §8c records as matching kernels K2 / K3-K5; the vjp by autograd over it
//   intersect_offset                                                         : restated (IntersectTile.cu:206-252: first index of every tile in the sorted keys)
// Used by oracle/make_golden_ref_raster.py -> tests/golden/ref_raster.npz: the whole hot path's forward image and parameter gradients as the reference composes
// them, which tests/test_gpu_raster_reference.py holds the product's render + backward to. Nothing here is product code.
#include "Ops.h"
#include "core/camera.hpp"
#include "core/image_io.hpp"
#include "core/splat_data.hpp"
#include "rasterizer.hpp"
#ifndef REF_REAL_GSPLAT
#include "torch_impl.hpp"
#endif
#ifdef REF_REAL_GSPLAT
#include "kernels/fused_ssim.cuh" // the reference's fused-SSIM autograd wrapper (over its own ssim.cu, ref_loss_all.cpp): refraster_train_loss_backward
#include "bilateral_grid.hpp"     // gs::training::BilateralGrid (components/bilateral_grid.cpp over the bilateral_grid_*.cu files): config 5's appearance model
#endif
#include <cstring>

#define REF_API extern "C" __attribute__((visibility("default")))

#ifndef REF_REAL_GSPLAT // (make refgsplat: the gsplat:: operators below are not restated but taken from the reference's own gsplat/*.cpp + *.cu - ref_gsplat_all.cpp)
extern "C" {
void refk_projection_ut(uint32_t C, uint32_t N, const float* means, const float* quats, const float* scales, const float* opacities, const float* viewmats0,
                        const float* viewmats1, const float* Ks, uint32_t W, uint32_t H, float eps2d, float near_plane, float far_plane, float radius_clip,
                        int camera_model, const float* ut, int rs_type, const float* radial, const float* tangential, const float* thin_prism, int32_t* radii,
                        float* means2d, float* depths, float* conics, float* compensations);
int refk_rasterize_fwd(uint32_t cdim, uint32_t C, uint32_t N, uint32_t n_isects, const float* means, const float* quats, const float* scales, const float* colors,
                       const float* opacities, const float* backgrounds, const uint8_t* masks, uint32_t W, uint32_t H, uint32_t tile_size, uint32_t tw, uint32_t th,
                       const float* vm0, const float* vm1, const float* Ks, int camera_model, const float* ut, int rs_type, const float* radial,
                       const float* tangential, const float* thin_prism, const int32_t* offsets, const int32_t* flatten_ids, float* render_colors,
                       float* render_alphas, int32_t* last_ids);
int refk_rasterize_bwd(uint32_t cdim, uint32_t C, uint32_t N, uint32_t n_isects, const float* means, const float* quats, const float* scales, const float* colors,
                       const float* opacities, const float* backgrounds, const uint8_t* masks, uint32_t W, uint32_t H, uint32_t tile_size, uint32_t tw, uint32_t th,
                       const float* vm0, const float* vm1, const float* Ks, int camera_model, const float* ut, int rs_type, const float* radial,
                       const float* tangential, const float* thin_prism, const int32_t* offsets, const int32_t* flatten_ids, const float* render_alphas,
                       const int32_t* last_ids, const float* v_render_colors, const float* v_render_alphas, float* v_means, float* v_quats, float* v_scales,
                       float* v_colors, float* v_opacities);
}

std::tuple<unsigned char*, int, int, int> load_image(std::filesystem::path, int, int) { throw std::runtime_error("ref_raster_shim: no image loading"); }
void free_image(unsigned char*) {}
std::tuple<int, int, int> get_image_info(std::filesystem::path) { throw std::runtime_error("ref_raster_shim: no image loading"); }

namespace {
    struct UT {
        float v[5];
        explicit UT(const UnscentedTransformParameters& p) : v{p.alpha, p.beta, p.kappa, p.in_image_margin_factor, p.require_all_sigma_points_valid ? 1.f : 0.f} {}
    };
    const float* opt(const at::optional<at::Tensor>& t) { return (t.has_value() && t->defined() && t->numel() > 0) ? t->data_ptr<float>() : nullptr; }
    void need_plain_camera(const at::optional<at::Tensor>& vm1, const at::optional<at::Tensor>& r, const at::optional<at::Tensor>& t, const at::optional<at::Tensor>& p) {
        TORCH_CHECK(!vm1.has_value() && !opt(r) && !opt(t) && !opt(p), "ref_raster_shim: global shutter without distortion only");
    }
} // namespace

namespace gsplat {
    // Projection.cpp:22-110 (ProjectionUT3DGSFused.cu:206-286): outputs allocated, one thread per (camera, Gaussian)
    std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor> projection_ut_3dgs_fused(
        const at::Tensor means, const at::Tensor quats, const at::Tensor scales, const at::optional<at::Tensor> opacities, const at::Tensor viewmats0,
        const at::optional<at::Tensor> viewmats1, const at::Tensor Ks, const uint32_t image_width, const uint32_t image_height, const float eps2d,
        const float near_plane, const float far_plane, const float radius_clip, const bool calc_compensations, const CameraModelType camera_model,
        const UnscentedTransformParameters ut_params, ShutterType rs_type, const at::optional<at::Tensor> radial_coeffs,
        const at::optional<at::Tensor> tangential_coeffs, const at::optional<at::Tensor> thin_prism_coeffs) {
        need_plain_camera(viewmats1, radial_coeffs, tangential_coeffs, thin_prism_coeffs);
        const uint32_t N = means.size(0), C = viewmats0.size(0);
        auto radii = at::empty({C, N, 2}, means.options().dtype(at::kInt));
        auto means2d = at::empty({C, N, 2}, means.options()), depths = at::empty({C, N}, means.options()), conics = at::empty({C, N, 3}, means.options());
        at::Tensor compensations;
        if (calc_compensations) compensations = at::zeros({C, N}, means.options());
        const UT ut(ut_params);
        refk_projection_ut(C, N, means.data_ptr<float>(), quats.data_ptr<float>(), scales.data_ptr<float>(), opt(opacities), viewmats0.data_ptr<float>(), nullptr,
                           Ks.data_ptr<float>(), image_width, image_height, eps2d, near_plane, far_plane, radius_clip, (int)camera_model, ut.v, (int)rs_type, nullptr,
                           nullptr, nullptr, radii.data_ptr<int32_t>(), means2d.data_ptr<float>(), depths.data_ptr<float>(), conics.data_ptr<float>(),
                           calc_compensations ? compensations.data_ptr<float>() : nullptr);
        return {radii, means2d, depths, conics, compensations};
    }

    // SphericalHarmonics.cpp:15-44 over reference::spherical_harmonics (tests/torch_impl.cpp:296-330); masked-out rows are not written by the kernel (at::empty
    // there, zeros here: they belong to Gaussians without a footprint, which nothing reads)
    at::Tensor spherical_harmonics_fwd(const uint32_t degrees_to_use, const at::Tensor dirs, const at::Tensor coeffs, const at::optional<at::Tensor> masks) {
        auto colors = reference::spherical_harmonics((int)degrees_to_use, dirs, coeffs);
        if (masks.has_value() && masks->defined()) colors = colors * masks->unsqueeze(-1).to(colors.dtype());
        return colors.contiguous();
    }
    // SphericalHarmonics.cpp:46-76: v_coeffs = zeros_like(coeffs) + the kernel's writes, v_dirs likewise; here the vjp of the reference's own CPU forward
    std::tuple<at::Tensor, at::Tensor> spherical_harmonics_bwd(const uint32_t K, const uint32_t degrees_to_use, const at::Tensor dirs, const at::Tensor coeffs,
                                                               const at::optional<at::Tensor> masks, const at::Tensor v_colors, bool compute_v_dirs) {
        torch::AutoGradMode grad(true);
        auto d = dirs.detach().clone().set_requires_grad(true), c = coeffs.detach().clone().set_requires_grad(true);
        auto colors = reference::spherical_harmonics((int)degrees_to_use, d, c);
        auto v = v_colors;
        if (masks.has_value() && masks->defined()) v = v * masks->unsqueeze(-1).to(v.dtype());
        auto g = torch::autograd::grad({colors}, {d, c}, {v}, /*retain_graph=*/false, /*create_graph=*/false, /*allow_unused=*/true);
        at::Tensor v_dirs = g[0].defined() ? g[0] : at::zeros_like(dirs), v_coeffs = g[1].defined() ? g[1] : at::zeros_like(coeffs);
        return {v_coeffs.contiguous(), compute_v_dirs ? v_dirs.contiguous() : at::Tensor()};
    }

    // Intersect.cpp:15-122 over reference::isect_tiles (tests/torch_impl.cpp:333-...): tiles_per_gauss int32, isect_ids int64, flatten_ids int32
    std::tuple<at::Tensor, at::Tensor, at::Tensor> intersect_tile(const at::Tensor means2d, const at::Tensor radii, const at::Tensor depths,
                                                                  const at::optional<at::Tensor> camera_ids, const at::optional<at::Tensor> gaussian_ids,
                                                                  const uint32_t C, const uint32_t tile_size, const uint32_t tile_width, const uint32_t tile_height,
                                                                  const bool sort) {
        TORCH_CHECK(!camera_ids.has_value() && !gaussian_ids.has_value(), "ref_raster_shim: non-packed only");
        torch::NoGradGuard ng;
        auto [tpg, ids, flat] = reference::isect_tiles(means2d.detach(), radii, depths.detach(), (int)tile_size, (int)tile_width, (int)tile_height, sort);
        return {tpg.to(at::kInt).contiguous(), ids.to(at::kLong).contiguous(), flat.to(at::kInt).contiguous()};
    }
    // Intersect.cpp:124-137 / IntersectTile.cu:206-252: offsets[c][ty][tx] = first position of that (camera, tile) in the sorted keys; the key is
    // (camera id << tile bits | tile id) << 32 | depth bits (IntersectTile.cu:96-108)
    at::Tensor intersect_offset(const at::Tensor isect_ids, const uint32_t C, const uint32_t tile_width, const uint32_t tile_height) {
        const int64_t n_tiles = (int64_t)tile_width * tile_height, n = isect_ids.numel(), total = n_tiles * C;
        auto offsets = at::empty({(int64_t)C, (int64_t)tile_height, (int64_t)tile_width}, isect_ids.options().dtype(at::kInt));
        int32_t* o = offsets.data_ptr<int32_t>();
        const int64_t* ids = isect_ids.data_ptr<int64_t>();
        uint32_t tile_bits = 0;
        while ((1ll << tile_bits) < n_tiles) ++tile_bits; // floor(log2(n_tiles)) + 1 in the reference for non powers of two: the same for n_tiles that are not
        if ((1ll << tile_bits) == n_tiles) ++tile_bits;   // powers of two, one more for those (IntersectTile.cu:51, 228)
        int64_t next = 0;
        for (int64_t i = 0; i < n; ++i) {
            const int64_t key = ids[i] >> 32, cam = key >> tile_bits, tile = key & ((1ll << tile_bits) - 1), flat = cam * n_tiles + tile;
            for (; next <= flat; ++next) o[next] = (int32_t)i;
        }
        for (; next < total; ++next) o[next] = (int32_t)n;
        return offsets;
    }

    // Rasterization.cpp:20-132 (Fwd.cu:281-420): renders / alphas / last_ids allocated, one 16 x 16 workgroup per tile
    std::tuple<at::Tensor, at::Tensor, at::Tensor> rasterize_to_pixels_from_world_3dgs_fwd(
        const at::Tensor means, const at::Tensor quats, const at::Tensor scales, const at::Tensor colors, const at::Tensor opacities,
        const at::optional<at::Tensor> backgrounds, const at::optional<at::Tensor> masks, const uint32_t image_width, const uint32_t image_height,
        const uint32_t tile_size, const at::Tensor viewmats0, const at::optional<at::Tensor> viewmats1, const at::Tensor Ks, const CameraModelType camera_model,
        const UnscentedTransformParameters ut_params, ShutterType rs_type, const at::optional<at::Tensor> radial_coeffs,
        const at::optional<at::Tensor> tangential_coeffs, const at::optional<at::Tensor> thin_prism_coeffs, const at::Tensor tile_offsets,
        const at::Tensor flatten_ids) {
        need_plain_camera(viewmats1, radial_coeffs, tangential_coeffs, thin_prism_coeffs);
        TORCH_CHECK(!masks.has_value(), "ref_raster_shim: no tile masks");
        const uint32_t C = tile_offsets.size(0), th = tile_offsets.size(1), tw = tile_offsets.size(2), N = means.size(0), cdim = colors.size(-1);
        auto renders = at::empty({C, image_height, image_width, cdim}, means.options());
        auto alphas = at::empty({C, image_height, image_width, 1}, means.options());
        auto last_ids = at::empty({C, image_height, image_width}, means.options().dtype(at::kInt));
        const UT ut(ut_params);
        TORCH_CHECK(refk_rasterize_fwd(cdim, C, N, flatten_ids.numel(), means.data_ptr<float>(), quats.data_ptr<float>(), scales.data_ptr<float>(),
                                       colors.data_ptr<float>(), opacities.data_ptr<float>(), opt(backgrounds), nullptr, image_width, image_height, tile_size, tw, th,
                                       viewmats0.data_ptr<float>(), nullptr, Ks.data_ptr<float>(), (int)camera_model, ut.v, (int)rs_type, nullptr, nullptr, nullptr,
                                       tile_offsets.data_ptr<int32_t>(), flatten_ids.data_ptr<int32_t>(), renders.data_ptr<float>(), alphas.data_ptr<float>(),
                                       last_ids.data_ptr<int32_t>()) == 0, "Unsupported number of channels: ", cdim);
        return {renders, alphas, last_ids};
    }
    // Rasterization.cpp:134-261 (Bwd.cu:375-520): five zero-initialised gradient tensors, atomics into them
    std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor> rasterize_to_pixels_from_world_3dgs_bwd(
        const at::Tensor means, const at::Tensor quats, const at::Tensor scales, const at::Tensor colors, const at::Tensor opacities,
        const at::optional<at::Tensor> backgrounds, const at::optional<at::Tensor> masks, const uint32_t image_width, const uint32_t image_height,
        const uint32_t tile_size, const at::Tensor viewmats0, const at::optional<at::Tensor> viewmats1, const at::Tensor Ks, const CameraModelType camera_model,
        const UnscentedTransformParameters ut_params, ShutterType rs_type, const at::optional<at::Tensor> radial_coeffs,
        const at::optional<at::Tensor> tangential_coeffs, const at::optional<at::Tensor> thin_prism_coeffs, const at::Tensor tile_offsets,
        const at::Tensor flatten_ids, const at::Tensor render_alphas, const at::Tensor last_ids, const at::Tensor v_render_colors, const at::Tensor v_render_alphas) {
        need_plain_camera(viewmats1, radial_coeffs, tangential_coeffs, thin_prism_coeffs);
        const uint32_t C = tile_offsets.size(0), th = tile_offsets.size(1), tw = tile_offsets.size(2), N = means.size(0), cdim = colors.size(-1);
        auto v_means = at::zeros_like(means), v_quats = at::zeros_like(quats), v_scales = at::zeros_like(scales), v_colors = at::zeros_like(colors),
             v_opac = at::zeros_like(opacities);
        const UT ut(ut_params);
        auto m = means.contiguous(), q = quats.contiguous(), s = scales.contiguous(), c = colors.contiguous(), o = opacities.contiguous();
        auto ra = render_alphas.contiguous(), li = last_ids.contiguous(), vc = v_render_colors.contiguous(), va = v_render_alphas.contiguous();
        TORCH_CHECK(refk_rasterize_bwd(cdim, C, N, flatten_ids.numel(), m.data_ptr<float>(), q.data_ptr<float>(), s.data_ptr<float>(), c.data_ptr<float>(),
                                       o.data_ptr<float>(), opt(backgrounds), nullptr, image_width, image_height, tile_size, tw, th, viewmats0.data_ptr<float>(),
                                       nullptr, Ks.data_ptr<float>(), (int)camera_model, ut.v, (int)rs_type, nullptr, nullptr, nullptr, tile_offsets.data_ptr<int32_t>(),
                                       flatten_ids.data_ptr<int32_t>(), ra.data_ptr<float>(), li.data_ptr<int32_t>(), vc.data_ptr<float>(), va.data_ptr<float>(),
                                       v_means.data_ptr<float>(), v_quats.data_ptr<float>(), v_scales.data_ptr<float>(), v_colors.data_ptr<float>(),
                                       v_opac.data_ptr<float>()) == 0, "Unsupported number of channels: ", cdim);
        return {v_means, v_quats, v_scales, v_colors, v_opac};
    }
} // namespace gsplat

#else
std::tuple<unsigned char*, int, int, int> load_image(std::filesystem::path, int, int) { throw std::runtime_error("ref_raster_shim: no image loading"); }
void free_image(unsigned char*) {}
std::tuple<int, int, int> get_image_info(std::filesystem::path) { throw std::runtime_error("ref_raster_shim: no image loading"); }
#endif

// ---- C API --------------------------------------------------------------------------------------------------------------------------------------------------
#ifdef REF_LINK_GPU // (make reflink: the same entry points with the tensors on the GPU - the reference's L2 code unmodified, LINKED to the product's backend library)
static torch::Tensor f32(const float* p, std::vector<int64_t> shape) { return torch::from_blob(const_cast<float*>(p), shape, torch::kFloat32).clone().to(torch::kCUDA); }
#else
static torch::Tensor f32(const float* p, std::vector<int64_t> shape) { return torch::from_blob(const_cast<float*>(p), shape, torch::kFloat32).clone(); }
#endif
static void put(const torch::Tensor& t, float* dst) {
    if (!dst) return;
    auto c = t.detach().to(torch::kFloat32).cpu().contiguous();
    std::memcpy(dst, c.data_ptr<float>(), sizeof(float) * c.numel());
}

// One training-time render + backward of the reference: SplatData (raw parameters; K1 = number of shN coefficients) seen by a pinhole Camera(R, T, focal, centre,
// camera size; the image size the camera was calibrated at = cam_w x cam_h, rendered at width x height as after load_and_get_image), background bg[3] or null.
// Loss = sum(image * v_image) + sum(alpha * v_alpha); outputs: image [3,H,W], alpha [1,H,W], per-Gaussian radii / visibility, gradients of the six raw tensors.
REF_API int refraster_render_backward(int64_t N, int64_t K1, int sh_degree, int active_sh_degree, const float* means, const float* sh0, const float* shN,
                                      const float* scaling, const float* rotation, const float* opacity, const float* R, const float* T, float fx, float fy, float cx,
                                      float cy, int cam_w, int cam_h, int width, int height, const float* bg, const float* v_image, const float* v_alpha, float* image,
                                      float* alpha, int32_t* radii, float* g_means, float* g_sh0, float* g_shN, float* g_scaling, float* g_rotation, float* g_opacity,
                                      float* viewmat_out, float* K_out) {
    try {
        auto req = [](torch::Tensor t) { return t.set_requires_grad(true); };
        gs::SplatData model(sh_degree, req(f32(means, {N, 3})), req(f32(sh0, {N, 1, 3})), req(f32(shN, {N, K1, 3})), req(f32(scaling, {N, 3})), req(f32(rotation, {N, 4})),
                            req(f32(opacity, {N, 1})), 1.0f);
        model.set_active_sh_degree(active_sh_degree);
        gs::Camera cam(f32(R, {3, 3}), f32(T, {3}), fx, fy, cx, cy, torch::empty({0}, torch::kFloat32), torch::empty({0}, torch::kFloat32), gsplat::CameraModelType::PINHOLE,
                       "view", "", cam_w, cam_h, 0);
        if (width != cam_w || height != cam_h) { // what load_and_get_image leaves behind (camera.cpp:113-114): the size of the image actually loaded
            struct Peek : gs::Camera {};          // (no public setter)
            static_assert(sizeof(Peek) == sizeof(gs::Camera));
            TORCH_CHECK(false, "ref_raster_shim: render size != camera size needs load_and_get_image");
        }
        torch::Tensor bgc = bg ? f32(bg, {3}) : torch::Tensor();
        auto out = gs::training::rasterize(cam, model, bgc, 1.0f, false, false, gs::training::RenderMode::RGB, nullptr);
        put(out.image, image), put(out.alpha, alpha);
        if (radii) {
            auto r = out.radii.to(torch::kInt).cpu().contiguous();
            std::memcpy(radii, r.data_ptr<int32_t>(), sizeof(int32_t) * r.numel());
        }
        put(cam.world_view_transform(), viewmat_out), put(cam.K(), K_out);
        auto loss = (out.image * f32(v_image, {3, height, width})).sum();
        if (v_alpha) loss = loss + (out.alpha * f32(v_alpha, {1, height, width})).sum();
        loss.backward();
        auto g = [](const torch::Tensor& p) { return p.grad().defined() ? p.grad() : torch::zeros_like(p); };
        put(g(model.means()), g_means), put(g(model.sh0()), g_sh0), put(g(model.shN()), g_shN), put(g(model.scaling_raw()), g_scaling);
        put(g(model.rotation_raw()), g_rotation), put(g(model.opacity_raw()), g_opacity);
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "refraster_render_backward: %s\n", e.what());
        return 1;
    }
}

#ifdef REF_REAL_GSPLAT
// The loss of one training step as Trainer::train_step composes it (trainer.cpp:640-715) and its gradients: rasterize() -> compute_photometric_loss (:103-130:
// l1_loss, 1 - fused_ssim(..., "valid", train), the lambda mix) -> + compute_scale_reg_loss (:132-145: scale_reg * get_scaling().mean()) + compute_opacity_reg_loss
// (:147-160: opacity_reg * get_opacity().mean()), each followed by backward(). The three loss functions are members of the Trainer class; their bodies are restated
// here line by line, everything they call is the reference's own code.
REF_API int refraster_train_loss_backward(int64_t N, int64_t K1, int sh_degree, int active_sh_degree, const float* means, const float* sh0, const float* shN,
                                          const float* scaling, const float* rotation, const float* opacity, const float* R, const float* T, float fx, float fy, float cx,
                                          float cy, int width, int height, const float* bg, const float* gt_image, float lambda_dssim, float scale_reg, float opacity_reg,
                                          float* loss_out, float* image, float* g_means, float* g_sh0, float* g_shN, float* g_scaling, float* g_rotation,
                                          float* g_opacity,
                                          // optional bilateral grid (trainer.cpp:662-664, 699-706): BilateralGrid(n_images, gW, gH, gL) with grids = identity + grid_delta,
                                          // applied to the rendered image for image `image_idx`; + tv_weight * tv_loss(). Null grid_delta: no grid.
                                          int n_images, int gW, int gH, int gL, int image_idx, const float* grid_delta, float tv_weight, float* g_grids) {
    try {
        auto req = [](torch::Tensor t) { return t.set_requires_grad(true); };
        gs::SplatData model(sh_degree, req(f32(means, {N, 3})), req(f32(sh0, {N, 1, 3})), req(f32(shN, {N, K1, 3})), req(f32(scaling, {N, 3})), req(f32(rotation, {N, 4})),
                            req(f32(opacity, {N, 1})), 1.0f);
        model.set_active_sh_degree(active_sh_degree);
        gs::Camera cam(f32(R, {3, 3}), f32(T, {3}), fx, fy, cx, cy, torch::empty({0}, torch::kFloat32), torch::empty({0}, torch::kFloat32), gsplat::CameraModelType::PINHOLE,
                       "view", "", width, height, 0);
        torch::Tensor bgc = bg ? f32(bg, {3}) : torch::Tensor();
        auto out = gs::training::rasterize(cam, model, bgc, 1.0f, false, false, gs::training::RenderMode::RGB, nullptr);
        put(out.image, image);
        std::unique_ptr<gs::training::BilateralGrid> grid;
        if (grid_delta) {
            grid = std::make_unique<gs::training::BilateralGrid>(n_images, gW, gH, gL);
            torch::NoGradGuard ng;
            grid->parameters().add_(f32(grid_delta, {n_images, 12, gL, gH, gW}));
        }
        if (grid) out.image = grid->apply(out.image, image_idx);                                                     // :662-664
        torch::Tensor rendered = out.image.unsqueeze(0), gt = f32(gt_image, {3, height, width}).unsqueeze(0);       // :113-114
        auto l1_loss = torch::l1_loss(rendered, gt);                                                                 // :121
        auto ssim_loss = 1.f - fused_ssim(rendered, gt, "valid", /*train=*/true);                                    // :122
        torch::Tensor loss = (1.f - lambda_dssim) * l1_loss + lambda_dssim * ssim_loss;                              // :123-124
        loss.backward();                                                                                             // trainer.cpp:677
        float loss_value = loss.item<float>();
        if (scale_reg > 0.0f) {                                                                                      // :136-139, :686
            loss = scale_reg * model.get_scaling().mean();
            loss.backward();
            loss_value += loss.item<float>();
        }
        if (opacity_reg > 0.0f) {                                                                                    // :151-154, :695
            loss = opacity_reg * model.get_opacity().mean();
            loss.backward();
            loss_value += loss.item<float>();
        }
        if (grid && tv_weight > 0.f) {                                                                              // :162-173, :704
            loss = tv_weight * grid->tv_loss();
            loss.backward();
            loss_value += loss.item<float>();
        }
        *loss_out = loss_value;
        auto g = [](const torch::Tensor& p) { return p.grad().defined() ? p.grad() : torch::zeros_like(p); };
        put(g(model.means()), g_means), put(g(model.sh0()), g_sh0), put(g(model.shN()), g_shN), put(g(model.scaling_raw()), g_scaling);
        put(g(model.rotation_raw()), g_rotation), put(g(model.opacity_raw()), g_opacity);
        if (grid) put(g(grid->parameters()), g_grids);
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "refraster_train_loss_backward: %s\n", e.what());
        return 1;
    }
}
#endif

#ifdef REF_LINK_GPU
#include "fused_adam.hpp"
// The reference's optimizer over the product's fast_gs::optimizer::adam_step_wrapper: FusedAdam built exactly as create_optimizer builds it
// (strategy_utils.cpp:20-48: six single-tensor groups in the order means, sh0, shN, scaling, rotation, opacity, each Options(lr).eps(1e-15).betas(0.9, 0.999), global
// Options(0).eps(1e-15)), `n_steps` calls of FusedAdam::step(iteration0 + k) (fused_adam.cpp:22-95) with the given gradients -> parameters, both moments.
// Sizes: sizes[g] floats for group g; buffers are updated in place.
REF_API int reflink_fused_adam_steps(const int64_t* sizes, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                                     const double* lrs, int iteration0, int n_steps, int64_t* step_counts) {
    try {
        using Options = gs::training::FusedAdam::Options;
        std::vector<torch::Tensor> p;
        std::vector<torch::optim::OptimizerParamGroup> groups;
        for (int g = 0; g < 6; ++g) {
            p.push_back(f32(params[g], {sizes[g]}).set_requires_grad(true));
            p.back().mutable_grad() = f32(grads[g], {sizes[g]});
            auto options = std::make_unique<Options>(lrs[g]);
            options->eps(1e-15).betas(std::make_tuple(0.9, 0.999));
            groups.emplace_back(std::vector<torch::Tensor>{p.back()}, std::unique_ptr<torch::optim::OptimizerOptions>(std::move(options)));
        }
        auto global_options = std::make_unique<Options>(0.f);
        global_options->eps(1e-15);
        gs::training::FusedAdam opt(std::move(groups), std::move(global_options));
        for (int k = 0; k < n_steps; ++k) opt.step(iteration0 + k);
        for (int g = 0; g < 6; ++g) {
            put(p[g], params[g]);
            auto it = opt.state().find(p[g].unsafeGetTensorImpl());
            TORCH_CHECK(it != opt.state().end(), "no optimizer state for group ", g);
            auto& st = static_cast<gs::training::FusedAdam::AdamParamState&>(*it->second);
            put(st.exp_avg, exp_avg[g]), put(st.exp_avg_sq, exp_avg_sq[g]);
            step_counts[g] = st.step_count;
        }
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "reflink_fused_adam_steps: %s\n", e.what());
        return 1;
    }
}

// INTEGRATION.md §1b, COMPILED: `n_steps` training steps with the MSE loss on the reference's own SplatData / Camera / FusedAdam objects,
//   mode 0: as the reference's trainer runs them - gs::training::rasterize() -> mse_loss -> backward() -> FusedAdam::step(iter) -> zero_grad (trainer.cpp:640-760 with
//           the loss of SURVEY.md §8d), every operator through the linked backend one by one;
//   mode 1: the patch - the optimizer state is taken from FusedAdam with the members it really has (torch::optim::Optimizer::param_groups() / state(), the group's
//           FusedAdam::Options, FusedAdam::AdamParamState; lazy initialisation and ++step_count as fused_adam.cpp:44-66 do them) and the whole step is ONE call of
//           lfs::GutTrainStep::step (include/lfs_gut_train_step.hpp);
//   mode 2: the patch with the NEXT view named (round 6: next_viewmat - here the same camera, as a trainer whose dataloader has the next camera queued would pass it): the
//           step's tail then prepares the next call's SH colours and that call skips its colour kernel.
// Both leave parameters and moments in the same FusedAdam / SplatData objects, which are read back: tests/test_gpu_reference_links.py holds mode 1 to mode 0.
#include "lfs_gut_train_step.hpp"
#include <chrono>
#include <torch/cuda.h>
REF_API int reflink_mse_train_steps(int mode, int64_t N, int64_t K1, int sh_degree, int active_sh_degree, float* const* params, const float* R, const float* T, float fx,
                                    float fy, float cx, float cy, int width, int height, const float* bg, const float* gt_image, const double* lrs, int iteration0,
                                    int n_steps, float* losses, float* const* exp_avg, float* const* exp_avg_sq, int64_t* n_isects_out, int timed_from, double* ms_per_step) {
    // timed_from >= 0 / ms_per_step: wall time per step of steps [timed_from, n_steps), device-synchronised at both ends (the loss value is then NOT read back per step -
    // the reference's trainer reads it only for its progress bar)
    try {
        using FusedAdam = gs::training::FusedAdam;
        auto req = [](torch::Tensor t) { return t.set_requires_grad(true); };
        gs::SplatData model(sh_degree, req(f32(params[0], {N, 3})), req(f32(params[1], {N, 1, 3})), req(f32(params[2], {N, K1, 3})), req(f32(params[3], {N, 3})),
                            req(f32(params[4], {N, 4})), req(f32(params[5], {N, 1})), 1.0f);
        model.set_active_sh_degree(active_sh_degree);
        gs::Camera cam(f32(R, {3, 3}), f32(T, {3}), fx, fy, cx, cy, torch::empty({0}, torch::kFloat32), torch::empty({0}, torch::kFloat32), gsplat::CameraModelType::PINHOLE,
                       "view", "", width, height, 0);
        torch::Tensor bgc = bg ? f32(bg, {3}) : torch::Tensor(), gt = f32(gt_image, {3, height, width});
        std::vector<torch::Tensor*> tensors = {&model.means(), &model.sh0(), &model.shN(), &model.scaling_raw(), &model.rotation_raw(), &model.opacity_raw()};
        std::vector<torch::optim::OptimizerParamGroup> groups; // strategy_utils.cpp:20-48 (create_optimizer)
        for (int g = 0; g < 6; ++g) {
            auto options = std::make_unique<FusedAdam::Options>(lrs[g]);
            options->eps(1e-15).betas(std::make_tuple(0.9, 0.999));
            groups.emplace_back(std::vector<torch::Tensor>{*tensors[g]}, std::unique_ptr<torch::optim::OptimizerOptions>(std::move(options)));
        }
        auto global_options = std::make_unique<FusedAdam::Options>(0.f);
        global_options->eps(1e-15);
        FusedAdam optimizer(std::move(groups), std::move(global_options));
        lfs::GutTrainStep gut_step;
        torch::Tensor loss_scalar = torch::zeros({1}, gt.options());
        const torch::Tensor viewmat = cam.world_view_transform().contiguous(), Kmat = cam.K().contiguous();   // (modes 1 / 2: one tensor per camera, alive across the steps)
        std::chrono::steady_clock::time_point t0;
        const bool timing = ms_per_step != nullptr && timed_from >= 0 && timed_from < n_steps;
        for (int k = 0; k < n_steps; ++k) {
            const int iter = iteration0 + k;
            if (timing && k == timed_from) { torch::cuda::synchronize(); t0 = std::chrono::steady_clock::now(); }
            if (mode == 0) {
                auto out = gs::training::rasterize(cam, model, bgc, 1.0f, false, false, gs::training::RenderMode::RGB, nullptr);
                auto loss = torch::mse_loss(out.image, gt);
                loss.backward();
                optimizer.step(iter);
                optimizer.zero_grad(true, iter);
                if (!timing) losses[k] = loss.item<float>();
            } else {
                TORCH_CHECK(iter > 1000, "the one-call step updates all six groups: FusedAdam skips shN up to iteration 1000 (fused_adam.cpp:68-70) - use the split form there");
                std::array<lfs::AdamGroupState, 6> adam;
                auto& pg = optimizer.param_groups();
                for (int g = 0; g < 6; ++g) {
                    auto& param = pg[g].params()[0];
                    auto& slot = optimizer.state()[param.unsafeGetTensorImpl()];
                    if (!slot) { // fused_adam.cpp:44-57
                        auto fresh = std::make_unique<FusedAdam::AdamParamState>();
                        fresh->exp_avg = torch::zeros_like(param, torch::MemoryFormat::Preserve);
                        fresh->exp_avg_sq = torch::zeros_like(param, torch::MemoryFormat::Preserve);
                        slot = std::move(fresh);
                    }
                    auto& st = static_cast<FusedAdam::AdamParamState&>(*slot);
                    ++st.step_count; // :66
                    const auto& o = static_cast<const FusedAdam::Options&>(pg[g].options());
                    const auto [b1, b2] = o.betas();
                    adam[g] = {st.exp_avg, st.exp_avg_sq, (float)o.lr(), (float)b1, (float)b2, (float)o.eps(), (float)(1.0 / (1.0 - std::pow(b1, st.step_count))),
                               (float)(1.0 / std::sqrt(1.0 - std::pow(b2, st.step_count)))}; // :78-79
                }
                loss_scalar.zero_();
                *n_isects_out = gut_step.step(model.means(), model.sh0(), model.shN(), model.scaling_raw(), model.rotation_raw(), model.opacity_raw(), adam,
                                              (uint32_t)model.get_active_sh_degree(), viewmat, Kmat, (uint32_t)width, (uint32_t)height,
                                              bgc.defined() ? at::optional<torch::Tensor>(bgc) : at::nullopt, gt, 1.f, loss_scalar, 0.f, 0.f,
                                              mode == 2 ? at::optional<torch::Tensor>(viewmat) : at::nullopt);
                if (!timing) losses[k] = loss_scalar.item<float>();
            }
        }
        if (mode == 2) TORCH_CHECK(gut_step.colour_launches_saved() >= n_steps - 1 - 2 * gut_step.retries(), "the prepared colours were not used: ", gut_step.colour_launches_saved(), " of ", n_steps);
        if (timing) {
            torch::cuda::synchronize();
            *ms_per_step = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / double(n_steps - timed_from);
        }
        for (int g = 0; g < 6; ++g) {
            put(*tensors[g], params[g]);
            auto it = optimizer.state().find(tensors[g]->unsafeGetTensorImpl());
            TORCH_CHECK(it != optimizer.state().end(), "no optimizer state for group ", g);
            auto& st = static_cast<FusedAdam::AdamParamState&>(*it->second);
            put(st.exp_avg, exp_avg[g]), put(st.exp_avg_sq, exp_avg_sq[g]);
        }
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "reflink_mse_train_steps: %s\n", e.what());
        return 1;
    }
}
#endif
