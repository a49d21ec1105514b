"""Developer script (CPU only): wave-evaluation counts and lane utilisation of the rasterizer's kernels on a WINDOW of SYN-B (same
Gaussian statistics as the benchmark: the full scene, a 160x96-pixel crop of the 1080p view), run on the wavefront emulator of tests/emul.
    python tools/emul_eval_counts.py [width height]"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import lichtfeld_studio_amd as lfs  # noqa: E402
from lichtfeld_studio_amd import ops, scenes  # noqa: E402
import oracle  # noqa: E402

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (160, 96)
tmp = tempfile.mkdtemp()
out = os.path.join(tmp, "liblfs_raster_emul.so")
# Round 5: histograms next to the sums - how many of the four 4x4 quarters of a cell, and how many of its 64 lanes, are live in an ACCUMULATING backward evaluation (the
# kill criterion of the quarter-packed walk). The counting macro is re-defined in a patched COPY of the source: csrc/raster.hip itself (and the library's source hash) stay as they are.
CSRC = os.path.join(ROOT, "lichtfeld-studio_amd", "csrc")
text = open(os.path.join(CSRC, "raster.hip")).read()
patch = """
#ifdef LFS_EMULATE
#include <map>
#include <array>
extern "C" { __attribute__((visibility("default"))) unsigned long long lfs_emul_quarter_hist[5] = {0, 0, 0, 0, 0}; __attribute__((visibility("default"))) unsigned long long lfs_emul_lane_hist[65] = {0};
             __attribute__((visibility("default"))) unsigned long long lfs_emul_pack[4] = {0, 0, 0, 0}; }
// per wavefront (= one 8x8 cell's walk): [0..3] accumulating evaluations that touch quarter q, [4] the open bundle's quarter set, [5] bundles closed so far
static std::map<unsigned long long, std::array<unsigned, 6>> lfs_emul_walks;
static inline void lfs_emul_note(unsigned long long key, unsigned qset) {
    auto& w = lfs_emul_walks[key];
    for (int q = 0; q < 4; ++q) w[q] += (qset >> q) & 1u;
    if (w[4] & qset) { ++w[5]; w[4] = qset; } else w[4] |= qset;      // greedy: the next list entry joins the iteration iff its live quarters are free in it
}
extern "C" __attribute__((visibility("default"))) void lfs_emul_pack_finish() {
    for (auto& kv : lfs_emul_walks) { auto& w = kv.second; lfs_emul_pack[0] += w[5] + (w[4] != 0); unsigned m = 0; for (int q = 0; q < 4; ++q) m = w[q] > m ? w[q] : m; lfs_emul_pack[1] += m; }
    lfs_emul_walks.clear();
}
#undef LFS_EMUL_LANES
#define LFS_EMUL_LANES(m) do { const unsigned long long m_ = (m); if ((threadIdx.x & 63) == 0) { lfs_emul_counters[4] += __builtin_popcountll(m_); \\
    lfs_emul_counters[5] += ((m_ & 0xffffffffull) != 0) + ((m_ >> 32) != 0); int ql_ = 0; unsigned qs_ = 0; \\
    for (int q_ = 0; q_ < 4; ++q_) { const unsigned long long qm_ = (0x0f0f0f0full << ((q_ & 1) * 4)) << ((q_ >> 1) * 32); ql_ += (m_ & qm_) != 0; qs_ |= ((m_ & qm_) != 0) << q_; } \\
    lfs_emul_counters[6] += ql_; ++lfs_emul_quarter_hist[ql_]; ++lfs_emul_lane_hist[__builtin_popcountll(m_)]; \\
    lfs_emul_note((((unsigned long long)blockIdx.y << 20 | blockIdx.x) << 8) | (threadIdx.x >> 6), qs_); } } while (0)
#endif
namespace lfs {
"""
assert text.count("\nnamespace lfs {\n") >= 1
text = text.replace("\nnamespace lfs {\n", patch, 1)
src = os.path.join(CSRC, ".emul_eval_counts_raster.hip")      # (next to the real one: the relative #includes resolve)
open(src, "w").write(text)
try:
    subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang++", "-x", "c++", "-std=c++17", "-O2", "-DLFS_EMULATE", "-fPIC", "-shared", "-ffp-contract=on",
                           "-I" + os.path.join(ROOT, "tests", "emul"), "-Wno-unused-value", "-Wno-unknown-attributes",
                           src, os.path.join(ROOT, "tests", "emul", "emul_stubs.cpp"), "-o", out])
finally:
    os.remove(src)
lib = C.CDLL(out)
lib.lfs_rasterize_workspace_bytes.restype = C.c_size_t
ops.load_library = lambda: lib
ops.require_gpu = lambda *a: None
ops.stream = lambda: None
ops.workspace = lambda nbytes, dev, tag: torch.zeros(max(int(nbytes), 256), dtype=torch.uint8)
counters = (C.c_ulonglong * 8).in_dll(lib, "lfs_emul_counters")
qhist = (C.c_ulonglong * 5).in_dll(lib, "lfs_emul_quarter_hist")
lhist = (C.c_ulonglong * 65).in_dll(lib, "lfs_emul_lane_hist")
pack = (C.c_ulonglong * 4).in_dll(lib, "lfs_emul_pack")

sc = scenes.syn_b(n=1_000_000, n_views=4)
quats = torch.nn.functional.normalize(sc.raw_quats, dim=-1).numpy(); scales = sc.raw_scales.exp().numpy(); opac = torch.sigmoid(sc.raw_opacities).numpy()
means = sc.means.numpy()
vm = sc.viewmats[:1].numpy().copy(); K = sc.Ks[:1].numpy().copy()
x0, y0 = (sc.width - W) // 2 + 200, (sc.height - H) // 2 + 100        # an off-centre window of the full view
K[0, 0, 2] -= x0; K[0, 1, 2] -= y0
radii, m2, d, _, _ = oracle.projection_ut_3dgs_fused(means, quats, scales, opac, vm, None, K, W, H, camera_model=0, rs_type=0, radial_coeffs=None,
                                                     tangential_coeffs=None, thin_prism_coeffs=None)
keep = np.nonzero((radii[0] > 0).all(axis=1))[0]
means, quats, scales, opac = means[keep], quats[keep], scales[keep], opac[keep]
radii, m2, d = radii[:, keep], m2[:, keep], d[:, keep]
tw, th = (W + 15) // 16, (H + 15) // 16
_, ids, flat = oracle.intersect_tile(m2, radii, d, 1, 16, tw, th, True)
offs = oracle.intersect_offset(ids, 1, tw, th)
N = len(keep)
print(f"window {W}x{H}: {N} Gaussians, {len(flat)} tile entries ({len(flat) / (tw * th):.0f} per tile; SYN-B full view: 546)")
rng = np.random.default_rng(0)
t = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dt)
colors = rng.random((1, N, 3)).astype(np.float32)
args = (t(means), t(quats), t(scales), t(colors), t(opac[None]), None, None, W, H, 16, t(vm), None, t(K), lfs.CameraModelType.PINHOLE, None,
        lfs.ShutterType.GLOBAL, None, None, None, t(offs, torch.int32), t(flat, torch.int32))
lib.lfs_set_debug_flags(0)
for i in range(8): counters[i] = 0
rc, ra, li = ops.rasterize_to_pixels_from_world_3dgs_fwd(*args)
ops.rasterize_to_pixels_from_world_3dgs_bwd(*args, ra, li, torch.randn_like(rc), torch.randn_like(ra))
c = list(counters)
print(f"8x8 cells: fwd evaluations {c[0]} (composited {c[1]}), bwd evaluations {c[2]} (accumulated {c[3]}) = {c[2] / len(flat):.3f} per tile entry")
print(f"accumulated bwd evaluations: {c[4] / c[3]:.1f} of 64 lanes live ({100 * c[4] / (64 * c[3]):.0f} %); 8x4 half cells with a live lane: {c[5] / c[3]:.2f} of 2; "
      f"4x4 quarters: {c[6] / c[3]:.2f} of 4")

q = np.array(list(qhist), np.float64); l = np.array(list(lhist), np.float64)
print("live 4x4 quarters per accumulating evaluation (0..4):", [f"{100 * x / q.sum():.1f} %" for x in q], f"-> <= 2 quarters live in {100 * q[:3].sum() / q.sum():.1f} % of the evaluations")
cum = np.cumsum(l) / l.sum()
print("live lanes per accumulating evaluation: <= 8 lanes", f"{100 * cum[8]:.1f} %, <= 16: {100 * cum[16]:.1f} %, <= 32: {100 * cum[32]:.1f} %, <= 48: {100 * cum[48]:.1f} %, all 64: {100 * l[64] / l.sum():.1f} %")
# what a quarter-packed walk could save at best: an evaluation with q live quarters costs q / 4 of a full one if quarters of two entries could always be paired
print(f"ideal quarter packing: {100 * (q * np.arange(5) / 4).sum() / q[1:].sum():.1f} % of today's accumulating evaluations (lower bound, perfect pairing, no overhead)")

# what packing achieves on the ACTUAL order of the walks (a pixel's entries must stay in list order, so only neighbours in a cell's culled list can share an iteration):
lib.lfs_emul_pack_finish()
print(f"greedy packing of consecutive entries with disjoint live quarters (up to four entries per iteration, one per DPP row): {pack[0]} iterations for {c[3]} evaluations "
      f"= {100 * pack[0] / c[3]:.1f} %")
print(f"four independent 4x4 walks per wavefront (each quarter skips the entries that miss it; the wavefront iterates max-over-quarters times): {pack[1]} iterations "
      f"= {100 * pack[1] / c[3]:.1f} %")
