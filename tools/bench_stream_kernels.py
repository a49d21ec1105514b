#!/usr/bin/env python3
"""The step's streaming kernels ALONE, through their C-ABI entry points, N = 1 M Gaussians, SH degree 3 (round 6):
    lfs_gut_finish_adam  (raster_finish_adam_kernel<true>: 29 streams)      lfs_sh_model_bwd_adam_all  (sh_bwd_kernel<16, true, false, true>)
each launched back to back `reps` times on random operands and timed with events - what the kernel does when nothing else has heated the package or
touched the caches, next to its figure inside the step (bench.py's kernel table) and to the hand-written stream ceilings (tools/hbm_stream.hip).
    python tools/bench_stream_kernels.py [--n 1000000] [--reps 50]"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lichtfeld_studio_amd as lfs  # noqa: E402,F401
from lichtfeld_studio_amd.capi import check, load_library, ptr, stream  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--visible", type=float, default=0.95)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = load_library()
    N, K = args.n, 16
    g = torch.Generator(device="cpu").manual_seed(1)
    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g) * scale).to(dev)
    means, raw_scales, raw_quats, raw_opac = rnd(N, 3), rnd(N, 3, scale=0.3) - 3.5, rnd(N, 4), rnd(N)
    quats, scales, opac = torch.nn.functional.normalize(raw_quats, dim=-1), raw_scales.exp(), torch.sigmoid(raw_opac)
    sh0, shN = rnd(N, 1, 3, scale=0.5), rnd(N, K - 1, 3, scale=0.1)
    colors = torch.rand(N, 3, generator=g).to(dev)
    radii = torch.where(torch.rand(N, generator=g) < args.visible, 5, 0).to(torch.int32).to(dev)[:, None].repeat(1, 2).contiguous()
    viewmat = torch.eye(4, device=dev); viewmat[2, 3] = 5.0
    v_dirs = rnd(N, 3, scale=1e-4)
    order = ["means", "raw_scales", "raw_quats", "raw_opacities"]
    params = dict(means=means, raw_scales=raw_scales, raw_quats=raw_quats, raw_opacities=raw_opac)
    mom = {k: (torch.zeros_like(p), torch.zeros_like(p)) for k, p in params.items()}
    m = (C.c_void_p * 4)(*[mom[k][0].data_ptr() for k in order])
    v = (C.c_void_p * 4)(*[mom[k][1].data_ptr() for k in order])
    sc = (C.c_float * 24)(*([1e-5, 0.9, 0.999, 1e-15, 1.0, 1.0] * 4))
    nbytes = lib.lfs_rasterize_workspace_bytes(C.c_uint32(1), C.c_uint32(N), C.c_uint32(3), C.c_uint32(1920), C.c_uint32(1080), C.c_uint32(16), C.c_int64(1))
    ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    acc_off = lib.lfs_rasterize_workspace_acc_offset(C.c_uint32(1), C.c_uint32(N))
    acc = ws[acc_off:acc_off + N * 64].view(torch.float32)
    acc.copy_((torch.randn(N * 16, generator=g) * 1e-3).to(dev))
    acc_rows = C.c_void_p(ws.data_ptr() + acc_off)
    loss = torch.zeros(1, device=dev)
    m0, v0, mN, vN = torch.zeros_like(sh0), torch.zeros_like(sh0), torch.zeros_like(shN), torch.zeros_like(shN)
    sc6 = (C.c_float * 6)(1e-4, 0.9, 0.999, 1e-15, 1.0, 1.0)

    def finish():
        check(lib.lfs_gut_finish_adam(C.c_uint32(N), ptr(means), ptr(raw_scales), ptr(raw_quats), ptr(raw_opac), ptr(quats), ptr(scales), ptr(opac), ptr(v_dirs), m, v, sc,
                                      C.c_float(0.0), C.c_float(0.0), ptr(loss), ptr(ws), C.c_size_t(ws.numel()), stream()), "gut_finish_adam")

    def sh_bwd():
        check(lib.lfs_sh_model_bwd_adam_all(C.c_uint32(N), C.c_uint32(K), C.c_uint32(3), ptr(means), ptr(viewmat), ptr(sh0), ptr(shN), ptr(radii), ptr(colors),
                                            acc_rows, ptr(v_dirs), ptr(m0), ptr(v0), sc6, ptr(mN), ptr(vN), sc6, stream()), "sh_model_bwd_adam_all")

    out = {"n": N, "reps": args.reps, "visible_fraction": args.visible}
    for name, fn, nbytes_alg in (("finish_adam", finish, 326e6 * N / 1e6), ("sh_bwd_adam", sh_bwd, 1154e6 * N / 1e6)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        out[name] = {"ms": round(ms, 4), "TBps_at_the_step's_PMC_bytes": round(nbytes_alg / (ms * 1e-3) / 1e12, 3)}
    # alternating, each launch timed on its own: what finish_adam does when sh_bwd_adam's 1.1 GB have just gone through the caches (as in the step)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.reps)]
    torch.cuda.synchronize()
    for a, b, c in evs:
        a.record(); sh_bwd(); b.record(); finish(); c.record()
    torch.cuda.synchronize()
    import statistics
    out["alternating"] = {"sh_bwd_adam_ms": round(statistics.median(a.elapsed_time(b) for a, b, c in evs), 4), "finish_adam_ms": round(statistics.median(b.elapsed_time(c) for a, b, c in evs), 4)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
