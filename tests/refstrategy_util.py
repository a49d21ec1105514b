"""Scenarios shared by oracle/make_golden_ref_strategy.py (which drives the REFERENCE's strategy layer on the CPU and records what it does) and
tests/test_gpu_strategy_reference.py (which drives the product's strategies through the same iterations with the same random draws). Everything the two sides must
agree on up front is generated here from integer hashes - exact in float64, so both sides see identical float32 inputs on any platform."""
import numpy as np

GOLD = "ref_strategy.npz"
NAMES = ("means", "sh0", "shN", "scaling", "rotation", "opacity")          # param-group order, strategy_utils.cpp:35-40


def hashed(shape, salt):
    """deterministic values in [-0.5, 0.5): a 32-bit integer mix of (index, salt)"""
    n = int(np.prod(shape))
    x = (np.arange(n, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(salt) * np.uint64(40503) + np.uint64(12345)) & np.uint64(0xFFFFFFFF)
    x = ((x ^ (x >> np.uint64(15))) * np.uint64(2246822519)) & np.uint64(0xFFFFFFFF)
    x = ((x ^ (x >> np.uint64(13))) * np.uint64(3266489917)) & np.uint64(0xFFFFFFFF)
    x = x ^ (x >> np.uint64(16))
    return (x.astype(np.float64) / 4294967296.0 - 0.5).reshape(shape)


SCENARIOS = {
    # MCMC: refinement (relocate + add) at iterations 1002, 1005, 1008; SH degree raised at 1004; dead Gaussians by opacity and by a zero quaternion; the cap reached at 1008, so that 1011 relocates without growing (the one refinement after which Adam still steps)
    # (iterations 1001 .. 1010: past the first 1000, where FusedAdam skips the shN group - fused_adam.cpp:68-70)
    "mcmc": dict(kind="mcmc", N=160, K=3, sh_degree=1, scene_scale=1.3, it0=1000, iters=12, full_state=(1002, 1005, 1008, 1011, 1012),
                 params=dict(iterations=2000, start_refine=2, refine_every=3, stop_refine=1500, sh_degree_interval=4, max_cap=180, min_opacity=0.005, opacity_lr=0.5)),
    # ADC: duplicate + split + prune at 3, 6, 9; opacity reset at 5 (after which "too big" pruning is active); SH degree raised at 4 and 8
    "default": dict(kind="default", N=90, K=3, sh_degree=1, scene_scale=1.3, it0=0, iters=10, full_state=(3, 5, 6, 9, 10),
                    params=dict(iterations=60, start_refine=1, refine_every=3, stop_refine=40, sh_degree_interval=4, reset_every=5, grad_threshold=2e-4,
                                grow_scale3d=0.01, prune_scale3d=0.1, prune_opacity=0.005, revised_opacity=0)),
    "default_revised_opacity": dict(kind="default", N=120, K=0, sh_degree=0, scene_scale=0.7, it0=0, iters=7, full_state=(3, 5, 6, 7),
                                    params=dict(iterations=40, start_refine=1, refine_every=3, stop_refine=40, sh_degree_interval=4, reset_every=5, grad_threshold=2e-4,
                                                grow_scale3d=0.01, prune_scale3d=0.1, prune_opacity=0.005, revised_opacity=1)),
}


def initial(sc):
    N, K = sc["N"], sc["K"]
    f = lambda shape, salt, scale=1.0, shift=0.0: (hashed(shape, salt) * scale + shift).astype(np.float32)
    init = dict(means=f((N, 3), 1, 4.0), sh0=f((N, 1, 3), 2, 2.0), shN=f((N, K, 3), 3, 0.4), scaling=f((N, 3), 4, 5.0, -3.5),     # exp: 0.0025 .. 0.37
                rotation=f((N, 4), 5, 2.0), opacity=f((N,), 6, 12.0, -1.0))                                                        # sigmoid: 0.0009 .. 0.99
    init["rotation"][7] = 0                        # a degenerate quaternion: dead (MCMC) / pruned (ADC)
    init["rotation"][N // 2] = 0
    return init


def grads(shapes, it):
    """synthetic parameter gradients of iteration `it` for the current parameter shapes"""
    g = [(hashed(s, 100 * it + i) * 2e-2).astype(np.float32) for i, s in enumerate(shapes)]
    g[5] = g[5] + np.float32(2e-2)      # a steady pull towards lower opacity: Gaussians keep dying, so every refinement has something to relocate / prune
    return g


def densification_info(n, it):
    """[2, n]: visibility counts 0 .. 7 and accumulated gradient norms such that about a third of the rows exceed grad_threshold = 2e-4"""
    cnt = np.floor((hashed((n,), 7000 + it) + 0.5) * 8).astype(np.float32)
    acc = ((hashed((n,), 8000 + it) + 0.5) * 6e-4 * np.maximum(cnt, 1)).astype(np.float32)
    return np.stack([cnt, acc])
