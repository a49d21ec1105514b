"""Generate tests/golden/ref_strategy.npz by running the REFERENCE'S OWN strategy layer and optimizer on the CPU (oracle/_ref/libref_strategy.so: mcmc.cpp,
default_strategy.cpp, strategy_utils.cpp, fused_adam.cpp, scheduler.cpp compiled in place against libtorch over the reference's kernels; `make -C oracle refk
refstrategy`) through the scenarios of tests/refstrategy_util.py: per iteration, in the trainer's order (trainer.cpp:744-756): set the gradients, [set densification_info], post_backward(iter), step(iter).
Recorded: the reference's parameter defaults, every random draw it made (multinomial indices, normal deviates), the Gaussian count / learning rate / SH degree after
every iteration, and the full state (6 parameters + both Adam moments + step counts) after the iterations that refine or reset and after the last one. Run in the build container:
    python oracle/make_golden_ref_strategy.py
tests/test_gpu_strategy_reference.py replays the draws into the product's strategies on the MI355X and compares - SURVEY.md §8f row 3."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
import refstrategy_util as U  # noqa: E402


def shapes_of(st, K):
    n = st.size()
    return [(n, 3), (n, 1, 3), (n, K, 3), (n, 3), (n, 4), (n, 1)]


def run(name, sc, out):
    init = U.initial(sc)
    st = oracle.RefStrategy(sc["kind"], init["means"], init["sh0"], init["shN"], init["scaling"], init["rotation"], init["opacity"], sc["scene_scale"], sc["sh_degree"],
                            **sc["params"])
    for it in range(sc["it0"] + 1, sc["it0"] + sc["iters"] + 1):
        st.set_grads(U.grads(shapes_of(st, sc["K"]), it))
        refining = st.is_refining(it) and it < sc["params"]["stop_refine"]
        if sc["kind"] == "default" and refining:
            info = U.densification_info(st.size(), it)
            st.set_densification_info(info)
            # the rows default_strategy.cpp:165-177 will split (before duplication), for the product's fused path which draws per original row
            g = info[1] / np.maximum(info[0], 1.0)
            big = np.exp(st.get(3).reshape(-1, 3)).max(-1) > np.float32(sc["params"]["grow_scale3d"]) * np.float32(sc["scene_scale"])
            out[f"{name}/it{it}/split_idx"] = np.nonzero((g > np.float32(sc["params"]["grad_threshold"])) & big)[0]
        n_before = st.size()
        draws = st.post_backward(it, 1000 + it)      # trainer.cpp:744-756: post_backward, then step - a refinement that replaces the parameter tensors leaves
        st.step(it)                                   # them without gradients, and FusedAdam::step skips them (fused_adam.cpp:46-48)
        out[f"{name}/it{it}/draws"] = np.array([d[0] for d in draws] or [""])
        for k, (_, a) in enumerate(draws):
            out[f"{name}/it{it}/draw{k}"] = a
        s = st.state()
        for k, v in s.items():
            if it in sc["full_state"] or k in ("lr", "active_sh_degree") or k.endswith(".step"):
                out[f"{name}/it{it}/{k}"] = v
        out[f"{name}/it{it}/N"] = np.int64(st.size())
        out[f"{name}/it{it}/refining"] = np.bool_(refining)
        print(f"{name} it {it}: N {n_before} -> {st.size()}, draws {[(d[0], d[1].size) for d in draws]}, sh degree {int(s['active_sh_degree'])}, lr {s['lr'][0]:.6e}")


if __name__ == "__main__":
    assert oracle.ref_strategy_lib() is not None, "build oracle/_ref/libref_strategy.so first (make -C oracle refk refstrategy)"
    out = {}
    for k, v in oracle.ref_strategy_default_params().items():
        out[f"defaults/{k}"] = np.asarray(v)
    for name, sc in U.SCENARIOS.items():
        run(name, sc, out)
    path = os.path.join(ROOT, "tests", "golden", U.GOLD)
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KiB")
