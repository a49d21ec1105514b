"""CPU: tests/golden/ref_strategy.npz (the reference's own strategy layer + FusedAdam run on CPU libtorch, oracle/make_golden_ref_strategy.py) - what the GPU test
tests/test_gpu_strategy_reference.py holds the product's strategies to. Here: the file's internal consistency with the rules it is supposed to exhibit (so that a
broken generator cannot silently weaken the GPU test), and - where oracle/_ref/libref_strategy.so exists (the build container) - that it regenerates bit for bit."""
import os

import numpy as np
import pytest

import oracle
import refstrategy_util as U

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", U.GOLD))


def test_golden_trajectories_show_the_rules_they_pin():
    # MCMC (mcmc.cpp): growth by 5 % up to the cap, relocation + growth only on refining iterations, noise every iteration, shN stepped past iteration 1000
    sc = U.SCENARIOS["mcmc"]
    n_prev, steps = sc["N"], 0
    for it in range(sc["it0"] + 1, sc["it0"] + sc["iters"] + 1):
        names = [str(x) for x in GOLD[f"mcmc/it{it}/draws"] if str(x)]
        n_now = int(GOLD[f"mcmc/it{it}/N"])
        refining = it > sc["params"]["start_refine"] and it % sc["params"]["refine_every"] == 0
        assert bool(GOLD[f"mcmc/it{it}/refining"]) == refining
        assert names[-1] == "randn_like" and GOLD[f"mcmc/it{it}/draw{len(names) - 1}"].size == 3 * n_now
        assert n_now == (min(sc["params"]["max_cap"], int(1.05 * n_prev)) if refining else n_prev)
        assert ("multinomial" in names) == refining
        # trainer order: post_backward, then step. Growth replaces all six tensors, the new ones carry no gradient, FusedAdam::step skips them (fused_adam.cpp:46-48)
        steps += 0 if n_now != n_prev else 1
        assert int(GOLD[f"mcmc/it{it}/shN.step"]) == int(GOLD[f"mcmc/it{it}/means.step"]) == steps, (it, steps)
        n_prev = n_now
    assert n_prev == sc["params"]["max_cap"]
    last = sc["it0"] + sc["iters"]
    assert np.abs(GOLD[f"mcmc/it{last}/shN.exp_avg"]).max() > 0          # iteration > 1000: the higher SH degrees are stepped (fused_adam.cpp:68-70)
    assert int(GOLD[f"mcmc/it{last}/active_sh_degree"]) == 1
    # the exponential schedule touches the means group only (mcmc.cpp:494-495): lr_k = lr_0 * gamma^k in double
    lr0 = float(np.float32(1.6e-4) * np.float32(sc["scene_scale"]))
    gamma = 0.01 ** (1.0 / sc["params"]["iterations"])
    lr = lr0
    for k in range(sc["iters"]):          # the scheduler steps every iteration, also when Adam skipped everything (mcmc.cpp:386-393)
        lr *= gamma
    assert GOLD[f"mcmc/it{last}/lr"][0] == lr and GOLD[f"mcmc/it{last}/lr"][2] == float(np.float32(2.5e-3) / np.float32(20))
    # ADC (default_strategy.cpp): a split draws 2 x n_split x 3 deviates for exactly the rows recorded; opacity reset clamps and zeroes the opacity moments only
    for name in ("default", "default_revised_opacity"):
        sc = U.SCENARIOS[name]
        for it in range(1, sc["iters"] + 1):
            names = [str(x) for x in GOLD[f"{name}/it{it}/draws"] if str(x)]
            if bool(GOLD[f"{name}/it{it}/refining"]):
                assert names == ["randn"] and GOLD[f"{name}/it{it}/draw0"].size == 6 * len(GOLD[f"{name}/it{it}/split_idx"])
            else:
                assert names == []
        # iteration 5: reset_opacity replaces the opacity tensor only - clamped, moments zeroed, and (no gradient on the new tensor) not stepped; the others are
        thr = np.log(0.01 / 0.99)
        assert GOLD[f"{name}/it5/opacity"].max() <= np.float32(thr) + 1e-6 and np.abs(GOLD[f"{name}/it5/opacity.exp_avg"]).max() == 0
        assert np.abs(GOLD[f"{name}/it5/means.exp_avg"]).max() > 0
        assert int(GOLD[f"{name}/it5/means.step"]) == 4 and int(GOLD[f"{name}/it5/opacity.step"]) == 3      # iteration 3 refined: nothing stepped then
        # shN is not stepped before iteration 1000 but its step count advances (fused_adam.cpp:64-70)
        if sc["K"]:
            assert np.abs(GOLD[f"{name}/it5/shN.exp_avg"]).max() == 0 and int(GOLD[f"{name}/it5/shN.step"]) == 4


@pytest.mark.skipif(not oracle.have_ref("libref_strategy.so"), reason="oracle/_ref/libref_strategy.so not built (needs /root/reference)")
def test_golden_file_regenerates_from_the_reference_strategy_layer():
    from oracle import make_golden_ref_strategy as mg
    out = {}
    for k, v in oracle.ref_strategy_default_params().items():
        out[f"defaults/{k}"] = np.asarray(v)
    for name, sc in U.SCENARIOS.items():
        mg.run(name, sc, out)
    assert sorted(out) == sorted(GOLD.files)
    for k in out:
        a, b = np.asarray(out[k]), GOLD[k]
        assert a.shape == b.shape and np.array_equal(a, b), k
