"""Generate tests/golden/ref_splat_io.npz from two host functions of the REFERENCE'S OWN src/core/splat_data.cpp compiled in place (oracle/_ref/libref_splat_io.so,
`make -C oracle refsplatio`): compute_mean_neighbor_distances (the nanoflann query behind every Gaussian's initial scale) on point sets that exercise the tree
(duplicates, a lattice, a plane, clusters, coordinates over six orders of magnitude, fewer points than neighbours), write_ply_impl (the exported splat PLY,
kept as bytes) and SplatData::init_model_from_pointcloud (point cloud -> initial Gaussians). Run in the build container, where /root/reference exists:   python oracle/make_golden_ref_splat_io.py
tests/test_loader_reference.py (oracle restatement, emulated product kernel, PLY writer) and tests/test_gpu_refk_golden.py (the GPU kernel) compare with it."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402


def point_sets():
    rng = np.random.default_rng(77)
    sets = {"gauss_1500": rng.standard_normal((1500, 3)), "n1": rng.standard_normal((1, 3)), "n2": rng.standard_normal((2, 3)), "n3": rng.standard_normal((3, 3)),
            "n11": rng.standard_normal((11, 3)), "triplicates_600": np.repeat(rng.standard_normal((200, 3)), 3, 0),
            "lattice_512": np.stack(np.meshgrid(*[np.arange(8.)] * 3), -1).reshape(-1, 3), "planar_800": np.concatenate([rng.standard_normal((800, 2)), np.zeros((800, 1))], 1),
            "clusters_1600": np.concatenate([rng.standard_normal((200, 3)) * 0.01 + c for c in rng.standard_normal((8, 3)) * 5]),
            "lognormal_1000": np.exp(rng.standard_normal((1000, 3)) * 3), "identical_40": np.ones((40, 3))}
    return {k: v.astype(np.float32) for k, v in sets.items()}


def init_cases():
    """point clouds for SplatData::init_model_from_pointcloud: (positions, colours u8, scene centre, sh degree, init_scaling, init_opacity)"""
    rng = np.random.default_rng(41)
    a = (rng.standard_normal((700, 3)) * [3.0, 1.0, 0.3]).astype(np.float32)
    b = np.repeat(rng.standard_normal((80, 3)), 2, 0).astype(np.float32)             # duplicates: the 1e-7 clamp of the neighbour distance
    return {"sfm_700_deg3_default_json": (a, rng.integers(0, 256, (700, 3)).astype(np.uint8), np.array([0.2, -0.1, 0.4], np.float32), 3, 1.0, 0.1),
            "dups_160_deg1_mcmc_json": (b, rng.integers(0, 256, (160, 3)).astype(np.uint8), np.array([0.0, 0.0, 0.0], np.float32), 1, 0.1, 0.5)}


def ply_reader_files(full_splat_bytes):
    """name -> bytes of a binary little-endian PLY: the full splat file; positions only; positions + opacity + one scale column missing rot; f_dc without f_rest"""
    hdr = lambda props, n: ("ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n + "".join(f"property float {p}\n" for p in props) + "end_header\n").encode()
    rng = np.random.default_rng(12)
    f = lambda n, k: rng.standard_normal((n, k)).astype("<f4").tobytes()
    return {"full_splat": full_splat_bytes,
            "positions_only": hdr(["x", "y", "z"], 5) + f(5, 3),
            "no_rotation_no_sh": hdr(["x", "y", "z", "opacity", "scale_0", "scale_1", "scale_2"], 7) + f(7, 7),
            "dc_without_rest_reordered": hdr(["opacity", "x", "y", "z", "f_dc_0", "f_dc_1", "f_dc_2", "rot_0", "rot_1", "rot_2", "rot_3"], 4) + f(4, 11),
            "degree1_rest": hdr(["x", "y", "z"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(9)], 6) + f(6, 15)}


def ply_case():
    rng = np.random.default_rng(3)
    N, K = 101, 9
    return dict(means=rng.standard_normal((N, 3)), sh0=rng.standard_normal((N, 1, 3)), shN=rng.standard_normal((N, K - 1, 3)), opacity=rng.standard_normal((N, 1)),
                scaling=rng.standard_normal((N, 3)), rotation=rng.standard_normal((N, 4)) * 3)


if __name__ == "__main__":
    assert oracle.ref_splat_io_lib() is not None, "build oracle/_ref/libref_splat_io.so first (make -C oracle refsplatio)"
    out = {}
    for name, pts in point_sets().items():
        out[f"knn/{name}/points"] = pts
        out[f"knn/{name}/mean_dist"] = oracle.ref_mean_neighbor_distances(pts)
    for name, (pos, col, center, deg, i_s, i_o) in init_cases().items():
        r = oracle.ref_init_model_from_pointcloud(pos, col, center, deg, i_s, i_o)
        out.update({f"init/{name}/positions": pos, f"init/{name}/colors": col, f"init/{name}/scene_center": center,
                    f"init/{name}/config": np.array([deg, i_s, i_o], np.float64)})
        # rotation (identity), opacity (one value) and shN (zeros) are constant: keep their first rows only
        out.update({f"init/{name}/out_{k}": (v if k in ("means", "sh0", "scaling", "scene_scale") else v[:2]) for k, v in r.items()})
    c = {k: v.astype(np.float32) for k, v in ply_case().items()}
    for k, v in c.items():
        out[f"ply/{k}"] = v
    with tempfile.TemporaryDirectory() as d:
        out["ply/file_bytes"] = np.frombuffer(oracle.ref_write_ply(d, "splat", c["means"], c["sh0"], c["shN"], c["opacity"], c["scaling"], c["rotation"]), np.uint8)
    # ---- the reference's PLY READER (oracle/_ref/libref_ply.so: src/loader/formats/ply.cpp) on the file its writer made above and on sparse files: which defaults
    # it fills in for columns a file lacks
    if oracle.ref_ply_lib() is not None:
        with tempfile.TemporaryDirectory() as d:
            for name, data in ply_reader_files(out["ply/file_bytes"].tobytes()).items():
                fn = os.path.join(d, name + ".ply")
                open(fn, "wb").write(data)
                r = oracle.ref_load_ply(fn)
                out[f"plyread/{name}/file_bytes"] = np.frombuffer(data, np.uint8)
                for k, v in r.items():
                    out[f"plyread/{name}/{k}"] = np.asarray(v)
    path = os.path.join(ROOT, "tests", "golden", "ref_splat_io.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KiB")
