#!/usr/bin/env python3
"""Developer A/B builds: one source recompiled with extra flags and linked against the standard objects.

    python tools/build_variant.py <name> <source.hip>[,<source2.hip>] [-DX=1 ...]   ->  lichtfeld-studio_amd/liblfs_gsplat_<name>.so
    python tools/build_variant.py <name> <source.hip> --git <rev> [...]   the source as it was at <rev> (A/B against an earlier form of a file: same headers, same flags)

Run a bench / test on it with LFS_GSPLAT_LIB=<that path> (capi.library_path). The .so files are git-ignored and travel to the GPU box."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("lfs_build", os.path.join(ROOT, "lichtfeld-studio_amd", "build.py"))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)


def main():
    name, src, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
    srcs = src.split(",")   # "raster.hip,projection_ut.hip": a switch that lives in a shared header (LFS_REC_ROT: the record's writer and its readers)
    rev = None
    if "--git" in extra:
        i = extra.index("--git"); rev = extra[i + 1]; del extra[i:i + 2]
    b.build()
    objs = [os.path.join(b.BUILD, s + ".o") for s in b.SOURCES if s not in srcs]
    for src in srcs:
        path = os.path.join(b.CSRC, src)
        if rev is not None:
            path = os.path.join(b.CSRC, f".variant_{name}_{src}")      # (next to the real one: the relative #includes resolve)
            text = subprocess.run(["git", "show", f"{rev}:lichtfeld-studio_amd/csrc/{src}"], cwd=ROOT, capture_output=True, text=True, check=True).stdout
            open(path, "w").write(text)
        vobj = os.path.join(b.BUILD, f"{src}.{name}.o")
        cmd = [b.HIPCC, *b.COMMON, *b.SOURCES[src], *extra, "-x", "hip", "-c", path, "-o", vobj]
        try:
            subprocess.run(cmd, check=True)
        finally:
            if path != os.path.join(b.CSRC, src):
                os.remove(path)
        objs.append(vobj)
    out = os.path.join(b.HERE, f"liblfs_gsplat_{name}.so")
    subprocess.run([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out], check=True)
    print(out)


if __name__ == "__main__":
    main()
