#!/usr/bin/env python3
"""Static resource table of the gfx950 kernels inside a built library / object file: VGPRs, SGPRs, scratch, LDS, workgroup size and the occupancy
(wavefronts per SIMD) the register allocation permits - read from the AMDGPU metadata note of every code object embedded in the file
(clang offload bundles in .hip_fatbin). No GPU needed.

    python tools/kernel_resources.py [lichtfeld-studio_amd/liblfs_gsplat.so] [--filter raster_] [--json]

Used by tests/test_kernel_resources.py (the hot kernels must not spill and must keep the occupancy DESIGN.md quotes)."""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def code_objects(path: str, arch: str = "gfx950"):
    """yields the bytes of every `arch` code object embedded in `path`"""
    data = open(path, "rb").read()
    for m in re.finditer(MAGIC, data):
        p = m.start()
        n = struct.unpack_from("<Q", data, p + 24)[0]
        o = p + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, o)
            o += 24
            triple = data[o:o + tl].decode(errors="replace")
            o += tl
            if arch in triple and size:
                yield data[p + off:p + off + size]


def occupancy(vgprs: int, agprs: int = 0, sgprs: int = 0) -> int:
    """wavefronts per SIMD the register allocation allows: unified 512-entry vector register file of gfx90a+ (granule 8), 800 SGPRs per SIMD (granule 16);
    capped at 8. (LDS and workgroup size can lower it further: not modelled here.)"""
    alloc = max(8, -(-(vgprs + agprs) // 8) * 8)
    by_v = 512 // alloc
    by_s = 800 // max(16, -(-sgprs // 16) * 16)
    return max(1, min(8, by_v, by_s))


def kernels(path: str) -> dict:
    import yaml
    out = {}
    for co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".elf") as f:
            f.write(co)
            f.flush()
            notes = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True, check=True).stdout
        a = notes.index("---")
        md = yaml.safe_load(notes[a + 3:notes.index("...", a)])
        for k in md.get("amdhsa.kernels", []):
            out[k[".name"]] = k
    names = list(out)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines() if names else []
    table = {}
    for mangled, d in zip(names, dem):
        k = out[mangled]
        short = re.sub(r"\(.*", "", d).replace("void ", "").replace("lfs::", "")
        table[short] = {"vgprs": k[".vgpr_count"], "agprs": k.get(".agpr_count", 0), "sgprs": k[".sgpr_count"], "scratch_bytes": k[".private_segment_fixed_size"],
                        "lds_bytes": k[".group_segment_fixed_size"], "max_workgroup": k[".max_flat_workgroup_size"],
                        "waves_per_simd": occupancy(k[".vgpr_count"], k.get(".agpr_count", 0), k[".sgpr_count"])}
    return table


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    path = args[0] if args else os.path.join(ROOT, "lichtfeld-studio_amd", "liblfs_gsplat.so")
    flt = sys.argv[sys.argv.index("--filter") + 1] if "--filter" in sys.argv else ""
    if flt in args:
        args.remove(flt)
    t = {k: v for k, v in kernels(path).items() if flt in k}
    if "--json" in sys.argv:
        print(json.dumps(t, indent=1, sort_keys=True))
        return
    print(f"{'kernel':88s} vgpr agpr sgpr scratch   lds  wg  waves/SIMD")
    for k, v in sorted(t.items()):
        print(f"{k[:88]:88s} {v['vgprs']:4d} {v['agprs']:4d} {v['sgprs']:4d} {v['scratch_bytes']:7d} {v['lds_bytes']:5d} {v['max_workgroup']:4d} {v['waves_per_simd']:3d}")


if __name__ == "__main__":
    main()
