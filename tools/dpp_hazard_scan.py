"""Developer check for hand-written DPP inline asm (csrc/lfs_raster_rows.cuh): on gfx9 a VALU instruction that writes a VGPR must be followed by two
wait states before a DPP instruction reads that VGPR (ANY operand: the first GPU run of round 2 showed the plain operand counts too); the compiler guarantees that for its own DPP instructions but cannot see
into inline asm. Scans the disassembly of the built rasterizer object, straight-line code only (the first DPP use after a loop back-edge is
covered by dpp_ready()'s s_nop).
    python tools/dpp_hazard_scan.py            # after python lichtfeld-studio_amd/build.py"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def main():
    obj = os.path.join(ROOT, "lichtfeld-studio_amd", "build", "raster.hip.o")
    tmp = tempfile.mkdtemp()
    subprocess.check_call(["cp", obj, tmp])
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "raster.hip.o"], cwd=tmp, capture_output=True)
    code = [f for f in os.listdir(tmp) if f.endswith("gfx950")][0]
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", code], cwd=tmp, capture_output=True, text=True).stdout
    total = bad = 0
    for part in re.split(r"\n(?=[0-9a-f]+ <)", dis):
        m = re.match(r"[0-9a-f]+ <([^>]+)>", part)
        if not m:
            continue
        lines = [l.split("//")[0].strip() for l in part.splitlines()[1:]]
        lines = [l for l in lines if l]
        for i, l in enumerate(lines):
            if "_dpp" not in l:
                continue
            total += 1
            ops_ = [o.split()[0] for o in l.split(None, 1)[1].split(",")]
            src0 = set().union(*[regs(o) for o in ops_[1:]])   # every VGPR the DPP instruction reads ...
            if l.startswith(("v_fmac", "v_mac")):
                src0 |= regs(ops_[0])                          # ... including the accumulator of v_fmac
            ws, j = 0, i - 1
            while j >= 0 and ws < 2:
                p = lines[j]
                if p.startswith("s_nop"):
                    ws += int(p.split()[1]) + 1
                else:
                    if p.startswith("v_") and not p.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
                        if regs(p.split(None, 1)[1].split(",")[0].strip()) & src0:
                            bad += 1
                            print(m.group(1)[:60], "HAZARD:", p, "->", l)
                    ws += 1
                j -= 1
    print(f"{total} DPP instructions scanned, {bad} read-after-VALU-write hazards")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
