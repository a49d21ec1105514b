#!/usr/bin/env python3
"""Disassemble one gfx950 kernel out of a built object / library and count its instructions by class (VALU / SALU / LDS / VMEM / SMEM) - whole kernel, or the
innermost loops (backward branches). No GPU needed.     python tools/isa.py lichtfeld-studio_amd/build/raster.hip.o 'raster_bwd_kernel<3, 1, true, 0>' [--dump]"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_resources import code_objects  # noqa: E402

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def disassemble(path):
    """-> {demangled kernel name: [instruction lines]}"""
    out = {}
    for co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".elf") as f:
            f.write(co); f.flush()
            text = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, check=True).stdout
        cur = None
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                cur = m.group(1)
                out[cur] = []
            elif cur is not None and line.strip():
                out[cur].append(line.strip())
    names = list(out)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return {re.sub(r"\(.*", "", d).replace("void ", "").replace("lfs::", ""): out[n] for n, d in zip(names, dem)}


def klass(ins):
    op = ins.split()[0]
    if op.startswith(("v_",)):
        return "VALU"
    if op.startswith(("s_load", "s_buffer_load", "s_store")):
        return "SMEM"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    return "OTHER"


def count(lines):
    c = {}
    for l in lines:
        ins = re.sub(r"\s*//.*", "", l)
        k = klass(ins)
        c[k] = c.get(k, 0) + 1
    return c


def main():
    path, name = sys.argv[1], sys.argv[2]
    ks = disassemble(path)
    match = [k for k in ks if k == name] or [k for k in ks if name in k]
    for k in match:
        lines = ks[k]
        print(f"== {k}: {len(lines)} instructions {count(lines)}")
        # loops: a backward branch to a label; objdump prints targets as addresses in comments '// 000000001234: ...' - use the address column instead
        addrs = []
        for l in lines:
            m = re.search(r"//\s*([0-9A-Fa-f]+):", l)
            addrs.append(int(m.group(1), 16) if m else None)
        for i, l in enumerate(lines):
            m = re.match(r"s_cbranch_\w+\s+(\d+)|s_branch\s+(\d+)", l)
            if not m:
                continue
            # signed 16-bit word offset relative to the next instruction
            off = int(m.group(1) or m.group(2))
            if off >= 32768:
                off -= 65536
                if addrs[i] is None:
                    continue
                target = addrs[i] + 4 + 4 * off
                j = next((x for x in range(i, -1, -1) if addrs[x] is not None and addrs[x] <= target), 0)
                body = lines[j:i + 1]
                print(f"   loop [{j}:{i}] {len(body)} instructions {count(body)}")
        if "--dump" in sys.argv:
            print("\n".join(lines))


if __name__ == "__main__":
    main()
