"""ORACLE/_ref - TEST INFRASTRUCTURE ONLY. Pipe one of the reference's CUDA sources through the edits that let a host compiler take it, for the recipes of
oracle/Makefile that compile WHOLE .cu files in place (kernels AND their launchers):
    python ref_cu_prep.py <in.cu> <out>
  1. the launch syntax, which no host compiler parses:    kernel<T...><<<grid, block[, shmem[, stream]]>>>(args)  ->  cuemu::launcher(kernel<T...>, grid, block)(args)
     (dynamic shared memory is one static arena under the emulator, streams do not exist there: the two optional launch parameters are dropped);
  2. `(uint32_t)floor(x)` / `(uint32_t)ceil(x)`: 0 for a negative float on the GPU (the conversion instruction saturates), undefined in C++ -> cuemu_f2u_floor / _ceil.
Nothing else is touched; the output goes to a scratch directory that the recipe deletes."""
import re
import sys

src = open(sys.argv[1]).read()
launch = re.compile(r"([A-Za-z_][\w:]*(?:\s*<[^<>;(){}]*>)?)\s*<<<(.*?)>>>\s*\(", re.S)


def split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{<":
            depth += 1
        elif ch in ")]}>":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    out.append(cur)
    return [x.strip() for x in out]


def repl(m):
    kernel, params = m.group(1), split_top(m.group(2))
    assert len(params) >= 2, params
    return f"cuemu::launcher({' '.join(kernel.split())}, {params[0]}, {params[1]})("


out, n = launch.subn(repl, src)
out = out.replace("(uint32_t)floor(", "cuemu_f2u_floor(").replace("(uint32_t)ceil(", "cuemu_f2u_ceil(")
assert "<<<" not in out, "a launch the pattern did not catch"
open(sys.argv[2], "w").write(out)
print(f"{sys.argv[1]}: {n} launches rewritten", file=sys.stderr)
