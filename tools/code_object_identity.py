#!/usr/bin/env python3
"""Are the gfx950 code objects of two builds of the product library the same machine code?   python tools/code_object_identity.py <old.so> <new.so>
Extracts every gfx950 code object (tools/kernel_resources.py: the clang offload bundles inside the .so), and compares, per kernel symbol, the bytes of the function in
.text and of its kernel descriptor in .rodata (llvm-objdump / llvm-readelf symbol tables). Used when a source edit that is compiled OUT by default changes the library's
source hash: counters measured on the old build (profiles/traffic.json) describe the new one exactly when every kernel is byte-identical."""
import hashlib
import subprocess
import sys
import tempfile

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
import kernel_resources as kr  # noqa: E402

READELF = kr.READELF


def kernel_bytes(path):
    out = {}
    for blob in kr.code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob); f.flush()
            secs = {}
            for line in subprocess.run([READELF, "-S", "-W", f.name], capture_output=True, text=True, check=True).stdout.splitlines():
                p = line.replace("[", " ").replace("]", " ").split()
                if len(p) >= 6 and p[0].isdigit():
                    secs[int(p[0])] = (p[1], int(p[3], 16), int(p[4], 16))          # name, address, file offset
            for line in subprocess.run([READELF, "-s", "-W", f.name], capture_output=True, text=True, check=True).stdout.splitlines():
                p = line.split()
                if len(p) == 8 and p[3] in ("FUNC", "OBJECT") and p[6].isdigit() and int(p[6]) in secs:
                    name, addr, off = secs[int(p[6])]
                    if name not in (".text", ".rodata"):
                        continue
                    value, size = int(p[1], 16), int(p[2])
                    start = off + (value - addr)
                    out[p[7]] = hashlib.sha1(blob[start:start + size]).hexdigest()
    return out


def main():
    a, b = kernel_bytes(sys.argv[1]), kernel_bytes(sys.argv[2])
    only_a, only_b = sorted(set(a) - set(b)), sorted(set(b) - set(a))
    diff = sorted(k for k in set(a) & set(b) if a[k] != b[k])
    print(f"{sys.argv[1]}: {len(a)} symbols; {sys.argv[2]}: {len(b)} symbols; identical bytes: {len(set(a) & set(b)) - len(diff)}; different: {len(diff)}; "
          f"only in the first: {len(only_a)}; only in the second: {len(only_b)}")
    for k in diff + only_a + only_b:
        print("  differs:", subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:160])
    sys.exit(1 if diff or only_a or only_b else 0)


if __name__ == "__main__":
    main()
