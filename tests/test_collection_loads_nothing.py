"""Collecting the test suite must not dlopen anything of ours: not the reference checker libraries under oracle/_ref (GPUTEST_r02: a module-level
`skipif(oracle.ref_raster_lib(full=True) is None)` loaded one into every pytest process, `-m gpu` runs included), not liboracle.so, not the product library.
Import-time skip marks use oracle.have_ref() (a file check); libraries are loaded by the fixtures / functions of the tests that use them."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_collection_of_the_gpu_run_loads_no_library():
    env = dict(os.environ, LFS_REPORT_MAPS_AFTER_COLLECTION="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/", "--collect-only", "-q", "-m", "gpu", "-p", "no:cacheprovider"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("LFS_MAPPED_AFTER_COLLECTION=")]
    assert line, r.stdout[-2000:]
    mapped = [p for p in line[-1].split("=", 1)[1].split(";") if p]
    assert mapped == [], f"collection mapped: {mapped}"
