import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lfs():
    import lichtfeld_studio_amd  # noqa: F401  (repo-root shim -> lichtfeld-studio_amd/)
    return lichtfeld_studio_amd


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.lib()
    return oracle


def pytest_collection_finish(session):
    """LFS_REPORT_MAPS_AFTER_COLLECTION=1 (tests/test_collection_loads_nothing.py): print which checker / product libraries collecting the suite mapped
    into the process. Collecting must load none: a `-m gpu` run imports every test module, and what those imports dlopen ends up next to the HIP runtime."""
    if os.environ.get("LFS_REPORT_MAPS_AFTER_COLLECTION") != "1":
        return
    libs = set()
    with open("/proc/self/maps") as f:
        for line in f:
            p = line.rsplit(" ", 1)[-1].strip()
            if "/oracle/" in p or "liblfs" in p or "_lfs_torch_ops" in p:
                libs.add(p)
    print("\nLFS_MAPPED_AFTER_COLLECTION=" + ";".join(sorted(libs)))
