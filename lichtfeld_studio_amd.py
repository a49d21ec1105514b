"""Import shim: the product package lives in `lichtfeld-studio_amd/` (a directory name Python
cannot import directly); `import lichtfeld_studio_amd` loads it from there."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lichtfeld-studio_amd")
_spec = importlib.util.spec_from_file_location(
    "lichtfeld_studio_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["lichtfeld_studio_amd"] = _mod
_spec.loader.exec_module(_mod)
