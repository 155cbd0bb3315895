"""Import shim: the package directory is named ``depth-estimation_amd`` (not a valid Python
identifier), so ``import depth_estimation_amd`` resolves to this file, which loads the package
from that directory under the importable name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "depth-estimation_amd")
_spec = importlib.util.spec_from_file_location(
    "depth_estimation_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["depth_estimation_amd"] = _mod
_spec.loader.exec_module(_mod)
