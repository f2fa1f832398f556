"""Import alias: the package directory is named `stereo-to-multiview-cuda_amd` (not a Python identifier),
so `import stm_amd` loads it by path and re-exports it under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stereo-to-multiview-cuda_amd")
_spec = importlib.util.spec_from_file_location("stm_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["stm_amd"] = _mod
_spec.loader.exec_module(_mod)
