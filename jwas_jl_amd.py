"""Import shim: the package directory is `jwas.jl_amd/` (a dot cannot appear in a Python module
name), so `import jwas_jl_amd` loads that directory as the package `jwas_jl_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "jwas.jl_amd")
_spec = importlib.util.spec_from_file_location(
    "jwas_jl_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["jwas_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
