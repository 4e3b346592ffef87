"""Import shim: the product package lives in the directory `gusto.jl_amd/` (a dot is not importable), so this
module loads it under the name `gusto_jl_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gusto.jl_amd")
_spec = importlib.util.spec_from_file_location("gusto_jl_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["gusto_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
