"""Import shim: the package directory is `cleanrl.jl_amd/` (not a valid identifier), so `import cleanrl_jl_amd`
loads it under this name."""
import importlib.util
import os
import sys

_pkg = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cleanrl.jl_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_pkg, "__init__.py"), submodule_search_locations=[_pkg])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
