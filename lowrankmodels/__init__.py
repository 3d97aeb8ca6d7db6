"""Import shim: the package directory is literally ``lowrankmodels.jl_amd/`` (a dot in a
directory name is not importable as such), so ``import lowrankmodels.jl_amd`` is wired
here with an explicit spec.  No code lives in this directory."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_root = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "lowrankmodels.jl_amd")
_spec = _ilu.spec_from_file_location(
    "lowrankmodels.jl_amd", _os.path.join(_root, "__init__.py"), submodule_search_locations=[_root]
)
jl_amd = _ilu.module_from_spec(_spec)
_sys.modules["lowrankmodels.jl_amd"] = jl_amd
_spec.loader.exec_module(jl_amd)
