"""Import shim: the product package lives in the directory ``petal-neighbors_amd/``
(a hyphen cannot appear in a Python module name), so ``import petal_neighbors_amd``
loads that directory as the package of the same name."""
import importlib.util as _u
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "petal-neighbors_amd")
_spec = _u.spec_from_file_location("petal_neighbors_amd", _os.path.join(_dir, "__init__.py"),
                                   submodule_search_locations=[_dir])
_mod = _u.module_from_spec(_spec)
_sys.modules["petal_neighbors_amd"] = _mod
_spec.loader.exec_module(_mod)
