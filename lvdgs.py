"""Importable alias for the package directory ``lvd_gs-slam_amd/``.

The directory name is fixed by the project layout and is not a valid Python
identifier, so this loader registers it under the canonical module name
``lvdgs`` (``import lvdgs``, ``from lvdgs.rasterizer import GaussianRasterizer``).
"""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "lvd_gs-slam_amd")
_spec = _ilu.spec_from_file_location(
    "lvdgs", _os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = _ilu.module_from_spec(_spec)
_sys.modules["lvdgs"] = _mod
_spec.loader.exec_module(_mod)
