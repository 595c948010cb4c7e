"""Import shim: the product package lives in the directory ``nerf-vo_amd/`` (a name Python cannot
import directly because of the hyphen).  ``import nerf_vo_amd`` loads that directory as a regular
package under the importable name ``nerf_vo_amd``."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_pkg_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "nerf-vo_amd")
_spec = _ilu.spec_from_file_location(
    "nerf_vo_amd", _os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir]
)
_module = _ilu.module_from_spec(_spec)
_sys.modules["nerf_vo_amd"] = _module
_spec.loader.exec_module(_module)
