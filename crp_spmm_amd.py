"""Import shim: makes the hyphenated directory ``crp-spmm_amd/`` importable as
the package ``crp_spmm_amd`` (``import crp_spmm_amd.engine`` etc.)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "crp-spmm_amd")
_spec = importlib.util.spec_from_file_location("crp_spmm_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["crp_spmm_amd"] = _mod
_spec.loader.exec_module(_mod)
