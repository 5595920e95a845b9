"""crp-spmm_amd -- MI355X-native CRP-SpMM hot path (host-side Python view).

The directory name carries a hyphen (the layout the build contract names), so
the package is imported as ``crp_spmm_amd`` through the loader shim
``crp_spmm_amd.py`` at the repository root.

Contents: ``csrc/`` (HIP kernels + the C ABI declared in ``include/``),
``_lib`` (ctypes binding, fails loudly when the library is missing),
``engine`` (RpSpmm / Para2dSpmm, mirrors of rp_spmm_* / para2d_spmm_*),
``comm`` (crp_comm_t over torch.distributed: gloo on CPU, nccl == RCCL on GPU),
``planner`` / ``mmio`` (host planner and Matrix Market ingest),
``hip`` (device-level kernel wrappers), ``gen`` (synthetic inputs).
"""
from . import _lib  # noqa: F401
from ._lib import CrpHipError, CrpLibraryError, load  # noqa: F401

__version__ = "0.1"
