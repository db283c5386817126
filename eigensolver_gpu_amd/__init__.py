"""MI355X-native generalized symmetric/Hermitian-definite eigensolver.

Drop-in for the dsygvdx_gpu / zhegvdx_gpu path of NVIDIA/Eigensolver_gpu: a C-ABI shared
library of hand-written HIP kernels for gfx950 (``lib/libeigsolve_gpu.so``, sources in
``csrc/``), Fortran module shims with the reference's names (``fortran/``) and this Python
host mirror (``api``).  No CPU fallback exists: the entry points raise if the library is absent.
"""
from . import api  # noqa: F401
from .api import (EigsolveLibraryMissing, Workspace, dsygvdx_gpu, hegvdx, init_eigsolve_gpu, nvtxEndRange,  # noqa: F401
                  nvtxStartRange, zhegvdx_gpu)

__all__ = ["api", "zhegvdx_gpu", "dsygvdx_gpu", "init_eigsolve_gpu", "nvtxStartRange", "nvtxEndRange", "hegvdx",
           "Workspace", "EigsolveLibraryMissing"]
