"""petsc-dev_amd -- MI355X (gfx950) native Krylov hot path behind PETSc's Vec/Mat/KSP plugin surface.

The product is two C-ABI shared libraries built in-tree:
  csrc/libmi355x_kernels.so   hand-written HIP kernels + RCCL transport   (include/mi355x_kernels.h, mi355x_comm.h)
  host/libpetschipmi355x.so   C host side mirroring PETSc's Vec/Mat/KSP/PC interface (include/petschipmi355x.h)
This Python package only builds and loads them (ctypes) for tests and bench.py; there is no
Python or CPU compute path, and loading fails loudly if a library is missing.
"""
from ._build import build_all, kernels_lib_path, host_lib_path, ROOT  # noqa: F401
from ._lib import load_kernels, load_host  # noqa: F401
