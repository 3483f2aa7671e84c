"""petsc-dev_amd -- MI355X (gfx950) native Krylov hot path behind PETSc's Vec/Mat/KSP plugin surface.

The product is two C-ABI shared libraries built in-tree:
  csrc/libmi355x_kernels.so   hand-written HIP kernels + RCCL transport   (include/mi355x_kernels.h, mi355x_comm.h)
  host/libpetschipmi355x.so   the plugin: HIPMI355X Vec/Mat types, MPIAIJ halo scatter, PCILU (include/petschipmi355x.h)
and, because no PETSc exists on the GPU box, a third one that stands in for PETSc's object model there:
  harness/libpetscharness.so  registries, Vec/Mat/PC/KSP wrappers, KSPSolve_CG/GMRES/BCGS (include/petscmini.h); no device code
This Python package only builds and loads them (ctypes) for tests and bench.py; there is no
Python or CPU compute path, and loading fails loudly if a library is missing.
"""
from ._build import build_all, kernels_lib_path, host_lib_path, harness_lib_path, ROOT  # noqa: F401
from ._lib import load_kernels, load_host, load_harness  # noqa: F401
