"""In-tree builds (make + hipcc/gcc).  No JIT cache: the .so files travel with the tree."""
import os
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)


def kernels_lib_path():
    # MI355X_KERNELS_LIB selects an alternative build of the same C ABI (tuning experiments)
    return os.environ.get("MI355X_KERNELS_LIB") or os.path.join(PKG, "csrc", "libmi355x_kernels.so")


def harness_lib_path():
    # PETSC_HARNESS_LIB selects an alternative build of the harness library (AddressSanitizer build for the CPU tests)
    return os.environ.get("PETSC_HARNESS_LIB") or os.path.join(PKG, "harness", "libpetscharness.so")


def host_lib_path():
    # PETSC_HIPMI355X_HOST_LIB selects an alternative build of the host library (e.g. an AddressSanitizer build for the CPU tests)
    return os.environ.get("PETSC_HIPMI355X_HOST_LIB") or os.path.join(PKG, "host", "libpetschipmi355x.so")


def _make(directory, jobs=8):
    subprocess.run(["make", "-j%d" % jobs, "-C", directory], check=True)


def build_all(oracle=True):
    """Compile every HIP extension for gfx950, the C host library and (test infrastructure) the oracle."""
    os.environ.setdefault("PYTORCH_ROCM_ARCH", "gfx950")
    _make(os.path.join(PKG, "csrc"))
    _make(os.path.join(PKG, "harness"))     # the stand-in PETSc object model (no device code)
    _make(os.path.join(PKG, "host"))        # the plugin: HIPMI355X Vec/Mat types over the kernel library
    if oracle:
        _make(os.path.join(ROOT, "oracle"))
    if os.path.isdir(os.path.join(ROOT, "examples")):
        _make(os.path.join(ROOT, "examples"))      # standalone C user programs on the host library
