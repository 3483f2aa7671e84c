"""Small device-array helper over the kernel library's C ABI (used by the -m gpu tests and bench.py)."""
import ctypes as C

import numpy as np

import petsc_dev_amd as pda


# The plug-in's own solvers are KSP types of their own (petsc-dev_amd/host/kspfused.c); "cg" / "gmres" / "bcgs" are the harness's
# plain restatement of the reference's KSPSolve_* (what an unchanged PETSc program drives over the same Vec/Mat ops).
PRODUCT_KSP = {"cg": "cghipmi355x", "gmres": "gmreshipmi355x", "bcgs": "bcgshipmi355x"}
_PLAIN_MARKS = {"cg": ("-ksp_cg_fused 0", "-ksp_cg_single_reduction"), "gmres": ("-ksp_gmres_fused 0",), "bcgs": ("-ksp_bcgs_fused 0",)}


def ksp_type_for(ksp, opts=""):
    """the product's solver for a method, unless the options ask for the reference's op-by-op sequence (-ksp_*_fused 0,
    -ksp_cg_single_reduction): then the plain type, which is the independent restatement the product is compared with"""
    if ksp in PRODUCT_KSP and not any(m in opts for m in _PLAIN_MARKS[ksp]):
        return PRODUCT_KSP[ksp]
    return ksp


class Dev:
    def __init__(self):
        self.k = pda.load_kernels()
        n = C.c_int()
        rc = self.k.mi355x_device_count(C.byref(n))
        if rc != 0 or n.value < 1:
            raise RuntimeError("no HIP device visible (mi355x_device_count rc=%d n=%d)" % (rc, n.value))
        self.chk(self.k.mi355x_set_device(0))
        h = C.c_void_p()
        self.chk(self.k.mi355x_handle_create(C.byref(h)))
        self.h = h
        self._live = []

    def chk(self, rc):
        if rc != 0:
            raise RuntimeError("HIP error %d: %s" % (rc, self.k.mi355x_error_string(rc).decode()))

    def alloc(self, nbytes):
        p = C.c_void_p()
        self.chk(self.k.mi355x_malloc(C.byref(p), nbytes))
        self._live.append(p)
        return p

    def put(self, a):
        a = np.ascontiguousarray(a)
        p = self.alloc(max(a.nbytes, 16))
        if a.nbytes:
            self.chk(self.k.mi355x_memcpy_h2d(self.h, p, a.ctypes.data, a.nbytes))
            self.sync()
        return p

    def get(self, p, n, dtype=np.float64):
        out = np.empty(n, dtype=dtype)
        if out.nbytes:
            self.chk(self.k.mi355x_memcpy_d2h(self.h, out.ctypes.data, p, out.nbytes))
        self.sync()
        return out

    def sync(self):
        self.chk(self.k.mi355x_handle_synchronize(self.h))

    def free(self, p):
        self.chk(self.k.mi355x_free(p))
        self._live = [q for q in self._live if q.value != p.value]

    def free_all(self):
        for p in self._live:
            self.k.mi355x_free(p)
        self._live = []

    def scalar_out(self, count=1):
        """Read `count` doubles of the handle's pinned scratch after syncing."""
        self.sync()
        addr = self.k.mi355x_handle_host_scratch(self.h)
        return np.ctypeslib.as_array((C.c_double * count).from_address(addr)).copy()

    def host_scratch(self):
        return C.c_void_p(self.k.mi355x_handle_host_scratch(self.h))

    def ptr_table(self, ptrs):
        return (C.c_void_p * len(ptrs))(*[p.value for p in ptrs])
