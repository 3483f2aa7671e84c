"""Thin ctypes front end of the two C libraries above the kernel library -- harness/libpetscharness.so (the stand-in
PETSc object model, include/petscmini.h) and host/libpetschipmi355x.so (the HIPMI355X plugin, include/petschipmi355x.h)
-- for tests and bench.py.

No arithmetic happens here: every call goes straight to the C host library, which dispatches through
the Vec/Mat function tables to the HIP kernels.  Errors raise PetscError with the C traceback text."""
import ctypes as C

import numpy as np

from ._lib import load_host, load_harness, load_kernels

vp, i32, dbl = C.c_void_p, C.c_int, C.c_double
P = C.POINTER

# name -> argtypes; all return PetscErrorCode (int)
_SIG = {
    "PetscHIPMI355XInitialize": [i32], "PetscHIPMI355XFinalize": [], "PetscHIPMI355XRegisterAll": [],
    "PetscCommCreate": [i32, i32, vp, vp, vp, vp, P(vp)], "PetscCommSetWorld": [vp], "PetscCommSetExchange": [vp, vp], "PetscCommSetDeviceComm": [vp, vp], "PetscCommSetDeviceComms": [vp, vp, vp], "PetscCommGetDeviceTransport": [vp, P(i32), P(i32), P(i32)],
    "PetscCommDestroy": [P(vp)], "PetscGetFlops": [P(dbl)],
    "PetscOptionsInsertString": [C.c_char_p], "PetscOptionsSetValue": [C.c_char_p, C.c_char_p], "PetscOptionsClear": [],
    "VecCreate": [vp, P(vp)], "VecSetSizes": [vp, i32, i32], "VecSetType": [vp, C.c_char_p], "VecSetFromOptions": [vp],
    "VecGetType": [vp, P(C.c_char_p)], "VecDuplicate": [vp, P(vp)], "VecDestroy": [P(vp)], "VecGetSize": [vp, P(i32)],
    "VecGetLocalSize": [vp, P(i32)], "VecGetOwnershipRange": [vp, P(i32), P(i32)],
    "VecSetValues": [vp, i32, vp, vp, i32], "VecAssemblyBegin": [vp], "VecAssemblyEnd": [vp],
    "VecGetArray": [vp, P(vp)], "VecRestoreArray": [vp, P(vp)], "VecGetArrayRead": [vp, P(vp)], "VecRestoreArrayRead": [vp, P(vp)],
    "VecPlaceArray": [vp, vp], "VecResetArray": [vp], "VecReplaceArray": [vp, vp], "PetscMallocFn": [C.c_size_t, P(vp)],
    "VecHIPMI355XGetArray": [vp, P(vp)], "VecHIPMI355XRestoreArray": [vp, P(vp)], "VecHIPMI355XGetArrayRead": [vp, P(vp)],
    "VecSet": [vp, dbl], "VecCopy": [vp, vp], "VecSwap": [vp, vp], "VecScale": [vp, dbl], "VecAXPY": [vp, dbl, vp],
    "VecAYPX": [vp, dbl, vp], "VecAXPBY": [vp, dbl, dbl, vp], "VecWAXPY": [vp, dbl, vp, vp],
    "VecAXPBYPCZ": [vp, dbl, dbl, dbl, vp, vp], "VecMAXPY": [vp, i32, vp, vp], "VecPointwiseMult": [vp, vp, vp],
    "VecPointwiseDivide": [vp, vp, vp], "VecReciprocal": [vp], "VecDot": [vp, vp, P(dbl)], "VecDotBegin": [vp, vp, P(dbl)], "VecDotEnd": [vp, vp, P(dbl)], "VecNormBegin": [vp, i32, P(dbl)], "VecNormEnd": [vp, i32, P(dbl)], "PetscCommSplitReductionBegin": [vp], "VecTDot": [vp, vp, P(dbl)],
    "VecMDot": [vp, i32, vp, vp], "VecMTDot": [vp, i32, vp, vp], "VecNorm": [vp, i32, P(dbl)], "VecNormalize": [vp, P(dbl)],
    "VecDotNorm2": [vp, vp, P(dbl), P(dbl)],
    "VecScatterBegin": [vp, vp, vp, i32, i32], "VecScatterEnd": [vp, vp, vp, i32, i32],
    "VecScatterGetLists": [vp, P(i32), P(vp), P(vp), P(vp), P(i32), P(vp), P(vp), P(vp), P(i32), P(vp), P(vp)],
    "MatCreate": [vp, P(vp)], "MatSetSizes": [vp, i32, i32, i32, i32], "MatSetType": [vp, C.c_char_p], "MatSetFromOptions": [vp],
    "MatGetType": [vp, P(C.c_char_p)], "MatSetUp": [vp], "MatSeqAIJSetPreallocation": [vp, i32, vp],
    "MatMPIAIJSetPreallocation": [vp, i32, vp, i32, vp], "MatSetValues": [vp, i32, vp, i32, vp, vp, i32],
    "MatAssemblyBegin": [vp, i32], "MatAssemblyEnd": [vp, i32],
    "MatCreateSeqAIJWithArrays": [vp, i32, i32, vp, vp, vp, P(vp)],
    "MatCreateMPIAIJWithArrays": [vp, i32, i32, i32, i32, vp, vp, vp, P(vp)],
    "MatCreateSeqBAIJWithArrays": [vp, i32, i32, i32, vp, vp, vp, P(vp)],
    "MatDestroy": [P(vp)], "MatGetSize": [vp, P(i32), P(i32)], "MatGetLocalSize": [vp, P(i32), P(i32)],
    "MatGetOwnershipRange": [vp, P(i32), P(i32)], "MatGetVecs": [vp, P(vp), P(vp)],
    "MatDuplicate": [vp, i32, P(vp)], "MatSetOptionsPrefix": [vp, C.c_char_p], "MatDiagonalScale": [vp, vp, vp], "MatSetValuesBatch": [vp, i32, i32, vp, vp], "MatMult": [vp, vp, vp], "MatMultAdd": [vp, vp, vp, vp], "MatMultTranspose": [vp, vp, vp],
    "MatMultTransposeAdd": [vp, vp, vp, vp], "MatGetDiagonal": [vp, vp], "MatScale": [vp, dbl], "MatZeroEntries": [vp],
    "MatSeqAIJGetArrays": [vp, P(i32), P(vp), P(vp), P(vp)], "MatMPIAIJGetSeqAIJ": [vp, P(vp), P(vp), P(vp)],
    "MatMPIAIJGetScatter": [vp, P(vp), P(vp), P(i32)],
    "MatHIPMI355XSetTiming": [vp, i32], "MatHIPMI355XGetTiming": [vp, P(i32), P(dbl)],
    "MatMPIAIJHIPMI355XSetHaloTiming": [vp, i32], "MatMPIAIJHIPMI355XGetHaloTiming": [vp, P(i32), P(dbl), P(dbl), P(dbl), P(dbl), P(i32)],
    "PetscCommDeviceAllreduceLatency": [vp, i32, P(dbl), P(dbl)],
    "MatHIPMI355XGetIndexCompression": [vp, P(i32)], "MatHIPMI355XGetRowPatterns": [vp, P(i32)], "MatHIPMI355XGetTiledInfo": [vp, P(i32), P(i32)], "MatHIPMI355XGetBlockedInfo": [vp, P(i32), P(i32)], "MatHIPMI355XGetValuePatterns": [vp, P(i32)], "MatHIPMI355XSetValuePatterns": [vp, i32], "VecHIPMI355XSetCGUpdateTiming": [i32], "VecHIPMI355XGetCGUpdateTiming": [P(i32), P(dbl)], "MatHIPMI355XGetInodeInfo": [vp, P(i32), P(i32), P(i32)], "MatHIPMI355XGetUploadCount": [vp, P(i32)], "MatHIPMI355XGetTransposeCounts": [vp, P(i32), P(i32)],
    "PetscMiniGenPoisson7": [i32, i32, i32, C.c_long, C.c_long, vp, vp, vp, P(C.c_long)],
    "PetscViewerBinaryOpen": [vp, C.c_char_p, i32, P(vp)], "PetscViewerDestroy": [P(vp)],
    "MatLoad": [vp, vp], "MatView": [vp, vp], "VecLoad": [vp, vp], "VecView": [vp, vp],
    "PCSetType": [vp, C.c_char_p], "PCILUGetLevels_HIPMI355X": [vp, P(i32), P(i32)], "PCILUGetShiftCount_HIPMI355X": [vp, P(i32)], "PCICCGetInfo_HIPMI355X": [vp, P(i32), P(i32), P(i32)], "PCILUGetSolver_HIPMI355X": [vp, P(i32), P(i32)], "PCFactorDebugSetAborted_HIPMI355X": [vp], "PCILUGetNodeInfo_HIPMI355X": [vp, P(i32), P(i32), P(i32)], "PCFactorGetMatrix": [vp, P(vp)], "MatSolve": [vp, vp, vp], "PCBJacobiGetSubKSP": [vp, P(i32), P(i32), P(vp)],
    "KSPCreate": [vp, P(vp)], "KSPSetType": [vp, C.c_char_p], "KSPSetOperators": [vp, vp, vp, i32], "KSPGetPC": [vp, P(vp)],
    "KSPSetTolerances": [vp, dbl, dbl, dbl, i32], "KSPSetNormType": [vp, i32], "KSPSetPCSide": [vp, i32], "KSPSetInitialGuessNonzero": [vp, i32], "KSPSetOptionsPrefix": [vp, C.c_char_p],
    "KSPSetFromOptions": [vp], "KSPGMRESSetRestart": [vp, i32], "KSPGMRESSetCGSRefinementType": [vp, i32],
    "KSPSetUp": [vp], "KSPSolve": [vp, vp, vp], "KSPGetIterationNumber": [vp, P(i32)], "KSPGetResidualNorm": [vp, P(dbl)],
    "KSPGetConvergedReason": [vp, P(i32)], "KSPSetResidualHistory": [vp, vp, i32, i32],
    "KSPGetResidualHistory": [vp, P(vp), P(i32)], "KSPDestroy": [P(vp)],
}

# names exported by the plugin library; everything else in _SIG lives in the harness
_PLUGIN = {n for n in _SIG if "HIPMI355X" in n} | {
    "PetscCommSetDeviceComm", "PetscCommSetDeviceComms", "PetscCommGetDeviceTransport", "PetscCommDeviceAllreduceLatency", "VecScatterBegin", "VecScatterEnd",
    "VecScatterGetLists", "MatSeqAIJGetArrays", "MatMPIAIJGetSeqAIJ", "MatMPIAIJGetScatter"}

INSERT_VALUES, ADD_VALUES = 1, 2
SCATTER_FORWARD, SCATTER_REVERSE = 0, 1
NORM_1, NORM_2, NORM_FROBENIUS, NORM_INFINITY, NORM_1_AND_2 = 0, 1, 2, 3, 4
MAT_FINAL_ASSEMBLY, MAT_FLUSH_ASSEMBLY = 0, 1
PETSC_DECIDE, PETSC_DEFAULT = -1, -2


class PetscError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("PetscErrorCode %d\n%s" % (code, text))
        self.code = code


class Lib:
    """Checked function access: lib.VecAXPY(y, a, x) raises PetscError on a non-zero return."""

    def __init__(self):
        load_kernels()
        self._h = load_harness()       # object model: VecCreate, MatMult, KSPSolve, ...
        self._p = load_host()          # plugin: type constructors, PetscHIPMI355X*, introspection
        self._h.PetscGetLastErrorMessage.restype = C.c_char_p
        self._p.PetscHIPMI355XVersion.restype = C.c_char_p
        for name, args in _SIG.items():
            fn = self._sym(name)
            fn.argtypes = args
            fn.restype = i32
        self.COMM_SELF = vp.in_dll(self._h, "PETSC_COMM_SELF")
        self.chk(self._p.PetscHIPMI355XInitialize(-1))

    def _sym(self, name):
        """a function of the plugin or of the harness (each name is exported by exactly one of them)"""
        try:
            return getattr(self._p, name) if name in _PLUGIN else getattr(self._h, name)
        except AttributeError:
            return getattr(self._h, name) if name in _PLUGIN else getattr(self._p, name)

    @property
    def COMM_WORLD(self):
        return vp.in_dll(self._h, "PETSC_COMM_WORLD")

    def chk(self, rc):
        if rc:
            raise PetscError(rc, self._h.PetscGetLastErrorMessage().decode(errors="replace"))

    def __getattr__(self, name):
        fn = self._sym(name)

        def call(*a):
            self.chk(fn(*a))
        return call

    def raw(self, name):
        return self._sym(name)


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = Lib()
    return _lib


def _ptr(a):
    return a.ctypes.data_as(vp) if a is not None else None


class Vec:
    def __init__(self, handle=None, own=True):
        self.h = handle if handle is not None else vp()
        self.own = own

    @classmethod
    def create(cls, n, N=None, comm=None, vtype=b"hipmi355x"):
        L = lib()
        v = cls()
        L.VecCreate(comm or L.COMM_WORLD, C.byref(v.h))
        L.VecSetSizes(v.h, n, PETSC_DECIDE if N is None else N)
        L.VecSetType(v.h, vtype)
        return v

    @classmethod
    def from_array(cls, a, comm=None, N=None, vtype=b"hipmi355x"):
        a = np.ascontiguousarray(a, dtype=np.float64)
        v = cls.create(a.size, N=N, comm=comm, vtype=vtype)
        v.set_array(a)
        return v

    def duplicate(self):
        w = Vec()
        lib().VecDuplicate(self.h, C.byref(w.h))
        return w

    @property
    def n(self):
        k = i32()
        lib().VecGetLocalSize(self.h, C.byref(k))
        return k.value

    def set_array(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        p = vp()
        lib().VecGetArray(self.h, C.byref(p))
        C.memmove(p, a.ctypes.data, a.nbytes)
        lib().VecRestoreArray(self.h, C.byref(p))

    def array(self):
        n = self.n
        p = vp()
        lib().VecGetArrayRead(self.h, C.byref(p))
        out = np.empty(n)
        if n:
            C.memmove(out.ctypes.data, p, out.nbytes)
        lib().VecRestoreArrayRead(self.h, C.byref(p))
        return out

    def norm(self, t=NORM_2):
        r = (dbl * 2)()
        lib().VecNorm(self.h, t, r)
        return (r[0], r[1]) if t == NORM_1_AND_2 else r[0]

    def dot(self, y):
        r = dbl()
        lib().VecDot(self.h, y.h, C.byref(r))
        return r.value

    def destroy(self):
        if self.own and self.h:
            lib().VecDestroy(C.byref(self.h))

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def vec_table(vecs):
    return (vp * len(vecs))(*[v.h.value for v in vecs])


class Mat:
    def __init__(self, handle=None, own=True):
        self.h = handle if handle is not None else vp()
        self.own = own
        self._keep = None

    @classmethod
    def from_csr(cls, ai, aj, aa, ncols=None, comm=None):
        """sequential: MatCreateSeqAIJWithArrays"""
        L = lib()
        ai = np.ascontiguousarray(ai, dtype=np.int32); aj = np.ascontiguousarray(aj, dtype=np.int32)
        aa = np.ascontiguousarray(aa, dtype=np.float64)
        m = ai.size - 1
        A = cls()
        L.MatCreateSeqAIJWithArrays(comm or L.COMM_SELF, m, m if ncols is None else ncols, _ptr(ai), _ptr(aj), _ptr(aa), C.byref(A.h))
        return A

    @classmethod
    def from_csr_mpi(cls, ai, aj, aa, n_local_cols, M=PETSC_DECIDE, N=PETSC_DECIDE, comm=None):
        """this rank's rows with global column indices: MatCreateMPIAIJWithArrays"""
        L = lib()
        ai = np.ascontiguousarray(ai, dtype=np.int32); aj = np.ascontiguousarray(aj, dtype=np.int32)
        aa = np.ascontiguousarray(aa, dtype=np.float64)
        A = cls()
        L.MatCreateMPIAIJWithArrays(comm or L.COMM_WORLD, ai.size - 1, n_local_cols, M, N, _ptr(ai), _ptr(aj), _ptr(aa), C.byref(A.h))
        return A

    @classmethod
    def from_bsr(cls, bs, ai, aj, aa, nbcols=None, comm=None):
        L = lib()
        ai = np.ascontiguousarray(ai, dtype=np.int32); aj = np.ascontiguousarray(aj, dtype=np.int32)
        aa = np.ascontiguousarray(aa, dtype=np.float64)
        mbs = ai.size - 1
        nb = mbs if nbcols is None else nbcols
        A = cls()
        L.MatCreateSeqBAIJWithArrays(comm or L.COMM_SELF, bs, mbs * bs, nb * bs, _ptr(ai), _ptr(aj), _ptr(aa), C.byref(A.h))
        return A

    def get_vecs(self):
        r, l = Vec(), Vec()
        lib().MatGetVecs(self.h, C.byref(r.h), C.byref(l.h))
        return r, l

    def mult(self, x, y):
        lib().MatMult(self.h, x.h, y.h)

    def local_size(self):
        m, n = i32(), i32()
        lib().MatGetLocalSize(self.h, C.byref(m), C.byref(n))
        return m.value, n.value

    def destroy(self):
        if self.own and self.h:
            lib().MatDestroy(C.byref(self.h))

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class KSP:
    def __init__(self, comm=None):
        self.h = vp()
        lib().KSPCreate(comm or lib().COMM_WORLD, C.byref(self.h))
        self._hist = None

    def set_operators(self, A, P_=None):
        lib().KSPSetOperators(self.h, A.h, (P_ or A).h, 0)

    def set_type(self, t):
        lib().KSPSetType(self.h, t.encode())

    def set_pc_type(self, t):
        pc = vp()
        lib().KSPGetPC(self.h, C.byref(pc))
        lib().PCSetType(pc, t.encode())

    def set_tolerances(self, rtol=PETSC_DEFAULT, abstol=PETSC_DEFAULT, dtol=PETSC_DEFAULT, max_it=PETSC_DEFAULT):
        lib().KSPSetTolerances(self.h, float(rtol), float(abstol), float(dtol), int(max_it))

    def set_from_options(self):
        lib().KSPSetFromOptions(self.h)

    def record_history(self, n=20000):
        self._hist = np.zeros(n)
        lib().KSPSetResidualHistory(self.h, _ptr(self._hist), n, 1)

    def solve(self, b, x):
        lib().KSPSolve(self.h, b.h, x.h)

    @property
    def its(self):
        k = i32()
        lib().KSPGetIterationNumber(self.h, C.byref(k))
        return k.value

    @property
    def reason(self):
        k = i32()
        lib().KSPGetConvergedReason(self.h, C.byref(k))
        return k.value

    @property
    def rnorm(self):
        r = dbl()
        lib().KSPGetResidualNorm(self.h, C.byref(r))
        return r.value

    def history(self):
        p = vp(); n = i32()
        lib().KSPGetResidualHistory(self.h, C.byref(p), C.byref(n))
        return self._hist[:n.value].copy()

    def destroy(self):
        if self.h:
            lib().KSPDestroy(C.byref(self.h))

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def gen_poisson7(nx, ny, nz, rstart=0, rend=None):
    """this rank's rows of P7(nx,ny,nz) as CSR with global columns (host arrays for MatCreate*WithArrays)"""
    L = lib()
    if rend is None:
        rend = nx * ny * nz
    nnz = C.c_long()
    L.PetscMiniGenPoisson7(nx, ny, nz, rstart, rend, None, None, None, C.byref(nnz))
    ai = np.zeros(rend - rstart + 1, dtype=np.int32)
    aj = np.zeros(nnz.value, dtype=np.int32)
    aa = np.zeros(nnz.value)
    L.PetscMiniGenPoisson7(nx, ny, nz, rstart, rend, _ptr(ai), _ptr(aj), _ptr(aa), C.byref(nnz))
    return ai, aj, aa
