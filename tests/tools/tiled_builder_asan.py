"""The column-tiled product's host-side builder under AddressSanitizer (CPU only; the GPU pool has no sanitizer runs): random rectangular
patterns, empty rows included, through mi355x_spmv_tiled_build, the layout walked back by tests/tiled.py.
    cd petsc-dev_amd/csrc && RT=$(ls -d /opt/rocm/lib/llvm/lib/clang/*/lib/linux)
    hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include -Xarch_host -fsanitize=address,undefined -c spmv_tiled.hip -o variants/spmv_tiled_asan.o
    hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -shared-libsan -o variants/libmi355x_kernels_asan.so runtime.o vec_kernels.o spmv_csr.o \
          variants/spmv_tiled_asan.o scatter_bsr.o trisolve.o trisolve_build.o comm_rccl.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -Wl,-rpath,$RT
    ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$RT/libclang_rt.asan-x86_64.so python3 tests/tools/tiled_builder_asan.py
(plain python: pytest + torch under the preloaded runtime did not finish in 15 minutes here)"""
import ctypes as C, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
print("loading", flush=True)
k = C.CDLL(os.path.join(ROOT, 'petsc-dev_amd', 'csrc', 'variants', 'libmi355x_kernels_asan.so'))
print("loaded", flush=True)
import tiled
k.mi355x_spmv_tiled_build.argtypes=[C.c_int,C.c_int,C.c_void_p,C.c_void_p,C.c_int,C.c_void_p]
k.mi355x_spmv_tiled_debug_get.argtypes=[C.c_void_p,C.c_int,C.c_void_p,C.c_size_t,C.c_void_p]
k.mi355x_spmv_tiled_info.argtypes=[C.c_void_p]+[C.c_void_p]*5
k.mi355x_spmv_tiled_destroy.argtypes=[C.c_void_p]
rng=np.random.default_rng(5)
import scipy.sparse as sp
for trial in range(12):
    m=int(rng.integers(1,30000)); n=int(rng.integers(1,600000)); d=float(rng.uniform(0.5,40))
    nnz=int(m*d)
    rows=rng.integers(0,m,nnz); cols=np.where(rng.random(nnz)<0.7, np.clip(rows*n//max(m,1)+rng.integers(-3000,3000,nnz),0,n-1), rng.integers(0,n,nnz))
    S=sp.csr_matrix((np.ones(nnz),(rows,cols)),shape=(m,n)); S.sum_duplicates(); S.sort_indices()
    ai=S.indptr.astype(np.int32); aj=S.indices.astype(np.int32)
    smin=int(rng.choice([1,16,128,1024]))
    plan=tiled.build(k,ai,aj,n,smin)
    inf=tiled.info(k,plan)
    assert inf["staged"]+inf["remainder"]==aj.size
    if m<3000: tiled.walk(k,plan,m)
    k.mi355x_spmv_tiled_destroy(plan)
    print(trial,m,n,aj.size,smin,inf,flush=True)
print("done")
