#!/bin/bash
# development build of the kernel library with timestamps in the sync-free triangular solves: variants/libmi355x_kernels_tritrace.so
cd "$(dirname "$0")/.." || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include -DMI355X_TRI_TRACE -c trisolve.hip -o variants/trisolve_trace.o 2>/dev/null || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libmi355x_kernels_tritrace.so runtime.o vec_kernels.o spmv_csr.o scatter_bsr.o variants/trisolve_trace.o trisolve_build.o comm_rccl.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
