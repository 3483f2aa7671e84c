#!/bin/bash
# build a variant of the kernel library that differs in spmv_tiled.hip only: build_tiled.sh NAME -DTL_TW=... -DTL_WAVES=...  (development aid for A/B runs on one box)
cd "$(dirname "$0")/.." || exit 1
name=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include "$@" -c spmv_tiled.hip -o variants/spmv_tiled_$name.o 2>/dev/null || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libmi355x_kernels_$name.so runtime.o vec_kernels.o spmv_csr.o variants/spmv_tiled_$name.o scatter_bsr.o trisolve.o trisolve_build.o comm_rccl.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
