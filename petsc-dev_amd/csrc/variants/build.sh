#!/bin/bash
# build a variant of the kernel library that differs in spmv_csr.hip only: build.sh NAME -DFOO=1 -DBAR=2   (development aid for A/B runs on one box)
cd "$(dirname "$0")/.." || exit 1
name=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include "$@" -c spmv_csr.hip -o variants/spmv_csr_$name.o 2>/dev/null || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libmi355x_kernels_$name.so runtime.o vec_kernels.o variants/spmv_csr_$name.o spmv_tiled.o scatter_bsr.o trisolve.o trisolve_build.o comm_rccl.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
