#!/bin/bash
# A/B of the column-tiled SpMV's build knobs on the IRR stand-in (one box, one process per variant)
R=${GRAFT_REPO_ROOT:-/root/repo}
export CFG4_CACHE=/tmp MI355X_TILED_DEBUG=1
V=$R/petsc-dev_amd/csrc/variants
run() { echo "== $1"; shift; timeout -k 10 300 env "$@" python3 $R/tests/tools/tiled_probe.py irr 1024 2>&1 | grep -v "^irr\|lines of x\|^row-block"; }
run "default build" A=1
for v in ${VARIANTS:-w4 w12}; do run "variant $v" MI355X_KERNELS_LIB=$V/libmi355x_kernels_$v.so; done
