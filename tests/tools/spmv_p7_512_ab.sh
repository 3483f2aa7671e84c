#!/bin/bash
# P7(512) on one GPU (bench.py --scaling strong): the shipped library against variants, several times in alternation on ONE box (boxes of the pool differ by more than the variants do)
R=${GRAFT_REPO_ROOT:-/root/repo}
V=$R/petsc-dev_amd/csrc/variants
one() { local label=$1; shift; env "$@" python3 $R/bench.py --scaling strong --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('%-34s %8.2f it/s  spmv %.4f ms = %.3f' % ('$label', d['ksp_its_per_sec'], r['avg_launch_ms'], r['frac']))
"; }
for rep in 1 2 3; do
  for v in ${VARIANTS:-default nty}; do
    if [ $v = default ]; then one "shipped library" A=1; else one "variant $v" MI355X_KERNELS_LIB=$V/libmi355x_kernels_$v.so; fi
  done
done
