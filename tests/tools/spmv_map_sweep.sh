#!/bin/bash
# A/B of the row-pattern SpMV's block -> XCD map and y-store policy on P7(256) (vectors in the Infinity Cache) and P7(512) (beyond it):
# one process per variant and size, the registered CG's timed solve (bench.py), SpMV and fused-update launch times from HIP events
R=${GRAFT_REPO_ROOT:-/root/repo}
V=$R/petsc-dev_amd/csrc/variants
one() {  # label, env...
  local label=$1; shift
  for mode in "--no-strong --headline-only --no-cpu-baseline" "--scaling strong --no-cpu-baseline"; do
    env "$@" python3 $R/bench.py $mode --steps 100 --warmup 5 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline'] if d['scaling'] == 'strong' else d['roofline_spmv']
        print('%-22s %-7s rows %10d  %8.2f it/s  %.5f ms/step  spmv %.5f ms = %.3f of 8 TB/s' % ('$label', d['scaling'], d['config']['rows_per_gpu'], d['ksp_its_per_sec'], d['ms_per_step'], r['avg_launch_ms'], r['frac']))
"
  done
}
one "default (CH=32)" A=1
for v in ch8 ch128 ch1024 remap0 nty nty_ch128; do one "$v" MI355X_KERNELS_LIB=$V/libmi355x_kernels_$v.so; done
