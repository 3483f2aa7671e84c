#!/bin/bash
# the adaptive run length of the row-pattern SpMV's block -> XCD map against fixed ones, same box, P7(256) and P7(512)
R=${GRAFT_REPO_ROOT:-/root/repo}
one() {
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
one "adaptive" A=1
for c in 32 64 128 256; do one "MI355X_SPMV_CH=$c" MI355X_SPMV_CH=$c; done
one "adaptive again" A=1
