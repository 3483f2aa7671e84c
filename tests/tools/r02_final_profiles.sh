#!/bin/bash
# Round-2 profile set for the kernels added this round (each rocprofv3 pass is its own process; PMC passes carry no trace domains
# besides --kernel-trace).  Outputs under gpurun_out/r02f/, summaries copied to profiles/ afterwards.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02f
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
T=$R/tests/tools
stats() {  # name, command...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${name}_stats -o s -- "$@" > $O/${name}_stats.log 2>&1 || return 1
  find $O/${name}_stats -name "*kernel_stats.csv" -exec cp {} $O/${name}_kernel_stats.csv \;
  rm -rf $O/${name}_stats
}
pmc() {    # name, counters (space separated), command...
  local name=$1; local ctrs=$2; shift 2
  local dirs=""
  for c in $ctrs; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${name}_pmc_$c -o p -- "$@" > $O/${name}_pmc_$c.log 2>&1 || echo "counter $c failed for $name"
    dirs="$dirs $O/${name}_pmc_$c"
  done
  python3 $T/pmc_summary.py $dirs > $O/${name}_pmc_summary.csv
  rm -rf $dirs
}
set -x
# config 4 (SpMV-only on the two stand-ins), final kernels
for w in irr fem; do
  python3 $T/cfg4_spmv.py $w 30 > $O/cfg4_$w.log 2>&1 || exit 1
  stats cfg4_$w python3 $T/cfg4_spmv.py $w 20 || exit 1
  pmc cfg4_$w "FETCH_SIZE WRITE_SIZE TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_LATENCY_sum" python3 $T/cfg4_spmv.py $w 10
done
# ILU(0) apply, sync-free and level-scheduled
python3 $T/ilu_bench.py 256 syncfree > $O/ilu_syncfree.log 2>&1 || exit 1
python3 $T/ilu_bench.py 256 level > $O/ilu_level.log 2>&1 || exit 1
stats ilu_syncfree python3 $T/ilu_bench.py 256 syncfree || exit 1
pmc ilu_syncfree "FETCH_SIZE WRITE_SIZE" python3 $T/ilu_bench.py 256 syncfree
# config 5 (BAIJ), all variants interleaved + traffic
python3 $T/cfg5_baij.py 128 5 > $O/cfg5.log 2>&1 || exit 1
stats cfg5 python3 $T/cfg5_baij.py 128 3 || exit 1
pmc cfg5 "FETCH_SIZE WRITE_SIZE" python3 $T/cfg5_baij.py 128 2
# GMRES(30) kernel budget
for mode in fused unfused; do
  opt=""; [ $mode = unfused ] && opt="-ksp_gmres_fused 0"
  stats gmres_$mode python3 $T/solver_bench.py 256 120 "$opt" gmres:jacobi || exit 1
done
tail -3 $O/cfg4_irr.log $O/cfg4_fem.log $O/ilu_syncfree.log $O/ilu_level.log $O/cfg5.log
