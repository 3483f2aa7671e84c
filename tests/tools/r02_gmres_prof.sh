#!/bin/bash
# per-kernel times of GMRES(30)+Jacobi on P7(256), fused and op-by-op (rocprofv3 --kernel-trace --stats)
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/gmres_prof
for mode in fused unfused; do
  opt=""; [ $mode = unfused ] && opt="-ksp_gmres_fused 0"
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/gmres_prof/$mode -o $mode -- python3 $R/tests/tools/solver_bench.py 256 120 "$opt" gmres:jacobi > $R/gpurun_out/gmres_prof/$mode.log 2>&1
  find $R/gpurun_out/gmres_prof/$mode -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/gmres_prof/${mode}_kernel_stats.csv \;
  rm -rf $R/gpurun_out/gmres_prof/$mode
done
