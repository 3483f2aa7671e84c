#!/bin/bash
# the triangular-solve part of tests/tools/r03_profiles.sh on its own (re-run after a change of csrc/trisolve.hip)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
T=$R/tests/tools
set -x
python3 $T/fem_ilu_apply.py > $O/ilu_fem.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ilu_fem_stats -o s -- python3 $T/fem_ilu_apply.py > $O/ilu_fem_stats.log 2>&1 || exit 1
find $O/ilu_fem_stats -name "*kernel_stats.csv" -exec cp {} $O/ilu_fem_kernel_stats.csv \;
rm -rf $O/ilu_fem_stats
dirs=""
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/ilu_fem_pmc_$c -o p -- python3 $T/fem_ilu_apply.py > $O/ilu_fem_pmc_$c.log 2>&1 || echo "counter $c failed"
  dirs="$dirs $O/ilu_fem_pmc_$c"
done
python3 $T/pmc_summary.py $dirs > $O/ilu_fem_pmc_summary.csv
rm -rf $dirs
python3 $T/cfg4_solve.py fem ilu > $O/cfg4_fem_ilu.log 2>&1 || exit 1
CFG4_OPTS="-pc_factor_hipmi355x_trisolve_order column" python3 $T/cfg4_solve.py fem ilu > $O/cfg4_fem_ilu_column.log 2>&1 || exit 1
CFG4_OPTS="-pc_factor_hipmi355x_trisolve_nodes 0" python3 $T/cfg4_solve.py fem ilu > $O/cfg4_fem_ilu_rows.log 2>&1 || exit 1
python3 $T/cfg4_solve.py irr ilu > $O/cfg4_irr_ilu.log 2>&1 || exit 1
for a in "p7:256 ilu" "p7:256 icc" "fem ilu" "fem icc"; do PETSC_HIPMI355X_SETUP_TIMING=1 python3 $T/factor_setup.py $a; done > $O/factor_setup.log 2>&1 || exit 1
tail -3 $O/cfg4_fem_ilu.log
