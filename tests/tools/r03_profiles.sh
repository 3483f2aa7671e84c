#!/bin/bash
# Round-3 profile set (each rocprofv3 pass is its own process; PMC passes carry no trace domain besides --kernel-trace).
# Outputs under gpurun_out/r03p/; the summaries are copied into profiles/ afterwards.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
T=$R/tests/tools
stats() {  # name, command...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${name}_stats -o s -- "$@" > $O/${name}_stats.log 2>&1 || return 1
  find $O/${name}_stats -name "*kernel_stats.csv" -exec cp {} $O/${name}_kernel_stats.csv \;
  rm -rf $O/${name}_stats
}
pmc() {    # name, counters (space separated), [--stamp], command...
  local name=$1; local ctrs=$2; shift 2
  local stamp=""; if [ "$1" = "--stamp" ]; then stamp="--stamp"; shift; fi
  local dirs=""
  for c in $ctrs; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${name}_pmc_$c -o p -- "$@" > $O/${name}_pmc_$c.log 2>&1 || echo "counter $c failed for $name"
    dirs="$dirs $O/${name}_pmc_$c"
  done
  python3 $T/pmc_summary.py $stamp $dirs > $O/${name}_pmc_summary.csv
  rm -rf $dirs
}
set -x
# the contract line: JSON, kernel summary of the same command, the two PMC passes behind roofline.traffic (stamped with the kernel sources)
stats bench python3 $R/bench.py --steps 20 --warmup 3 || exit 1
pmc bench "FETCH_SIZE WRITE_SIZE" --stamp python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline
# ILU(0) application on the FEM stand-in: node-blocked sync-free solves
python3 $T/fem_ilu_apply.py > $O/ilu_fem.log 2>&1 || exit 1
stats ilu_fem python3 $T/fem_ilu_apply.py || exit 1
pmc ilu_fem "FETCH_SIZE WRITE_SIZE" python3 $T/fem_ilu_apply.py
# config 4 end to end on the FEM stand-in: GMRES(30) + ILU(0) (level order = the default, then the reference's inode order), set-up cost
python3 $T/cfg4_solve.py fem ilu > $O/cfg4_fem_ilu.log 2>&1 || exit 1
CFG4_OPTS="-pc_factor_hipmi355x_trisolve_order column" python3 $T/cfg4_solve.py fem ilu > $O/cfg4_fem_ilu_column.log 2>&1 || exit 1
for a in "p7:256 ilu" "p7:256 icc" "fem ilu" "fem icc"; do PETSC_HIPMI355X_SETUP_TIMING=1 python3 $T/factor_setup.py $a; done > $O/factor_setup.log 2>&1 || exit 1
# config 5 (BAIJ): the two forms of the row-block kernel, traffic and texture / L1 counters
stats cfg5 python3 $T/cfg5_baij.py 128 3 bs3x,bs3g,bs4x,bs4mfma || exit 1
pmc cfg5 "FETCH_SIZE WRITE_SIZE TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_FLAT_READ_WAVEFRONTS_sum SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" python3 $T/cfg5_baij.py 128 2 bs3x,bs3g
# GMRES(30) kernel budget with the registered solver, and the other solvers' iteration times
python3 $T/solver_bench.py 256 120 "-mat_hipmi355x_value_patterns 0" > $O/solver_bench_value_streamed.log 2>&1 || exit 1
python3 $T/solver_bench.py 256 120 > $O/solver_bench_value_patterns.log 2>&1 || exit 1
# the plain KSP types = the call sequences of an unchanged PETSc program, with the Vec type's noted operations (default) and without
{ echo "== plain types, -vec_hipmi355x_defer 1 (default)"; python3 $T/solver_bench.py 256 120 "-mat_hipmi355x_value_patterns 0 -ksp_cg_fused 0 -ksp_gmres_fused 0 -ksp_bcgs_fused 0" cg:jacobi,gmres:jacobi,bcgs:jacobi,cg:none;
  echo "== plain types, -vec_hipmi355x_defer 0 (every call a kernel of its own)"; python3 $T/solver_bench.py 256 120 "-mat_hipmi355x_value_patterns 0 -ksp_cg_fused 0 -ksp_gmres_fused 0 -ksp_bcgs_fused 0 -vec_hipmi355x_defer 0" cg:jacobi,gmres:jacobi,bcgs:jacobi,cg:none; } > $O/solver_bench_unchanged_program.log 2>&1 || exit 1
stats gmres python3 $T/solver_bench.py 256 120 "-mat_hipmi355x_value_patterns 0" gmres:jacobi || exit 1
# last, so that the JSON reads the freshly stamped counters
cp $O/bench_pmc_summary.csv $R/profiles/bench_pmc_summary.csv
python3 $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err || exit 1
tail -c 1500 $O/bench_n1.json
