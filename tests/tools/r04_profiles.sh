#!/bin/bash
# Round-4 profile set (each rocprofv3 pass is its own process; PMC passes carry no trace domain besides --kernel-trace).
# Outputs under gpurun_out/r04p/; the summaries are copied into profiles/ afterwards.  Parts: bench | irr | solvers | sizes | cfg5 | fem (default: all)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
T=$R/tests/tools
PARTS=${*:-bench irr solvers sizes cfg5 fem}
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
has() { case " $PARTS " in *" $1 "*) return 0;; esac; return 1; }
set -x
if has bench; then
  # the contract line: kernel summary of the same command, the two PMC passes behind roofline.traffic (stamped with the kernel sources), then the JSON
  stats bench python3 $R/bench.py --steps 20 --warmup 3 --no-strong || exit 1
  pmc bench "FETCH_SIZE WRITE_SIZE" --stamp python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-strong
  cp $O/bench_pmc_summary.csv $R/profiles/bench_pmc_summary.csv
  python3 $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err || exit 1
  tail -c 1200 $O/bench_n1.json
  # the strong-scaling point alone (P7(512) on this one GPU): kernel summary and traffic
  stats bench_p7_512 python3 $R/bench.py --scaling strong --steps 20 --warmup 3 --no-cpu-baseline || exit 1
  pmc bench_p7_512 "FETCH_SIZE WRITE_SIZE" python3 $R/bench.py --scaling strong --steps 6 --warmup 2 --no-cpu-baseline
fi
if has irr; then
  # config 4's irregular stand-in: the row-block kernel (before) against the column-tiled product (after), time, traffic, L1 -> L2 requests, L2 misses
  export CFG4_CACHE=/tmp
  PETSC_OPTIONS_EXTRA="-mat_hipmi355x_tiled 0" python3 $T/cfg4_spmv.py irr 30 > $O/cfg4_irr_rowblock.log 2>&1 || exit 1
  python3 $T/cfg4_spmv.py irr 30 > $O/cfg4_irr_tiled.log 2>&1 || exit 1
  python3 $T/tiled_probe.py irr 1024 > $O/cfg4_irr_tiled_parts.log 2>&1 || exit 1
  PETSC_OPTIONS_EXTRA="-mat_hipmi355x_tiled 0" stats cfg4_irr_rowblock python3 $T/cfg4_spmv.py irr 10 || exit 1
  stats cfg4_irr_tiled python3 $T/cfg4_spmv.py irr 10 || exit 1
  PETSC_OPTIONS_EXTRA="-mat_hipmi355x_tiled 0" pmc cfg4_irr_rowblock "FETCH_SIZE WRITE_SIZE TCP_TCC_READ_REQ_sum TCC_MISS_sum TCC_HIT_sum" python3 $T/cfg4_spmv.py irr 5
  pmc cfg4_irr_tiled "FETCH_SIZE WRITE_SIZE TCP_TCC_READ_REQ_sum TCC_MISS_sum TCC_HIT_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY TA_BUSY_avr" python3 $T/cfg4_spmv.py irr 5
  python3 $T/cfg4_solve.py irr jacobi > $O/cfg4_irr_solve.log 2>&1
fi
if has solvers; then
  python3 $T/solver_bench.py 256 120 "-mat_hipmi355x_value_patterns 0" > $O/solver_bench_value_streamed.log 2>&1 || exit 1
  python3 $T/solver_bench.py 256 120 > $O/solver_bench_value_patterns.log 2>&1 || exit 1
  { echo "== plain types, -vec_hipmi355x_defer 1 (default)"; python3 $T/solver_bench.py 256 120 "-mat_hipmi355x_value_patterns 0 -ksp_cg_fused 0 -ksp_gmres_fused 0 -ksp_bcgs_fused 0" cg:jacobi,gmres:jacobi,bcgs:jacobi,cg:none;
    echo "== plain types, -vec_hipmi355x_defer 0 (every call a kernel of its own)"; python3 $T/solver_bench.py 256 120 "-mat_hipmi355x_value_patterns 0 -ksp_cg_fused 0 -ksp_gmres_fused 0 -ksp_bcgs_fused 0 -vec_hipmi355x_defer 0" cg:jacobi,gmres:jacobi,bcgs:jacobi,cg:none; } > $O/solver_bench_unchanged_program.log 2>&1 || exit 1
  stats gmres python3 $T/solver_bench.py 256 120 "-mat_hipmi355x_value_patterns 0" gmres:jacobi || exit 1
fi
if has sizes; then
  for n in 128 384; do python3 $R/bench.py --grid-n $n --headline-only --no-cpu-baseline --no-strong; done > $O/bench_other_sizes.jsonl 2> $O/bench_other_sizes.err
  python3 $R/bench.py --scaling strong --no-cpu-baseline >> $O/bench_other_sizes.jsonl 2>> $O/bench_other_sizes.err
fi
if has cfg5; then
  python3 $T/cfg5_baij.py 128 3 bs3x,bs3g,bs4x,bs4mfma > $O/cfg5.log 2>&1 || exit 1
  stats cfg5 python3 $T/cfg5_baij.py 128 3 bs3x,bs4x,bs4mfma || exit 1
fi
if has fem; then
  export CFG4_CACHE=/tmp
  python3 $T/cfg4_spmv.py fem 30 > $O/cfg4_fem_spmv.log 2>&1 || exit 1
  python3 $T/fem_ilu_apply.py > $O/ilu_fem.log 2>&1 || exit 1
  python3 $T/cfg4_solve.py fem ilu > $O/cfg4_fem_ilu.log 2>&1 || exit 1
fi
echo done
