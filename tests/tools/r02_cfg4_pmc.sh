set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/r02a
rocprofv3 -L > gpurun_out/r02a/avail.txt 2>&1 || true
python3 tests/tools/cfg4_spmv.py irr 30 > gpurun_out/r02a/irr_base.log 2>&1 &&
python3 tests/tools/cfg4_spmv.py fem 30 > gpurun_out/r02a/fem_base.log 2>&1 &&
for w in irr fem; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02a/${w}_stats -o s -- python3 tests/tools/cfg4_spmv.py $w 20 > gpurun_out/r02a/${w}_stats.log 2>&1 || exit 1
  for c in FETCH_SIZE WRITE_SIZE TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/r02a/${w}_pmc_$c -o p -- python3 tests/tools/cfg4_spmv.py $w 10 > gpurun_out/r02a/${w}_pmc_$c.log 2>&1 || echo "counter $c failed"
  done
done
python3 tests/tools/pmc_summary.py gpurun_out/r02a/irr_pmc_* > gpurun_out/r02a/irr_pmc_summary.csv
python3 tests/tools/pmc_summary.py gpurun_out/r02a/fem_pmc_* > gpurun_out/r02a/fem_pmc_summary.csv
cat gpurun_out/r02a/irr_base.log gpurun_out/r02a/fem_base.log
