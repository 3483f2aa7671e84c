R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02g; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- python3 $R/tests/tools/solver_bench.py 256 62 "" gmres:jacobi > $O/pmc_$c.log 2>&1 || echo "counter $c failed"
done
python3 $R/tests/tools/pmc_summary.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/gmres_fused_pmc_summary.csv
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
