#!/bin/bash
# bench.py on one GPU: the JSON line, the rocprofv3 kernel summary of the same command, and the two PMC passes behind roofline.traffic
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_bench
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/bench.py --steps 20 --warmup 3 > $O/stats.log 2>&1
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
find $O/stats -name "*domain_stats.csv" -exec cp {} $O/bench_domain_stats.csv \;
rm -rf $O/stats
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- python3 $R/bench.py --steps 10 --warmup 2 > $O/pmc_$c.log 2>&1
done
python3 $R/tests/tools/pmc_summary.py --stamp $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/bench_pmc_summary.csv
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
cat $O/bench_n1.json
