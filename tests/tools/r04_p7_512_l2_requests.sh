#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04p
mkdir -p $O; cd /tmp && export TMPDIR=/tmp
dirs=""
for c in TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_LATENCY_sum; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/p7req_$c -o p -- python3 $R/bench.py --scaling strong --steps 6 --warmup 2 --no-cpu-baseline > $O/p7req_$c.log 2>&1 || echo "counter $c failed"
  dirs="$dirs $O/p7req_$c"
done
python3 $R/tests/tools/pmc_summary.py $dirs | grep -v "fillBuffer\|copyBuffer" > $O/bench_p7_512_l2_requests.csv
rm -rf $dirs
grep "pat_kernel\|CGUpdateDev\|aypx" $O/bench_p7_512_l2_requests.csv | cut -c1-60,150-400
