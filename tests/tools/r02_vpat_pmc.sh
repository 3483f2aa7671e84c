#!/bin/bash
# counters of the value-pattern SpMV alone (tests/tools/vpat_bench.py), one rocprofv3 --pmc pass per counter
# usage: r02_vpat_pmc.sh [variant-library-name]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/vpat_pmc${1:+_$1}
mkdir -p $O
[ -n "$1" ] && export MI355X_KERNELS_LIB=$R/petsc-dev_amd/csrc/variants/libmi355x_kernels_$1.so
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr \
         TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum \
         TCP_UTCL1_TRANSLATION_HIT_sum GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES TA_FLAT_READ_WAVEFRONTS_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- python3 $R/tests/tools/vpat_bench.py > $O/pmc_$c.log 2>&1 || echo "counter $c failed"
done
python3 $R/tests/tools/pmc_summary.py $O/pmc_* > $O/summary.csv
grep -E "valpat|pat_kernel" $O/summary.csv | cut -c1-60,180-400
