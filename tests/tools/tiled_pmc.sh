#!/bin/bash
# counters of the column-tiled SpMV on the IRR stand-in: request latency, instruction mix, texture-addresser load (one rocprofv3 pass per counter)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp CFG4_CACHE=/tmp
T=$R/tests/tools
dirs=""
for c in ${COUNTERS:-GRBM_GUI_ACTIVE TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_BUSY_avr TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LEVEL_WAVES}; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/tpmc_$c -o p -- python3 $T/tiled_probe.py irr 1024 > $O/tpmc_$c.log 2>&1 || echo "counter $c failed"
  dirs="$dirs $O/tpmc_$c"
done
python3 $T/pmc_summary.py $dirs | grep -v "fillBuffer\|copyBuffer\|gather_values" > $O/tiled_pmc_${1:-v}.csv
rm -rf $dirs
cut -c1-140 $O/tiled_pmc_${1:-v}.csv
