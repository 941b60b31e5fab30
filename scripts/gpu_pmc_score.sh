#!/bin/bash
# memory-path counters of the scoring (and stencil) kernel: average VMEM/LDS latency and TA/TCP stalls
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_score
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for cnt in "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum" "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_REQ_sum TCC_ATOMIC_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $cnt -d $OUT/p$i -o c --output-format csv -- python $R/bench.py --config ${CFG:-chr1_10kb} --steps 4 --warmup 1 --cpu-rows 0 > $OUT/p$i.log 2>&1
  for k in hpk_score hpk_stencil; do echo "== $k"; python $R/scripts/pmc_summary.py $OUT/p$i $k; done
done
