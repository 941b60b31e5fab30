#!/bin/bash
# issue / wait counters of the scoring kernel.  (A pass with TA_* counters hung the profiler on this
# pool for 20 minutes: every pass runs under its own timeout.)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_score
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for cnt in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $cnt -d $OUT/p$i -o c --output-format csv -- python $R/bench.py --config ${CFG:-chr1_10kb} --steps 4 --warmup 1 --cpu-rows 0 > $OUT/p$i.log 2>&1
  for k in ${KERNELS:-hpk_score}; do echo "== $k"; python $R/scripts/pmc_summary.py $OUT/p$i $k; done
  find $OUT/p$i -name '*kernel_trace*' -delete
done
