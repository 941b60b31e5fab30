#!/bin/bash
# what the stencil's waves wait on: LDS pipe (instructions, bank conflicts, busy cycles), VALU / SALU busy, wave cycles
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_lds
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in ${CFGS:-chr1_10kb}; do
i=0
for cnt in "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_IFETCH SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $cnt -d $OUT/${cfg}_p$i -o c --output-format csv -- python $R/bench.py --config $cfg --pipeline-depth 1 --steps 4 --warmup 1 --cpu-rows 0 --stencil-only > $OUT/${cfg}_p$i.log 2>&1
  echo "== $cfg: $cnt"; python $R/scripts/pmc_summary.py $OUT/${cfg}_p$i hpk_stencil
  find $OUT/${cfg}_p$i -name '*kernel_trace*' -delete
done
done
