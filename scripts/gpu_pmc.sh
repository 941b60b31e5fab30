#!/bin/bash
# PMC passes for the stencil kernel (counters only, no tracing flags besides kernel-trace)
R=$GRAFT_REPO_ROOT
TAG=${1:-pmc}
shift
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_INSTS_SENDMSG" \
           "$@"; do
  [ -z "$set" ] && continue
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/$TAG -o p$i --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --cpu-rows 0 --stencil-only > $R/gpurun_out/$TAG.p$i.log 2>&1
  tail -2 $R/gpurun_out/$TAG.p$i.log
done
ls $R/gpurun_out/$TAG
