#!/bin/bash
# Dynamic instruction counts of the stencil kernel, round-2 tree (build/r02) against this one, one chromosome per launch
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_ab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in ${CFGS:-chr1_5kb}; do
for side in r02 new; do
  if [ $side = r02 ]; then B="$R/build/r02/bench.py --batch 2"; else B="$R/bench.py --batch 2 --group 1"; fi
  i=0
  for cnt in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
    i=$((i+1))
    (cd $(dirname ${B%% *}); timeout 300 rocprofv3 --kernel-trace --pmc $cnt -d $OUT/${cfg}_${side}_p$i -o c --output-format csv -- python $B --config $cfg --pipeline-depth 1 --steps 2 --warmup 1 --cpu-rows 0 --stencil-only > $OUT/${cfg}_${side}_p$i.log 2>&1)
    echo "== $cfg $side: $cnt"; python $R/scripts/pmc_summary.py $OUT/${cfg}_${side}_p$i hpk_stencil
    find $OUT/${cfg}_${side}_p$i -name '*kernel_trace*' -delete
  done
done
done
