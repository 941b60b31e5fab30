#!/bin/bash
# instruction mix of the fused stencil under the ablation stops (0: all, 8: no scoring, 9: search + packing only) and of the two kernels
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_fused
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # tag, env...
  tag=$1; shift
  i=0
  for cnt in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    i=$((i+1))
    env "$@" timeout 300 rocprofv3 --kernel-trace --pmc $cnt -d $OUT/${tag}_p$i -o c --output-format csv -- python $R/bench.py --depths 60 --steps 2 --warmup 1 --cpu-rows 0 --no-extra --no-probes --batch 128 --group 8 > $OUT/${tag}_p$i.log 2>&1
    echo "== $tag: $cnt"; python $R/scripts/pmc_summary.py $OUT/${tag}_p$i hpk_stencil; python $R/scripts/pmc_summary.py $OUT/${tag}_p$i hpk_score
    rm -rf $OUT/${tag}_p$i
  done
}
for t in ${TAGS:-fused fused8 fused9 two}; do
  case $t in
    fused) run fused HPK_FUSE=1;;
    fused8) run fused8 HPK_FUSE=1 HPK_DBG_STOP=8;;
    fused9) run fused9 HPK_FUSE=1 HPK_DBG_STOP=9;;
    two) run two HPK_FUSE=0;;
  esac
done
