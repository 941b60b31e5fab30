#!/bin/bash
# LDS pipe of the stencil: instructions, active cycles, bank conflicts - whole kernel and tables only (HPK_DBG_STOP=2: no batches)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_lds
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for stop in 0 2; do
 i=0
 for cnt in "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  HPK_DBG_STOP=$stop timeout 300 rocprofv3 --kernel-trace --pmc $cnt -d $OUT/s${stop}_p$i -o c --output-format csv -- python $R/bench.py --depths 60 --steps 2 --warmup 1 --cpu-rows 0 --no-extra --no-probes --batch 128 --group 8 > $OUT/s${stop}_p$i.log 2>&1
  echo "== dbg_stop $stop: $cnt"; python $R/scripts/pmc_summary.py $OUT/s${stop}_p$i hpk_stencil
  rm -rf $OUT/s${stop}_p$i
 done
done
