#!/bin/bash
# round 5: dynamic instruction counts of the stencil, lean tiles on / off (mixed default workload and chr1 @5 kb, launches of 8 / 4 chromosomes)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_lean; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in chr1_10kb chr1_5kb; do
  G=8; [ $cfg = chr1_5kb ] && G=4
  for lean in 0 1; do
    for cnt in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
      HPK_LEAN=$lean timeout 300 rocprofv3 --kernel-trace --pmc $cnt -d $OUT/p -o c --output-format csv -- python $R/bench.py --config $cfg --steps 1 --warmup 1 --batch $G --group $G --cpu-rows 0 --no-probes --no-extra > $OUT/log.txt 2>&1
      echo "== $cfg lean=$lean (launches of $G chromosomes)"; python $R/scripts/measure/pmc_summary.py $OUT/p hpk_stencil
      rm -rf $OUT/p
    done
  done
done
