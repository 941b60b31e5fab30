#!/bin/bash
# HBM bytes of the fused stencil (HPK_FUSE=1) next to the two-kernel path: FETCH_SIZE / WRITE_SIZE, separate passes, one depth
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_ft
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for fuse in 1 0; do
 for cnt in FETCH_SIZE WRITE_SIZE; do
  HPK_FUSE=$fuse timeout 300 rocprofv3 --kernel-trace --pmc $cnt -d $OUT/f${fuse}_$cnt -o c --output-format csv -- python $R/bench.py --depths 60 --steps 2 --warmup 1 --cpu-rows 0 --no-extra --no-probes --batch 128 --group 8 > $OUT/f${fuse}_$cnt.log 2>&1
  echo "== fuse $fuse: $cnt (KiB per launch of 8 chromosomes)"; python $R/scripts/pmc_summary.py $OUT/f${fuse}_$cnt hpk_stencil; python $R/scripts/pmc_summary.py $OUT/f${fuse}_$cnt hpk_score
  rm -rf $OUT/f${fuse}_$cnt
 done
done
