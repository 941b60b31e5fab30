#!/bin/bash
# phase clocks of the stencil, library before the row-per-DPP-row layout (libhpk_old*.so) against the tree's
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/clk; mkdir -p $OUT
P="import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stencil_ms/chrom %.4f ms/chrom %.4f frac %.3f' % (d['roofline']['kernel_ms_per_chromosome'], d['config']['ms_per_chromosome'], d['roofline']['frac']))"
for v in ${VARIANTS:-old new}; do
  if [ $v = old ]; then L=$PWD/hicpeaks_amd/libhpk_old.so; LC=$PWD/hicpeaks_amd/libhpk_old_clk.so; else L=$PWD/hicpeaks_amd/libhpk.so; LC=$PWD/hicpeaks_amd/libhpk_clk.so; fi
  for args in "--depths 60" "--config chr1_5kb"; do
    echo "== $v $args"
    HPK_LIB=$L python bench.py $args --no-extra --steps 5 --warmup 2 --cpu-rows 0 2>/dev/null | python -c "$P"
    HPK_LIB=$LC HPK_CLK_DUMP=$OUT/ab.bin python bench.py $args --no-extra --steps 2 --warmup 1 --batch 8 --group 8 --distinct 8 --cpu-rows 0 --pipeline-depth 1 > /dev/null 2>&1
    python scripts/clk_summary.py $OUT/ab.bin 8
  done
done
