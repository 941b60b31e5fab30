#!/bin/bash
# Per-phase cycle accounting of hpk_stencil and the effect of the tile order (HPK_TILE_ORDER 0 | 1).
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/clk; mkdir -p $OUT
for cfg in ${CFGS:-chr1_10kb chr1_5kb}; do
 for ord in 0 1; do
  echo "== $cfg order $ord"
  HPK_TILE_ORDER=$ord python bench.py --config $cfg --steps 200 --warmup 20 --cpu-rows 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stencil_ms %.4f ms_per_step %.4f frac %.3f' % (d['roofline']['kernel_ms'], d['ms_per_step'], d['roofline']['frac']), d['phases_ms'])"
  HPK_TILE_ORDER=$ord HPK_LIB=$PWD/hicpeaks_amd/libhpk_clk.so HPK_CLK_DUMP=$OUT/${cfg}_o$ord.bin python bench.py --config $cfg --steps 3 --warmup 1 --cpu-rows 0 --pipeline-depth 1 > /dev/null 2>&1
  python scripts/clk_summary.py $OUT/${cfg}_o$ord.bin
 done
done
