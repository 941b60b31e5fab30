#!/bin/bash
# round 5: where a lean tile's time goes - kernel time and phase clocks of the lean tiles alone (ablation 10), the full tiles
# alone (11), and all (0)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/clk; mkdir -p $OUT
P="import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('stencil_ms/chrom %.4f ms/chrom %.4f frac %.3f tiles %d lean %d' % (d['roofline']['kernel_ms_per_chromosome'], c['ms_per_chromosome'], d['roofline']['frac'], c['tiles'], c['lean_tiles']))"
for args in "--depths 60" "--config chr1_5kb"; do
  for stop in ${STOPS:-0 10 11}; do
    echo "== $args dbg_stop $stop"
    HPK_DBG_STOP=$stop python bench.py $args --no-extra --steps 5 --warmup 2 --cpu-rows 0 --no-probes 2>/dev/null | python -c "$P"
    HPK_DBG_STOP=$stop HPK_LIB=$PWD/hicpeaks_amd/libhpk_clk.so HPK_CLK_DUMP=$OUT/ab.bin python bench.py $args --no-extra --steps 2 --warmup 1 --batch 8 --group 8 --distinct 8 --cpu-rows 0 --pipeline-depth 1 --no-probes > /dev/null 2>&1
    python scripts/clk_summary.py $OUT/ab.bin 8
  done
done
