#!/bin/bash
# round 5: phase 1 of lean / full tiles split (libhpk_clk1.so: -DHPK_PHASE_CLOCK -DHPK_CLK_P1).  Slots: wait | cells | list | prefixes+stores | phase 2 and
# barriers | batches | end barrier
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/clk; mkdir -p $OUT
for args in "--depths 60" "--config chr1_5kb"; do
  for stop in ${STOPS:-0 10 11}; do
    echo "== $args dbg_stop $stop (bound forced to ${FORCE:-6})"
    HPK_SPEC_FORCE=${FORCE:-6} HPK_DBG_STOP=$stop HPK_LIB=$PWD/hicpeaks_amd/libhpk_clk1.so HPK_CLK_DUMP=$OUT/ab.bin python bench.py $args --no-extra --steps 2 --warmup 1 --batch 8 --group 8 --distinct 8 --cpu-rows 0 --pipeline-depth 1 --no-probes 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('stencil_ms/chrom %.4f tiles/chrom %.0f lean/chrom %.0f' % (d['roofline']['kernel_ms_per_chromosome'], c['tiles']/16., c['lean_tiles']/16.))"
    python scripts/clk_summary.py $OUT/ab.bin 8 | head -9
  done
done
