#!/bin/bash
# Per-phase cycle accounting of the stencil kernel (libhpk_clk.so = -DHPK_PHASE_CLOCK build: make -C hicpeaks_amd/csrc
# OUT=../libhpk_clk.so EXTRA=-DHPK_PHASE_CLOCK) next to the plain timing.  GROUP: chromosomes per launch.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/clk; mkdir -p $OUT
for cfg in ${CFGS:-chr1_10kb chr1_5kb}; do
 for g in ${GRPS:-1 8}; do
  echo "== $cfg group $g"
  python bench.py --config $cfg --steps ${STEPS:-5} --warmup 2 --batch ${BATCH:-64} --group $g --cpu-rows 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stencil_ms/chrom %.4f ms/chrom %.4f frac %.3f' % (d['roofline']['kernel_ms_per_chromosome'], d['config']['ms_per_chromosome'], d['roofline']['frac']), {k: round(v, 4) for k, v in d['phases_ms'].items()})"
  HPK_LIB=$PWD/hicpeaks_amd/libhpk_clk.so HPK_CLK_DUMP=$OUT/${cfg}_$g.bin python bench.py --config $cfg --steps 2 --warmup 1 --batch $g --group $g --cpu-rows 0 --pipeline-depth 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  (clock build: stencil_ms/chrom %.4f)' % d['roofline']['kernel_ms_per_chromosome'])"
  python scripts/measure/clk_summary.py $OUT/${cfg}_$g.bin $g $PERWAVE
 done
done
