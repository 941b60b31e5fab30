#!/bin/bash
# Per-phase cycle accounting of the stencil kernels (libhpk_clk.so = -DHPK_PHASE_CLOCK build) next to the plain timing.
# VARIANTS: space-separated "name:ENV=VAL,ENV=VAL" items.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/clk; mkdir -p $OUT
for cfg in ${CFGS:-chr1_10kb chr1_5kb}; do
 for var in ${VARIANTS:-old:HPK_OLD_STENCIL=1 new:HPK_X=0}; do
  name=${var%%:*}; envs=$(echo ${var#*:} | tr ',' ' ')
  echo "== $cfg $name ($envs)"
  env $envs python bench.py --config $cfg --steps ${STEPS:-200} --warmup 20 --cpu-rows 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stencil_ms %.4f ms_per_step %.4f frac %.3f' % (d['roofline']['kernel_ms'], d['ms_per_step'], d['roofline']['frac']), {k: round(v, 4) for k, v in d['phases_ms'].items()})"
  env $envs HPK_LIB=$PWD/hicpeaks_amd/libhpk_clk.so HPK_CLK_DUMP=$OUT/${cfg}_$name.bin python bench.py --config $cfg --steps 3 --warmup 1 --cpu-rows 0 --pipeline-depth 1 > /dev/null 2>&1
  python scripts/clk_summary.py $OUT/${cfg}_$name.bin
 done
done
