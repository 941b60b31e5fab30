#!/bin/bash
# round 5: where a wave of hpk_score spends its life (libhpk_clk.so = -DHPK_PHASE_CLOCK build): prologue / loop / epilogue ticks per wave
cd $GRAFT_REPO_ROOT
for cfg in ${CFGS:-chr1_10kb chr1_10kb_union chr1_10kb_bhfdr}; do
  echo "== $cfg"
  HPK_LIB=$PWD/hicpeaks_amd/libhpk_clk.so HPK_CLK_DUMP=/tmp/clk.bin python bench.py --config $cfg --steps 2 --warmup 2 --cpu-rows 0 --no-extra --no-probes --pipeline-depth 1 2>&1 >/dev/null | grep 'hpk_score clock' | tail -3
done
