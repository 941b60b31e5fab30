#!/bin/bash
# quick A/B of the stencil: parity slice + stencil time on two configs (+ phase clocks with CLK=1)
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
for cfg in ${CFGS:-chr1_10kb chr1_5kb chr1_10kb_union}; do
  python bench.py --config $cfg --steps 300 --warmup 30 --cpu-rows 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg stencil_ms %.4f ms_per_step %.4f frac %.3f' % (d['roofline']['kernel_ms'], d['ms_per_step'], d['roofline']['frac']))"
  if [ -n "$CLK" ]; then
    HPK_LIB=$PWD/hicpeaks_amd/libhpk_clk.so HPK_CLK_DUMP=gpurun_out/clk_$cfg.bin python bench.py --config $cfg --steps 3 --warmup 1 --cpu-rows 0 --pipeline-depth 1 > /dev/null 2>&1
    python scripts/clk_summary.py gpurun_out/clk_$cfg.bin
  fi
done
