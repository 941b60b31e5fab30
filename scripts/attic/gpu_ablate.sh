#!/bin/bash
# stencil-only ablation: time after loads / after SAT / full
cd $GRAFT_REPO_ROOT
for s in ${ABL:-1 2 4 3 0}; do
  echo "== HPK_DBG_STOP=$s"
  HPK_DBG_STOP=$s python bench.py --config ${CFG:-chr1_10kb} --pipeline-depth 1 --steps 10 --warmup 2 --cpu-rows 0 --stencil-only 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stencil_ms',d['roofline']['kernel_ms'])"
done
