#!/bin/bash
cd $GRAFT_REPO_ROOT
for s in ${ABL:-0 1 2 3 4}; do
  echo "== HPK_DBG_SCORE=$s"
  HPK_DBG_SCORE=$s python bench.py --pipeline-depth 1 --steps 10 --warmup 2 --cpu-rows 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('score_ms', round(d['phases_ms']['score'],3))"
done
