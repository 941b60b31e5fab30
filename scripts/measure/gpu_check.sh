#!/bin/bash
# GPU parity tests + the bench lines of the three single-GPU configurations (phases in ms)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for c in ${CFGS:-chr1_10kb chr1_10kb_union chr1_5kb}; do
  python bench.py --config $c --steps 20 --warmup 3 --cpu-rows 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['name'], '%.3g' % d['value'], round(d['ms_per_step'],3), 'frac', round(d['roofline']['frac'],3), {k: round(v,3) for k,v in d['phases_ms'].items()})"
done
