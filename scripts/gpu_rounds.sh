#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 2 3 4; do for c in chr1_10kb chr1_10kb_union; do
  HPK_ROUNDS=$r python bench.py --config $c --steps 20 --warmup 3 --cpu-rows 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print($r, d['config']['name'], '%.3g' % d['value'], round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['phases_ms'].items() if k in ('tighten','host_bh','d2h')})"
done; done
