#!/bin/bash
# bench.py against the chromosomes per library call (hpk_submit_batch): CFG, GRPS, STEPS, BATCH
cd $GRAFT_REPO_ROOT
for g in ${GRPS:-1 4 16 32}; do
  python bench.py --config ${CFG:-chr1_10kb} --steps ${STEPS:-5} --warmup 2 --batch ${BATCH:-128} --group $g --cpu-rows 0 $EXTRA 2>&1 | tail -1 | python -c "
import json,sys
l=sys.stdin.read()
try:
    d=json.loads(l)
    print('group', $g, 'ms/chrom %.4f' % d['config'].get('ms_per_chromosome', d['ms_per_step']), 'stencil/chrom %.4f' % d['roofline'].get('kernel_ms_per_chromosome', d['roofline']['kernel_ms']), 'frac %.3f' % d['roofline']['frac'], 'value %.3g' % d['value'], {k: round(v,4) for k,v in d.get('phases_ms',{}).items()})
except Exception as e:
    print('group', $g, 'FAILED', l[-600:])
"
done
