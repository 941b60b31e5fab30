#!/bin/bash
# A/B of the speculative halo (HPK_SPEC_HALO=0 | 1): GPU parity tests, then stencil time per chromosome on the bench configurations
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
for cfg in ${CFGS:-chr1_10kb chr1_10kb_union chr1_5kb deep_1kb wg_10kb_union wg_5kb}; do
  for h in 0 1; do
    HPK_SPEC_HALO=$h timeout 600 python bench.py --config $cfg --steps 5 --warmup 2 --cpu-rows 0 2>&1 | tail -1 | python -c "
import json,sys
l=sys.stdin.read()
try:
    d=json.loads(l)
    r=d['roofline']; c=d['config']
    print('$cfg halo=$h value %.4g ms/step %.4f stencil/chrom %s frac %.3f ms/chrom %s redone %s' % (d['value'], d['ms_per_step'], r.get('kernel_ms_per_chromosome'), r['frac'], c.get('ms_per_chromosome'), c.get('passes_redone_in_full', c.get('redone_in_full_rank0'))), {k: round(v,4) for k,v in d.get('phases_ms',{}).items() if k in ('stencil','score','tighten')})
except Exception as e:
    print('$cfg halo=$h FAILED', l[-800:])
"
  done
done
