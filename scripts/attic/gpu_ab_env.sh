#!/bin/bash
# A/B of an environment switch of the library: VAR=0 | default, ms per chromosome of the bench configurations
cd $GRAFT_REPO_ROOT
VAR=${VAR:-HPK_SPEC}
for cfg in ${CFGS:-chr1_10kb chr1_10kb_union chr1_5kb}; do
  for v in 0 1; do
    env $VAR=$v python bench.py --config $cfg --steps ${STEPS:-100} --warmup 10 --cpu-rows 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg $VAR=$v stencil_ms %.4f ms_per_chrom %.4f value %.3g' % (d['roofline']['kernel_ms'], d['config']['ms_per_chromosome'], d['value']), {k: round(v,3) for k,v in d['phases_ms'].items() if k in ('score','tighten')}, d['config']['candidates'], d['config']['significant_px'])"
  done
done
