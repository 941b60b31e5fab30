#!/bin/bash
# round 5: the scoring kernel's share - per-phase times (instrumented launches: events between the kernels) of the default, union and bhfdr lines
cd $GRAFT_REPO_ROOT
for c in ${CFGS:-chr1_10kb chr1_10kb_union chr1_10kb_bhfdr}; do
  HPK_LIB=${HPK_LIB:-$PWD/hicpeaks_amd/libhpk.so} python bench.py --config $c --steps ${STEPS:-10} --warmup 3 --cpu-rows 0 --no-extra 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-18s value %.4g ms/chrom %.4f stencil/chrom %.4f' % (d['config']['name'], d['value'], d['config']['ms_per_chromosome'], d['roofline']['kernel_ms_per_chromosome']), {k: round(v,4) for k,v in d['phases_ms'].items()})"
done
