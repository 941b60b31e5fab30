#!/bin/bash
# A/B of experimental builds (hicpeaks_amd/libhpk_exp*.so) against the production library
cd $GRAFT_REPO_ROOT
for cfg in ${CFGS:-chr1_10kb chr1_5kb deep_1kb}; do
  for lib in libhpk.so $(cd hicpeaks_amd; ls libhpk_exp*.so 2>/dev/null); do
    HPK_LIB=$PWD/hicpeaks_amd/$lib timeout 120 python bench.py --config $cfg --steps ${STEPS:-100} --warmup 10 --cpu-rows 0 ${BENCH_FLAGS} 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg $lib stencil_ms %.4f ms_per_chrom %.4f frac %.3f' % (d['roofline']['kernel_ms'], d['config']['ms_per_chromosome'], d['roofline']['frac']), {k: round(v,3) for k,v in d['phases_ms'].items() if k in ('score','tighten')})"
  done
done
