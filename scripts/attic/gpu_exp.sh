#!/bin/bash
# A/B of experimental builds (hicpeaks_amd/libhpk_exp*.so) against the production library: stencil time per chromosome
cd $GRAFT_REPO_ROOT
for cfg in ${CFGS:-chr1_10kb chr1_5kb deep_1kb}; do
  b=64; [ $cfg = deep_1kb ] && b=4; [ $cfg = chr1_5kb ] && b=16
  g=${G:-16}; [ $g -gt $b ] && g=$b
  for lib in libhpk.so $(cd hicpeaks_amd; ls libhpk_exp*.so 2>/dev/null); do
    HPK_LIB=$PWD/hicpeaks_amd/$lib timeout 300 python bench.py --config $cfg --steps ${STEPS:-5} --warmup 2 --batch $b --group $g --cpu-rows 0 ${BENCH_FLAGS} 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg g=$g %-18s stencil/chrom %.4f ms/chrom %.4f frac %.3f' % ('$lib', d['roofline']['kernel_ms_per_chromosome'], d['config']['ms_per_chromosome'], d['roofline']['frac']), {k: round(v,4) for k,v in d['phases_ms'].items() if k in ('score','tighten')})"
  done
done
