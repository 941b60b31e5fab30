#!/bin/bash
# A/B of two builds of the library on one box: abtmp/libhpk_old.so (HPK_LIB) against the tree's, bench configurations in turn
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in ${CFGS:-chr1_10kb chr1_10kb_balanced_f64}; do
  extra=""; c=$cfg
  if [ "$cfg" = chr1_10kb_balanced_f64 ]; then c=chr1_10kb; extra="--balanced-f64"; fi
  for lib in abtmp/libhpk_old.so hicpeaks_amd/libhpk.so; do
    HPK_LIB=$PWD/$lib python bench.py --config $c $extra --steps ${STEPS:-20} --warmup 2 --cpu-rows 0 --no-probes 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg $lib stencil_ms/chrom %.5f ms_per_chrom %.5f value %.4g frac %.3f' % (d['roofline']['kernel_ms_per_chromosome'], d['config']['ms_per_chromosome'], d['value'], d['roofline']['frac']))"
  done
done
done
