#!/bin/bash
# A/B against the round-2 tree (build/r02: `git archive 08946ff | tar -x -C build/r02`, make in its csrc): plain stencil time
# per chromosome, same box, same process order.  CFGS, new side runs with --group 1 and --group ${G:-16}.
cd $GRAFT_REPO_ROOT
for cfg in ${CFGS:-chr1_10kb chr1_5kb deep_1kb}; do
  b=64; [ $cfg = deep_1kb ] && b=4; [ $cfg = chr1_5kb ] && b=16
  (cd build/r02 && python bench.py --config $cfg --steps 5 --warmup 2 --batch $b --cpu-rows 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg r02      stencil %.4f ms/chrom %.4f' % (d['roofline']['kernel_ms'], d['config']['ms_per_chromosome']))")
  for g in 1 ${G:-16}; do
    [ $g -gt $b ] && g=$b
    python bench.py --config $cfg --steps 5 --warmup 2 --batch $b --group $g --cpu-rows 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg new g=%-3d stencil %.4f ms/chrom %.4f' % ($g, d['roofline']['kernel_ms_per_chromosome'], d['config']['ms_per_chromosome']))"
  done
done
