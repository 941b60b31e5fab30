#!/bin/bash
# what a record bound wider than a band's own freeze width costs (the mixed workload lays every band out for the deepest sample's):
# bands of one depth under their own bound and under a forced bound of 8, the halo held at the plan's own in both (spec_halo = 0)
cd $GRAFT_REPO_ROOT
P='import sys,json
o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%s ms/chrom %.4f stencil %.4f score %.4f bound %s frozen %s" % (sys.argv[1], o["config"]["ms_per_chromosome"], o["roofline"]["kernel_ms_per_chromosome"], o["phases_ms"]["score"], o["config"]["record_bound_w"], o["config"]["frozen_w"]))'
for d in 15 40 60; do
  HPK_SPEC_HALO=0 python bench.py --no-extra --cpu-rows 0 --depths $d --steps 3 --warmup 2 2>/dev/null | python -c "$P" "depth $d own bound "
  HPK_SPEC_HALO=0 HPK_SPEC_FORCE=8 python bench.py --no-extra --cpu-rows 0 --depths $d --steps 3 --warmup 2 2>/dev/null | python -c "$P" "depth $d bound 8   "
done
