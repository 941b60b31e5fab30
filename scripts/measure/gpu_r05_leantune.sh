#!/bin/bash
# round 5: lean-tile thresholds (share of min_local_reads below which a chunk is lean | candidates summed cell by cell before a tile is recomputed)
cd $GRAFT_REPO_ROOT
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; c=d["config"]; print("%-16s %-22s stencil/chrom %.4f ms/chrom %.4f lean %d/%d redo %d expl %d" % (sys.argv[1], sys.argv[2], r["kernel_ms_per_chromosome"], c.get("ms_per_chromosome", 0), c.get("lean_tiles", 0), c.get("tiles", 0), c.get("lean_redone", 0), c.get("lean_explicit", 0)))'
for fm in ${COMBOS:-"50 24" "35 24" "70 24" "50 8" "50 64" "100 48" "0 0"}; do
  set -- $fm
  for cfg in ${CFGS:-chr1_10kb chr1_10kb_union}; do
    HPK_LEAN_FRAC=$1 HPK_LEAN_MAX=$2 timeout 600 python bench.py --config $cfg --cpu-rows 0 --no-extra --no-probes --steps ${STEPS:-6} --warmup 2 2>/dev/null | python -c "$P" $cfg "frac$1_max$2"
  done
done
