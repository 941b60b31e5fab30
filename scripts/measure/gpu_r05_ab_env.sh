#!/bin/bash
# round 5: one library, lean kernel on / off (HPK_LEAN), same box, interleaved repeats
cd $GRAFT_REPO_ROOT
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; c=d["config"]; print("%-18s %-8s stencil/chrom %.4f ms/chrom %.4f frac %.3f lean %d/%d redo %d expl %d" % (sys.argv[1], sys.argv[2], r["kernel_ms_per_chromosome"], c.get("ms_per_chromosome", 0), r["frac"], c.get("lean_tiles", 0), c.get("tiles", 0), c.get("lean_redone", 0), c.get("lean_explicit", 0)))'
for rep in 1 2; do
for lean in 1 0; do
  HPK_LEAN=$lean timeout 600 python bench.py --cpu-rows 0 --no-extra --no-probes --steps ${STEPS:-10} 2>/dev/null | python -c "$P" mixed lean$lean
  HPK_LEAN=$lean timeout 600 python bench.py --cpu-rows 0 --no-extra --no-probes --steps ${STEPS:-10} --depths 60 2>/dev/null | python -c "$P" depth60 lean$lean
  for cfg in ${CFGS:-chr1_5kb chr1_10kb_union}; do
    HPK_LEAN=$lean timeout 600 python bench.py --config $cfg --cpu-rows 0 --no-extra --no-probes --steps 5 --warmup 2 2>/dev/null | python -c "$P" $cfg lean$lean
  done
done
done
