#!/bin/bash
# round 5: lean tiles only where the far field is most of the band - HPK_LEAN_SHARE (least share of a band's column chunks that must be lean) 50 against 60
# (at 10 kb a band has four chunks: 50 lets the shallowest class' two far chunks through, 60 does not; 5 kb: 12 of 15, 1 kb: nearly all), same box
cd $GRAFT_REPO_ROOT
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; c=d["config"]; print("%-18s share %s  value %.4g  ms/chrom %.4f  stencil/chrom %.4f  lean %d/%d redo %d" % (sys.argv[1], sys.argv[2], d["value"], c["ms_per_chromosome"], r["kernel_ms_per_chromosome"], c["lean_tiles"], c["tiles"], c["lean_redone"]))'
for rep in 1 2; do
for sh in 50 60; do
  HPK_LEAN_SHARE=$sh python bench.py --cpu-rows 0 --no-extra --no-probes --steps 10 2>/dev/null | python -c "$P" mixed $sh
  HPK_LEAN_SHARE=$sh python bench.py --cpu-rows 0 --no-extra --no-probes --steps 10 --structure 2>/dev/null | python -c "$P" mixed_structure $sh
  HPK_LEAN_SHARE=$sh python bench.py --config chr1_10kb_union --cpu-rows 0 --no-extra --no-probes --steps 5 --warmup 2 2>/dev/null | python -c "$P" union $sh
  HPK_LEAN_SHARE=$sh python bench.py --config chr1_10kb_union --structure --cpu-rows 0 --no-extra --no-probes --steps 5 --warmup 2 2>/dev/null | python -c "$P" union_structure $sh
done
done
