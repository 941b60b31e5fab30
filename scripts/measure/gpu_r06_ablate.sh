#!/bin/bash
# round 6: cumulative ablation of hpk_stencil_s in production mode (HPK_DBG_STOP: 2 = the tables only - phases 1 and 2 -, 0 = everything); stencil ms per chromosome
cd $GRAFT_REPO_ROOT
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print("%-14s %-6s stencil/chrom %.4f" % (sys.argv[1], sys.argv[2], r["kernel_ms_per_chromosome"]))'
for cfg in "" "--depths 60"; do
  for st in 0 2; do
    HPK_DBG_STOP=$st timeout 600 python bench.py --cpu-rows 0 --no-extra --no-probes --steps 5 $cfg 2>/dev/null | python -c "$P" "${cfg:-mixed}" dbg$st
  done
done
