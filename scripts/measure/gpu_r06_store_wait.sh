#!/bin/bash
# round 6: what the tile loop's first wait (s_waitcnt vmcnt(0) ahead of phase 1) waits for.  (a) phase clocks with the record stores'
# acknowledgements drained behind the batches (libhpk_clk.so = -DHPK_PHASE_CLOCK -DHPK_CLK_P1: slot 4 = wait at the top of the tile +
# gap rows/lists; the variant that drained the stores behind the batches - slot 5 - is in the round's history, not in the tree); (b) the kernel without its record stores (libhpk_abl.so = -DHPK_ABLATE, HPK_DBG_STOP=3)
cd $GRAFT_REPO_ROOT
CFGS="chr1_10kb" GRPS="8" bash scripts/measure/gpu_phase_clock.sh 2>&1 | tail -13
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print("%-14s %-6s stencil/chrom %.4f" % (sys.argv[1], sys.argv[2], r["kernel_ms_per_chromosome"]))'
for rep in 1 2; do
for st in 0 3 2; do
  HPK_LIB=$PWD/hicpeaks_amd/libhpk_abl.so HPK_DBG_STOP=$st timeout 600 python bench.py --cpu-rows 0 --no-extra --no-probes --steps 5 2>/dev/null | python -c "$P" mixed dbg$st
done
done
