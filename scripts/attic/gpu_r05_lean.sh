#!/bin/bash
# round 5: lean tiles - parity slice, then the judged bench lines with and without them
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_lean; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lean" 2>&1 | tail -5 | tee $O/pytest_lean.txt
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest_all.txt
P='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; c=d["config"]; print(sys.argv[1], "value %.4g ms/chrom %.4f stencil/chrom %.4f frac %.3f" % (d["value"], c.get("ms_per_chromosome", 0), r["kernel_ms_per_chromosome"], r["frac"]), {k: c.get(k) for k in ("lean_tiles", "lean_redone", "lean_explicit", "passes_redone_in_full", "passes_rescored")})'
for lean in 1 0; do
  HPK_LEAN=$lean timeout 600 python bench.py --cpu-rows 0 --no-extra 2>/dev/null | tail -1 | python -c "$P" mixed_lean$lean | tee -a $O/bench.txt
  HPK_LEAN=$lean timeout 600 python bench.py --cpu-rows 0 --no-extra --depths 60 2>/dev/null | tail -1 | python -c "$P" d60_lean$lean | tee -a $O/bench.txt
  for cfg in chr1_5kb chr1_10kb_union deep_1kb; do
    HPK_LEAN=$lean timeout 600 python bench.py --config $cfg --cpu-rows 0 --no-extra --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "$P" ${cfg}_lean$lean | tee -a $O/bench.txt
  done
done
