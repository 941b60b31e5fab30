#!/bin/bash
# round 5: the whole GPU suite, then the lines the verdict asked for: structured inputs, the seam's general f64 variant
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_breadth; mkdir -p $O
[ -n "$SKIP_TESTS" ] || timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $O/pytest.txt
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; c=d["config"]; print("%-34s value %.4g stencil/chrom %.4f ms/chrom %.4f frac %.3f" % (sys.argv[1], d["value"], r["kernel_ms_per_chromosome"], c.get("ms_per_chromosome", 0), r["frac"]), {k: c.get(k) for k in ("frozen_w_by_depth", "passes_redone_in_full", "passes_rescored", "tiles", "lean_tiles", "lean_redone", "lean_explicit")})'
python bench.py --structure --cpu-rows 0 --no-extra --steps 10 2>/dev/null | tail -1 > $O/bench_structure.json; python -c "$P" mixed_structure < $O/bench_structure.json | tee -a $O/lines.txt
python bench.py --config chr1_5kb --structure --cpu-rows 0 --no-extra --steps 5 --warmup 2 2>/dev/null | tail -1 > $O/bench_5kb_structure.json; python -c "$P" chr1_5kb_structure < $O/bench_5kb_structure.json | tee -a $O/lines.txt
python bench.py --config chr1_10kb_union --structure --cpu-rows 0 --no-extra --steps 5 --warmup 2 2>/dev/null | tail -1 > $O/bench_union_structure.json; python -c "$P" union_structure < $O/bench_union_structure.json | tee -a $O/lines.txt
python bench.py --config chr1_10kb_union --balanced-f64 --cpu-rows 0 --no-extra --steps 5 --warmup 2 2>/dev/null | tail -1 > $O/bench_union_f64.json; python -c "$P" union_balanced_f64 < $O/bench_union_f64.json | tee -a $O/lines.txt
python bench.py --balanced-f64 --cpu-rows 0 --no-extra --steps 5 --warmup 2 2>/dev/null | tail -1 > $O/bench_f64.json; python -c "$P" p2w5_balanced_f64 < $O/bench_f64.json | tee -a $O/lines.txt
