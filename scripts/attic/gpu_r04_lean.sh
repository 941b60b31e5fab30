#!/bin/bash
# hpk_score with and without the lean path (option lean_scoring / HPK_LEAN): parity slice, then the bench lines
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ref_big.py -m gpu -x -q --show-capture=no 2>&1 | tail -4
P='import sys,json
o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%s ms/chrom %.4f  stencil/chrom %.4f value %.4g" % (sys.argv[1], o["config"].get("ms_per_chromosome") or 0, o["roofline"]["kernel_ms_per_chromosome"], o["value"]), {k: round(v, 4) for k, v in o["phases_ms"].items() if k in ("stencil", "score", "tighten")})'
for lean in 1 0; do
  HPK_LEAN=$lean timeout 600 python bench.py --no-extra --cpu-rows 0 --depths 60 --steps 5 --warmup 2 2>/dev/null | python -c "$P" lean${lean}_d60
  HPK_LEAN=$lean timeout 600 python bench.py --no-extra --cpu-rows 0 --steps 5 --warmup 2 2>/dev/null | python -c "$P" lean${lean}_mixed
  HPK_LEAN=$lean timeout 600 python bench.py --no-extra --cpu-rows 0 --config chr1_10kb_union --steps 5 --warmup 2 2>/dev/null | python -c "$P" lean${lean}_union
done
