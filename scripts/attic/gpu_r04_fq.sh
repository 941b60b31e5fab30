cd $GRAFT_REPO_ROOT
P='import sys,json
o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%s ms/chrom %.4f  stencil/chrom %.4f frac %.3f value %.4g" % (sys.argv[1], o["config"].get("ms_per_chromosome") or 0, o["roofline"]["kernel_ms_per_chromosome"], o["roofline"]["frac"], o["value"]), {k: round(v, 4) for k, v in o["phases_ms"].items()})'
HPK_FUSE=1 timeout 600 python bench.py --no-extra --cpu-rows 0 --depths 60 --steps 5 --warmup 2 2>/dev/null | python -c "$P" fused_d60
HPK_FUSE=1 timeout 600 python bench.py --no-extra --cpu-rows 0 --steps 5 --warmup 2 2>/dev/null | python -c "$P" fused_mixed
timeout 600 python bench.py --no-extra --cpu-rows 0 --depths 60 --steps 5 --warmup 2 2>/dev/null | python -c "$P" two_d60
