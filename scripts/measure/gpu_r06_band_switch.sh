cd $GRAFT_REPO_ROOT
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print("%-14s %-6s stencil/chrom %.4f" % (sys.argv[1], sys.argv[2], r["kernel_ms_per_chromosome"]))'
for rep in 1 2; do
for st in 0 6; do
  HPK_LIB=$PWD/hicpeaks_amd/libhpk_abl.so HPK_DBG_STOP=$st timeout 600 python bench.py --cpu-rows 0 --no-extra --no-probes --steps 5 2>/dev/null | python -c "$P" mixed dbg$st
done
done
