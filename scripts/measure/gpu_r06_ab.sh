#!/bin/bash
# round 6: A/B of experimental builds (hicpeaks_amd/libhpk_exp*.so) against the tree's library on the judged lines, same box:
# stencil / whole path per chromosome.  TESTS=1: the GPU suite on the tree's library first.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ "${TESTS:-0}" = 1 ]; then timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5; fi
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; c=d["config"]; print("%-18s %-22s stencil/chrom %.4f ms/chrom %.4f frac %.3f value %.4g lean %d/%d redo %d" % (sys.argv[1], sys.argv[2], r["kernel_ms_per_chromosome"], c.get("ms_per_chromosome", 0), r["frac"], d["value"], c.get("lean_tiles", 0), c.get("tiles", 0), c.get("lean_redone", 0)))'
for rep in $(seq 1 ${REPS:-1}); do
for lib in ${LIBS:-libhpk.so $(cd hicpeaks_amd; ls libhpk_exp*.so 2>/dev/null)}; do
  HPK_LIB=$PWD/hicpeaks_amd/$lib timeout 600 python bench.py --cpu-rows 0 --no-extra --no-probes --steps ${STEPS:-10} 2>/dev/null | python -c "$P" mixed $lib
  for cfg in ${CFGS:-chr1_5kb chr1_10kb_union}; do
    HPK_LIB=$PWD/hicpeaks_amd/$lib timeout 600 python bench.py --config $cfg --cpu-rows 0 --no-extra --no-probes --steps 5 --warmup 2 2>/dev/null | python -c "$P" $cfg $lib
  done
done
done 2>&1 | tee gpurun_out/r06_ab_${TAG:-run}.txt
