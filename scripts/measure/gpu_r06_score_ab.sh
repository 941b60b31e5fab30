#!/bin/bash
# round 6: A/B of hpk_score builds (LIBS) on the default workload, --depths 60, the union and bhfdr: score / whole path per chromosome, same box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; c=d["config"]; s=d.get("roofline_score") or {}; print("%-18s %-18s score/chrom %.4f stencil/chrom %.4f ms/chrom %.4f value %.4g" % (sys.argv[1], sys.argv[2], s.get("kernel_ms_per_chromosome", 0), r["kernel_ms_per_chromosome"], c.get("ms_per_chromosome", 0), d["value"]))'
for rep in $(seq 1 ${REPS:-2}); do
for lib in ${LIBS:-libhpk.so libhpk_exp1.so}; do
  HPK_LIB=$PWD/hicpeaks_amd/$lib timeout 600 python bench.py --cpu-rows 0 --no-extra --no-probes --steps ${STEPS:-10} 2>/dev/null | python -c "$P" mixed $lib
  for cfg in ${CFGS:-chr1_10kb_union chr1_10kb_bhfdr}; do
    HPK_LIB=$PWD/hicpeaks_amd/$lib timeout 600 python bench.py --config $cfg --cpu-rows 0 --no-extra --no-probes --steps 5 --warmup 2 2>/dev/null | python -c "$P" $cfg $lib
  done
done
done 2>&1 | tee gpurun_out/r06_score_ab_${TAG:-run}.txt
