#!/bin/bash
# calibrate FETCH_SIZE on the stencil's own access pattern: loads-only variant (HPK_DBG_STOP=1), stencil-only bench
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_loads
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for stop in 1 2 0; do
for cnt in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  n=$(echo $cnt | tr ' ' '_' | cut -c1-30)
  HPK_DBG_STOP=$stop rocprofv3 --kernel-trace --pmc $cnt -d $OUT/s${stop}_$n -o c --output-format csv -- python $R/bench.py --config ${CFG:-chr1_10kb} --pipeline-depth 1 --steps 4 --warmup 1 --cpu-rows 0 --stencil-only > $OUT/s${stop}_$n.log 2>&1
  echo "== stop=$stop $cnt"; python $R/scripts/pmc_summary.py $OUT/s${stop}_$n hpk_stencil
done
done
