#!/bin/bash
# Round profile: bench JSON + rocprofv3 kernel trace/stats + PMC (FETCH_SIZE / WRITE_SIZE in separate passes).
R=$GRAFT_REPO_ROOT
TAG=${1:-r01}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python bench.py > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o k --output-format csv -- python $R/bench.py --steps 20 --warmup 3 --cpu-rows 0 > $OUT/trace.log 2>&1
for cnt in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU"; do
  n=$(echo $cnt | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $cnt -d $OUT/pmc_$n -o c --output-format csv -- python $R/bench.py --steps 5 --warmup 1 --cpu-rows 0 > $OUT/pmc_$n.log 2>&1
done
ls -R $OUT | head -40
