#!/bin/bash
# Round profile: bench JSON + rocprofv3 kernel trace/stats + PMC (FETCH_SIZE / WRITE_SIZE in separate passes).
# Everything lands in gpurun_out/$TAG; the summaries judged are copied to profiles/ by hand afterwards.
R=$GRAFT_REPO_ROOT
TAG=${1:-r01}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python bench.py > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
for c in chr1_10kb_union chr1_5kb deep_1kb wg_10kb_union wg_5kb; do
  timeout 900 python bench.py --config $c --steps 5 --warmup 1 --cpu-rows 0 2>/dev/null | tail -1 > $OUT/bench_$c.json
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o k --output-format csv -- python $R/bench.py --steps 20 --warmup 3 --cpu-rows 0 > $OUT/trace.log 2>&1
for cnt in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU"; do
  n=$(echo $cnt | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $cnt -d $OUT/pmc_$n -o c --output-format csv -- python $R/bench.py --steps 5 --warmup 1 --cpu-rows 0 > $OUT/pmc_$n.log 2>&1
done
cd $R
{
  echo "# rocprofv3 --pmc passes (bench.py --steps 5 --warmup 1 --cpu-rows 0, config chr1_10kb), mean per dispatch"
  for k in hpk_stencil hpk_score; do
    echo "## $k"
    for d in $OUT/pmc_*/; do python scripts/pmc_summary.py $d $k; done
  done
} > $OUT/pmc_summary.txt 2>&1
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
ls $OUT | head -40
