#!/bin/bash
# Round profile: bench JSON lines + rocprofv3 kernel trace/stats + PMC (FETCH_SIZE / WRITE_SIZE in separate passes, per
# configuration).  Everything lands in gpurun_out/$TAG; scripts/measure/collect_profiles.py copies the summaries to profiles/.
R=$GRAFT_REPO_ROOT
TAG=${1:-r06}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python bench.py > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json | cut -c1-600
for c in chr1_10kb_union chr1_5kb deep_1kb wg_10kb_union wg_5kb; do
  timeout 900 python bench.py --config $c --steps 5 --warmup 1 --cpu-rows 0 2>/dev/null | tail -1 > $OUT/bench_$c.json
done
timeout 600 python bench.py --balanced-f64 --steps 5 --warmup 1 --cpu-rows 0 2>/dev/null | tail -1 > $OUT/bench_chr1_10kb_balanced_f64.json
HPK_SPEC=0 timeout 600 python bench.py --cpu-rows 0 --no-extra 2>/dev/null | tail -1 > $OUT/bench_chr1_10kb_no_record_bound.json
# one depth (round 3's workload: 64 distinct bands, all of depth 60); the lean kernel off (every tile through hpk_stencil_s); bands
# with structure (TAD blocks, compartments, far-field patches); the seam's general f64 variant (union plan on an f64 balanced band)
timeout 600 python bench.py --depths 60 --cpu-rows 0 --no-extra 2>/dev/null | tail -1 > $OUT/bench_chr1_10kb_depth60.json
HPK_LEAN=0 timeout 600 python bench.py --cpu-rows 0 --no-extra 2>/dev/null | tail -1 > $OUT/bench_chr1_10kb_no_lean_kernel.json
HPK_LEAN=0 timeout 600 python bench.py --config chr1_5kb --steps 5 --warmup 1 --cpu-rows 0 --no-extra 2>/dev/null | tail -1 > $OUT/bench_chr1_5kb_no_lean_kernel.json
timeout 600 python bench.py --structure --cpu-rows 0 --no-extra 2>/dev/null | tail -1 > $OUT/bench_chr1_10kb_structure.json
timeout 600 python bench.py --config chr1_5kb --structure --steps 5 --warmup 1 --cpu-rows 0 --no-extra 2>/dev/null | tail -1 > $OUT/bench_chr1_5kb_structure.json
timeout 600 python bench.py --config chr1_10kb_union --structure --steps 5 --warmup 1 --cpu-rows 0 --no-extra 2>/dev/null | tail -1 > $OUT/bench_chr1_10kb_union_structure.json
timeout 600 python bench.py --config chr1_10kb_union --balanced-f64 --steps 5 --warmup 1 --cpu-rows 0 --no-extra 2>/dev/null | tail -1 > $OUT/bench_chr1_10kb_union_balanced_f64.json
timeout 600 python bench.py --config chr1_10kb_bhfdr --steps 5 --warmup 1 --cpu-rows 0 2>/dev/null | tail -1 > $OUT/bench_chr1_10kb_bhfdr.json
timeout 600 python bench.py --host-inputs --steps 3 --warmup 1 --batch 20 --cpu-rows 0 2>/dev/null | tail -1 > $OUT/bench_chr1_10kb_host_inputs.json
cd /tmp && export TMPDIR=/tmp
# kernel trace of the default command's launch shape: every stencil / scoring launch carries a whole group of 64
# chromosomes (--no-probes leaves the single-chromosome probes out), so the averages are those of bench.json's kernel_ms
PB="--steps 20 --warmup 1 --batch 128 --group 64 --cpu-rows 0 --no-probes --no-extra"
# (the trace pass runs the default command's own steps - 20 library calls of 64 chromosomes each: with two calls per step, as the
# counter passes run, every other stencil launch meets the pipeline's ramp and the average comes out 4 % above bench.json's)
PT="--steps 2 --warmup 1 --cpu-rows 0 --no-probes --no-extra"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o k --output-format csv -- python $R/bench.py $PT > $OUT/trace.log 2>&1
# HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes, launches of G chromosomes (collect_profiles.py divides by G)
for c in chr1_10kb chr1_10kb_union chr1_5kb deep_1kb; do
  G=8; [ $c = chr1_5kb ] && G=4; [ $c = deep_1kb ] && G=1
  echo $G > $OUT/pmc_group_$c.txt
  PBc="--config $c --steps 1 --warmup 1 --batch $G --group $G --cpu-rows 0 --no-probes --no-extra"
  # (+ the vector and scalar instructions of the same launches: what roofline_valu / salu_frac price)
  for cnt in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_SALU; do
    rocprofv3 --kernel-trace --pmc $cnt -d $OUT/pmc_${c}_$cnt -o c --output-format csv -- python $R/bench.py $PBc > $OUT/pmc_${c}_$cnt.log 2>&1
  done
done
PB="--steps 1 --warmup 1 --batch 8 --group 8 --cpu-rows 0 --no-probes --no-extra"
echo 8 > $OUT/pmc_group_sq.txt
for cnt in "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"; do
  nm=$(echo $cnt | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $cnt -d $OUT/pmc_sq_$nm -o c --output-format csv -- python $R/bench.py $PB > $OUT/pmc_sq_$nm.log 2>&1
done
cd $R
python scripts/measure/collect_profiles.py $TAG --no-copy
# the raw traces are tens of MB (gpurun copies back at most 64 MiB): keep the summaries and the tails of the logs
for f in $OUT/*.log; do tail -5 $f > $f.tail; rm $f; done
rm -rf $OUT/trace $OUT/pmc_*/
ls -la $OUT | head -60
