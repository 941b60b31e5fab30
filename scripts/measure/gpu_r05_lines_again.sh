#!/bin/bash
# round 5: the bench lines a late change touches, measured again into gpurun_out/$TAG like profile_round.sh (LINES: the configurations)
R=$GRAFT_REPO_ROOT; TAG=${1:-r05}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
python bench.py > $OUT/bench.json 2> $OUT/bench.err
for c in ${LINES:-chr1_10kb_union wg_10kb_union wg_5kb chr1_10kb_bhfdr chr1_5kb deep_1kb}; do
  timeout 900 python bench.py --config $c --steps 5 --warmup 1 --cpu-rows 0 2>/dev/null | tail -1 > $OUT/bench_$c.json
done
timeout 600 python bench.py --config chr1_10kb_union --structure --steps 5 --warmup 1 --cpu-rows 0 --no-extra 2>/dev/null | tail -1 > $OUT/bench_chr1_10kb_union_structure.json
timeout 600 python bench.py --config chr1_10kb_union --balanced-f64 --steps 5 --warmup 1 --cpu-rows 0 --no-extra 2>/dev/null | tail -1 > $OUT/bench_chr1_10kb_union_balanced_f64.json
timeout 600 python bench.py --depths 60 --cpu-rows 0 --no-extra 2>/dev/null | tail -1 > $OUT/bench_chr1_10kb_depth60.json
timeout 600 python bench.py --structure --cpu-rows 0 --no-extra 2>/dev/null | tail -1 > $OUT/bench_chr1_10kb_structure.json
ls $OUT | head -40
