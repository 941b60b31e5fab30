#!/bin/bash
# round 5: the bench lines on bands with structure again (after the survivors' pinned landing area), into gpurun_out/$TAG like profile_round.sh
R=$GRAFT_REPO_ROOT; TAG=${1:-r05}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
timeout 600 python bench.py --structure --cpu-rows 0 --no-extra 2>/dev/null | tail -1 > $OUT/bench_chr1_10kb_structure.json
timeout 600 python bench.py --config chr1_5kb --structure --steps 5 --warmup 1 --cpu-rows 0 --no-extra 2>/dev/null | tail -1 > $OUT/bench_chr1_5kb_structure.json
timeout 600 python bench.py --config chr1_10kb_union --structure --steps 5 --warmup 1 --cpu-rows 0 --no-extra 2>/dev/null | tail -1 > $OUT/bench_chr1_10kb_union_structure.json
timeout 600 python bench.py --config chr1_10kb_union --steps 5 --warmup 1 --cpu-rows 0 2>/dev/null | tail -1 > $OUT/bench_chr1_10kb_union.json
timeout 600 python bench.py --config chr1_10kb_bhfdr --steps 5 --warmup 1 --cpu-rows 0 2>/dev/null | tail -1 > $OUT/bench_chr1_10kb_bhfdr.json
for f in structure 5kb_structure union_structure; do :; done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('OUTDIR','gpurun_out/r05')+'/bench_*structure*.json')+glob.glob('gpurun_out/r05/bench_chr1_10kb_union.json')+glob.glob('gpurun_out/r05/bench_chr1_10kb_bhfdr.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), '%.4g' % d['value'], round(d['config']['ms_per_chromosome'],4), round(d['roofline']['kernel_ms_per_chromosome'],4), round(d['roofline']['frac'],3))
PY
