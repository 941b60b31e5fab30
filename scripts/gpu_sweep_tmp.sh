cd $GRAFT_REPO_ROOT
for gm in 1 2 3 4 5; do
  HPK_SCORE_GM=$gm python bench.py --config chr1_10kb --steps 100 --warmup 10 --cpu-rows 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('GM=$gm ms_per_chrom %.4f' % d['config']['ms_per_chromosome'], {k: round(v,3) for k,v in d['phases_ms'].items() if k in ('score','tighten')})"
done
