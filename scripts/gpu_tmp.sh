cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15
for d in 4 6 8 12; do
for cfg in chr1_10kb chr1_10kb_union; do
HPK_SCORE_DIV=$d timeout 600 python bench.py --config $cfg --steps 5 --warmup 2 --cpu-rows 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']
print('$cfg div=$d value %.4g stencil/chrom %.4f frac %.3f ms/chrom %.4f' % (d['value'], r.get('kernel_ms_per_chromosome'), r['frac'], c.get('ms_per_chromosome')), {k: round(v,4) for k,v in d.get('phases_ms',{}).items() if k in ('stencil','score','tighten')})"
done; done
