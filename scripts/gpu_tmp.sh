cd $GRAFT_REPO_ROOT
HPK_HALF_TILES=1 timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -5
for cfg in chr1_10kb chr1_10kb_union chr1_5kb deep_1kb; do
for h in 0 1; do
HPK_HALF_TILES=$h python bench.py --config $cfg --steps 5 --warmup 2 --cpu-rows 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']
print('$cfg half=$h value %.4g stencil/chrom %.4f frac %.3f ms/chrom %.4f' % (d['value'], r.get('kernel_ms_per_chromosome'), r['frac'], c.get('ms_per_chromosome')), {k: round(v,4) for k,v in d.get('phases_ms',{}).items() if k in ('stencil','score','tighten')})"
done; done
