cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in chr1_10kb chr1_10kb_union chr1_5kb deep_1kb; do
for cap in 64 127; do
HPK_TR_CAP=$cap python bench.py --config $cfg --steps 5 --warmup 2 --cpu-rows 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']
print('$cfg cap=$cap value %.4g stencil/chrom %.4f frac %.3f ms/chrom %.4f' % (d['value'], r.get('kernel_ms_per_chromosome'), r['frac'], c.get('ms_per_chromosome')), {k: round(v,4) for k,v in d.get('phases_ms',{}).items() if k in ('stencil','score','tighten')})"
done; done; done
