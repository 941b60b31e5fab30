cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python scripts/gpu_hostcost.py chr1_10kb 32; HPK_HOST_THREADS=1 python scripts/gpu_hostcost.py chr1_10kb 32; python scripts/gpu_hostcost.py chr1_10kb_union 12
for cfg in chr1_10kb chr1_10kb_union wg_10kb_union; do
python bench.py --config $cfg --steps 5 --warmup 2 --cpu-rows 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']
print('$cfg value %.4g ms/step %.3f stencil/chrom %s frac %.3f ms/chrom %s' % (d['value'], d['ms_per_step'], r.get('kernel_ms_per_chromosome'), r['frac'], c.get('ms_per_chromosome')), {k: round(v,4) for k,v in d.get('phases_ms',{}).items() if k in ('stencil','score','tighten','host_bh')})"
done
