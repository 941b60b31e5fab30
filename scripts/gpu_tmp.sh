cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12
for cfg in chr1_10kb chr1_10kb_union chr1_5kb wg_10kb_union; do
for sp in 0 1; do
HPK_SPEC_SURV=$sp python bench.py --config $cfg --steps 5 --warmup 2 --cpu-rows 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']
print('$cfg spec_surv=$sp value %.4g ms/step %.3f ms/chrom %s sig %s copied %s' % (d['value'], d['ms_per_step'], c.get('ms_per_chromosome'), c.get('significant_px'), c.get('records_copied_back')), {k: round(v,4) for k,v in d.get('phases_ms',{}).items() if k in ('stencil','score','tighten','host_bh')})"
done; done
