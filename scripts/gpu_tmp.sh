cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3
python bench.py --config chr1_10kb_bhfdr --steps 5 --warmup 2 --cpu-rows 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('bhfdr value %.4g ms/chrom %.4f sig %s copied %s' % (d['value'], c['ms_per_chromosome'], c['significant_px'], c['records_copied_back']), {k: round(v,4) for k,v in d['phases_ms'].items()})"
