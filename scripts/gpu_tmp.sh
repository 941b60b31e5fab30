cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3
for c in wg_5kb wg_10kb_union; do python bench.py --config $c --steps 5 --warmup 2 --cpu-rows 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(c['name'], '%.4g' % d['value'], d['ms_per_step'])"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/wg -o k --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --config wg_5kb --steps 3 --warmup 1 --cpu-rows 0 > /tmp/wg.log 2>&1
f=$(find /tmp/wg -name "*kernel_stats.csv" | head -1); grep "hpk_ir\|hpk_etab" $f | cut -c1-140
