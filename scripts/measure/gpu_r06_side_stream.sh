#!/bin/bash
# round 6: hpk_etab_edge + hpk_band_class beside the previous batch's hpk_score (the lane's side stream: today) against the
# same kernels in line on the compute stream (HPK_SIDE_SERIAL=1): wall per chromosome un-profiled, kernel averages from rocprofv3's trace
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/side; mkdir -p $OUT
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; c=d["config"]; s=d.get("roofline_score",{}); print("%-34s ms/chrom %.4f stencil %.4f score %.4f value %.4g" % (sys.argv[1], c["ms_per_chromosome"], r["kernel_ms_per_chromosome"], s.get("kernel_ms_per_chromosome",0), d["value"]))'
for rep in 1 2; do
for ser in 0 1; do
  for dp in "" "--depths 60"; do
    HPK_SIDE_SERIAL=$ser timeout 600 python bench.py --cpu-rows 0 --no-extra --no-probes --steps 10 $dp 2>/dev/null | python -c "$P" "side_serial=$ser ${dp:-mixed}"
  done
done
done
cd /tmp && export TMPDIR=/tmp
for ser in 0 1; do
  for dp in "" "--depths 60"; do
    tag=ser${ser}_$(echo ${dp:-mixed} | tr -d ' -')
    HPK_SIDE_SERIAL=$ser rocprofv3 --kernel-trace --stats -d $OUT/$tag -o k --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-rows 0 --no-probes --no-extra $dp > $OUT/$tag.log 2>&1
    echo "== kernel trace, side_serial=$ser ${dp:-mixed}: calls, average / min / max us"
    python - $OUT/$tag <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'hpk_' in r['Name']:
        print('  %-60s %6s %9.1f %9.1f %9.1f' % (__import__('re').search(r'hpk_\w+(<[^>]*>)?', r['Name']).group(0)[:60], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
PY
    rm -rf $OUT/$tag
  done
done
