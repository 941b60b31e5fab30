#!/bin/bash
# round 5: per-kernel durations (rocprofv3 --kernel-trace --stats) of the default launch shape, lean kernel on / off
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/trace5; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for args in "--depths 60" "" "--config chr1_5kb"; do
for lean in 1 0; do
  rm -rf $OUT/t
  HPK_LEAN=$lean timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/t -o k --output-format csv -- python $R/bench.py $args --steps 2 --warmup 1 --cpu-rows 0 --no-probes --no-extra > $OUT/log.txt 2>&1
  echo "== bench.py $args lean=$lean"
  python - <<PY
import csv,glob
f=glob.glob('$OUT/t/**/*kernel_stats.csv',recursive=True)
for row in csv.DictReader(open(f[0])):
    n=row['Name']
    if n.startswith('void (anonymous namespace)::') or 'hpk' in n:
        print('%-60s calls %5s avg_us %10.1f total_ms %9.2f' % (n.replace('void (anonymous namespace)::','')[:60], row['Calls'], float(row['AverageNs'])/1e3, float(row['TotalDurationNs'])/1e6))
PY
done
done
