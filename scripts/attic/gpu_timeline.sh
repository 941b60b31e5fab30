#!/bin/bash
# kernel timeline of one steady-state step: start offsets / durations / gaps (rocprofv3 --kernel-trace)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/timeline
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/t -o k --output-format csv -- python $R/bench.py --config ${CFG:-chr1_10kb} --steps ${STEPS:-6} --warmup ${WARM:-2} --cpu-rows 0 > $OUT/log 2>&1
python - <<PY
import csv, glob
ks = []
for f in glob.glob('$OUT/t/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ks.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:40]))
for f in glob.glob('$OUT/t/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ks.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '')[:30]))
ks.sort()
idx = [i for i, k in enumerate(ks) if 'hpk_stencil' in k[2]]
a, b = idx[-2], idx[-1]
t0 = ks[a][0]
prev_end = None
for s, e, nme in ks[a - 6:b + 1]:
    gap = (s - prev_end) / 1e3 if prev_end else 0
    print('%9.1f us  dur %7.1f  gap %7.1f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, gap, nme))
    prev_end = e
PY
