#!/bin/bash
# roofline.kernel_ms (HIP events inside bench.py) against rocprofv3's kernel trace IN THE SAME RUN, and an untraced run beside it
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/kms; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P='import sys,json
o=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print("%s: bench.py roofline.kernel_ms %.4f ms per launch of %d chromosomes (%d launches timed)" % (sys.argv[2], o["roofline"]["kernel_ms"], o["config"]["chromosomes_per_launch"], o["roofline"]["launches_timed"]))'
python $R/bench.py --steps 5 --warmup 2 --no-extra --no-probes --cpu-rows 0 > $OUT/plain.log 2>/dev/null
python -c "$P" $OUT/plain.log "untraced run"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o k --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-extra --no-probes --cpu-rows 0 > $OUT/traced.log 2>/dev/null
python -c "$P" $OUT/traced.log "traced run  "
python - <<PY
import csv, glob
f = glob.glob('$OUT/trace/**/k_kernel_stats.csv', recursive=True)[0]
for r in csv.reader(open(f)):
    if r and 'hpk_stencil_s' in r[0]: print('traced run  : rocprofv3 --stats hpk_stencil_s calls %s average %.4f ms min %.4f max %.4f' % (r[1], float(r[3]) / 1e6, float(r[5]) / 1e6, float(r[6]) / 1e6))
t = glob.glob('$OUT/trace/**/k_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(t)))
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for r in rows if 'hpk_stencil_s' in r['Kernel_Name']]
d = d[2 * 20:]                # the timed region's launches: behind the two warm-up steps of 20 library calls
print('traced run  : the %d stencil launches of the timed region in the trace: mean %.4f ms  min %.4f  max %.4f' % (len(d), sum(d) / len(d), min(d), max(d)))
PY
rm -rf $OUT/trace
