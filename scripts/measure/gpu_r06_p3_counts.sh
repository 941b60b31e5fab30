#!/bin/bash
# round 6: scalar / vector / LDS instructions of the stencil with and without phase 3 (HPK_DBG_STOP=2: the tables only) - what the batches
# issue, against the time they take (profiles/r06_stencil_ablation.txt)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/p3cnt; mkdir -p $OUT
PB="--steps 1 --warmup 1 --batch 8 --group 8 --cpu-rows 0 --no-probes --no-extra"
for st in 0 2; do
  HPK_DBG_STOP=$st rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d $OUT/s$st -o c --output-format csv -- python $R/bench.py $PB > $OUT/s$st.log 2>&1
  python - $OUT/s$st $st <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if 'hpk_stencil_s' in r['Kernel_Name'] and 'Lb1EEE' not in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
print('dbg_stop', sys.argv[2], {k: '%.4g per launch of 8 (n=%d)' % (sum(v) / len(v), len(v)) for k, v in sorted(acc.items())})
PY
done
rm -rf $OUT/s0 $OUT/s2
