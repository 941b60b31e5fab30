#!/bin/bash
# round 5: hpk_score's instruction counters in the steady state (bounds inherited: launches of 64 chromosomes after two warm-up steps) -
# the round's profile passes run cold contexts, whose launches carry four times the records
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_score; rm -rf $O; mkdir -p $O
for cnt in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  nm=$(echo $cnt | tr ' ' '_' | cut -c1-30)
  rocprofv3 --kernel-trace --pmc $cnt -d $O/$nm -o c --output-format csv -- python $R/bench.py ${CFG:-} --steps 1 --warmup 2 --cpu-rows 0 --no-probes --no-extra > $O/$nm.log 2>&1
done
python - <<'PY'
import csv, glob, collections, os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/pmc_score'
for f in sorted(glob.glob(O+'/*/*counter_collection.csv')+glob.glob(O+'/*/*/*counter_collection.csv')):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'hpk_score' in r['Kernel_Name']:
            acc[r['Counter_Name']].append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
    for k,v in sorted(acc.items()):
        v.sort(); last=[x for _,x in v[-20:]]
        print('%-24s launches %3d  mean of the last 20: %.4g' % (k, len(v), sum(last)/len(last)))
PY
rm -rf $O/*/
