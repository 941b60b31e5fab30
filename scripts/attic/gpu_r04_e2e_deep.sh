#!/bin/bash
# End to end at real depth: a deep whole-genome 5 kb .mcool (scripts/make_cool_deep.py), the host stages per
# chromosome, the command line's wall time on ONE GPU, and the share of that wall time the GPU is busy (rocprofv3 kernel trace).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04e; mkdir -p $O
DEPTH=${DEPTH:-500}
F=/tmp/hpk_deep.mcool
CH="1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 17 18 19 20 21 22 X"
{
echo "# host: $(nproc) cores, $(free -g | awk '/Mem:/{print $2}') GiB RAM; /tmp: $(df -h /tmp | tail -1 | awk '{print $4}') free"
cd /tmp && export TMPDIR=/tmp
t0=$(date +%s)
PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 $R/scripts/make_cool_deep.py $F --res 5000 --num 2011 --depth $DEPTH --far --threads $(( $(nproc) < 48 ? $(nproc) : 48 )) 2>/dev/null | tail -1
echo "# (written in $(( $(date +%s) - t0 )) s)"
python $R/scripts/host_e2e.py --deep --depth $DEPTH --file $F --chroms $CH
echo "## the command line under rocprofv3 --kernel-trace --stats (warm page cache): kernel time against wall"
t0=$(date +%s.%N)
rocprofv3 --kernel-trace --stats -d $O/prof -o e2e --output-format csv -- python $R/scripts/pyHICCUPS -p $F::/resolutions/5000 -O /tmp/deep.bedpe --pw 4 --ww 7 --maxapart 10000000 --logFile /tmp/deep.log > /dev/null 2>&1
t1=$(date +%s.%N)
python - <<PY
import csv, glob
tot = 0.0; rows = []
for f in glob.glob('$O/prof/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        tot += float(r['TotalDurationNs']); rows.append((float(r['TotalDurationNs']), r['Name'][:70], r['Calls']))
wall = $t1 - $t0
print('wall (traced) %.1f s, kernels %.2f s -> GPU busy %.1f %% of the wall time' % (wall, tot / 1e9, 100 * tot / 1e9 / wall))
for t, n, c in sorted(rows, reverse=True)[:8]:
    print('  %8.3f s  %5s calls  %s' % (t / 1e9, c, n))
PY
wc -l /tmp/deep.bedpe
} 2>&1 | tee $O/host_e2e_deep.txt
rm -rf $O/prof
