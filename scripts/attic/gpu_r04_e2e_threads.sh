#!/bin/bash
# decoding threads of the cooler reader against the command line's wall time on the deep map (file from gpu_r04_e2e_deep.sh's recipe)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04e; mkdir -p $O
DEPTH=${DEPTH:-500}
F=/tmp/hpk_deep.mcool
cd /tmp && export TMPDIR=/tmp
[ -f $F ] || PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 $R/scripts/make_cool_deep.py $F --res 5000 --num 2011 --depth $DEPTH --far --threads 48 2>/dev/null | tail -1
{
for T in 16 32 64 128; do
  for rep in 1 2; do
    t0=$(date +%s.%N)
    HPK_READ_THREADS=$T python $R/scripts/pyHICCUPS -p $F::/resolutions/5000 -O /tmp/deep_$T.bedpe --pw 4 --ww 7 --maxapart 10000000 --logFile /tmp/deep.log > /dev/null 2>&1
    t1=$(date +%s.%N)
    echo "HPK_READ_THREADS=$T  wall $(python -c "print('%.2f' % ($t1 - $t0))") s  lines $(wc -l < /tmp/deep_$T.bedpe)"
  done
done
cmp /tmp/deep_16.bedpe /tmp/deep_128.bedpe && echo "identical output"
} 2>&1 | tee $O/e2e_threads.txt
