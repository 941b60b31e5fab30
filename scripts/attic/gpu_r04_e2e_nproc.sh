#!/bin/bash
# reader processes: --nproc N workers around one queue, all on the one GPU (HPK_CLI_SHARE_GPU=1), deep map
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04e; mkdir -p $O
F=/tmp/hpk_deep.mcool
cd /tmp && export TMPDIR=/tmp
[ -f $F ] || PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 $R/scripts/make_cool_deep.py $F --res 5000 --num 2011 --depth ${DEPTH:-500} --far --threads 48 2>/dev/null | tail -1
{
for N in 1 2 4 8; do
  t0=$(date +%s.%N)
  HPK_CLI_SHARE_GPU=1 python $R/scripts/pyHICCUPS -p $F::/resolutions/5000 -O /tmp/deep_n$N.bedpe --pw 4 --ww 7 --maxapart 10000000 --nproc $N --deterministic --logFile /tmp/deep.log > /dev/null 2>/tmp/err_$N.txt
  rc=$?
  t1=$(date +%s.%N)
  echo "--nproc $N (one GPU shared)  wall $(python -c "print('%.2f' % ($t1 - $t0))") s  rc $rc  lines $(wc -l < /tmp/deep_n$N.bedpe)"
  [ $rc -ne 0 ] && tail -3 /tmp/err_$N.txt
done
cmp /tmp/deep_n1.bedpe /tmp/deep_n8.bedpe && echo "identical output (--deterministic)"
} 2>&1 | tee $O/e2e_nproc.txt
