#!/bin/bash
# round 5: where the command line's wall time goes on the deep whole-genome 5 kb .mcool (1.07 G pixels): HPK_CLI_TIMELINE - every
# stage of every chromosome per thread - plus the process' own start-up and tail.  VARIANTS: environment settings to compare on
# the same box ("-" = none), interleaved.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out; F=/tmp/hpk_deep.mcool
cd /tmp && export TMPDIR=/tmp
echo "# host: $(nproc) cores"
PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 $R/scripts/make_cool_deep.py $F --res 5000 --num 2011 --depth ${DEPTH:-500} --far --threads $(( $(nproc) < 48 ? $(nproc) : 48 )) 2>/dev/null | tail -1
for rep in 1 2 3; do
  for v in ${VARIANTS:--}; do
    t0=$(date +%s.%N)
    env $( [ "$v" = "-" ] || echo $v ) HPK_READ_THREADS=${THREADS:-64} HPK_CLI_TIMELINE=$O/r05_timeline_${v//[^A-Za-z0-9_]/}_$rep.txt python $R/scripts/pyHICCUPS -p $F::/resolutions/5000 -O /tmp/deep.bedpe --pw 4 --ww 7 --maxapart 10000000 --logFile /tmp/deep.log > /tmp/deep.out 2>&1 || tail -5 /tmp/deep.out
    t1=$(date +%s.%N)
    echo "$v  wall $(python -c "print('%.2f' % ($t1 - $t0))") s  lines $(wc -l < /tmp/deep.bedpe)  md5 $(md5sum < /tmp/deep.bedpe | cut -c1-8)"
  done
done
