#!/bin/bash
# End to end through the command line: synthetic hg38 chr1-22,X (band archive on local disk) -> scripts/pyHICCUPS.
#   RES=10000 (default): union of (1,3)/(2,5)/(4,7), 5 Mb band (BASELINE configs[2]);  RES=5000: (4,7), 10 Mb band (configs[3]).
# Prints the wall time of the command (interpreter start + read + upload + kernels + clustering + BEDPE): the part of the
# north_star wall-time target that includes the host.  Output is kept in gpurun_out/cli_wg_$RES.txt.
R=$GRAFT_REPO_ROOT
cd $R
RES=${RES:-10000}
ARC=/tmp/wg$RES.npz
if [ "$RES" = "5000" ]; then PW="4"; WW="7"; MAXAPART=10000000; DEPTH=25.0; NL=800; else PW="1 2 4"; WW="3 5 7"; MAXAPART=5000000; DEPTH=60.0; NL=400; fi
{
python - <<PY
import numpy as np, time, torch
from hicpeaks_amd import synthetic, io, bandgen
res, W = $RES, 10
D = $MAXAPART // res
num = D + W + 1
t = time.time()
bands = {}
sizes = synthetic.hg38_bins(res)
nmax = max(sizes.values())
for i, (c, n) in enumerate(sizes.items()):
    raw_d, w_d, _, _ = bandgen.device_band(n, num, num, $( [ "$RES" = "5000" ] && echo 7 || echo 3 ), depth=$DEPTH, nloops=max(1, $NL * n // nmax), seed=i,
                                           device=torch.device('cuda', 0), want_expected=False)
    bands['chr' + c] = (raw_d.cpu().numpy(), w_d.cpu().numpy())
    del raw_d, w_d
io.save_band_archive('$ARC', res, bands, compressed=False)
print('archive of %d chromosomes, %.2f GB of bands, written in %.0f s' % (len(bands), sum(b[0].nbytes for b in bands.values()) / 1e9, time.time() - t))
PY
ls -la $ARC
for rep in 1 2 3; do
  t0=$(date +%s.%N)
  python scripts/pyHICCUPS -O /tmp/wg$RES.bedpe -p $ARC --pw $PW --ww $WW --maxww 10 --maxapart $MAXAPART --logFile /tmp/wg.log > /dev/null 2>&1
  t1=$(date +%s.%N)
  python -c "print('scripts/pyHICCUPS --pw $PW --ww $WW --maxapart $MAXAPART on the $RES bp genome: wall %.2f s' % ($t1 - $t0))"
done
wc -l /tmp/wg$RES.bedpe; head -3 /tmp/wg$RES.bedpe
rm -f $ARC
} 2>&1 | tee gpurun_out/cli_wg_$RES.txt
