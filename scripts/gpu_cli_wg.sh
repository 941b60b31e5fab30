#!/bin/bash
# End to end through the command line: synthetic hg38 chr1-22,X at 10 kb (band archive on local disk) -> pyHICCUPS,
# union of (1,3)/(2,5)/(4,7), 5 Mb band.  Prints the wall time of the command (read + upload + kernels + clustering +
# BEDPE), the part of the north_star metric that includes the host.
R=$GRAFT_REPO_ROOT
cd $R
ARC=/tmp/wg10k.npz
python - <<PY
import numpy as np, time
from hicpeaks_amd import synthetic, io
res, D, W = 10000, 500, 10
num = D + W + 1
t = time.time()
bands = {}
for i, (c, n) in enumerate(synthetic.hg38_bins(res).items()):
    raw, w, _ = synthetic.synth_band(n, num, depth=60.0, nloops=max(1, 400 * n // 24896), seed=i)
    bands['chr' + c] = (raw.astype(np.float32), w)
io.save_band_archive('$ARC', res, bands, compressed=False)
print('archive written in %.0f s' % (time.time() - t))
PY
ls -la $ARC
for rep in 1 2; do
  t0=$(date +%s.%N)
  python scripts/pyHICCUPS -O /tmp/wg10k.bedpe -p $ARC --pw 1 2 4 --ww 3 5 7 --maxww 10 --maxapart 5000000 --logFile /tmp/wg.log > /dev/null 2>&1
  t1=$(date +%s.%N)
  python -c "print('pyHICCUPS wall %.2f s' % ($t1 - $t0))"
done
wc -l /tmp/wg10k.bedpe; head -3 /tmp/wg10k.bedpe
