#!/bin/bash
# The command line on the synthetic genome with and without the record bound (HPK_SPEC=0): the BEDPE files must be identical.
R=$GRAFT_REPO_ROOT; cd $R
RES=${RES:-10000}
ARC=/tmp/wgs$RES.npz; rm -f /tmp/a.log /tmp/b.log /tmp/c.log
if [ "$RES" = "5000" ]; then PW="4"; WW="7"; MAXAPART=10000000; DEPTH=25.0; NL=800; MW=7; else PW="1 2 4"; WW="3 5 7"; MAXAPART=5000000; DEPTH=60.0; NL=400; MW=3; fi
python - <<PY
import numpy as np, torch
from hicpeaks_amd import synthetic, io, bandgen
res, W = $RES, 10
num = $MAXAPART // res + W + 1
bands = {}
sizes = synthetic.hg38_bins(res)
nmax = max(sizes.values())
for i, (c, n) in enumerate(sizes.items()):
    # depth varies from chromosome to chromosome so that the frozen width moves between neighbours
    raw_d, w_d, _, _ = bandgen.device_band(n, num, num, $MW, depth=$DEPTH * (0.03 if i % 3 == 1 else 1.0), nloops=max(1, $NL * n // nmax), seed=i,
                                           device=torch.device('cuda', 0), want_expected=False)
    bands['chr' + c] = (raw_d.cpu().numpy(), w_d.cpu().numpy())
io.save_band_archive('$ARC', res, bands, compressed=False)
PY
HPK_SPEC=0 python scripts/pyHICCUPS -O /tmp/a.bedpe -p $ARC --pw $PW --ww $WW --maxww 10 --maxapart $MAXAPART --logFile /tmp/a.log > /dev/null 2>&1
python scripts/pyHICCUPS -O /tmp/b.bedpe -p $ARC --pw $PW --ww $WW --maxww 10 --maxapart $MAXAPART --logFile /tmp/b.log > /dev/null 2>&1
wc -l /tmp/a.bedpe /tmp/b.bedpe
cmp /tmp/a.bedpe /tmp/b.bedpe && echo "IDENTICAL with and without the record bound"
echo "chromosomes logged: $(grep -c "Observed Contact Number" /tmp/b.log), computed once more: $(grep -c "computed once more" /tmp/b.log)"
# and with a bound forced to the narrowest width: every chromosome that widens at all is computed once more
HPK_SPEC_FORCE=$MW python scripts/pyHICCUPS -O /tmp/c.bedpe -p $ARC --pw $PW --ww $WW --maxww 10 --maxapart $MAXAPART --logFile /tmp/c.log > /dev/null 2>&1
cmp /tmp/a.bedpe /tmp/c.bedpe && echo "IDENTICAL with the bound forced to $MW; computed once more: $(grep -c "computed once more" /tmp/c.log) of $(grep -c "Observed Contact Number" /tmp/c.log)"
rm -f $ARC
