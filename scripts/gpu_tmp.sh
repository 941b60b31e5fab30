cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -k "poisson or bhfdr or boundary" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
cat > /tmp/pt.py <<PY
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
from hicpeaks_amd import _lib, synthetic
raw, w, _ = synthetic.synth_band(800, 61, depth=40.0, nloops=5, seed=1)
prm = _lib.make_params(_lib.MODE_HICCUPS, [2], [5], 10, 0.05, 500000, 10000, 16, 0)
c = _lib.Context(0)
for i in range(3):
    c.score_host(raw.astype(np.float32), None, None, None, prm, weight=w)
    c._check(c.lib.hpk_set_chunk_bounds(c.h, c.bounds.ctypes.data, 128))
PY
rocprofv3 --kernel-trace -d /tmp/pt -o k --output-format csv -- python /tmp/pt.py > /tmp/pt.log 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob('/tmp/pt/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'ptab' in r['Kernel_Name']:
            print('hpk_ptab', (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, 'us')
PY
