#!/bin/bash
# scripts/pyHICCUPS on a real .mcool of the whole synthetic genome @5 kb: bands built on the GPU from the pixel table
# (default) against dense bands built on the host (HPK_HOST_BANDS=1); wall seconds, four runs each
cd $GRAFT_REPO_ROOT
W="1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 17 18 19 20 21 22 X"
[ -f /tmp/hpk_e2e_wg.mcool ] || PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 scripts/make_cool.py /tmp/hpk_e2e_wg.mcool --genome hg38 --res 5000 --num 2011 --group /resolutions/5000 --depth 25 --chroms $W > /dev/null 2>&1
python - <<'PY'
import os, subprocess, sys, time
W = "1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 17 18 19 20 21 22 X".split()
for rep in range(4):
    for hb in (0, 1):
        env = dict(os.environ)
        env.pop('HPK_HOST_BANDS', None)
        if hb: env['HPK_HOST_BANDS'] = '1'
        t = time.perf_counter()
        subprocess.call([sys.executable, 'scripts/pyHICCUPS', '-p', '/tmp/hpk_e2e_wg.mcool::/resolutions/5000', '-O', '/tmp/o_%d.bedpe' % hb,
                         '--pw', '4', '--ww', '7', '--maxww', '10', '--maxapart', '10000000', '-C'] + W + ['--logFile', '/tmp/l.log'],
                        env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        print('bands on the %s: %.2f s wall, %d lines' % ('host' if hb else 'GPU ', time.perf_counter() - t, sum(1 for _ in open('/tmp/o_%d.bedpe' % hb))))
print('identical output:', open('/tmp/o_0.bedpe').read() == open('/tmp/o_1.bedpe').read())
PY
