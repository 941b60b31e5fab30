#!/usr/bin/env python
"""Where the SGPR spills of hpk_stencil_s live: v_writelane / v_readlane (an SGPR parked in / fetched from a lane of
a VGPR) per stretch of the tile loop, next to the stretch's other instructions.  Static counts over the code of one tile pass
(both variants of phase 1's cell loop - all cells inside the band / some masked - are in the listing; a tile executes one).

usage: spill_census.py <hpk_kernels-hip-amdgcn-amd-amdhsa-gfx950.s built with -gline-tables-only> <mangled-name substring>
       (make -C hicpeaks_amd/csrc asm EXTRA=-gline-tables-only; the marks below are source lines of hpk_kernels.hip)"""
import bisect
import collections
import re
import sys

src = open(sys.argv[2]).read().splitlines() if len(sys.argv) > 3 else None
text = open(sys.argv[1]).read()
pat = sys.argv[-1]
names = [m.group(1) for m in re.finditer(r'^(_Z\w+):', text, re.M) if pat in m.group(1)]
name = names[0]
i = text.index(name + ':')
body = text[i:text.index('.Lfunc_end', i)].splitlines()
hip = open(__file__.replace('scripts/measure/spill_census.py', 'hicpeaks_amd/csrc/hpk_kernels.hip')).read().splitlines()


def line_of(marker, after=0):
    for n, l in enumerate(hip[after:], after + 1):
        if marker in l:
            return n
    raise SystemExit('marker not found: ' + marker)


k0 = line_of('hpk_stencil_s(HpkStencilArgs a')
marks = [(0, 'helpers (inlined)'), (k0, 'prologue'), (line_of('auto flush_hist = ', k0), 'flush of a band\'s counts'),
         (line_of('    while (have) {', k0), 'band top'), (line_of('    do {', k0), 'tile top (walk, next tile)'),
         (line_of('// ---- phase 1 (rows)', k0), 'phase 1: setup'), (line_of('#define HPK_CELLS(MASKED)', k0), 'phase 1: cells (x2 variants)'),
         (line_of('    cm &= cmask;', k0), 'phase 1: list slot + f64 prefix'), (line_of('    if (nrow != 0u) {', k0), 'phase 1: list entries'),
         (line_of('// ... and of the packed plane', k0), 'phase 1: packed prefix'), (line_of("// The next tile's rows start moving now", k0), 'prefetch'),
         (line_of('// ---- phase 2 (columns)', k0), 'phase 2'), (line_of("// the next tile's column weights are in", k0), 'column weights, gap rows'),
         (line_of('// ---- phase 3: batches', k0), 'phase 3: search'), (line_of('// ---- sums at the resolving step', k0), 'phase 3: sums, records'),
         (line_of('    tnext = lds_u32(lds0 +', line_of('// ---- sums at the resolving step', k0)), 'tile end (work list)'),
         (line_of('hpk_stencil_lean(HpkStencilArgs', k0) - 40, 'after')]
keys = [m[0] for m in marks]
tot = collections.OrderedDict((m[1], collections.Counter()) for m in marks)
cur = 0
for l in body:
    t = l.strip()
    m = re.match(r'\.loc\s+\d+\s+(\d+)', t)
    if m:
        cur = int(m.group(1))
        continue
    if not t or t[0] in ';.' or t.split()[0].endswith(':'):
        continue
    op = t.split()[0]
    kind = 'lane' if op.startswith(('v_readlane', 'v_writelane')) else 'valu' if op.startswith('v_') else \
        'wait' if op.startswith(('s_waitcnt', 's_nop', 's_barrier')) else 'salu' if op.startswith('s_') else \
        'lds' if op.startswith('ds_') else 'vmem' if op.startswith(('buffer_', 'global_', 'flat_', 'scratch_')) else 'other'
    tot[marks[max(0, bisect.bisect_right(keys, cur) - 1)][1]][kind] += 1
print('# %s' % name)
print('%-34s %6s %6s %6s %6s %5s %5s' % ('stretch', 'lane', 'valu', 'salu', 'wait', 'lds', 'vmem'))
s = collections.Counter()
for k, c in tot.items():
    if sum(c.values()):
        print('%-34s %6d %6d %6d %6d %5d %5d' % (k, c['lane'], c['valu'], c['salu'], c['wait'], c['lds'], c['vmem']))
        s.update(c)
print('%-34s %6d %6d %6d %6d %5d %5d' % ('all', s['lane'], s['valu'], s['salu'], s['wait'], s['lds'], s['vmem']))
