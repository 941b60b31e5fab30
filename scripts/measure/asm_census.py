#!/usr/bin/env python
"""Static instruction census of one kernel in the -save-temps assembly: VALU / SALU / LDS / VMEM per basic block.
usage: asm_census.py file.s <substring of the mangled kernel name>"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
start = end = None
for i, l in enumerate(lines):
    if start is None and l.startswith('_Z') and pat in l.split(':')[0] and ':' in l:
        start = i
    elif start is not None and l.strip().startswith('.Lfunc_end'):
        end = i
        break
print('lines', start, end)
seg, cur = [], ['entry', 0, 0, 0, 0, start]
for i, l in enumerate(lines[start + 1:end], start + 1):
    t = l.strip()
    if re.match(r'^\.LBB\d+_\d+:', t):
        seg.append(cur)
        cur = [t.split(':')[0], 0, 0, 0, 0, i]
        continue
    if t.startswith('v_'): cur[1] += 1
    elif t.startswith('s_'): cur[2] += 1
    elif t.startswith('ds_'): cur[3] += 1
    elif t.startswith('global_') or t.startswith('buffer_') or t.startswith('flat_'): cur[4] += 1
seg.append(cur)
print('%-12s %6s %6s %6s %6s  line' % ('block', 'valu', 'salu', 'lds', 'vmem'))
for s in seg:
    if s[1] + s[2] + s[3] + s[4] > int(sys.argv[3]) if len(sys.argv) > 3 else 10:
        print('%-12s %6d %6d %6d %6d  %d' % tuple(s))
print('total', [sum(s[k] for s in seg) for k in range(1, 5)])
