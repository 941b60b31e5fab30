#!/usr/bin/env python
"""Instruction census of one kernel of build/asm/*.s (make -C hicpeaks_amd/csrc asm): scripts/asm_kernel.py <substring> [--dump]"""
import collections
import re
import sys

s = open('build/asm/hpk_kernels-hip-amdgcn-amd-amdhsa-gfx950.s').read()
key = sys.argv[1]
names = [m.group(1) for m in re.finditer(r'^(_Z\w+):\s*; @', s, re.M) if key in m.group(1)]
for name in names:
    i0 = s.index('\n' + name + ':')
    i1 = s.index('.Lfunc_end', i0)
    body = s[i0:i1]
    if '--dump' in sys.argv:
        print(body)
        continue
    c = collections.Counter()
    for l in body.split('\n'):
        t = l.strip().split()
        if t and not t[0].startswith(('.', ';')) and not t[0].endswith(':'):
            c[t[0]] += 1
    tot = sum(c.values())
    sel = ['global_load', 'global_store', 'global_atomic', 's_load', 'scratch_load', 'scratch_store', 'v_readlane', 'v_writelane',
           'buffer_load', 'ds_read', 'ds_write', 's_waitcnt', 's_barrier', 'v_readfirstlane']
    print(name[:90], 'instr', tot, {k: sum(v for n, v in c.items() if n.startswith(k)) for k in sel})
