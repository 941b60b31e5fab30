#!/usr/bin/env python
"""Where the command line's wall time goes outside its own timeline: interpreter start -> imports -> main() -> exit.
usage: cli_wall.py <pyHICCUPS arguments...>   (prints to stderr; run under `time`)"""
import os, sys, time
t_py = time.time()
boot = None
try:        # the process' start (clock ticks since boot) against now
    st = open('/proc/self/stat').read().rsplit(')', 1)[1].split()
    start_ticks = int(st[19]); hz = os.sysconf('SC_CLK_TCK')
    up = float(open('/proc/uptime').read().split()[0])
    boot = up - start_ticks / hz            # seconds the process has lived when this line runs
except Exception:
    pass
import atexit
marks = [('interpreter up (since exec)', boot if boot is not None else 0.0)]
def mark(name):
    marks.append((name, (boot or 0.0) + time.time() - t_py))
def report():
    mark('atexit')
    for n, t in marks:
        sys.stderr.write('[cli wall] %-34s %.3f s\n' % (n, t))
atexit.register(report)
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, '..', '..'))
import hicpeaks_amd.cli as cli
mark('cli imported')
import numpy
mark("numpy imported")
from hicpeaks_amd import _lib
c = _lib.default_context(0) if os.environ.get("CLI_WALL_CTX_FIRST") else None
mark('context created')
rc = cli.main_hiccups(sys.argv[1:])
mark('main returned')
sys.exit(rc)
