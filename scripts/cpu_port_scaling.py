#!/usr/bin/env python3
"""Thread scaling of the CPU port (oracle/cpu_port.c) on this host: C3 box (98 304 atoms), RESPA [4,2,1] at 4 fs.
    python scripts/cpu_port_scaling.py [threads ...]     (each count in a child process: OMP_NUM_THREADS is read at start-up)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = '''
import sys, os
sys.path.insert(0, %r)
from atomsmm_amd.testing import tip3p_box
from oracle import cpu_port
c = tip3p_box(32)
sec, st = cpu_port.time_port(c, warmup=2, steps=int(os.environ.get("PORT_STEPS", "6")), loops=(4, 2, 1), dt=0.004, skin=0.1)
print("%%d threads: %%.1f ms/step (%%d builds, %%d list entries)" %% (cpu_port.threads(), sec * 1e3, st["builds"], st["list_entries"]))
''' % ROOT
counts = [int(a) for a in sys.argv[1:]] or [1, 16, 64, os.cpu_count()]
for t in counts:
    env = dict(os.environ, OMP_NUM_THREADS=str(t), OMP_PROC_BIND='spread', OMP_PLACES='cores', PORT_STEPS='3' if t < 8 else '10')
    subprocess.run([sys.executable, '-c', CHILD], env=env, check=False)
