"""Energy conservation of the bench workload (C3, RESPA [4,2,1]) over a longer NVE run: total energy every `block` steps.
usage: python scripts/probe_drift.py [outer_step_fs=4.0] [n_blocks=20] [block=250]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench

dt_fs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
nblocks = int(sys.argv[2]) if len(sys.argv) > 2 else 20
block = int(sys.argv[3]) if len(sys.argv) > 3 else 250
sim, case = bench.build_simulation(32, (4, 2, 1), dt_fs, 'damped', None)
eng = sim.context._engine
bench.relax(sim, torch)
n = eng.n
rows = []
for b in range(nblocks + 1):
    st = sim.context.getState(getEnergy=True)
    pe, ke = st.getPotentialEnergy()._value, st.getKineticEnergy()._value
    rows.append((b * block * dt_fs * 1e-3, pe, ke, pe + ke))
    if b < nblocks:
        sim.step(block)
t = np.array([r[0] for r in rows])
e = np.array([r[3] for r in rows])
slope = np.polyfit(t, e, 1)[0]
ke_mean = np.mean([r[2] for r in rows])
print('outer step %.1f fs, %d atoms, %.1f ps' % (dt_fs, n, t[-1]))
print('total energy: first %.3f last %.3f kJ/mol, rms fluctuation %.3f, drift %.4f kJ/mol/ps = %.3e kT/ps per DOF'
      % (e[0], e[-1], np.std(e - np.polyval(np.polyfit(t, e, 1), t)), slope, slope / (2 * ke_mean)))
print('temperature %.1f K' % (2 * ke_mean / (3 * n * bench.KB)))
eng.ctx.check()
