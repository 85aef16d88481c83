"""Throughput of the thermostatted RESPA integrators (Langevin_R, NHL_R, SIN_R; RESPA [4,2,1], 4 fs) on the C3 box against the
plain NVE RespaPropagator step: python scripts/probe_thermostats.py [names...]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import atomsmm_amd as atomsmm
from atomsmm_amd import openmm, unit
from atomsmm_amd.openmm import app
from atomsmm_amd.testing import system_from_arrays, tip3p_box

case = tip3p_box(32)
fs, K, ps = unit.femtoseconds, unit.kelvin, unit.picoseconds
MAKE = {'NVE': lambda: atomsmm.RespaPropagator([4, 2, 1]).integrator(4 * fs),
        'Langevin_R': lambda: atomsmm.Langevin_R_Integrator(4 * fs, [4, 2, 1], 300 * K, 1 / ps),
        'NHL_R': lambda: atomsmm.NHL_R_Integrator(4 * fs, [4, 2, 1], 300 * K, 10 * fs, 1 / ps),
        'SIN_R': lambda: atomsmm.SIN_R_Integrator(4 * fs, [4, 2, 1], 300 * K, 10 * fs, 1 / ps)}
for name in (sys.argv[1:] or list(MAKE)):
    system = system_from_arrays(case, nonbondedMethod='CutoffPeriodic', cutoff=1.0, switch=0.9)
    respa = atomsmm.RESPASystem(system, 0.7 * unit.nanometers, 0.5 * unit.nanometers)
    nb = atomsmm.hijackForce(respa, atomsmm.findNonbondedForce(respa))
    outer = atomsmm.DampedSmoothedForce(2.9 / unit.nanometers, 1.0 * unit.nanometers, 0.9 * unit.nanometers).importFrom(nb)
    outer.setForceGroup(2)
    outer.addTo(respa)
    integ = MAKE[name]()
    integ.setRandomNumberSeed(5) if hasattr(integ, 'setRandomNumberSeed') else None
    sim = app.Simulation(app.Topology(len(case['positions'])), respa, integ, openmm.Platform.getPlatformByName('HIP'))
    sim.context.setPositions(case['positions'] * unit.nanometers)
    sim.context.setVelocities(case['velocities'])
    sim.step(100)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sim.step(200)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 200
    ke = sim.context.getState(getEnergy=True).getKineticEnergy()._value
    print('%-11s ms/step %.3f  ns/day %6.1f  T = %.1f K  interpreted = %s' % (
        name, t * 1e3, 4e-6 * 86400 / t, 2 * ke / (3 * len(case['mass']) * 0.0083144626), sim.context._engine._interpreted), flush=True)
    del sim
