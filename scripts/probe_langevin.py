"""Throughput of Langevin_R (RESPA [4,2,1] + Ornstein-Uhlenbeck bath, 'middle' scheme) on the C3 box: general path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import atomsmm_amd as atomsmm
from atomsmm_amd import openmm, unit
from atomsmm_amd.openmm import app
from atomsmm_amd.testing import system_from_arrays, tip3p_box

case = tip3p_box(32)
system = system_from_arrays(case, nonbondedMethod='CutoffPeriodic', cutoff=1.0, switch=0.9)
respa = atomsmm.RESPASystem(system, 0.7 * unit.nanometers, 0.5 * unit.nanometers)
nb = atomsmm.hijackForce(respa, atomsmm.findNonbondedForce(respa))
outer = atomsmm.DampedSmoothedForce(2.9 / unit.nanometers, 1.0 * unit.nanometers, 0.9 * unit.nanometers).importFrom(nb)
outer.setForceGroup(2); outer.addTo(respa)
integ = atomsmm.Langevin_R_Integrator(4 * unit.femtoseconds, [4, 2, 1], 300 * unit.kelvin, 1 / unit.picoseconds)
print(integ)
sim = app.Simulation(app.Topology(len(case['positions'])), respa, integ, openmm.Platform.getPlatformByName('HIP'))
sim.context.setPositions(case['positions'] * unit.nanometers)
sim.context.setVelocities(case['velocities'])
sim.step(100)
torch.cuda.synchronize(); t0 = time.perf_counter()
sim.step(200)
torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 200
ke = sim.context.getState(getEnergy=True).getKineticEnergy()._value
print('ms/step %.3f  ns/day %.1f  T=%.1f' % (t * 1e3, 4e-6 * 86400 / t, 2 * ke / (3 * len(case['mass']) * 0.0083144626)))
