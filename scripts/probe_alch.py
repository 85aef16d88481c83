import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import atomsmm_amd as atomsmm
from atomsmm_amd import openmm, unit
from atomsmm_amd.openmm import app
from atomsmm_amd.testing import system_from_arrays
d=np.load('/root/repo/tests/golden/phenol-in-water.npz', allow_pickle=True); c={k:d[k] for k in d.keys()}
system=system_from_arrays(c, nonbondedMethod='PME', cutoff=1.0, switch=0.9)
solute=set(int(i) for i in np.where(c['resname']=='aaa')[0])
s=atomsmm.AlchemicalRespaSystem(system, 7*unit.angstroms, 5*unit.angstroms, solute, coupling_function='lambda^4*(5-4*lambda)')
comp=atomsmm.splitPotentialEnergy(s, app.Topology(len(c['positions'])), c['positions']*unit.nanometers, **{'lambda':0.5,'respa_switch':1})
exp={'HarmonicBondForce': 2621.3223922886677, 'HarmonicAngleForce': 1525.1006876561419,'PeriodicTorsionForce': 18.767576693568476, 'Real-Space': 80089.51116719692,'Reciprocal-Space': -107038.52551657759, 'CustomNonbondedForce': 5037.152491649265,'CustomBondForce': -53.526446723139806, 'CustomBondForce(1)': -53.374675325650806,'CustomCVForce': -7.114065227572182, 'CustomCVForce(1)': -6.301336948673654,'Total': -17866.987725318053}
for k,v in comp.items(): print(k, v._value, exp.get(k), v._value-exp.get(k,0))
print('sum solute q', c['charge'][sorted(solute)].sum(), 'sum q2 solute', (c['charge'][sorted(solute)]**2).sum())
