"""Drop-in check: scripts written for the reference run unchanged once the repository root is on sys.path.

The import lines below are the reference's own (tests/test_respa_forces.py:3-8) and every test starts from the calls its
tests start from -- app.PDBFile(...), app.ForceField(...), forcefield.createSystem(pdb.topology, ...) -- on the
reference's data files (copied, as data, to tests/golden/data/).  The step sequences re-type the bodies of
tests/test_respa_forces.py:11-79, tests/test_DampedSmoothedForce.py:11-35, tests/test_ExceptionNonbondedForce.py:11-24
and tests/test_systems.py:11-25,131-152; the expected values are the literals those files hold.  The platform name
'Reference' resolves to the HIP path (there is no other platform here)."""
import os

import pytest
from simtk import openmm
from simtk import unit
from simtk.openmm import app

import atomsmm

pytestmark = pytest.mark.gpu
DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'data')


def static_energy(pdb, system):
    integrator = openmm.VerletIntegrator(0.0*unit.femtoseconds)
    platform = openmm.Platform.getPlatformByName('Reference')
    simulation = app.Simulation(pdb.topology, system, integrator, platform)
    simulation.context.setPositions(pdb.positions)
    potential = simulation.context.getState(getEnergy=True).getPotentialEnergy()
    return potential/potential.unit


def spcfw_with(force_factory):
    case = os.path.join(DATA, 'q-SPC-FW')
    pdb = app.PDBFile(case + '.pdb')
    forcefield = app.ForceField(case + '.xml')
    system = forcefield.createSystem(pdb.topology, nonbondedMethod=app.CutoffPeriodic)
    force = force_factory()
    force.importFrom(atomsmm.hijackForce(system, atomsmm.findNonbondedForce(system))).addTo(system)
    return pdb, system


@pytest.mark.parametrize('adjustment,target', [(None, -24955.845391462222),              # G1  test_respa_forces.py:30
                                               ('shift', -26451.885982885935),           # G2  :34
                                               ('force-switch', -26516.68871844118)])    # G3  :38
def test_near_force(adjustment, target):
    rcut, rswitch = 10*unit.angstroms, 9.5*unit.angstroms
    pdb, system = spcfw_with(lambda: atomsmm.NearNonbondedForce(rcut, rswitch, adjustment))
    assert static_energy(pdb, system) == pytest.approx(target)


@pytest.mark.parametrize('degree,target', [(1, -25074.251664020387),                     # G4  test_DampedSmoothedForce.py:31
                                           (2, -25074.342992954276)])                    # G5  :35
def test_damped_smoothed_force(degree, target):
    rcut, rswitch, alpha = 10*unit.angstroms, 9.5*unit.angstroms, 0.29/unit.angstroms
    pdb, system = spcfw_with(lambda: atomsmm.DampedSmoothedForce(alpha, rcut, rswitch, degree=degree))
    assert static_energy(pdb, system) == pytest.approx(target)


@pytest.mark.parametrize('adjustment', [None, 'shift', 'force-switch'])
def test_far_force(adjustment):
    """near + far == the plain PME system (tests/test_respa_forces.py:41-79: self-consistency, no literal)."""
    rswitch_inner, rcut_inner = 6.5*unit.angstroms, 7.0*unit.angstroms
    rswitch, rcut = 9.5*unit.angstroms, 10*unit.angstroms
    case = os.path.join(DATA, 'q-SPC-FW')
    pdb = app.PDBFile(case + '.pdb')
    forcefield = app.ForceField(case + '.xml')
    system = forcefield.createSystem(pdb.topology, nonbondedMethod=openmm.app.PME)
    nbforce = atomsmm.hijackForce(system, atomsmm.findNonbondedForce(system))
    innerforce = atomsmm.NearNonbondedForce(rcut_inner, rswitch_inner, adjustment)
    innerforce.importFrom(nbforce).addTo(system)
    outerforce = atomsmm.FarNonbondedForce(innerforce, rcut, rswitch).setForceGroup(2)
    outerforce.importFrom(nbforce).addTo(system)
    potential = atomsmm.splitPotentialEnergy(system, pdb.topology, pdb.positions)['Total']
    refsys = forcefield.createSystem(pdb.topology, nonbondedMethod=openmm.app.PME, nonbondedCutoff=rcut, removeCMMotion=True)
    force = refsys.getForce(refsys.getNumForces()-2)
    force.setUseSwitchingFunction(True)
    force.setSwitchingDistance(rswitch)
    refpot = atomsmm.splitPotentialEnergy(refsys, pdb.topology, pdb.positions)['Total']
    assert potential/potential.unit == pytest.approx(refpot/refpot.unit)


def test_exceptions_force():
    """G11 (tests/test_ExceptionNonbondedForce.py:11-24): NonbondedExceptionsForce + the bonded terms of EMIM-B(CN)4."""
    case = os.path.join(DATA, 'emim_BCN4_Jiung2014')
    pdb = app.PDBFile(case + '.pdb')
    forcefield = app.ForceField(case + '.xml')
    system = forcefield.createSystem(pdb.topology, nonbondedMethod=app.CutoffPeriodic)
    force = atomsmm.forces.NonbondedExceptionsForce()
    force.importFrom(atomsmm.hijackForce(system, atomsmm.findNonbondedForce(system))).addTo(system)
    assert static_energy(pdb, system) == pytest.approx(-27616.298459208883)


def test_respa_system_with_special_bonds():
    """tests/test_systems.py:131-152 from its own readSystem (:11-25): per-term energies of RESPASystem(7 A, 5 A) over
    q-SPC-FW with redefined bonds and angles, through splitPotentialEnergy."""
    case = os.path.join(DATA, 'q-SPC-FW')
    pdb = app.PDBFile(case + '.pdb')
    forcefield = app.ForceField(case + '.xml')
    system = forcefield.createSystem(pdb.topology, nonbondedMethod=openmm.app.PME, nonbondedCutoff=10*unit.angstroms,
                                     rigidWater=False, constraints=None, removeCMMotion=False)
    nbforce = system.getForce(atomsmm.findNonbondedForce(system))
    nbforce.setUseSwitchingFunction(True)
    nbforce.setSwitchingDistance(9*unit.angstroms)
    respa_info = dict(rcutIn=7*unit.angstroms, rswitchIn=5*unit.angstroms)
    respa_system = atomsmm.RESPASystem(system, *respa_info.values())
    respa_system.redefine_bond(pdb.topology, 'HOH', 'H[1-2]', 'O', 1.05*unit.angstroms)
    respa_system.redefine_angle(pdb.topology, 'HOH', 'H[1-2]', 'O', 'H[1-2]', 113*unit.degrees)
    components = atomsmm.splitPotentialEnergy(respa_system, pdb.topology, pdb.positions)
    potential = dict()
    potential['HarmonicBondForce'] = 3665.684696323676
    potential['HarmonicAngleForce'] = 1811.197218501007
    potential['PeriodicTorsionForce'] = 0.0
    potential['Real-Space'] = 84694.39953220935
    potential['Reciprocal-Space'] = -111582.71281220087
    potential['CustomNonbondedForce'] = -25531.129587235544
    potential['CustomNonbondedForce(1)'] = 25531.129587235544
    potential['CustomBondForce'] = 0.0
    potential['CustomBondForce(1)'] = -1175.253817235862
    potential['CustomAngleForce'] = -305.0221912655623
    potential['Total'] = -22891.707373668243
    for term, value in components.items():
        assert value/value.unit == pytest.approx(potential[term])
