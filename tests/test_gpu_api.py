"""GPU tests through the AtomsMM-shaped Python API (atomsmm_amd as atomsmm): these read like the
reference's own tests (tests/test_respa_forces.py, test_DampedSmoothedForce.py, test_systems.py) with
`ForceField.createSystem` replaced by the array-built equivalent and the platform being HIP.
Tolerance vs the reference literals: pytest.approx default (rel 1e-6), as in the reference."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')

import atomsmm_amd as atomsmm  # noqa: E402
from atomsmm_amd import openmm, unit  # noqa: E402
from atomsmm_amd.openmm import app  # noqa: E402
from atomsmm_amd.testing import system_from_arrays  # noqa: E402
from oracle import oracle as O  # noqa: E402  (checker only)


def create_system(case, **kw):
    return system_from_arrays(case, **kw), case['positions'] * unit.nanometers, app.Topology(len(case['positions']))


def executeNearForceTest(spcfw, adjustment, target):
    rcut = 10 * unit.angstroms
    rswitch = 9.5 * unit.angstroms
    system, positions, topology = create_system(spcfw, nonbondedMethod='CutoffPeriodic', flexible=False)
    force = atomsmm.NearNonbondedForce(rcut, rswitch, adjustment)
    force.importFrom(atomsmm.hijackForce(system, atomsmm.findNonbondedForce(system))).addTo(system)
    integrator = openmm.VerletIntegrator(0.0 * unit.femtoseconds)
    platform = openmm.Platform.getPlatformByName('Reference')      # resolves to the HIP platform
    assert platform.getName() == 'HIP'
    simulation = app.Simulation(topology, system, integrator, platform)
    simulation.context.setPositions(positions)
    state = simulation.context.getState(getEnergy=True)
    potential = state.getPotentialEnergy()
    assert potential / potential.unit == pytest.approx(target)


def test_unshifted_near(spcfw):
    executeNearForceTest(spcfw, None, -24955.845391462222)           # tests/test_respa_forces.py:30


def test_shifted_near(spcfw):
    executeNearForceTest(spcfw, 'shift', -26451.885982885935)         # tests/test_respa_forces.py:34


def test_force_switched_near(spcfw):
    executeNearForceTest(spcfw, 'force-switch', -26516.68871844118)   # tests/test_respa_forces.py:38


@pytest.mark.parametrize('degree,target', [(1, -25074.251664020387), (2, -25074.342992954276)])
def test_damped_smoothed(spcfw, degree, target):                      # tests/test_DampedSmoothedForce.py:30-35
    rcut = 10 * unit.angstroms
    rswitch = 9.5 * unit.angstroms
    alpha = 0.29 / unit.angstroms
    system, positions, topology = create_system(spcfw, nonbondedMethod='CutoffPeriodic', flexible=False)
    force = atomsmm.DampedSmoothedForce(alpha, rcut, rswitch, degree=degree)
    force.importFrom(atomsmm.hijackForce(system, atomsmm.findNonbondedForce(system))).addTo(system)
    simulation = app.Simulation(topology, system, openmm.VerletIntegrator(0.0), openmm.Platform.getPlatformByName('HIP'))
    simulation.context.setPositions(positions)
    potential = simulation.context.getState(getEnergy=True).getPotentialEnergy()
    assert potential / potential.unit == pytest.approx(target)


def test_exceptions_force_plus_bonded(emim, goldens):                 # tests/test_ExceptionNonbondedForce.py:11-24
    system, positions, topology = create_system(emim, nonbondedMethod='CutoffPeriodic')
    force = atomsmm.forces.NonbondedExceptionsForce()
    force.importFrom(atomsmm.hijackForce(system, atomsmm.findNonbondedForce(system))).addTo(system)
    simulation = app.Simulation(topology, system, openmm.VerletIntegrator(0.0), openmm.Platform.getPlatformByName('HIP'))
    simulation.context.setPositions(positions)
    potential = simulation.context.getState(getEnergy=True).getPotentialEnergy()
    assert potential / potential.unit == pytest.approx(-27616.298459208883)


def test_RESPASystem_split_energies(spcfw, goldens):
    """tests/test_systems.py:128-152 (q-SPC-FW, flexible) with a CutoffPeriodic NonbondedForce so that no
    reciprocal-space term is needed: bonded, near (G6), -near, exceptions and the direct-space total."""
    system, positions, topology = create_system(spcfw, nonbondedMethod='CutoffPeriodic', switch=0.9)
    respa_system = atomsmm.RESPASystem(system, 7 * unit.angstroms, 5 * unit.angstroms)
    components = atomsmm.splitPotentialEnergy(respa_system, topology, positions)
    assert set(components) == {'HarmonicBondForce', 'HarmonicAngleForce', 'Real-Space', 'Reciprocal-Space',
                               'CustomNonbondedForce', 'CustomNonbondedForce(1)', 'CustomBondForce', 'Total'}
    assert components['Reciprocal-Space']._value == 0.0      # CutoffPeriodic: no reciprocal-space term
    value = {k: v / v.unit for k, v in components.items()}
    assert value['HarmonicBondForce'] == pytest.approx(goldens['G_bonds']['value'])
    assert value['HarmonicAngleForce'] == pytest.approx(goldens['G_angles']['value'])
    assert value['CustomNonbondedForce'] == pytest.approx(-25531.129587235544)
    assert value['CustomNonbondedForce(1)'] == pytest.approx(25531.129587235544)
    assert value['CustomBondForce'] == 0.0
    c = spcfw
    rf = O.desc(O.NONBONDED, rc=1.0, rswitch=0.9, flags=O.COULOMB_RF | O.SWITCH,
                krf=(78.3 - 1) / ((2 * 78.3 + 1) * 1.0 ** 3), crf=3 * 78.3 / ((2 * 78.3 + 1) * 1.0))
    e_rf = O.pair_eval(rf, c['positions'], c['box'], c['charge'], c['sigma'], c['epsilon'], c['exc_pairs'], want_forces=False)[0]
    e_rf += O.dispersion_correction(c['sigma'], c['epsilon'], c['box'], 1.0, 0.9)
    assert value['Real-Space'] == pytest.approx(e_rf, rel=1e-10)
    assert value['Total'] == pytest.approx(sum(v for k, v in value.items() if k != 'Total'))


def test_pme_direct_and_reciprocal_groups_G7_G8(spcfw, goldens):
    """Group-2 NonbondedForce of RESPASystem on a PME system: the direct-space group (pair erfc + exclusion erf
    term + dispersion constant) reproduces tests/test_systems.py:142, the reciprocal-space group :143."""
    system, positions, topology = create_system(spcfw, nonbondedMethod='PME', switch=0.9)
    respa_system = atomsmm.RESPASystem(system, 7 * unit.angstroms, 5 * unit.angstroms)
    nb = respa_system.getForce(atomsmm.findNonbondedForce(respa_system))
    nb.setForceGroup(5)
    nb.setReciprocalSpaceForceGroup(6)
    context = openmm.Context(respa_system, openmm.VerletIntegrator(0.0))
    context.setPositions(positions)
    e = context.getState(getEnergy=True, groups={5}).getPotentialEnergy()
    assert e / e.unit == pytest.approx(goldens['G7']['value'])
    e = context.getState(getEnergy=True, groups={6}).getPotentialEnergy()
    assert e / e.unit == pytest.approx(goldens['G8']['value'])


def test_RESPASystem_split_energies_pme(spcfw, goldens):
    """tests/test_systems.py:128-152 without the special-bond redefinitions: every component of the PME system."""
    system, positions, topology = create_system(spcfw, nonbondedMethod='PME', switch=0.9)
    respa_system = atomsmm.RESPASystem(system, 7 * unit.angstroms, 5 * unit.angstroms)
    components = atomsmm.splitPotentialEnergy(respa_system, topology, positions)
    value = {k: v / v.unit for k, v in components.items()}
    assert value['Real-Space'] == pytest.approx(84694.39953220935)
    assert value['Reciprocal-Space'] == pytest.approx(-111582.71281220087)
    assert value['CustomNonbondedForce'] == pytest.approx(-25531.129587235544)
    assert value['CustomNonbondedForce(1)'] == pytest.approx(25531.129587235544)
    assert value['CustomBondForce'] == 0.0
    # the reference redefines bonds/angles and adds the difference as CustomBond/CustomAngle corrections
    # (G_bonds = 3665.68... - 1175.25..., G_angles = 1811.19... - 305.02...), so its Total is this system's Total
    assert value['HarmonicBondForce'] + value['HarmonicAngleForce'] == pytest.approx(goldens['G_bonds']['value'] + goldens['G_angles']['value'])
    assert value['Total'] == pytest.approx(-22891.707373668243)


def test_solvation_offsets_G9(heaq, goldens):
    """tests/test_systems.py:63-79: near force of RESPASystem over a SolvationSystem-prepared NonbondedForce
    (solute LJ off, solute charges as lambda_coul offsets, solute-solute pairs as exceptions), lambda_coul = 0.5."""
    import itertools
    system, positions, topology = create_system(heaq, nonbondedMethod='CutoffPeriodic', cutoff=1.0, switch=0.9)
    nb = system.getForce(atomsmm.findNonbondedForce(system))
    solute = [int(i) for i in np.where(heaq['resname'] == 'aaa')[0]]
    have = {tuple(sorted(nb.getExceptionParameters(k)[:2])) for k in range(nb.getNumExceptions())}
    for i, j in itertools.combinations(solute, 2):               # systems.py:286-296
        if (i, j) not in have:
            q1, s1, e1 = nb.getParticleParameters(i)
            q2, s2, e2 = nb.getParticleParameters(j)
            nb.addException(i, j, q1 * q2, (s1 + s2) / 2, (e1 * e2).sqrt())
    nb.addGlobalParameter('lambda_coul', 1.0)
    for i in solute:                                             # systems.py:298-311
        q, s, e = nb.getParticleParameters(i)
        nb.setParticleParameters(i, 0.0, 0.0, 0.0)
        if q / q.unit != 0.0:
            nb.addParticleParameterOffset('lambda_coul', i, q, 0.0, 0.0)
    respa_system = atomsmm.RESPASystem(system, 7 * unit.angstroms, 5 * unit.angstroms)
    components = atomsmm.splitPotentialEnergy(respa_system, topology, positions, lambda_coul=0.5)
    value = {k: v / v.unit for k, v in components.items()}
    assert value['CustomNonbondedForce'] == pytest.approx(-17294.836032921234)
    assert value['CustomNonbondedForce(1)'] == pytest.approx(17294.836032921194)
    assert value['CustomBondForce'] == pytest.approx(112.25315524350334)
    assert value['HarmonicBondForce'] == pytest.approx(1815.1848188179738)
    assert value['HarmonicAngleForce'] == pytest.approx(1111.5544374007236)
    assert value['PeriodicTorsionForce'] == pytest.approx(1.5998609986459567)


@pytest.mark.parametrize('method', ['CutoffPeriodic', 'PME'])
def test_far_plus_near_equals_total(spcfw, method):
    """tests/test_respa_forces.py:41-79 (G12; the reference runs it on the PME source): near + FarNonbondedForce ==
    the plain switched NonbondedForce, for the three adjustments."""
    for adjustment in (None, 'shift', 'force-switch'):
        system, positions, topology = create_system(spcfw, nonbondedMethod=method, flexible=False)
        nbforce = atomsmm.hijackForce(system, atomsmm.findNonbondedForce(system))
        inner = atomsmm.NearNonbondedForce(7.0 * unit.angstroms, 6.5 * unit.angstroms, adjustment)
        inner.importFrom(nbforce).addTo(system)
        outer = atomsmm.FarNonbondedForce(inner, 10 * unit.angstroms, 9.5 * unit.angstroms).setForceGroup(2)
        outer.importFrom(nbforce).addTo(system)
        potential = atomsmm.splitPotentialEnergy(system, topology, positions)['Total']
        refsys, _, _ = create_system(spcfw, nonbondedMethod=method, cutoff=1.0, switch=0.95, flexible=False)
        refpot = atomsmm.splitPotentialEnergy(refsys, topology, positions)['Total']
        assert potential / potential.unit == pytest.approx(refpot / refpot.unit)


@pytest.mark.parametrize('rows', ['molecule', 'atom'])
@pytest.mark.parametrize('method', ['CutoffPeriodic', 'PME'])
def test_far_force_shares_one_list(spcfw, method, rows):
    """FarNonbondedForce = total + discount (forces.py:710-724): the reference -- and OpenMM -- run two passes over two neighbour
    lists; here the discount (a near force guarded by step(rc0 - r), sign -1) shares the total's neighbour list, its displacement
    check and its sorted copies, and accumulates into the same force rows: ONE kernel launch for both, under the total's id, with
    per-atom rows (option cluster = 0) and with molecule rows (water, the default: the fused pass walks the front parts of the rows
    for both forces).  Either way the forces equal the separate (energy-carrying, analytic) evaluations."""
    system, positions, topology = create_system(spcfw, nonbondedMethod=method, flexible=False)
    nbforce = atomsmm.hijackForce(system, atomsmm.findNonbondedForce(system))
    inner = atomsmm.NearNonbondedForce(7.0 * unit.angstroms, 6.5 * unit.angstroms, 'force-switch')
    inner.importFrom(nbforce).addTo(system)
    outer = atomsmm.FarNonbondedForce(inner, 10 * unit.angstroms, 9.5 * unit.angstroms).setForceGroup(2)
    outer.importFrom(nbforce).addTo(system)
    integrator = openmm.CustomIntegrator(0.001)
    integrator.addComputePerDof('v', 'v + dt*f2/m')          # a force-only evaluation of group 2
    context = openmm.Context(system, integrator, openmm.Platform.getPlatformByName('HIP'),
                             {'Option.cluster': '1' if rows == 'molecule' else '0'})
    context.setPositions(positions)
    eng = context._engine
    ids = eng.pair_force_ids(2)
    assert len(ids) == 2
    total, discount = ids
    eng.ctx.profile_enable(True)
    integrator.step(1)
    launches = {pid: eng.ctx.profile_read(pid)[0] for pid in ids}
    eng.ctx.profile_enable(False)
    assert eng.ctx.pair_stats(discount)['rode_along'] == 1       # computed inside the total's launch
    assert launches == {total: 1, discount: 0}
    assert eng.ctx.pair_stats(total)['list_kind'] == (1 if rows == 'molecule' else 0)
    assert eng.ctx.pair_stats(discount)['shares_list'] == 1 and eng.ctx.pair_stats(discount)['n_evals'] == 1
    fused = eng._buffers['f2'].cpu().numpy().copy()
    separate = context.getState(getForces=True, groups={2}).getForces(asNumpy=True)._value
    assert np.abs(fused - separate).max() <= 1e-11 * np.abs(separate).max()


def test_forces_from_getState_match_oracle(spcfw):
    c = spcfw
    system, positions, topology = create_system(c, nonbondedMethod='CutoffPeriodic', switch=0.9)
    respa = atomsmm.RESPASystem(system, 7 * unit.angstroms, 5 * unit.angstroms)
    context = openmm.Context(respa, openmm.VerletIntegrator(0.0))
    context.setPositions(positions)
    f1 = context.getState(getForces=True, groups={1}).getForces(asNumpy=True)
    ref = O.pair_eval(O.desc(O.NEAR_FSWITCH, rc=0.7, rc0=0.7, rs0=0.5), c['positions'], c['box'], c['charge'], c['sigma'],
                      c['epsilon'], c['exc_pairs'])[1]
    f1 = f1.value_in_unit(unit.kilojoules_per_mole / unit.nanometer)
    assert np.abs(f1 - ref).max() <= 1e-9 * np.abs(ref).max()
    f0 = context.getState(getForces=True, groups=1 << 0).getForces(asNumpy=True)._value
    ref0 = (O.harmonic_bonds(c['bonds'], c['bond_r0'], c['bond_k'], c['positions'], c['box'])[1] +
            O.harmonic_angles(c['angles'], c['angle_theta0'], c['angle_k'], c['positions'], c['box'])[1])
    assert np.abs(f0 - ref0).max() <= 1e-10 * np.abs(ref0).max()
    both = context.getState(getForces=True, groups={0, 1}).getForces(asNumpy=True)._value
    assert np.abs(both - (ref + ref0)).max() <= 1e-9 * np.abs(ref + ref0).max()


def test_respa_dynamics_through_api_vs_oracle(spcfw):
    """RESPASystem + DampedSmoothedForce outer force + RespaPropagator([4,2,1]), 3 steps of 4 fs:
    positions/velocities/energy agree with the same step program driven on the CPU oracle."""
    c = spcfw
    system, positions, topology = create_system(c, nonbondedMethod='CutoffPeriodic')
    respa = atomsmm.RESPASystem(system, 7 * unit.angstroms, 5 * unit.angstroms)
    nb = atomsmm.hijackForce(respa, atomsmm.findNonbondedForce(respa))
    outer = atomsmm.DampedSmoothedForce(0.29 / unit.angstroms, 10 * unit.angstroms, 9 * unit.angstroms).importFrom(nb)
    outer.setForceGroup(2)
    outer.addTo(respa)
    integrator = atomsmm.RespaPropagator([4, 2, 1]).integrator(4 * unit.femtoseconds)
    simulation = app.Simulation(topology, respa, integrator, openmm.Platform.getPlatformByName('HIP'))
    simulation.context.setPositions(positions)
    simulation.context.setVelocitiesToTemperature(300 * unit.kelvin, 1)
    v0 = simulation.context.getState(getVelocities=True).getVelocities(asNumpy=True)._value.copy()
    nsteps = 3
    simulation.step(nsteps)
    state = simulation.context.getState(getPositions=True, getVelocities=True, getEnergy=True, groups={0, 1, 2})
    # oracle-driven program
    dt, m = 0.004, c['mass']
    dn = O.desc(O.NEAR_FSWITCH, rc=0.7, rc0=0.7, rs0=0.5)
    dd = O.desc(O.DAMPED, rc=1.0, rswitch=0.9, alpha=2.9, degree=1)
    pe = lambda d, p, wf=True: O.pair_eval(d, p, c['box'], c['charge'], c['sigma'], c['epsilon'], c['exc_pairs'], want_forces=wf)
    f0 = lambda p: (O.harmonic_bonds(c['bonds'], c['bond_r0'], c['bond_k'], p, c['box'])[1] +
                    O.harmonic_angles(c['angles'], c['angle_theta0'], c['angle_k'], p, c['box'])[1])
    x, v = c['positions'].copy(), v0.copy()
    F1 = pe(dn, x)[1]
    for _ in range(nsteps):
        F2 = pe(dd, x)[1]
        O.kick(v, F2, m, 0.5 * dt, fsub=F1)
        for _n1 in range(2):
            O.kick(v, F1, m, 0.25 * dt)
            F0 = f0(x)
            for _n0 in range(4):
                O.kick(v, F0, m, 0.0625 * dt)
                O.move(x, v, 0.125 * dt)
                F0 = f0(x)
                O.kick(v, F0, m, 0.0625 * dt)
            F1 = pe(dn, x)[1]
            O.kick(v, F1, m, 0.25 * dt)
        F2 = pe(dd, x)[1]
        O.kick(v, F2, m, 0.5 * dt, fsub=F1)
    assert np.abs(state.getPositions(asNumpy=True)._value - x).max() < 1e-11
    assert np.abs(state.getVelocities(asNumpy=True)._value - v).max() < 1e-9
    e_ref = (pe(dn, x, False)[0] + pe(dd, x, False)[0] +
             O.harmonic_bonds(c['bonds'], c['bond_r0'], c['bond_k'], x, c['box'], want_forces=False)[0] +
             O.harmonic_angles(c['angles'], c['angle_theta0'], c['angle_k'], x, c['box'], want_forces=False)[0])
    assert state.getPotentialEnergy()._value == pytest.approx(e_ref, rel=1e-10)
    assert state.getKineticEnergy()._value == pytest.approx(0.5 * O.mvv(v, m), rel=1e-12)
    st = simulation.context._engine.ctx.pair_stats(simulation.context._engine.pair_force_ids(1)[0])
    assert st['n_evals'] == 1 + 2 * nsteps + 1     # 1 + 2/step (force cache) + the final getState


def test_time_reversibility_and_energy_conservation(spcfw):
    """Size-independent properties: RESPA is time-reversible (run, flip v, run back) and the
    total energy drifts little over 50 steps of [4,2,1] at 2 fs."""
    c = spcfw
    system, positions, topology = create_system(c, nonbondedMethod='CutoffPeriodic')
    respa = atomsmm.RESPASystem(system, 7 * unit.angstroms, 5 * unit.angstroms)
    nb = atomsmm.hijackForce(respa, atomsmm.findNonbondedForce(respa))
    outer = atomsmm.DampedSmoothedForce(0.29 / unit.angstroms, 10 * unit.angstroms, 9 * unit.angstroms).importFrom(nb)
    outer.setForceGroup(2)
    outer.addTo(respa)
    integrator = atomsmm.RespaPropagator([4, 2, 1]).integrator(2 * unit.femtoseconds)
    context = openmm.Context(respa, integrator)
    context.setPositions(positions)
    context.setVelocitiesToTemperature(300 * unit.kelvin, 7)

    def total():
        s = context.getState(getEnergy=True)      # all groups: 31 (-near) cancels 1, leaving E0 + E2 = the Hamiltonian
        return s.getPotentialEnergy()._value + s.getKineticEnergy()._value
    e0 = total()
    integrator.step(50)
    e1 = total()
    ke = context.getState(getEnergy=True).getKineticEnergy()._value
    assert abs(e1 - e0) < 0.02 * ke
    s = context.getState(getVelocities=True)
    context.setVelocities(-s.getVelocities(asNumpy=True)._value)
    integrator.step(50)
    back = context.getState(getPositions=True).getPositions(asNumpy=True)._value
    assert np.abs(back - c['positions']).max() < 1e-7


def test_pme_respa_system_forces_and_dynamics(spcfw):
    """RESPASystem on the PME source exactly as the reference builds it (systems.py:62-82: group 2 = the PME
    NonbondedForce, reciprocal space included): (i) the group-2 forces are the gradient of the group-2 energy
    (central differences; the B-spline derivative in the gather is analytic), (ii) [4,2,1] RESPA at 2 fs conserves
    energy and is time-reversible."""
    c = spcfw
    system, positions, topology = create_system(c, nonbondedMethod='PME', switch=0.9)
    respa = atomsmm.RESPASystem(system, 7 * unit.angstroms, 5 * unit.angstroms)
    integrator = atomsmm.RespaPropagator([4, 2, 1]).integrator(2 * unit.femtoseconds)
    context = openmm.Context(respa, integrator)
    context.setPositions(positions)
    f = context.getState(getForces=True, groups={2}).getForces(asNumpy=True)._value
    x0 = c['positions'].copy()
    h = 1e-5
    for atom, k in ((0, 0), (1, 2), (700, 1), (1535, 0)):
        e = []
        for sgn in (+1, -1):
            x = x0.copy()
            x[atom, k] += sgn * h
            context.setPositions(x * unit.nanometers)
            e.append(context.getState(getEnergy=True, groups={2}).getPotentialEnergy()._value)
        assert -(e[0] - e[1]) / (2 * h) == pytest.approx(f[atom, k], abs=2e-5 * np.abs(f).max())
    context.setPositions(positions)
    context.setVelocitiesToTemperature(300 * unit.kelvin, 7)

    def total():
        s = context.getState(getEnergy=True)
        return s.getPotentialEnergy()._value + s.getKineticEnergy()._value
    e0 = total()
    integrator.step(50)
    e1 = total()
    ke = context.getState(getEnergy=True).getKineticEnergy()._value
    assert abs(e1 - e0) < 0.02 * ke
    s = context.getState(getVelocities=True)
    context.setVelocities(-s.getVelocities(asNumpy=True)._value)
    integrator.step(50)
    back = context.getState(getPositions=True).getPositions(asNumpy=True)._value
    assert np.abs(back - c['positions']).max() < 1e-7


def _heaq_system(heaq):
    """readSystem of tests/test_systems.py:13-26: PME, rc 1.0 nm, switch 0.9 nm, flexible, no constraints."""
    system, positions, topology = create_system(heaq, nonbondedMethod='PME', cutoff=1.0, switch=0.9)
    solute = set(int(i) for i in np.where(heaq['resname'] == 'aaa')[0])
    return system, positions, topology, solute


def _check(components, potential):
    assert set(components) == set(potential)
    for term, value in components.items():
        assert value / value.unit == pytest.approx(potential[term]), term


def test_SolvationSystem(heaq):                                    # tests/test_systems.py:28-42
    system, positions, topology, solute = _heaq_system(heaq)
    solvation_system = atomsmm.SolvationSystem(system, solute)
    components = atomsmm.splitPotentialEnergy(solvation_system, topology, positions, lambda_vdw=0.5, lambda_coul=0.5)
    _check(components, {'HarmonicBondForce': 1815.1848188179738, 'HarmonicAngleForce': 1111.5544374007236,
                        'PeriodicTorsionForce': 1.5998609986459567, 'Real-Space': 58273.35327317236,
                        'Reciprocal-Space': -76436.3982762784, 'CustomNonbondedForce': -64.67189605331785,
                        'Total': -15299.377781942014})


def test_SolvationSystem_with_lj_parameter_scaling(heaq):          # tests/test_systems.py:45-58
    system, positions, topology, solute = _heaq_system(heaq)
    solvation_system = atomsmm.SolvationSystem(system, solute, use_softcore=False)
    components = atomsmm.splitPotentialEnergy(solvation_system, topology, positions, lambda_vdw=0.5, lambda_coul=0.5)
    _check(components, {'HarmonicBondForce': 1815.1848188179738, 'HarmonicAngleForce': 1111.5544374007236,
                        'PeriodicTorsionForce': 1.5998609986459567, 'Real-Space': 58235.03496195241,
                        'Reciprocal-Space': -76436.3982762784, 'Total': -15273.024197108643})


def test_AlchemicalSystem_softcore_meets_the_SolvationSystem_literal(heaq):
    """AlchemicalSystem (systems.py:318-410) with the softcore coupling and the long-range correction: its coupling force is
    SolvationSystem's softcore force -- same expression, interaction group and exclusions -- so at lambda_vdw = 0.5 it
    must return the literal the reference's test holds for that force (tests/test_systems.py:28-42)."""
    system, positions, topology, solute = _heaq_system(heaq)
    alchemical_system = atomsmm.AlchemicalSystem(system, solute, group=5, use_lrc=True)
    components = atomsmm.splitPotentialEnergy(alchemical_system, topology, positions, lambda_vdw=0.5)
    assert components['CustomNonbondedForce'] / components['CustomNonbondedForce'].unit == pytest.approx(-64.67189605331785)
    # the solute carries no charge any more: the rest is SolvationSystem at lambda_coul = 0
    reference = atomsmm.splitPotentialEnergy(atomsmm.SolvationSystem(system, solute), topology, positions,
                                             lambda_vdw=0.5, lambda_coul=0.0)
    for term in ('Real-Space', 'Reciprocal-Space', 'HarmonicBondForce', 'Total'):
        assert components[term]._value == pytest.approx(reference[term]._value, rel=1e-12), term


@pytest.mark.parametrize('coupling', ['spline', 'art', 'lambda_vdw^2'])
def test_AlchemicalSystem_coupling_functions(heaq, coupling):
    """Lennard-Jones times ((gt0-gt1)*S + gt1) (systems.py:353-365): the energy of the coupling force is S(lambda) times the
    solute-solvent Lennard-Jones energy -- which is the softcore force at lambda = 1, pinned to the oracle elsewhere -- and
    deriv(energy, lambda_vdw) is S'(lambda) times it."""
    system, positions, topology, solute = _heaq_system(heaq)
    S = {'spline': lambda x: x ** 3 * (10 - 15 * x + 6 * x * x), 'art': lambda x: x - np.sin(2 * np.pi * x) / (2 * np.pi),
         'lambda_vdw^2': lambda x: x * x}[coupling]
    dS = {'spline': lambda x: 30 * x * x * (1 - x) ** 2, 'art': lambda x: 1 - np.cos(2 * np.pi * x),
          'lambda_vdw^2': lambda x: 2 * x}[coupling]

    def coupling_energy(sys_, lam):
        context = openmm.Context(sys_, openmm.VerletIntegrator(0.0))
        context.setPositions(positions)
        context.setParameter('lambda_vdw', lam)
        return context, context.getState(getEnergy=True, groups={7}).getPotentialEnergy()._value

    _, full = coupling_energy(atomsmm.AlchemicalSystem(system, solute, coupling='softcore', group=7), 1.0)
    codes = np.where(heaq['resname'] == 'aaa', 1.0, 2.0)
    d = O.desc(O.SOFTCORE, rc=1.0, rswitch=0.9, alpha=1.0, flags=O.SWITCH, Kc=1.0)
    nb = system.getForce(atomsmm.findNonbondedForce(system))
    excl = [nb.getExceptionParameters(k)[:2] for k in range(nb.getNumExceptions())]
    ref = O.pair_eval(d, heaq['positions'], heaq['box'], codes, heaq['sigma'], heaq['epsilon'], excl, want_forces=False)[0]
    assert full == pytest.approx(ref, rel=1e-11)
    alch = atomsmm.AlchemicalSystem(system, solute, coupling=coupling, group=7)
    for lam in (0.0, 0.3, 0.8, 1.0, 1.4):
        context, value = coupling_energy(alch, lam)
        factor = 0.0 if lam < 0 else (1.0 if lam >= 1 else S(lam))
        assert value == pytest.approx(factor * full, rel=1e-12, abs=1e-10), lam
        if 0 < lam < 1:
            derivatives = context.getState(getParameterDerivatives=True).getEnergyParameterDerivatives()
            assert derivatives['lambda_vdw'] == pytest.approx(dS(lam) * full, rel=1e-6), lam


def test_energy_derivative_of_charge_offsets(heaq):
    """deriv(energy, lambda_coul) for SolvationSystem's charge offsets (systems.py:289-308): the energy is a quadratic form of
    the charges, so a unit central step is exact.  Checked (i) against the same difference quotient taken through the public
    API at another step, on PME; (ii) against the oracle's pair energies at lambda +- 1 on the reaction-field system, whose
    lambda-dependent part is the direct-space sum alone."""
    system, positions, topology, solute = _heaq_system(heaq)
    solv = atomsmm.SolvationSystem(system, solute)
    context = openmm.Context(solv, openmm.VerletIntegrator(0.0))
    context.setPositions(positions)
    context.setParameter('lambda_coul', 0.35)
    d = context._engine.energy_derivative('lambda_coul')
    assert context.getParameter('lambda_coul') == 0.35
    e = []
    for lam in (0.6, 0.1, 0.35):
        context.setParameter('lambda_coul', lam)
        e.append(context.getState(getEnergy=True).getPotentialEnergy()._value)
    assert d == pytest.approx((e[0] - e[1]) / 0.5, rel=1e-9)
    assert context._engine.energy_derivative('lambda_coul') == pytest.approx(d, rel=1e-12)     # state restored: same answer again
    # reaction field: against the oracle
    rf_system, _, _ = create_system(heaq, nonbondedMethod='CutoffPeriodic', cutoff=1.0, switch=0.9)
    rf = atomsmm.SolvationSystem(rf_system, solute)
    context = openmm.Context(rf, openmm.VerletIntegrator(0.0))
    context.setPositions(positions)
    lam = 0.35
    context.setParameter('lambda_coul', lam)
    d = context._engine.energy_derivative('lambda_coul')
    nb = rf.getForce(atomsmm.findNonbondedForce(rf))
    excl = [nb.getExceptionParameters(k)[:2] for k in range(nb.getNumExceptions())]
    sig = heaq['sigma'].copy()
    eps = heaq['epsilon'].copy()
    member = heaq['resname'] == 'aaa'
    sig[member], eps[member] = 0.0, 0.0
    eps_rf = nb.getReactionFieldDielectric()
    krf, crf = (eps_rf - 1) / (2 * eps_rf + 1), 3 * eps_rf / (2 * eps_rf + 1)
    dd = O.desc(O.NONBONDED, rc=1.0, rswitch=0.9, flags=O.SWITCH | O.COULOMB_RF, krf=krf, crf=crf)
    ref = []
    for point in (lam + 1.0, lam - 1.0):
        q = np.where(member, point * heaq['charge'], heaq['charge'])
        ref.append(O.pair_eval(dd, heaq['positions'], heaq['box'], q, sig, eps, excl, want_forces=False)[0])
    assert d == pytest.approx((ref[0] - ref[1]) / 2.0, rel=1e-9)


def test_RESPASystem_on_SolvationSystem(heaq):                     # tests/test_systems.py:61-80
    system, positions, topology, solute = _heaq_system(heaq)
    solvation_system = atomsmm.SolvationSystem(system, solute)
    respa_system = atomsmm.RESPASystem(solvation_system, 7 * unit.angstroms, 5 * unit.angstroms)
    components = atomsmm.splitPotentialEnergy(respa_system, topology, positions, lambda_vdw=0.5, lambda_coul=0.5)
    _check(components, {'HarmonicBondForce': 1815.1848188179738, 'HarmonicAngleForce': 1111.5544374007236,
                        'PeriodicTorsionForce': 1.5998609986459567, 'Real-Space': 58161.10011792888,
                        'Reciprocal-Space': -76436.3982762784, 'CustomNonbondedForce': -64.67189605331785,
                        'CustomNonbondedForce(1)': -17294.836032921234, 'CustomNonbondedForce(2)': 17294.836032921194,
                        'CustomBondForce': 112.25315524350334, 'Total': -15299.377781942032})


def test_RESPASystem_with_exception_offsets(heaq):                 # tests/test_systems.py:83-107
    system, positions, topology, solute = _heaq_system(heaq)
    solvation_system = atomsmm.SolvationSystem(system, solute)
    nbforce = solvation_system.getForce(atomsmm.findNonbondedForce(solvation_system))
    for index in range(nbforce.getNumExceptions()):
        i, j, chargeprod, sigma, epsilon = nbforce.getExceptionParameters(index)
        nbforce.setExceptionParameters(index, i, j, 0.0, sigma, epsilon)
        nbforce.addExceptionParameterOffset('lambda_coul', index, chargeprod, 0.0, 0.0)
    respa_system = atomsmm.RESPASystem(solvation_system, 7 * unit.angstroms, 5 * unit.angstroms)
    components = atomsmm.splitPotentialEnergy(respa_system, topology, positions, lambda_vdw=0.5, lambda_coul=0.5)
    _check(components, {'HarmonicBondForce': 1815.1848188179738, 'HarmonicAngleForce': 1111.5544374007236,
                        'PeriodicTorsionForce': 1.5998609986459567, 'Real-Space': 58201.09912379701,
                        'Reciprocal-Space': -76436.3982762784, 'CustomNonbondedForce': -64.67189605331785,
                        'CustomNonbondedForce(1)': -17294.836032921234, 'CustomNonbondedForce(2)': 17294.836032921194,
                        'CustomBondForce': 72.25414937535754, 'Total': -15299.377781942048})


def test_RESPASystem_with_lj_parameter_scaling(heaq):              # tests/test_systems.py:110-128
    system, positions, topology, solute = _heaq_system(heaq)
    solvation_system = atomsmm.SolvationSystem(system, solute, use_softcore=False)
    respa_system = atomsmm.RESPASystem(solvation_system, 7 * unit.angstroms, 5 * unit.angstroms)
    components = atomsmm.splitPotentialEnergy(respa_system, topology, positions, lambda_vdw=0.5, lambda_coul=0.5)
    _check(components, {'HarmonicBondForce': 1815.1848188179738, 'HarmonicAngleForce': 1111.5544374007236,
                        'PeriodicTorsionForce': 1.5998609986459567, 'Real-Space': 58122.78180670893,
                        'Reciprocal-Space': -76436.3982762784, 'CustomNonbondedForce': -17317.054135213173,
                        'CustomNonbondedForce(1)': 17317.054135213126, 'CustomBondForce': 112.25315524350334,
                        'Total': -15273.024197108669})


def test_RESPASystem_with_special_bonds(spcfw):                    # tests/test_systems.py:131-152
    system, positions, _ = create_system(spcfw, nonbondedMethod='PME', cutoff=1.0, switch=0.9)
    topology = app.Topology.from_arrays(spcfw['atomname'], spcfw['resname'], spcfw['residue'])
    respa_system = atomsmm.RESPASystem(system, 7 * unit.angstroms, 5 * unit.angstroms)
    respa_system.redefine_bond(topology, 'HOH', 'H[1-2]', 'O', 1.05 * unit.angstroms)
    respa_system.redefine_angle(topology, 'HOH', 'H[1-2]', 'O', 'H[1-2]', 113 * unit.degrees)
    components = atomsmm.splitPotentialEnergy(respa_system, topology, positions)
    potential = {'HarmonicBondForce': 3665.684696323676, 'HarmonicAngleForce': 1811.197218501007,
                 'Real-Space': 84694.39953220935, 'Reciprocal-Space': -111582.71281220087,
                 'CustomNonbondedForce': -25531.129587235544, 'CustomNonbondedForce(1)': 25531.129587235544,
                 'CustomBondForce': 0.0, 'CustomBondForce(1)': -1175.253817235862, 'CustomAngleForce': -305.0221912655623,
                 'Total': -22891.707373668243}          # (the reference also lists an empty PeriodicTorsionForce: 0.0)
    _check(components, potential)


def _pressure_case(case, temperature):
    system, positions, topology = create_system(case, nonbondedMethod='PME')
    platform = openmm.Platform.getPlatformByName('Reference')
    computer = atomsmm.PressureComputer(system, topology, platform, temperature=temperature)
    context = openmm.Context(system, openmm.CustomIntegrator(0), platform)
    context.setPositions(positions)
    return computer, context


def test_pressure_with_bath_temperature(spcfw):                    # tests/test_computers.py:22-38
    computer, context = _pressure_case(spcfw, 300 * unit.kelvin)
    state = context.getState(getPositions=True, getVelocities=True, getForces=True)
    computer.import_configuration(state)
    atomic_virial = computer.get_atomic_virial()
    assert atomic_virial / atomic_virial.unit == pytest.approx(-11661.677650154408)
    atomic_pressure = computer.get_atomic_pressure()
    assert atomic_pressure / atomic_pressure.unit == pytest.approx(-58.64837784125407)
    molecular_virial = computer.get_molecular_virial(state.getForces())
    assert molecular_virial / molecular_virial.unit == pytest.approx(-5418.629781093525)
    molecular_pressure = computer.get_molecular_pressure(state.getForces())
    assert molecular_pressure / molecular_pressure.unit == pytest.approx(-554.9525554206972)


def test_pressure_with_kinetic_temperature(spcfw):                 # tests/test_computers.py:41-57
    """The virials do not depend on the velocities; the two pressures of the reference test do (OpenMM's random
    velocities, not reproducible here) and are checked against their definitions instead."""
    computer, context = _pressure_case(spcfw, None)
    context.setVelocitiesToTemperature(300 * unit.kelvin, 1234)
    state = context.getState(getPositions=True, getVelocities=True, getForces=True)
    computer.import_configuration(state)
    W = computer.get_atomic_virial()
    assert W / W.unit == pytest.approx(-11661.677650154408)
    Wm = computer.get_molecular_virial(state.getForces())
    assert Wm / Wm.unit == pytest.approx(-5418.629781093525)
    v = state.getVelocities(asNumpy=True)._value
    m = spcfw['mass']
    volume = float(np.prod(spcfw['box']))
    atm = unit.md_value(1 * unit.atmospheres)
    p_atomic = computer.get_atomic_pressure()
    assert p_atomic / p_atomic.unit == pytest.approx((float((m[:, None] * v * v).sum()) + W._value) / (3 * volume) / atm)
    vcm = (m[:, None] * v).reshape(-1, 3, 3).sum(1) / m.reshape(-1, 3).sum(1)[:, None]
    kmol = 0.5 * float((m.reshape(-1, 3).sum(1)[:, None] * vcm ** 2).sum())
    p_mol = computer.get_molecular_pressure(state.getForces())
    assert p_mol / p_mol.unit == pytest.approx((2 * kmol + Wm._value) / (3 * volume) / atm)


def test_pressure_with_exceptions(emim):                           # tests/test_computers.py:60-74
    computer, context = _pressure_case(emim, 300 * unit.kelvin)
    state = context.getState(getPositions=True, getVelocities=True, getForces=True)
    computer.import_configuration(state)
    atomic_virial = computer.get_atomic_virial()
    assert atomic_virial / atomic_virial.unit == pytest.approx(-22827.477810819175)
    atomic_pressure = computer.get_atomic_pressure()
    assert atomic_pressure / atomic_pressure.unit == pytest.approx(-282.7243180164338)
    molecular_virial = computer.get_molecular_virial(state.getForces())
    assert molecular_virial / molecular_virial.unit == pytest.approx(-23272.958585794207)
    molecular_pressure = computer.get_molecular_pressure(state.getForces())
    assert molecular_pressure / molecular_pressure.unit == pytest.approx(-3283.563262288828)


def _phenol_system(phenol):
    system, positions, topology = create_system(phenol, nonbondedMethod='PME', cutoff=1.0, switch=0.9)
    solute = set(int(i) for i in np.where(phenol['resname'] == 'aaa')[0])
    return system, positions, topology, solute


def test_AlchemicalRespaSystem(phenol):                             # tests/test_systems.py:155-181
    system, positions, topology, solute = _phenol_system(phenol)
    solvation_system = atomsmm.AlchemicalRespaSystem(system, 7 * unit.angstroms, 5 * unit.angstroms, solute,
                                                     coupling_function='lambda^4*(5-4*lambda)')
    components = atomsmm.splitPotentialEnergy(solvation_system, topology, positions, **{'lambda': 0.5, 'respa_switch': 1})
    _check(components, {'HarmonicBondForce': 2621.3223922886677, 'HarmonicAngleForce': 1525.1006876561419,
                        'PeriodicTorsionForce': 18.767576693568476, 'Real-Space': 80089.51116719692,
                        'Reciprocal-Space': -107038.52551657759, 'CustomNonbondedForce': 5037.152491649265,
                        'CustomBondForce': -53.526446723139806, 'CustomBondForce(1)': -53.374675325650806,
                        'CustomCVForce': -7.114065227572182, 'CustomCVForce(1)': -6.301336948673654,
                        'Total': -17866.987725318053})


def test_AlchemicalRespaSystem_without_middle_scale(phenol):       # tests/test_systems.py:184-208
    system, positions, topology, solute = _phenol_system(phenol)
    solvation_system = atomsmm.AlchemicalRespaSystem(system, 7 * unit.angstroms, 5 * unit.angstroms, solute,
                                                     coupling_function='lambda^4*(5-4*lambda)', middle_scale=False)
    components = atomsmm.splitPotentialEnergy(solvation_system, topology, positions, **{'lambda': 0.5})
    _check(components, {'HarmonicBondForce': 2621.3223922886677, 'HarmonicAngleForce': 1525.1006876561419,
                        'PeriodicTorsionForce': 18.767576693568476, 'Real-Space': 80089.51116719692,
                        'Reciprocal-Space': -107038.52551657759, 'CustomBondForce': -53.526446723139806,
                        'CustomCVForce': -7.114065227572182, 'Total': -22844.464204692995})


def test_AlchemicalRespaSystem_with_coulomb_scaling(phenol):       # tests/test_systems.py:211-249
    system, positions, topology, solute = _phenol_system(phenol)
    solvation_system = atomsmm.AlchemicalRespaSystem(system, 7 * unit.angstroms, 5 * unit.angstroms, solute,
                                                     coupling_function='lambda^4*(5-4*lambda)', coulomb_scaling=True,
                                                     lambda_coul=0.5)
    components = atomsmm.splitPotentialEnergy(solvation_system, topology, positions, **{'lambda': 0.5, 'respa_switch': 1})
    simulation = app.Simulation(topology, solvation_system, openmm.CustomIntegrator(0),
                                openmm.Platform.getPlatformByName("Reference"))
    simulation.context.setPositions(positions)
    force = solvation_system.get_alchemical_coul_force()
    assert force.getNumCollectiveVariables() == 1 and force.getCollectiveVariableName(0) == 'alchemical_coulomb_energy'
    before = simulation.context.getState(getEnergy=True, groups=2 ** 2).getPotentialEnergy()
    Ecoul = force.getCollectiveVariableValues(simulation.context)
    components['Ecoul'] = Ecoul[0] * unit.kilojoule_per_mole
    _check(components, {'HarmonicBondForce': 2621.3223922886677, 'HarmonicAngleForce': 1525.1006876561419,
                        'PeriodicTorsionForce': 18.767576693568476, 'Real-Space': 80078.91697014398,
                        'Reciprocal-Space': -107074.49398119976, 'CustomNonbondedForce': 5037.152491649265,
                        'CustomBondForce': -53.526446723139806, 'CustomBondForce(1)': -53.374675325650806,
                        'CustomNonbondedForce(1)': -23.447733015522058, 'CustomCVForce': -7.114065227572182,
                        'CustomCVForce(1)': -6.301336948673654, 'Total': -17936.998120008684,
                        'Ecoul': -93.14060537915793})
    # the scaling factor was put back: the outer group's energy is what it was
    again = simulation.context.getState(getEnergy=True, groups=2 ** 2).getPotentialEnergy()
    assert again / again.unit == pytest.approx(before / before.unit, rel=1e-13)


def test_AlchemicalRespaSystem_with_softcore(phenol):              # tests/test_systems.py:252-292 (system components)
    system, positions, topology, solute = _phenol_system(phenol)
    solvation_system = atomsmm.AlchemicalRespaSystem(system, 7 * unit.angstroms, 5 * unit.angstroms, solute, use_softcore=True)
    components = atomsmm.splitPotentialEnergy(solvation_system, topology, positions, **{'lambda': 0.5, 'respa_switch': 1})
    _check(components, {'HarmonicBondForce': 2621.3223922886677, 'HarmonicAngleForce': 1525.1006876561419,
                        'PeriodicTorsionForce': 18.767576693568476, 'Real-Space': 80089.51116719692,
                        'Reciprocal-Space': -107038.52551657759, 'CustomNonbondedForce': 5037.152491649265,
                        'CustomBondForce': -53.526446723139806, 'CustomBondForce(1)': -53.374675325650806,
                        'CustomNonbondedForce(1)': -24.140118811594814, 'CustomNonbondedForce(2)': -24.140118811594814,
                        'Total': -17901.852560765})
    # the collective variables E0..E5: the softcore energy at lambda = 0, 0.2, ..., 1 (systems.py:412-470)
    context = openmm.Context(system, openmm.CustomIntegrator(0), openmm.Platform.getPlatformByName('Reference'))
    context.setPositions(positions)
    force = solvation_system.get_alchemical_vdw_force([i / 5 for i in range(6)])
    values = force.getCollectiveVariableValues(context)
    assert [force.getCollectiveVariableName(i) for i in range(force.getNumCollectiveVariables())] == ['E%d' % i for i in range(6)]
    assert values == pytest.approx([0.0, -10.071581499620784, -19.66450283710424, -28.284595753200428, -35.004158250494505,
                                    -37.9416812137183])


def test_alchemical_respa_trajectory_fused_equals_unfused(phenol):
    """AlchemicalRespaSystem over a PME system, RESPA [2,2,1]: the near force (Kc = 138.935456637, systems.py:572) and the
    NonbondedForce (OpenMM's 138.935456) share a neighbour list but must NOT share a pass (the pass forms Kc q_i q_j once);
    groups 1 and 2 hold several members each.  Every fusion of amm_run_ops on == off, bit for bit."""
    system, positions, topology, solute = _phenol_system(phenol)
    results = []
    for fuse in (True, False):
        alch = atomsmm.AlchemicalRespaSystem(system, 7 * unit.angstroms, 5 * unit.angstroms, solute,
                                             coupling_function='lambda^4*(5-4*lambda)')
        integrator = atomsmm.RespaPropagator([2, 2, 1]).integrator(2 * unit.femtoseconds)
        context = openmm.Context(alch, integrator)
        context._engine.ctx.set_fuse_inner(fuse)
        context.setParameter('lambda', 0.5)
        context.setParameter('respa_switch', 1)
        context.setPositions(positions)
        context.setVelocitiesToTemperature(300 * unit.kelvin, 11)
        integrator.step(4)
        st = context.getState(getPositions=True, getVelocities=True)
        results.append((st.getPositions(asNumpy=True)._value, st.getVelocities(asNumpy=True)._value))
    assert np.array_equal(results[0][0], results[1][0]) and np.array_equal(results[0][1], results[1][1])
    assert np.isfinite(results[0][0]).all()


def test_pme_respa_trajectory_fused_equals_unfused(spcfw):
    """RESPASystem over a PME water box, RESPA [4,2,1]: group 2 holds the NonbondedForce's pair force, its exclusion terms
    and the reciprocal space; the near force rides on the pair force's pass (dual evaluation) and the other members are
    added afterwards -- every fusion of amm_run_ops on == off, bit for bit, list rebuilds included."""
    system = system_from_arrays(spcfw, nonbondedMethod='PME')
    results = []
    for fuse in (True, False):
        respa = atomsmm.RESPASystem(system, 7 * unit.angstroms, 5 * unit.angstroms)
        integrator = atomsmm.RespaPropagator([4, 2, 1]).integrator(4 * unit.femtoseconds)
        context = openmm.Context(respa, integrator)
        context._engine.ctx.set_fuse_inner(fuse)
        context.setPositions(spcfw['positions'] * unit.nanometers)
        context.setVelocitiesToTemperature(300 * unit.kelvin, 5)
        integrator.step(25)
        st = context.getState(getPositions=True, getVelocities=True)
        eng = context._engine
        results.append((st.getPositions(asNumpy=True)._value, st.getVelocities(asNumpy=True)._value,
                        eng.ctx.pair_stats(eng.pair_force_ids(2)[0])['n_builds']))
    assert np.array_equal(results[0][0], results[1][0]) and np.array_equal(results[0][1], results[1][1])
    assert results[0][2] == results[1][2] and results[0][2] >= 2          # the list was rebuilt on the way
