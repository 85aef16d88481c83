"""CPU tests of the host side: the AtomsMM-shaped API (names, strings, step programs, errors) and the
engine's translation / program unrolling, with a call recorder in place of the HIP library."""
import copy
import os
import re

import numpy as np
import pytest

import atomsmm_amd as atomsmm
from atomsmm_amd import backend as B
from atomsmm_amd import engine as E
from atomsmm_amd import forces as F
from atomsmm_amd import openmm, unit
from atomsmm_amd.openmm import app
from atomsmm_amd.testing import system_from_arrays
from fake_backend import RecordingContext


@pytest.fixture()
def recorder(monkeypatch):
    made = []

    def factory(*a, **k):
        made.append(RecordingContext(*a, **k))
        return made[-1]
    monkeypatch.setattr(E, '_context_factory', factory)
    return made


# ------------------------------------------------------------------------------------ step programs
def test_respa_program_text_matches_reference_capture(goldens):
    integ = atomsmm.RespaPropagator([4, 2, 1]).integrator(4 * unit.femtoseconds)
    g = goldens['programs']['respa_4_2_1']
    assert integ.pretty_steps() == g['steps']
    assert [integ.getPerDofVariableName(i) for i in range(integ.getNumPerDofVariables())] == g['per_dof']
    assert [integ.getGlobalVariableName(i) for i in range(integ.getNumGlobalVariables())] == g['globals']
    assert 'Computation steps:' in repr(integ) and '   8:       v <- v + (0.0625*dt)*(f0)/m' in repr(integ)


def test_constrained_respa_program_matches_reference_capture(goldens):
    nve = atomsmm.RespaPropagator([4, 1], boost=atomsmm.VelocityBoostPropagator(constrained=True),
                                  move=atomsmm.TranslationPropagator(constrained=True))
    integ = atomsmm.GlobalThermostatIntegrator(1 * unit.femtoseconds, nve)
    assert integ.pretty_steps() == goldens['programs']['respa_4_1_constrained']['steps']


def test_memory_respa_program_matches_reference_capture(goldens):
    integ = atomsmm.RespaPropagator([2, 1], has_memory=True).integrator(1 * unit.femtoseconds)
    assert integ.pretty_steps() == goldens['programs']['respa_2_1_memory']['steps']


def test_respa_switch_and_loop_of_one():
    integ = atomsmm.RespaPropagator([1, 1], use_respa_switch=True).integrator(1 * unit.femtoseconds)
    steps = integ.pretty_steps()
    assert steps[0] == 'respa_switch <- 1' and steps[-1] == 'respa_switch <- 0'
    assert not any('while' in s for s in steps)          # loop counts of 1 emit no counter (propagators.py:960-967)
    assert steps.count('allow forces to update the context state') == 1


def test_composition_propagators():
    kick, move = atomsmm.VelocityBoostPropagator(constrained=False), atomsmm.TranslationPropagator(constrained=False)
    ts = atomsmm.TrotterSuzukiPropagator(move, kick).integrator(2 * unit.femtoseconds)
    assert ts.pretty_steps()[1:] == ['v <- v + (0.5*dt)*f/m', 'x <- x + (1.0*dt)*v', 'v <- v + (0.5*dt)*f/m']
    sp = atomsmm.SplitPropagator(move, 3).integrator(1 * unit.femtoseconds)
    assert sp.pretty_steps() == ['nSplit <- 0', 'while (nSplit < 3):', '   x <- x + (0.3333333333333333*dt)*v',
                                 '   nSplit <- nSplit + 1', 'end']
    sy = atomsmm.SuzukiYoshidaPropagator(move, 3).integrator(1 * unit.femtoseconds)
    w = 1.3512071919596578
    assert sy.pretty_steps() == ['x <- x + ({}*dt)*v'.format(c) for c in (w, 1 - 2 * w, w)]
    with pytest.raises(atomsmm.InputError):
        atomsmm.SuzukiYoshidaPropagator(move, 5)
    with pytest.raises(atomsmm.InputError):
        atomsmm.RespaPropagator([2, 1], shell={5: kick})
    from atomsmm_amd.propagators import Propagator
    a, b = Propagator(), Propagator()
    a.globalVariables['k'] = 1
    b.globalVariables['k'] = 2
    with pytest.raises(atomsmm.InputError):
        atomsmm.ChainedPropagator([a, b])


def test_mts_schemes():
    bath = atomsmm.VelocityBoostPropagator(constrained=False)     # any propagator serves as a stand-in bath
    mid = atomsmm.MultipleTimeScaleIntegrator(2 * unit.femtoseconds, [2, 1], bath=bath, scheme='middle').pretty_steps()
    assert '   x <- x + (0.25*dt)*v' in mid                      # move split x, bath, x (propagators.py:970-973)
    xo = atomsmm.MultipleTimeScaleIntegrator(2 * unit.femtoseconds, [2, 1], bath=bath, scheme='xo-respa').pretty_steps()
    assert xo[1] == 'v <- v + (0.5*dt)*(f1)/m' and xo[2] == 'v <- v + (0.5*dt)*(f1)/m'
    with pytest.raises(atomsmm.InputError):
        atomsmm.MultipleTimeScaleIntegrator(2 * unit.femtoseconds, [2, 1], scheme='nope')
    with pytest.raises(atomsmm.InputError):
        atomsmm.RespaPropagator([2]).integrator(1 * unit.femtoseconds).addComputeGlobal('mvv', '1')


# ------------------------------------------------------------------------------------ energy strings
def test_energy_strings_match_reference_captures():
    """SURVEY.md Appendix C.5."""
    rc, rs = 7 * unit.angstroms, 5 * unit.angstroms
    lines = repr(atomsmm.NearNonbondedForce(rc, rs, None)).split('\n')
    assert lines == ['S*(4*epsilon*((sigma/r)^12-(sigma/r)^6) + Kc*chargeprod/r)',
                     'S = 1 + step(r - rs0)*u^3*(15*u - 6*u^2 - 10)', 'u=(r-rs0)/(rc0-rs0)']
    lines = repr(atomsmm.NearNonbondedForce(rc, rs, 'shift')).split('\n')
    assert lines[0] == ('S*(4*epsilon*((sigma/r)^12-(sigma/r)^6-((sigma/rc0)^12-(sigma/rc0)^6))'
                        '+Kc*chargeprod*(1/r-1/rc0))')
    fs = atomsmm.NearNonbondedForce(rc, rs, 'force-switch')
    lines = repr(fs).split('\n')
    assert lines[0] == ('4*epsilon*(f12*(sigma/r)^12-f6*(sigma/r)^6) + Kc*chargeprod*f1/r-(4*epsilon*'
                        '(f12c*(sigma/rc0)^12-f6c*(sigma/rc0)^6) + Kc*chargeprod*f1c/rc0)')
    assert lines[4] == 'R=u/b+1' and lines[5] == 'b=2.5' and lines[-1] == 'u=(r-rs0)/(rc0-rs0)'
    consts = {l.split('=')[0]: float(l.split('=')[1]) for l in lines[5:9]}
    assert consts['f12c'] == pytest.approx(8.685770457212126, rel=1e-14)
    assert consts['f6c'] == pytest.approx(2.744, rel=1e-14)
    assert consts['f1c'] == pytest.approx(1.171339712719476, rel=1e-12)
    assert fs.getGlobalParameters() == {'Kc': 138.935456, 'rc0': pytest.approx(0.7), 'rs0': pytest.approx(0.5)}
    assert repr(atomsmm.NearNonbondedForce(10 * unit.angstroms, 9.5 * unit.angstroms, 'force-switch')).split('\n')[5] == 'b=19.0'
    d1 = atomsmm.DampedSmoothedForce(0.29 / unit.angstroms, 10 * unit.angstroms, 9.5 * unit.angstroms)
    assert d1.getEnergyFunction() == '4*epsilon*((sigma/r)^12 - (sigma/r)^6) + erfc(alpha*r)*Kc*chargeprod/r'
    assert d1.getUseSwitchingFunction() and d1.getSwitchingDistance() / unit.nanometer == pytest.approx(0.95)
    assert not d1.getUseLongRangeCorrection()
    d2 = atomsmm.DampedSmoothedForce(0.29 / unit.angstroms, 10 * unit.angstroms, 9.5 * unit.angstroms, degree=2)
    assert repr(d2).split('\n') == ['S*(4*epsilon*((sigma/r)^12 - (sigma/r)^6) + erfc(alpha*r)*Kc*chargeprod/r)',
                                    'S = 1 + step(r - rswitch)*u^3*(15*u - 6*u^2 - 10)',
                                    'u = (r^d - rswitch^d)/(rcut^d - rswitch^d)', 'd=2']
    assert not d2.getUseSwitchingFunction()
    assert atomsmm.NonbondedExceptionsForce().getEnergyFunction() == '4*epsilon*x*(x-1) + Kc*chargeprod/r; x=(sigma/r)^6'
    sub = atomsmm.NearNonbondedForce(rc, rs, None, subtract=True, actual_cutoff=10 * unit.angstroms)
    assert sub.getEnergyFunction().startswith('-(step(rc0-r)*(S*(')
    assert sub.getCutoffDistance() / unit.nanometer == pytest.approx(1.0)
    exp = F.nearForceExpressions(rc, rs, 'force-switch')
    assert exp[-3:] == ['rs0=0.5', 'rc0={}'.format(unit.md_value(rc)), 'Kc=138.935456']


def test_input_errors():
    with pytest.raises(atomsmm.InputError):
        atomsmm.DampedSmoothedForce(0.29 / unit.angstroms, 10 * unit.angstroms, 10 * unit.angstroms)
    with pytest.raises(atomsmm.InputError):
        atomsmm.NearNonbondedForce(10 * unit.angstroms, 9 * unit.angstroms, 'bogus')
    with pytest.raises(atomsmm.InputError):
        atomsmm.FarNonbondedForce(atomsmm.NonbondedExceptionsForce(), 10 * unit.angstroms)
    assert str(atomsmm.InputError('x')) == '\033[1;31mx\033[0m'


@pytest.mark.parametrize('adj', [None, 'shift', 'force-switch'])
def test_describe_energy_recognises_reference_strings(adj):
    rc, rs = 7 * unit.angstroms, 5 * unit.angstroms
    near = atomsmm.NearNonbondedForce(rc, rs, adj)
    d = F.describe_energy(near.getEnergyFunction(), near.getGlobalParameters())
    assert d['family'] == near._amm['family'] and d['sign'] == 1.0 and not d['guard']
    assert d['rc0'] == pytest.approx(0.7) and d['rs0'] == pytest.approx(0.5)
    far = atomsmm.FarNonbondedForce(near, 10 * unit.angstroms, 9 * unit.angstroms)
    dd = F.describe_energy(far[1].getEnergyFunction(), far[1].getGlobalParameters())
    assert dd['family'] == d['family'] and dd['sign'] == -1.0 and dd['guard']
    assert far[1]._amm['sign'] == -1.0 and far[1]._amm['guard']
    respa_strings = ';'.join(F.nearForceExpressions(rc, rs, adj))
    dr = F.describe_energy(respa_strings)
    assert dr['family'] == d['family'] and dr['rc0'] == pytest.approx(0.7) and dr['Kc'] == 138.935456
    minus = respa_strings.split(';')
    minus[0] = '-step(rc0-r)*({})'.format(minus[0])
    dm = F.describe_energy(';'.join(minus))
    assert dm['sign'] == -1.0 and dm['guard']
    assert F.describe_energy('k*r^2') is None


# ------------------------------------------------------------------------------------ systems / forces plumbing
def water_system(spcfw, **kw):
    return system_from_arrays(spcfw, **kw)


def test_import_from_and_compound_force(spcfw):
    system = water_system(spcfw, nonbondedMethod='PME', flexible=False)
    n_forces = system.getNumForces()
    nb = atomsmm.hijackForce(system, atomsmm.findNonbondedForce(system))
    assert system.getNumForces() == n_forces - 1
    near = atomsmm.NearNonbondedForce(7 * unit.angstroms, 6.5 * unit.angstroms, 'shift')
    assert near.importFrom(nb).addTo(system) is near
    assert near.getNumParticles() == 1536 and near.getNumExclusions() == 1536
    assert near.getNonbondedMethod() == openmm.CustomNonbondedForce.CutoffPeriodic
    assert near.getEnergyFunction().endswith(';chargeprod = (charge1)*(charge2);sigma = 0.5*(sigma1+sigma2);'
                                             'epsilon = sqrt((epsilon1)*(epsilon2))')
    far = atomsmm.FarNonbondedForce(near, 10 * unit.angstroms, 9.5 * unit.angstroms).setForceGroup(2)
    assert isinstance(far, atomsmm.FarNonbondedForce) and far.getForceGroup() == 2
    far.importFrom(nb).addTo(system)
    total, discount = far[0], far[1]
    assert [f for f in far] == [total, discount]
    assert total.getNonbondedMethod() == openmm.NonbondedForce.PME and total.getNumExceptions() == 1536
    assert all(total.getExceptionParameters(i)[2] / unit.elementary_charge ** 2 == 0 for i in range(5))
    assert discount.getCutoffDistance() / unit.nanometer == pytest.approx(1.0)
    assert system.getNumForces() == n_forces + 2
    far.enableExceptions()
    assert len(far.forces) == 3 and far[2].getForceGroup() == 2
    assert atomsmm.countDegreesOfFreedom(system) == 3 * 1536 - 3


def test_respa_system_groups(spcfw):
    system = water_system(spcfw, nonbondedMethod='PME', switch=0.9)
    respa = atomsmm.RESPASystem(system, 7 * unit.angstroms, 5 * unit.angstroms)
    assert system.getNumForces() == 3                       # input untouched (deep copy, systems.py:63)
    groups = [(f.__class__.__name__, f.getForceGroup()) for f in respa.getForces()]
    assert groups == [('HarmonicBondForce', 0), ('HarmonicAngleForce', 0), ('NonbondedForce', 2),
                      ('_AtomsMM_CustomNonbondedForce', 1), ('_AtomsMM_CustomNonbondedForce', 31),
                      ('_AtomsMM_CustomBondForce', 0)]
    nb = respa.getForce(2)
    assert nb.getReciprocalSpaceForceGroup() == 2
    near, minus, exc = respa.getForce(3), respa.getForce(4), respa.getForce(5)
    assert near._amm['family'] == 'near-force-switch' and minus._amm['sign'] == -1.0 and minus._amm['guard']
    assert minus.getEnergyFunction().startswith('-step(rc0-r)*(4*epsilon*(f12*')
    assert exc.getNumBonds() == 1536 and exc.getEnergyFunction().startswith('4*epsilon*x*(x-1) + Kc*chargeprod/r;x=(sigma/r)^6')
    # fastExceptions extracted (zeroed) the exceptions of the NonbondedForce (forces.py:387-388)
    i, j, qq, s, e = nb.getExceptionParameters(0)
    assert (qq / qq.unit, s / s.unit, e / e.unit) == (0.0, 1.0, 0.0)
    slow = atomsmm.RESPASystem(system, 7 * unit.angstroms, 5 * unit.angstroms, fastExceptions=False)
    assert [f.getForceGroup() for f in slow.getForces()] == [0, 0, 2, 1, 31, 1, 31]


def test_unit_module():
    a = 10 * unit.angstroms
    assert a.value_in_unit(unit.nanometers) == pytest.approx(1.0) and a / a.unit == 10
    assert (0.29 / unit.angstroms).value_in_unit(unit.nanometer ** -1) == pytest.approx(2.9)
    assert 9.5 * unit.angstroms < a and a >= 1.0 * unit.nanometer
    assert (4 * unit.femtoseconds)._md() == pytest.approx(0.004)
    # kB*NA of simtk.unit in the OpenMM 7.x behind the reference's literals (CODATA 2006), pinned by tests/test_computers.py
    assert (atomsmm.utils.kB * 300 * unit.kelvin)._md() == pytest.approx(1.3806504e-23 * 6.02214179e23 * 0.3, rel=1e-14)
    with pytest.raises(TypeError):
        a + 1.0
    with pytest.raises(TypeError):
        a.value_in_unit(unit.picosecond)


# ------------------------------------------------------------------------------------ engine (host logic, recorder)
def respa_context(spcfw, recorder, loops=(4, 2, 1), outer='damped'):
    system = water_system(spcfw, nonbondedMethod='CutoffPeriodic')
    respa = atomsmm.RESPASystem(system, 7 * unit.angstroms, 5 * unit.angstroms)
    if outer == 'damped':       # composition recipe of SURVEY.md 8d C1
        nb = atomsmm.hijackForce(respa, atomsmm.findNonbondedForce(respa))
        f = atomsmm.DampedSmoothedForce(0.29 / unit.angstroms, 10 * unit.angstroms, 9 * unit.angstroms).importFrom(nb)
        f.setForceGroup(2)
        f.addTo(respa)
    integ = atomsmm.RespaPropagator(list(loops)).integrator(4 * unit.femtoseconds)
    sim = app.Simulation(app.Topology(), respa, integ, openmm.Platform.getPlatformByName('HIP'))
    sim.context.setPositions(spcfw['positions'] * unit.nanometers)
    return sim, recorder[-1]


def test_engine_translation(spcfw, recorder):
    sim, rec = respa_context(spcfw, recorder)
    fams = sorted((p['family'], p['sign'], p['rc'], bool(p['flags'] & B.GUARD_RC0)) for p in rec.pairs)
    assert fams == [(B.NEAR_FSWITCH, -1.0, pytest.approx(0.7), True), (B.NEAR_FSWITCH, 1.0, pytest.approx(0.7), False),
                    (B.DAMPED, 1.0, pytest.approx(1.0), False)]
    damped = [p for p in rec.pairs if p['family'] == B.DAMPED][0]
    assert damped['alpha'] == pytest.approx(2.9) and damped['rswitch'] == pytest.approx(0.9) and damped['degree'] == 1
    assert damped['n_excl'] == 1536 and damped['q'][1] == pytest.approx(-0.84)
    kinds = sorted(t[0] for b in rec.bonded for t in b['terms'])
    assert kinds == [B.BOND_HARMONIC, B.ANGLE_HARMONIC, B.BOND_LJC]
    # both near copies (groups 1 and 31) traverse the damped force's neighbour list
    shares = [c for c in rec.calls if c[0] == 'pair_share_list']
    assert len(shares) == 2 and {c[2] for c in shares} == {damped['id']}


def test_engine_unrolls_respa_with_force_caches(spcfw, recorder):
    sim, rec = respa_context(spcfw, recorder)
    sim.step(5)
    assert len(rec.runs) == 2 and rec.runs[0][1] == 1 and rec.runs[1][1] == 4   # first step, then steady state x4
    first, steady = rec.runs[0][0], rec.runs[1][0]
    g = {grp: slot for grp, (slot, ids) in rec.groups.items()}

    def evals(ops):
        return [o[1] for o in ops if o[0] == B.OP_EVAL]
    # steady state: 1 / 2 / 8 evaluations of groups 2 / 1 / 0, first step one more of f2 and f1
    assert sorted(evals(steady)) == [0] * 8 + [1] * 2 + [2]
    assert sorted(evals(first)) == [0] * 9 + [1] * 3 + [2] * 2
    kicks = [o for o in steady if o[0] == B.OP_KICK]
    moves = [o for o in steady if o[0] == B.OP_MOVE]
    assert len(kicks) == 2 + 4 + 16 and len(moves) == 8
    assert moves[0][4] == pytest.approx(0.125 * 0.004)
    # `_f2_ <- f2 ; v <- v + (0.5*dt)*(_f2_-f1)/m` (integrators.py:134-145): while _f2_ mirrors f2 the kick reads the group's
    # buffer itself, and the copy -- left without a reader -- is dropped from the op list
    assert kicks[0][1] == g[2] and kicks[0][2] == g[1] and kicks[0][3] == 0 and kicks[0][4] == pytest.approx(0.5 * 0.004)
    assert not [o for o in steady if o[0] == B.OP_COPY]
    # group 0 = one merged bonded set (bonds + angles + exceptions), groups 1/2 = one pair force each
    assert len(rec.groups[0][1]) == 1 and len(rec.groups[1][1]) == 1 and len(rec.groups[2][1]) == 1
    integ = sim.integrator
    assert integ.getGlobalVariableByName('n0RESPA') == 4 and integ.getGlobalVariableByName('n1RESPA') == 2
    assert integ.getGlobalVariableByName('NDOF') == 3 * 1536
    # new positions invalidate the caches: next step is a 'first' step again
    sim.context.setPositions(spcfw['positions'] * unit.nanometers)
    sim.step(1)
    assert sorted(evals(rec.runs[-1][0])) == [0] * 9 + [1] * 3 + [2] * 2


def test_engine_memory_program_and_unsupported_steps(spcfw, recorder):
    system = water_system(spcfw, nonbondedMethod='CutoffPeriodic')
    respa = atomsmm.RESPASystem(system, 7 * unit.angstroms, 5 * unit.angstroms)
    nb = respa.getForce(atomsmm.findNonbondedForce(respa))
    nb.setForceGroup(1)            # two time scales: 0 bonded, 1 everything else... keep near in 1 as well
    integ = atomsmm.RespaPropagator([2, 1], has_memory=True).integrator(1 * unit.femtoseconds)
    ctx = openmm.Context(respa, integ)
    ctx.setPositions(spcfw['positions'])
    integ.step(1)
    ops = recorder[-1].runs[0][0]
    assert [o[0] for o in ops if o[0] in (B.OP_COPY, B.OP_KICK)][0] == B.OP_COPY            # fm1 <- f1
    plus = [o for o in ops if o[0] == B.OP_KICK and o[3] == 1]
    assert len(plus) == 4                                                                    # (f0+fm1) kicks
    assert [o for o in ops if o[0] == B.OP_KICK][-1][3] == 0                                 # (f1-fm1)
    # constrained propagators on a constraint-free system: `v <- (x - x0)/((1.0*dt)*...)` is no kick or move, so the
    # program takes the general path (host-walked steps, per-DOF expressions compiled for amm_expr_eval)
    nve = atomsmm.RespaPropagator([2, 1], move=atomsmm.TranslationPropagator(constrained=True))
    integ2 = atomsmm.GlobalThermostatIntegrator(1 * unit.femtoseconds, nve)
    ctx2 = openmm.Context(respa, integ2)
    ctx2.setPositions(spcfw['positions'])
    integ2.step(1)
    assert ctx2._engine._interpreted is False       # static control flow: unrolled, the odd assignment as an EXPR op
    ops2 = recorder[-1].runs[0][0]
    assert sum(1 for o in ops2 if o[0] == B.OP_EXPR) == 2 and len(recorder[-1].exprs) == 1
    # a ComputeSum (mvv) makes the globals data dependent: general, host-walked path
    nh = atomsmm.NoseHooverPropagator(300 * unit.kelvin, 4605, 10 * unit.femtoseconds)
    integ3 = atomsmm.GlobalThermostatIntegrator(1 * unit.femtoseconds, atomsmm.UnconstrainedVelocityVerletPropagator(), nh)
    ctx3 = openmm.Context(respa, integ3)
    ctx3.setPositions(spcfw['positions'])
    integ3.step(1)
    assert ctx3._engine._interpreted is True
    sums = [c for c in recorder[-1].calls if c[0] == 'expr_eval' and c[6]]
    assert len(sums) == 2                             # mvv before each half thermostat step


def test_engine_rejects_what_it_cannot_run(spcfw, recorder):
    system = water_system(spcfw, nonbondedMethod='PME')
    integ = atomsmm.RespaPropagator([2, 2, 1]).integrator(1 * unit.femtoseconds)
    respa = atomsmm.RESPASystem(system, 7 * unit.angstroms, 5 * unit.angstroms)
    ctx = openmm.Context(respa, integ)
    ctx.setPositions(spcfw['positions'])
    # PME reciprocal space: one mesh object, OpenMM's grid rule, evaluated with the group-2 force
    rec = ctx._engine.ctx
    assert len(rec.pme) == 1
    alpha = np.sqrt(-np.log(2 * 5e-4)) / 1.0
    assert rec.pme[0]['alpha'] == pytest.approx(alpha)
    assert rec.pme[0]['grid'] == [int(np.ceil(2 * alpha * 2.5 / (3 * 5e-4 ** 0.2)))] * 3 == [21, 21, 21]
    integ.step(1)
    index2, ids2 = [v for k, v in rec.groups.items() if k == 2][0]
    assert rec.pme[0]['id'] in ids2
    custom = openmm.CustomNonbondedForce('k*r^2')
    custom.setNonbondedMethod(custom.CutoffPeriodic)
    for _ in range(1536):
        custom.addParticle([])
    bad = copy.deepcopy(system)
    bad.addForce(custom)
    with pytest.raises(atomsmm.InputError, match='not recognised'):
        openmm.Context(bad, openmm.VerletIntegrator(0.0))
    nobox = openmm.System()
    nobox.addParticle(1.0)
    with pytest.raises(atomsmm.InputError):
        openmm.Context(nobox, openmm.VerletIntegrator(0.0))
    with pytest.raises(openmm.OpenMMException):
        openmm.VerletIntegrator(0.0).step(1)


def test_parameter_offsets_update_backend(heaq, recorder):
    """SolvationSystem-style charge offsets (systems.py:303-311): q_eff = q + lambda_coul*chargeScale."""
    system = system_from_arrays(heaq, nonbondedMethod='CutoffPeriodic')
    nb = system.getForce(atomsmm.findNonbondedForce(system))
    solute = np.where(heaq['resname'] == 'aaa')[0]
    nb.addGlobalParameter('lambda_coul', 1.0)
    for i in solute:
        q, s, e = nb.getParticleParameters(int(i))
        nb.setParticleParameters(int(i), 0.0, 0.0, 0.0)
        nb.addParticleParameterOffset('lambda_coul', int(i), q, 0.0, 0.0)
    respa = atomsmm.RESPASystem(system, 7 * unit.angstroms, 5 * unit.angstroms)
    near = respa.getForce(respa.getNumForces() - 3)
    assert 'charge1+lambda_coul*chargeScale_lambda_coul1' in near.getEnergyFunction()
    assert near.getNumPerParticleParameters() == 6 and near.getGlobalParameters()['lambda_coul'] == 1.0
    ctx = openmm.Context(respa, openmm.VerletIntegrator(0.0))
    rec = recorder[-1]
    i0 = int(solute[0])
    assert rec.pairs[1]['q'][i0] == pytest.approx(heaq['charge'][i0])
    ctx.setParameter('lambda_coul', 0.5)
    updates = [c for c in rec.calls if c[0] == 'pair_set_params']
    assert len(updates) == 3 and all(u[2][i0] == pytest.approx(0.5 * heaq['charge'][i0]) for u in updates)
    assert ctx.getParameter('lambda_coul') == 0.5
    with pytest.raises(openmm.OpenMMException):
        ctx.setParameter('nope', 1.0)
    # the bond-list sets that a parameter change replaces are freed: the number of live sets does not grow
    # (the plain system: its NonbondedForce keeps the exceptions, whose terms are rebuilt with the charges)
    ctx = openmm.Context(system, openmm.VerletIntegrator(0.0))
    rec = recorder[-1]
    live = len(rec.bonded)
    for value in (0.25, 0.75, 0.5):
        ctx.setParameter('lambda_coul', value)
    assert len(rec.bonded) == live and sum(c[0] == 'bonded_release' for c in rec.calls) >= 3


def test_distributed_markers(spcfw, recorder, monkeypatch):
    """world = 2: EVAL of a group holding a pair force is followed by an all-reduce of its buffer; group 0
    (bond lists only, computed redundantly) is not reduced; bond lists sharing a reduced group are sliced."""
    import torch.distributed as dist
    monkeypatch.setattr(dist, 'is_initialized', lambda: True)
    monkeypatch.setattr(dist, 'get_world_size', lambda *a: 2)
    monkeypatch.setattr(dist, 'get_rank', lambda *a: 1)
    monkeypatch.setattr(dist, 'get_backend', lambda *a: 'gloo')
    reduced = []
    # (the int32 all-reduce is the collective amm_check of engine._check, not a force exchange)
    monkeypatch.setattr(dist, 'all_reduce', lambda t, *a, **k: reduced.append(t.data_ptr()) if t.dtype.is_floating_point else None)
    sim, rec = respa_context(spcfw, recorder)
    assert rec.rank == 1 and rec.world == 2
    sim.step(2)
    eng = sim.context._engine
    f1, f2 = eng._buffers['f1'].data_ptr(), eng._buffers['f2'].data_ptr()
    assert reduced == [f2, f1, f1, f1, f2] + [f1, f1, f2]
    assert all(not b['sliced'] for b in rec.bonded)            # group 0 holds no pair force -> redundant, unsliced


# ------------------------------------------------------------------------------------ SolvationSystem (systems.py:240-313)
def test_solvation_system_structure_and_softcore_translation(heaq, recorder, goldens):
    from oracle import oracle as O
    system = system_from_arrays(heaq, nonbondedMethod='PME', cutoff=1.0, switch=0.9)
    solute = set(int(i) for i in np.where(heaq['resname'] == 'aaa')[0])
    n_exc0 = system.getForce(atomsmm.findNonbondedForce(system)).getNumExceptions()
    solv = atomsmm.SolvationSystem(system, solute)
    nb = solv.getForce(atomsmm.findNonbondedForce(solv))
    soft = [f for f in solv.getForces() if isinstance(f, openmm.CustomNonbondedForce)]
    assert len(soft) == 1 and soft[0].getNumInteractionGroups() == 1
    assert soft[0].getEnergyFunction().startswith('4*lambda_vdw*epsilon*(1-x)/x^2; x=(r/sigma)^6+0.5*(1-lambda_vdw)')
    assert soft[0].getUseSwitchingFunction() and soft[0].getUseLongRangeCorrection()       # imported (forces.py:284-291)
    assert soft[0].getCutoffDistance() == 1.0 * unit.nanometers
    # every solute-solute pair is an exception now, and an exclusion of the softcore force
    assert nb.getNumExceptions() == len(heaq['exc_pairs']) + (len(solute) * (len(solute) - 1) // 2
                                                            - sum(1 for a, b in heaq['exc_pairs'] if a in solute and b in solute))
    assert nb.getNumExceptions() > n_exc0 and soft[0].getNumExclusions() == nb.getNumExceptions()
    for i in solute:
        q, s, e = nb.getParticleParameters(i)
        assert (q._value, s._value, e._value) == (0.0, 0.0, 0.0)
    assert nb.getNumParticleParameterOffsets() == sum(1 for i in solute if heaq['charge'][i] != 0.0)
    # use_softcore=False: sigma / epsilon offsets instead of the softcore force
    solv2 = atomsmm.SolvationSystem(system, solute, use_softcore=False)
    assert not [f for f in solv2.getForces() if isinstance(f, openmm.CustomNonbondedForce)]
    nb2 = solv2.getForce(atomsmm.findNonbondedForce(solv2))
    assert {nb2.getParticleParameterOffset(k)[0] for k in range(nb2.getNumParticleParameterOffsets())} == {'lambda_coul', 'lambda_vdw'}
    # engine translation: one softcore pair force with group codes in the charge slot, lambda in alpha, and the
    # host-side long-range correction == the oracle's (scipy) == what the reference literal needs
    ctx = openmm.Context(solv, openmm.VerletIntegrator(0.0))
    rec = recorder[-1]
    sc = [p for p in rec.pairs if p['family'] == B.SOFTCORE]
    assert len(sc) == 1 and sc[0]['alpha'] == 1.0 and sc[0]['rswitch'] == 0.9 and sc[0]['flags'] & B.SWITCH
    codes = np.where(heaq['resname'] == 'aaa', 1.0, 2.0)
    assert np.array_equal(sc[0]['q'], codes)
    ctx.setParameter('lambda_vdw', 0.5)
    assert ('pair_set_lambda', sc[0]['id'], 0.5) in rec.calls
    entry = [e for e in ctx._engine.entries if sc[0]['id'] in e.pair_ids][0]
    lrc_oracle = O.softcore_lrc(heaq['sigma'], heaq['epsilon'], codes, heaq['box'], 1.0, 0.9, 0.5)
    # the product reproduces OpenMM's own quadrature (1e-5 stopping rule), the oracle integrates exactly
    assert entry.constant == pytest.approx(lrc_oracle, rel=1e-6)
    d = O.desc(O.SOFTCORE, rc=1.0, rswitch=0.9, alpha=0.5, flags=O.SWITCH, Kc=1.0)
    e_pair = O.pair_eval(d, heaq['positions'], heaq['box'], codes, heaq['sigma'], heaq['epsilon'], heaq['exc_pairs'],
                         want_forces=False)[0]
    assert e_pair + entry.constant == pytest.approx(goldens['G15']['value'], rel=1e-9)


def test_redefine_bond_and_angle(spcfw, recorder):
    system = system_from_arrays(spcfw, nonbondedMethod='PME', cutoff=1.0, switch=0.9)
    topology = app.Topology.from_arrays(spcfw['atomname'], spcfw['resname'], spcfw['residue'])
    respa = atomsmm.RESPASystem(system, 7 * unit.angstroms, 5 * unit.angstroms)
    n0 = respa.getNumForces()
    respa.redefine_bond(topology, 'HOH', 'H[1-2]', 'O', 1.05 * unit.angstroms)
    respa.redefine_angle(topology, 'HOH', 'H[1-2]', 'O', 'H[1-2]', 113 * unit.degrees)
    assert respa.getNumForces() == n0 + 2
    bond = [f for f in respa.getForces() if isinstance(f, openmm.HarmonicBondForce)][0]
    assert all(bond.getBondParameters(k)[2] == 1.05 * unit.angstroms for k in range(bond.getNumBonds()))
    special = respa.getForce(n0)
    assert special.getEnergyFunction() == '0.5*(K0*(r - r0)^2 - Kn*(r - rn)^2)' and special.getForceGroup() == 1
    assert special.getNumBonds() == bond.getNumBonds() == 1024
    i, j, (r0, K0, rn, Kn) = special.getBondParameters(0)
    assert (r0, rn, K0) == (pytest.approx(spcfw['bond_r0'][0]), pytest.approx(0.105), Kn)
    angle_force = respa.getForce(n0 + 1)
    assert isinstance(angle_force, openmm.CustomAngleForce) and angle_force.getNumAngles() == 512
    openmm.Context(respa, openmm.VerletIntegrator(0.0))
    rec = recorder[-1]
    kinds = sorted(t[0] for b in rec.bonded for t in b['terms'])
    assert kinds.count(B.BOND_HARMONIC) == 3 and kinds.count(B.ANGLE_HARMONIC) == 3      # original + (K0, -Kn) pair each
    with pytest.raises(ValueError):
        respa.redefine_bond(app.Topology(1536), 'HOH', 'H[1-2]', 'O', 1.05 * unit.angstroms)


def test_afed_program_matches_reference_capture(goldens):
    """AdiabaticDynamicsIntegrator over RespaPropagator([2,1]) with one ExtendedSystemVariable (integrators.py:642-860):
    the 44-step program captured from the reference."""
    inner = atomsmm.RespaPropagator([2, 1]).integrator(1 * unit.femtoseconds)
    lam = atomsmm.ExtendedSystemVariable('lambda_vdw', 1000, 5, 40 * unit.femtoseconds)
    integ = atomsmm.AdiabaticDynamicsIntegrator(inner, 2, [lam])
    g = goldens['programs']['afed_respa_2_1_nsteps_2']
    assert integ.pretty_steps() == g['steps']
    names = [integ.getGlobalVariableName(i) for i in range(integ.getNumGlobalVariables())]
    assert names == ['mvv', 'NDOF'] + g['globals']
    assert integ.getStepSize() == 4 * unit.femtoseconds
    assert integ.getGlobalVariableByName('_Q_eta_lambda_vdw') == pytest.approx(5 * 0.04 ** 2)
    # Langevin bath on lambda and periodic boundaries: the other two branches of integrators.py:701-733
    lam2 = atomsmm.ExtendedSystemVariable('lambda_vdw', 1000, 5, 40 * unit.femtoseconds, periodic=True, thermostat='Langevin')
    text = '\n'.join(atomsmm.AdiabaticDynamicsIntegrator(inner, 1, [lam2]).pretty_steps())
    assert 'lambda_vdw <- lambda_vdw + select(step(lambda_vdw-(0)),-1,1)' in text
    assert '_v_lambda_vdw <- z*_v_lambda_vdw+sqrt((1-z*z)*_kTbym_lambda_vdw)*gaussian; z=exp(-dt*_gamma_lambda_vdw)' in text
    assert '_nsteps_counter' not in text


def test_alchemical_respa_coulomb_scaling_host_logic(phenol, recorder):
    """AlchemicalRespaSystem(coulomb_scaling=True) (systems.py:686-708, 794-815, 848-856): the force-switched
    electrostatic force over the (solute, solvent) group, its translation (Coulomb-only interaction group: the sigma
    slot carries twice the set code), and reset_coulomb_scaling_factor through updateParametersInContext."""
    system = system_from_arrays(phenol, nonbondedMethod='PME', cutoff=1.0, switch=0.9)
    solute = [int(i) for i in np.where(phenol['resname'] == 'aaa')[0]]
    assert len(solute) == 13
    sys0 = atomsmm.AlchemicalRespaSystem(system, 7 * unit.angstroms, 5 * unit.angstroms, solute, coulomb_scaling=True)
    q0 = phenol['charge'][solute[0]]
    # reference quirk (systems.py:781-783, 806): with the default lambda_coul = 0 nothing is reset, and the short-ranged
    # force keeps the charges it was imported with
    assert sys0._switched_coulomb_force.getParticleParameters(solute[0])[0] == pytest.approx(q0)
    assert sys0._nonbonded_force.getParticleParameters(solute[0])[0]._value == 0.0
    alch = atomsmm.AlchemicalRespaSystem(system, 7 * unit.angstroms, 5 * unit.angstroms, solute, coulomb_scaling=True,
                                         lambda_coul=0.5)
    fsep = alch._switched_coulomb_force
    text = fsep.getEnergyFunction()
    assert text.startswith('respa_switch*(1 + step(r-0.5)*f1)*138.935456637*chargeprod/r; f1 = ')
    d = F.describe_energy(text, {"respa_switch": 0})
    assert d['family'] == 'near-force-switch' and d['coulomb_only'] and d['noshift'] and d['scale_name'] == 'respa_switch'
    assert d['rs0'] == 0.5 and d['Kc'] == 138.935456637
    assert fsep.getForceGroup() == 1 and fsep.getNumInteractionGroups() == 1
    assert fsep.getParticleParameters(solute[0])[0] == pytest.approx(0.5 * q0)
    assert alch._nonbonded_force.getParticleParameters(solute[0])[0]._value == pytest.approx(0.5 * q0)
    ctx = openmm.Context(alch, openmm.VerletIntegrator(0.0))
    rec = recorder[-1]
    mine = [p for p in rec.pairs if p['flags'] & B.GROUP_Q]
    assert len(mine) == 1 and mine[0]['flags'] & B.NO_SHIFT
    codes = mine[0]['sigma'] / 2
    assert set(np.where(codes == 1.0)[0]) == set(solute) and (codes[[i for i in range(len(codes)) if i not in solute]] == 2.0).all()
    assert (mine[0]['eps'] == 0.0).all() and mine[0]['q'][solute[0]] == pytest.approx(0.5 * q0)
    n_before = len([c for c in rec.calls if c[0] == 'pair_set_params'])
    alch.reset_coulomb_scaling_factor(1.0, ctx)
    updates = [c for c in rec.calls if c[0] == 'pair_set_params'][n_before:]
    assert len(updates) == 2 and all(u[2][solute[0]] == pytest.approx(q0) for u in updates)
    assert [c for c in rec.calls if c[0] == 'pme_set_charges'][-1][2][solute[0]] == pytest.approx(q0)
    with pytest.raises(openmm.OpenMMException):
        openmm.NonbondedForce().updateParametersInContext(ctx)


def test_sin_r_program_structure():
    """SIN_R_Integrator (integrators.py:358-416): the step program that SIN_R_Propagator / MassiveIsokineticPropagator
    (propagators.py:276-355, 1045-1105) emit -- isokinetic kicks in place of the RESPA boosts, the bath Trotter-split
    around the force-independent isokinetic step in the middle of the innermost loop."""
    integ = atomsmm.SIN_R_Integrator(2 * unit.femtoseconds, [2, 2, 1], 300 * unit.kelvin, 10 * unit.femtoseconds,
                                     1 / unit.picoseconds, L=2)
    text = [line.split(': ', 1)[1].strip() for line in repr(integ).split('Computation steps:')[1].strip().split('\n')]
    kT = unit.BOLTZMANN_CONSTANT_kB._value * 300
    assert integ.getGlobalVariableByName('LkT') == pytest.approx(2 * kT)
    assert integ.getGlobalVariableByName('Q1') == pytest.approx(kT * 0.01 ** 2) == integ.getGlobalVariableByName('Q2')
    assert text[2] == 'v <- v*cosh(z) + sqrt(LkT/m)*sinh(z); z = (0.5*dt)*(_f2_-f1)/sqrt(m*LkT)'
    rescale = ['H <- sqrt(LkT/(m*v^2 + 0.6666666666666666*Q1*(v1_0^2+v1_1^2)))', 'v <- H*v', 'v1_0 <- H*v1_0', 'v1_1 <- H*v1_1']
    assert text[3:7] == rescale
    inner = text.index('while (n0RESPA < 2):')
    body = text[inner + 1:text.index('n0RESPA <- n0RESPA + 1')]
    assert body[0] == 'v <- v*cosh(z) + sqrt(LkT/m)*sinh(z); z = (0.125*dt)*(f0)/sqrt(m*LkT)' and body[1:5] == rescale
    assert body[5] == 'x <- x + (0.125*dt)*v'
    assert body[6:8] == ['v1_0 <- v1_0*exp(-(0.125*dt)*v2_0)', 'v1_1 <- v1_1*exp(-(0.125*dt)*v2_1)'] and body[8:12] == rescale
    assert body[12] == ('v2_0 <- z*v2_0 + sqrt(kT*(1 - z*z)/mass)*gaussian + force*(1 - z)/(mass*friction); '
                        'force = Q1*v1_0^2 - kT; mass = Q2; z = exp(-(0.25*dt)*friction)')
    assert body[14:16] == body[6:8] and body[16:20] == rescale and body[20] == body[5] and body[21] == body[0]
    assert len(body) == 26 and len(text) == 57
    # split=True: the drive of v2 becomes a separate boost, Trotter-split around the force-free Ornstein-Uhlenbeck step
    split = atomsmm.SIN_R_Integrator(2 * unit.femtoseconds, [2, 2, 1], 300 * unit.kelvin, 10 * unit.femtoseconds,
                                     1 / unit.picoseconds, split=True)
    lines = repr(split)
    assert 'v2_0 <- v2_0 + (0.125*dt)*F/M; F = Q1*v1_0^2 - kT; M = Q2' in lines
    assert 'v2_0 <- z*v2_0 + sqrt(kT*(1 - z*z)/mass)*gaussian; mass = Q2; z = exp(-(0.25*dt)*friction)' in lines
    iso = atomsmm.MassiveIsokineticPropagator(300 * unit.kelvin, 10 * unit.femtoseconds, 1, forceDependent=False)
    assert set(iso.perDofVariables) == {'v1_0', 'v2_0', 'H'} and unit.md_value(iso.perDofVariables['v1_0']) == pytest.approx(100.0)


# ------------------------------------------------------------------------------------------------ drop-in import surface
def test_alias_packages_resolve_to_the_native_implementation():
    """`import atomsmm`, `from simtk import openmm, unit`, `from simtk.openmm import app`: the reference's import lines
    (/root/reference/tests/test_respa_forces.py:3-8, src/atomsmm/__init__.py:3-43)."""
    import importlib
    import simtk
    from simtk import openmm as mm2, unit as unit2
    from simtk.openmm import app as app2
    import atomsmm as alias
    import atomsmm_amd
    assert mm2 is atomsmm_amd.openmm and unit2 is atomsmm_amd.unit and app2 is atomsmm_amd.openmm.app
    assert importlib.import_module('simtk.openmm.app') is app2 and simtk.openmm is mm2
    assert alias.RespaPropagator is atomsmm_amd.RespaPropagator and alias.forces is atomsmm_amd.forces
    assert importlib.import_module('atomsmm.propagators') is atomsmm_amd.propagators
    for name in atomsmm_amd.__all__:
        assert getattr(alias, name) is getattr(atomsmm_amd, name)
    assert mm2.app.PME == app2.PME and app2.HBonds is not None


@pytest.mark.parametrize('case', ['q-SPC-FW', 'hydroxyethylaminoanthraquinone-in-water', 'emim_BCN4_Jiung2014', 'phenol-in-water'])
def test_pdbfile_and_forcefield_reproduce_the_fixtures(case):
    """app.PDBFile + app.ForceField.describe on the reference's data files give the committed .npz arrays exactly."""
    import os
    from conftest import GOLDEN, load_case
    from atomsmm_amd.openmm import app as A
    pdb = A.PDBFile(os.path.join(GOLDEN, 'data', case + '.pdb'))
    d = A.ForceField(os.path.join(GOLDEN, 'data', case + '.xml')).describe(pdb.topology)
    ref = load_case(case)
    assert np.array_equal(np.array([list(v) for v in pdb.positions._value]), ref['positions'])
    assert np.array_equal(np.array(list(pdb.topology.getUnitCellDimensions()._value)), ref['box'])
    assert [a.name for a in pdb.topology.atoms()] == list(ref['atomname'])
    assert [a.residue.name for a in pdb.topology.atoms()] == list(ref['resname'])
    for key in ('charge', 'sigma', 'epsilon', 'mass', 'bonds', 'bond_r0', 'bond_k', 'angles', 'angle_theta0', 'angle_k',
                'torsions', 'torsion_n', 'torsion_phase', 'torsion_k', 'exc_pairs', 'exc_chargeprod', 'exc_sigma', 'exc_epsilon'):
        assert np.array_equal(d[key], ref[key]), key


def test_create_system_options():
    """createSystem's keyword arguments as the reference's tests use them (tests/test_systems.py:14-19,
    tests/test_propagators.py:14-17, tests/test_respa_forces.py:58-61)."""
    import os
    from conftest import GOLDEN
    from atomsmm_amd.openmm import app as A
    pdb = A.PDBFile(os.path.join(GOLDEN, 'data', 'q-SPC-FW.pdb'))
    ff = A.ForceField(os.path.join(GOLDEN, 'data', 'q-SPC-FW.xml'))
    rigid = ff.createSystem(pdb.topology, nonbondedMethod=A.CutoffPeriodic)
    assert rigid.getNumConstraints() == 1536 and isinstance(rigid.getForce(rigid.getNumForces() - 1), openmm.CMMotionRemover)
    assert sum(f.getNumBonds() for f in rigid.getForces() if isinstance(f, openmm.HarmonicBondForce)) == 0
    flexible = ff.createSystem(pdb.topology, nonbondedMethod=A.PME, nonbondedCutoff=10 * unit.angstroms, rigidWater=False,
                               constraints=None, removeCMMotion=False)
    assert flexible.getNumConstraints() == 0
    nb = flexible.getForce(atomsmm.findNonbondedForce(flexible))
    assert nb.getNonbondedMethod() == openmm.NonbondedForce.PME and nb.getCutoffDistance() == 1.0 * unit.nanometers
    assert nb.getNumExceptions() == 1536 and flexible.getForce(0).getNumBonds() == 1024 and flexible.getForce(1).getNumAngles() == 512
    box = flexible.getDefaultPeriodicBoxVectors()
    assert [float(unit.md_value(box[k])[k]) for k in range(3)] == [2.5, 2.5, 2.5]
    emim_pdb = A.PDBFile(os.path.join(GOLDEN, 'data', 'emim_BCN4_Jiung2014.pdb'))
    emim = A.ForceField(os.path.join(GOLDEN, 'data', 'emim_BCN4_Jiung2014.xml'))
    hbonds = emim.createSystem(emim_pdb.topology, nonbondedMethod=A.PME, constraints=A.HBonds, removeCMMotion=True)
    free = emim.createSystem(emim_pdb.topology, nonbondedMethod=A.PME, constraints=None, removeCMMotion=False)
    n_h_bonds = hbonds.getNumConstraints()
    assert n_h_bonds > 0 and free.getForce(0).getNumBonds() - hbonds.getForce(0).getNumBonds() == n_h_bonds
    with pytest.raises(ValueError):
        A.ForceField(os.path.join(GOLDEN, 'data', 'q-SPC-FW.xml')).createSystem(emim_pdb.topology)


# ------------------------------------------------------------------------------------ AlchemicalSystem (systems.py:318-410)
def test_alchemical_system_structure_and_coupling_translation(heaq, recorder):
    system = system_from_arrays(heaq, nonbondedMethod='PME', cutoff=1.0, switch=0.9)
    solute = set(int(i) for i in np.where(heaq['resname'] == 'aaa')[0])
    alch = atomsmm.AlchemicalSystem(system, solute, group=3, use_lrc=True)
    nb = alch.getForce(atomsmm.findNonbondedForce(alch))
    pair = [f for f in alch.getForces() if isinstance(f, openmm.CustomNonbondedForce)]
    assert len(pair) == 1 and pair[0].getForceGroup() == 3 and pair[0].getUseLongRangeCorrection()
    text = pair[0].getEnergyFunction()
    assert text == ('U_softcore; U_softcore = 4*lambda_vdw*epsilon*(1 - x)/x^2; x = (r/sigma)^6 + 0.5*(1 - lambda_vdw)'
                    '; sigma = 0.5*(sigma1 + sigma2); epsilon = sqrt(epsilon1*epsilon2)')
    assert [pair[0].getPerParticleParameterName(k) for k in range(2)] == ['sigma', 'epsilon']
    assert pair[0].getNumExclusions() == nb.getNumExceptions() and pair[0].getNumInteractionGroups() == 1
    for i in solute:
        q, s, e = nb.getParticleParameters(i)
        assert (q._value, s._value, e._value) == (0.0, 1.0, 0.0)
    assert nb.getNumParticleParameterOffsets() == 0            # no lambda_coul: the solute charges are simply gone
    d = F.describe_energy(text)
    assert d['family'] == 'softcore' and d['lambda_name'] == 'lambda_vdw'
    ctx = openmm.Context(alch, openmm.VerletIntegrator(0.0))
    sc = [p for p in recorder[-1].pairs if p['family'] == B.SOFTCORE]
    assert len(sc) == 1 and np.array_equal(sc[0]['q'], np.where(heaq['resname'] == 'aaa', 1.0, 2.0))
    assert np.array_equal(sc[0]['sigma'], heaq['sigma'])      # two per-particle columns, mapped by name
    ctx.setParameter('lambda_vdw', 0.25)
    assert ('pair_set_lambda', sc[0]['id'], 0.25) in recorder[-1].calls

    # Lennard-Jones times a coupling function: the factor goes to the backend as the overall scale of a plain LJ pair force
    spline = atomsmm.AlchemicalSystem(system, solute, coupling='spline')
    text = [f for f in spline.getForces() if isinstance(f, openmm.CustomNonbondedForce)][0].getEnergyFunction()
    assert text.startswith('U_spline; U_spline = 4*((gt0-gt1)*S + gt1)*epsilon*x*(x - 1); x = (sigma/r)^6; gt0 = step(lambda_vdw)'
                           '; gt1 = step(lambda_vdw-1); S = lambda_vdw^3*(10 - 15*lambda_vdw + 6*lambda_vdw^2)')
    d = F.describe_energy(text)
    assert d['family'] == 'lj' and 'S=lambda_vdw^3' in d['scale_text']
    ctx = openmm.Context(spline, openmm.VerletIntegrator(0.0))
    rec = recorder[-1]
    lj = [p for p in rec.pairs if p['family'] == B.NONBONDED and p['flags'] & B.GROUP_LJ]
    assert len(lj) == 1 and lj[0]['sign'] == 1.0
    ctx.setParameter('lambda_vdw', 0.5)
    assert ('pair_set_scale', lj[0]['id'], 0.5) in rec.calls         # S(1/2) = 1/2
    ctx.setParameter('lambda_vdw', 1.5)
    assert ('pair_set_scale', lj[0]['id'], 1.0) in rec.calls         # beyond the end point: fully coupled
    custom = atomsmm.AlchemicalSystem(system, solute, coupling='lambda_vdw^2')
    assert 'U_general = ' in [f for f in custom.getForces() if isinstance(f, openmm.CustomNonbondedForce)][0].getEnergyFunction()
    ctx = openmm.Context(custom, openmm.VerletIntegrator(0.0))
    ctx.setParameter('lambda_vdw', 0.5)
    assert any(c[0] == 'pair_set_scale' and c[2] == 0.25 for c in recorder[-1].calls)
    # the reference's `linear` text leaves two_pi undefined (systems.py:359): no Context can be made of it
    with pytest.raises(openmm.OpenMMException):
        openmm.Context(atomsmm.AlchemicalSystem(system, solute, coupling='linear'), openmm.VerletIntegrator(0.0))
    # RESPA splitting of the coupling force (systems.py:83-95) is not built
    with pytest.raises(NotImplementedError):
        atomsmm.RESPASystem(spline, 7 * unit.angstroms, 5 * unit.angstroms)


def test_deferred_globals_keep_sums_and_multiples_only():
    """expr.Deferred: a global that waits on device scalars stays a linear form under +, -, * number, / number (what AFED's
    velocity update of the extended variable does with deriv(energy, lambda), integrators.py:735-737); anything else asks
    for the number (NeedsValue) -- the engine then reads the device buffer once and evaluates again."""
    from atomsmm_amd import expr as X
    d = X.Deferred(1.5, {0: 2.0})
    env = {'v': 0.25, 'dt': 0.004, 'm': 50.0, '__deriv__': lambda what, name: d}
    out = X.eval_global('v - 0.5*(dt/4)*deriv(energy,lambda_vdw)/m', env)
    assert isinstance(out, X.Deferred)
    assert out.resolve([3.0]) == pytest.approx(0.25 - 0.5 * 0.001 * (1.5 + 2.0 * 3.0) / 50.0, rel=1e-15)
    env['v'] = out
    again = X.eval_global('lam + 0.5*dt*v - v*2 + (-v)', dict(env, lam=0.6))
    assert again.resolve([3.0]) == pytest.approx(0.6 + (0.002 - 3.0) * out.resolve([3.0]), rel=1e-14)
    other = X.Deferred(0.0, {1: 1.0})
    both = X.eval_global('v + w', dict(env, w=other))
    assert both.resolve([3.0, 7.0]) == pytest.approx(out.resolve([3.0]) + 7.0)
    for text in ('v^2', 'exp(-dt*v)', 'step(v)', 'select(step(v),2,0)-v', '1/v', 'v*v', 'max(v, 0)', 'abs(v)'):
        with pytest.raises(X.NeedsValue):
            X.eval_global(text, env)
    with pytest.raises(X.NeedsValue):
        float(out)


def test_softcore_force_classes(heaq, recorder):
    """forces.py:727-758: SoftcoreLennardJonesForce over an interaction group is the softcore pair family."""
    system = system_from_arrays(heaq, nonbondedMethod='PME', cutoff=1.0, switch=0.9)
    nb = system.getForce(atomsmm.findNonbondedForce(system))
    solute = set(int(i) for i in np.where(heaq['resname'] == 'aaa')[0])
    solvent = set(range(nb.getNumParticles())) - solute
    force = atomsmm.SoftcoreLennardJonesForce(parameter='lambda_vdw').importFrom(nb)
    assert force.getEnergyFunction().startswith('4*lambda_vdw*epsilon*x*(x-1);x = 1/((r/sigma)^6 + 0.5*(1-lambda_vdw))')
    assert force.getCutoffDistance() == nb.getCutoffDistance() and force.getUseSwitchingFunction()
    force.addInteractionGroup(solute, solvent)
    force.setForceGroup(4)
    force.addTo(system)
    openmm.Context(system, openmm.VerletIntegrator(0.0)).setParameter('lambda_vdw', 0.5)
    sc = [p_ for p_ in recorder[-1].pairs if p_['family'] == B.SOFTCORE]
    assert len(sc) == 1 and ('pair_set_lambda', sc[0]['id'], 0.5) in recorder[-1].calls
    # the reference's SoftcoreForce (softcore Lennard-Jones plus scaled Coulomb in one expression, forces.py:761-793) exists with the
    # reference's text and globals (test_energy_string_equals_the_reference_capture[softcore]); the HIP engine has no kernel for it
    # and says so when a Context is created over it
    both = atomsmm.SoftcoreForce(1.0 * unit.nanometers, 0.9 * unit.nanometers).importFrom(nb)
    assert both.getUseSwitchingFunction() and both.getGlobalParameterName(0) == 'Kc'
    system2 = openmm.System()
    for i in range(nb.getNumParticles()):
        system2.addParticle(1.0)
    system2.setDefaultPeriodicBoxVectors(*system.getDefaultPeriodicBoxVectors())
    both.addTo(system2)
    with pytest.raises(Exception, match='not recognised'):
        openmm.Context(system2, openmm.VerletIntegrator(0.0))


def test_simulation_serves_reporters(spcfw, recorder):
    """app.Simulation.step follows OpenMM's reporter protocol (describeNextReport / report), so scripts that attach reporters
    keep working: reports land on the right steps whatever the chunks step() is called in; app.StateDataReporter writes
    the requested columns."""
    import io
    system = system_from_arrays(spcfw, nonbondedMethod='CutoffPeriodic')
    respa = atomsmm.RESPASystem(system, 7 * unit.angstroms, 5 * unit.angstroms)
    integrator = atomsmm.RespaPropagator([2, 2, 1]).integrator(1 * unit.femtoseconds)
    simulation = app.Simulation(app.Topology(), respa, integrator, openmm.Platform.getPlatformByName('HIP'))
    simulation.context.setPositions(spcfw['positions'] * unit.nanometers)

    class Probe:
        def __init__(self, every):
            self.every, self.seen = every, []

        def describeNextReport(self, simulation):
            return (self.every - simulation.currentStep % self.every, True, False, False, False)

        def report(self, simulation, state):
            self.seen.append(simulation.currentStep)
            assert state.getPositions(asNumpy=True)._value.shape == (len(spcfw['positions']), 3)

    three, five = Probe(3), Probe(5)
    text = io.StringIO()
    simulation.reporters += [three, five, app.StateDataReporter(text, 4, step=True, time=True, potentialEnergy=True,
                                                                temperature=True, volume=True, density=True, speed=True)]
    for chunk in (1, 6, 2, 7):
        simulation.step(chunk)
    assert simulation.currentStep == 16
    assert three.seen == [3, 6, 9, 12, 15] and five.seen == [5, 10, 15]
    lines = text.getvalue().strip().splitlines()
    assert lines[0].startswith('#"Step","Time (ps)","Potential Energy (kJ/mole)","Temperature (K)"')
    assert [int(line.split(',')[0]) for line in lines[1:]] == [4, 8, 12, 16]
    assert float(lines[-1].split(',')[1]) == pytest.approx(0.016)
    density = float(lines[-1].split(',')[5])
    assert 0.9 < density < 1.1               # q-SPC/Fw water


# ------------------------------------------------------------------------------------ reference-generated text fixtures
# tests/golden/programs.json is written by scripts/capture_reference_text.py (build container only): the reference's own Python
# layer run under a recording stand-in for simtk.  Every constructor expression is evaluated again here, against atomsmm_amd.
def _captured():
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'programs.json')) as fh:
        return json.load(fh)


def _numbers_to_12_digits(text):
    import re

    def norm(m):
        return repr(float('%.12g' % float(m.group(0))))
    return re.sub(r'(?<![A-Za-z_0-9])\d+\.?\d*(?:[eE][-+]?\d+)?', norm, text)


@pytest.mark.parametrize('name', sorted(_captured()['programs']))
def test_step_program_equals_the_reference_capture(name):
    """Per-DOF variables, global variables (names, order and initial values) and every line of the emitted step program of each
    propagator / integrator class the repository restates -- composition classes, RESPA and its schemes, velocity Verlet, the
    thermostat propagators, Langevin_R / NHL_R / SIN_R, AFED -- against what the reference emits for the same constructor call."""
    import atomsmm_amd as atomsmm
    from atomsmm_amd import unit
    g = _captured()['programs'][name]
    integ = eval(g['ctor'], {'atomsmm': atomsmm, 'unit': unit})
    assert [integ.getPerDofVariableName(i) for i in range(integ.getNumPerDofVariables())] == g['per_dof']
    names = [integ.getGlobalVariableName(i) for i in range(integ.getNumGlobalVariables())]
    assert names == g['globals']
    for i, gname in enumerate(names):
        assert integ.getGlobalVariable(i) == pytest.approx(g['global_values'][gname], rel=1e-9, abs=1e-300), gname
    assert integ.pretty_steps() == g['steps']


@pytest.mark.parametrize('name', sorted(_captured()['forces']))
def test_energy_string_equals_the_reference_capture(name):
    """forces.py:400-724: the energy expression after importFrom(nonbonded), literals compared to 12 digits (the capture's unit
    stand-in and this package's unit module may differ in the last digit of a converted number)."""
    import atomsmm_amd as atomsmm
    from atomsmm_amd import openmm, unit
    g = _captured()['forces'][name]
    nb = openmm.NonbondedForce()
    nb.setNonbondedMethod(nb.CutoffPeriodic)
    nb.addParticle(0.5, 0.3, 0.7)
    nb.addParticle(-0.5, 0.25, 0.2)
    nb.addParticle(0.1, 0.2, 0.1)
    nb.addException(0, 1, -0.1, 0.27, 0.3)
    force = eval(g['ctor'], {'atomsmm': atomsmm, 'unit': unit})
    force.importFrom(nb)
    assert _numbers_to_12_digits(force.getEnergyFunction()) == _numbers_to_12_digits(g['energy'])
    mine = {force.getGlobalParameterName(i): force.getGlobalParameterDefaultValue(i) for i in range(force.getNumGlobalParameters())}
    assert set(mine) == set(g['globals'])
    for key, value in g['globals'].items():
        assert mine[key] == pytest.approx(value, rel=1e-12)
