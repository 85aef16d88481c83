"""GPU tests of the general step-program path: per-DOF / sum expressions compiled by
atomsmm_amd/expr.py and interpreted by amm_expr_eval, against the numpy restatement in oracle/expr_oracle.py (same
postfix programs, same Philox-4x32-10 stream), and the thermostat propagators built on it.

Parity status: UNPINNED against the reference -- its thermostat tests (tests/test_propagators.py:51-111) need OpenMM's
random number generator, HBonds constraints and PME on the EMIM fixture; what is checked here is the program text
semantics (deterministic thermostats step for step against a numpy evaluation of the same program) and the physics
(energy conservation without a bath, equipartition with one)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')

import atomsmm_amd as atomsmm  # noqa: E402
from atomsmm_amd import backend as B  # noqa: E402
from atomsmm_amd import expr as X  # noqa: E402
from atomsmm_amd import openmm, unit  # noqa: E402
from atomsmm_amd.testing import system_from_arrays  # noqa: E402
from oracle import expr_oracle as XO  # noqa: E402  (checker only)
from oracle import oracle as O  # noqa: E402

KB = unit.BOLTZMANN_CONSTANT_kB._value        # kB*NA in kJ/mol/K as the unit module defines it


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device='cuda')


EXPRESSIONS = [
    'v + 0.5*1.0*dt*f0/m',
    'vscaling*v; vscaling = sqrt(A+C*B*(gaussian^2+sumRs)+2*sqrt(C*B*A)*gaussian); C = 2.49/mvv; B = 1-A; A = exp(-dt*10.0); sumRs = 7*V+X^2',
    'z*v + sqrt(kT*(1 - z*z)/mass)*gaussian; mass = m; z = exp(-(0.5*dt)*friction)',
    'select(step(x-1.2), min(v, 0.5), max(v, -0.5))*abs(f0)^1.5 + delta(0*x)*uniform - erfc(v)/(1+x^2) + v^-3 + atan2(x, v)',
    'p + (0.25*dt)*(m*v^2 - kT)',
]


@pytest.mark.parametrize('expression', EXPRESSIONS)
def test_expression_interpreter_vs_numpy(expression):
    rng = np.random.default_rng(11)
    n = 1000
    ctx = B.HipContext(n, np.array([3.0, 3.0, 3.0]))
    arrays = dict(x=rng.uniform(0.5, 2.5, (n, 3)), v=rng.normal(0, 0.6, (n, 3)), f0=rng.normal(0, 300, (n, 3)),
                  p=rng.normal(0, 1, (n, 3)))
    mass = rng.choice([1.008, 15.9994, 12.011], n)
    t = {k: dev(a) for k, a in arrays.items()}
    ctx.bind_state(t['x'], t['v'], dev(mass))
    slots = {'x': B.SLOT_X, 'v': B.SLOT_V, 'f0': 0, 'p': 1}
    ctx.bind_buffer(0, t['f0'])
    ctx.bind_buffer(1, t['p'])
    env = dict(dt=0.002, mvv=8123.4, V=0.93, X=-0.41, kT=2.494, friction=10.0)

    def resolve(name):
        if name == 'm':
            return ('mass',)
        if name in slots:
            return ('buf', slots[name])
        return ('global',) if name in env else None
    prog = X.compile_per_dof(expression, resolve)
    gvals = [env[g] for g in prog.globals_]
    out = torch.zeros((n, 3), dtype=torch.float64, device='cuda')
    total = torch.zeros(1, dtype=torch.float64, device='cuda')
    ctx.expr_eval(prog.code, prog.consts, gvals, seed=20240521, counter=7, dst=out, total=total)
    ctx.check()
    ref = XO.run(prog.code, prog.consts, gvals, {slots[k]: arrays[k] for k in arrays}, mass, 20240521, 7)
    got = out.cpu().numpy()
    assert np.allclose(got, ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max())
    assert total.item() == pytest.approx(ref.sum(), rel=1e-11, abs=1e-9)
    # a different launch counter gives a different random stream, the same one the same stream
    if 'gaussian' in expression or 'uniform' in expression:
        ctx.expr_eval(prog.code, prog.consts, gvals, seed=20240521, counter=8, dst=out)
        assert not np.allclose(out.cpu().numpy(), got)
        ctx.expr_eval(prog.code, prog.consts, gvals, seed=20240521, counter=7, dst=out)
        assert np.array_equal(out.cpu().numpy(), got)
    # in place: the destination may be one of the operands
    ctx.expr_eval(prog.code, prog.consts, gvals, seed=20240521, counter=7, dst=t['v'])
    assert np.array_equal(t['v'].cpu().numpy(), got)
    with pytest.raises(B.HipError):
        ctx.expr_eval([X.OPCODES['ADD']], [], [], 0, 0, dst=out)          # stack underflow is caught on the host
    ctx.close()


def test_gaussian_stream_statistics():
    n = 200000
    ctx = B.HipContext(n, np.array([3.0, 3.0, 3.0]))
    x = torch.zeros((n, 3), dtype=torch.float64, device='cuda')
    ctx.bind_state(x, x.clone(), torch.ones(n, dtype=torch.float64, device='cuda'))
    out = torch.zeros((n, 3), dtype=torch.float64, device='cuda')
    prog = X.compile_per_dof('gaussian', lambda name: None)
    ctx.expr_eval(prog.code, prog.consts, [], seed=3, counter=1, dst=out)
    g = out.cpu().numpy().ravel()
    assert abs(g.mean()) < 0.01 and abs(g.std() - 1.0) < 0.01 and abs((g ** 4).mean() - 3.0) < 0.1
    assert abs(np.corrcoef(g[:-1], g[1:])[0, 1]) < 0.01
    ctx.close()


def _water(spcfw):
    system = system_from_arrays(spcfw, nonbondedMethod='CutoffPeriodic')
    nb = atomsmm.hijackForce(system, atomsmm.findNonbondedForce(system))
    force = atomsmm.DampedSmoothedForce(0.29 / unit.angstroms, 10 * unit.angstroms, 9 * unit.angstroms).importFrom(nb)
    force.addTo(system)
    return system


def test_unconstrained_velocity_verlet_vs_oracle(spcfw):
    """UnconstrainedVelocityVerletPropagator (propagators.py:1136-1153): its per-DOF assignments are bare products,
    `v + 0.5*1.0*dt*f/m` and `x + 1.0*dt*v` -- the kick's and the move's arithmetic ((coef * f) / m with the force as the
    last factor), so they run as native KICK / MOVE ops, not through the expression interpreter (which took 3 x 55 us per
    step at 32 768 atoms: bench.py --config c2); 5 steps against numpy."""
    c = spcfw
    system = _water(c)
    integrator = atomsmm.UnconstrainedVelocityVerletPropagator().integrator(0.5 * unit.femtoseconds)
    context = openmm.Context(system, integrator)
    context.setPositions(c['positions'] * unit.nanometers)
    context.setVelocitiesToTemperature(300 * unit.kelvin, 5)
    v = context.getState(getVelocities=True).getVelocities(asNumpy=True)._value.copy()
    x = c['positions'].copy()
    dd = O.desc(O.DAMPED, rc=1.0, rswitch=0.9, alpha=2.9, degree=1)

    def force(p):
        return (O.pair_eval(dd, p, c['box'], c['charge'], c['sigma'], c['epsilon'], c['exc_pairs'])[1] +
                O.harmonic_bonds(c['bonds'], c['bond_r0'], c['bond_k'], p, c['box'])[1] +
                O.harmonic_angles(c['angles'], c['angle_theta0'], c['angle_k'], p, c['box'])[1])
    dt, m = 0.0005, c['mass'][:, None]
    f = force(x)
    for _ in range(5):
        v = v + 0.5 * 1.0 * dt * f / m
        x = x + 1.0 * dt * v
        f = force(x)
        v = v + 0.5 * 1.0 * dt * f / m
    integrator.step(5)
    assert context._engine._interpreted is False      # static program, replayed by amm_run_ops
    assert not context._engine._expr_ids              # ... of native ops only: nothing was registered with the interpreter
    state = context.getState(getPositions=True, getVelocities=True)
    assert np.abs(state.getPositions(asNumpy=True)._value - x).max() < 1e-12
    assert np.abs(state.getVelocities(asNumpy=True)._value - v).max() < 1e-10


def test_nose_hoover_program_step_for_step(spcfw):
    """NoseHooverPropagator alone (propagators.py:1230-1273): mvv <- sum(m*v*v) on the GPU, the globals on the host,
    v <- vscaling*v on the GPU -- against a numpy evaluation of the same program text."""
    c = spcfw
    system = _water(c)
    dof = atomsmm.countDegreesOfFreedom(system)
    thermostat = atomsmm.NoseHooverPropagator(300 * unit.kelvin, dof, 10 * unit.femtoseconds, 3)
    integrator = thermostat.integrator(2 * unit.femtoseconds)
    context = openmm.Context(system, integrator)
    context.setPositions(c['positions'] * unit.nanometers)
    context.setVelocitiesToTemperature(450 * unit.kelvin, 9)
    v = context.getState(getVelocities=True).getVelocities(asNumpy=True)._value.copy()
    m = c['mass'][:, None]
    LkT, Q, dt, p_eta = dof * KB * 300.0, dof * KB * 300.0 * 0.01 ** 2, 0.002, 0.0
    sub = 1.0 / 3
    for _ in range(4):
        mvv = float((m * v * v).sum())
        p_eta = p_eta + (0.5 * sub * dt) * (mvv - LkT)
        vs = np.exp(-(sub * dt) * p_eta / Q)
        for _k in range(2):
            p_eta = p_eta + (sub * dt) * (vs ** 2 * mvv - LkT)
            vs = vs * np.exp(-(sub * dt) * p_eta / Q)
        p_eta = p_eta + (0.5 * sub * dt) * (vs ** 2 * mvv - LkT)
        v = vs * v
    integrator.step(4)
    got = context.getState(getVelocities=True).getVelocities(asNumpy=True)._value
    assert np.abs(got - v).max() < 1e-11 * np.abs(v).max()
    assert integrator.getGlobalVariableByName('p_eta') == pytest.approx(p_eta, rel=1e-11)


def test_nose_hoover_chain_and_ggm_programs_step_for_step(spcfw):
    """NoseHooverChainPropagator (propagators.py:1362-1449: globals on the host, one sum and one scaling on the GPU) and
    MassiveGeneralizedGaussianMomentPropagator (:1314-1359: three per-DOF expressions with auxiliary definitions, inside a
    while block) against numpy evaluations of the same program text."""
    c = spcfw
    system = _water(c)
    dof = atomsmm.countDegreesOfFreedom(system)
    m = c['mass'][:, None]
    kT, tau, dt = KB * 300.0, 0.01, 0.002
    # --- Nose-Hoover chain
    integrator = atomsmm.NoseHooverChainPropagator(300 * unit.kelvin, dof, 10 * unit.femtoseconds).integrator(2 * unit.femtoseconds)
    context = openmm.Context(system, integrator)
    context.setPositions(c['positions'] * unit.nanometers)
    context.setVelocitiesToTemperature(450 * unit.kelvin, 9)
    v = context.getState(getVelocities=True).getVelocities(asNumpy=True)._value.copy()
    NkT = dof * kT
    Q1, Q2, p1, p2 = NkT * tau ** 2, kT * tau ** 2, 0.0, 0.0
    for _ in range(5):
        p2 = p2 + (p1 ** 2 / Q1 - kT) * 0.5 * dt
        p1 = p1 * np.exp(-(0.5 / Q2) * p2 * dt)
        mvv = float((m * v * v).sum())
        p1 = p1 + (mvv - NkT) * 0.5 * dt
        vs = np.exp(-(1.0 / Q1) * p1 * dt)
        p1 = p1 + (vs ** 2 * mvv - NkT) * 0.5 * dt
        p1 = p1 * np.exp(-(0.5 / Q2) * p2 * dt)
        p2 = p2 + (p1 ** 2 / Q1 - kT) * 0.5 * dt
        v = vs * v
    integrator.step(5)
    got = context.getState(getVelocities=True).getVelocities(asNumpy=True)._value
    assert np.abs(got - v).max() < 1e-11 * np.abs(v).max()
    assert integrator.getGlobalVariableByName('p_NHC_1') == pytest.approx(p1, rel=1e-10)
    assert integrator.getGlobalVariableByName('p_NHC_2') == pytest.approx(p2, rel=1e-10)
    # --- massive generalized Gaussian moment thermostat, 2 sub-steps per step
    integrator = atomsmm.MassiveGeneralizedGaussianMomentPropagator(300 * unit.kelvin, 10 * unit.femtoseconds, 2).integrator(2 * unit.femtoseconds)
    context = openmm.Context(system, integrator)
    context.setPositions(c['positions'] * unit.nanometers)
    context.setVelocitiesToTemperature(450 * unit.kelvin, 9)
    v = context.getState(getVelocities=True).getVelocities(asNumpy=True)._value.copy()
    Q1, Q2 = kT * tau ** 2, 2 * kT ** 3 * tau ** 2
    p1, p2 = np.zeros_like(v), np.zeros_like(v)
    sub = 0.5
    for _ in range(3 * 2):
        p1 = p1 + (sub / 2 * dt) * (m * v ** 2 - kT)
        p2 = p2 + (sub / 2 * dt) * (m ** 2 * v ** 4 / 3 - kT ** 2)
        scale = np.exp(-(sub / 2) * dt * (p1 / Q1 + kT * p2 / Q2))
        v1 = v * scale
        alpha = p2 / (3 * m * Q2)
        v = v1 / np.sqrt(1 + 2 * v1 ** 2 * alpha * sub * dt) * scale
        p2 = p2 + (sub / 2 * dt) * (m ** 2 * v ** 4 / 3 - kT ** 2)
        p1 = p1 + (sub / 2 * dt) * (m * v ** 2 - kT)
    integrator.step(3)
    got = context.getState(getVelocities=True).getVelocities(asNumpy=True)._value
    assert np.abs(got - v).max() < 1e-11 * np.abs(v).max()
    gp2 = np.array([list(row) for row in integrator.getPerDofVariableByName('p2')])
    assert np.abs(gp2 - p2).max() < 1e-10 * np.abs(p2).max()


def test_langevin_middle_scheme_equilibrates(spcfw):
    """B A O A B with the Ornstein-Uhlenbeck bath (propagators.py:685-741): a cold start reaches the bath temperature."""
    c = spcfw
    system = _water(c)
    nve = atomsmm.UnconstrainedVelocityVerletPropagator()
    bath = atomsmm.OrnsteinUhlenbeckPropagator(300 * unit.kelvin, 20 / unit.picoseconds)
    integrator = atomsmm.TrotterSuzukiPropagator(nve, bath).integrator(1 * unit.femtoseconds)
    integrator.setRandomNumberSeed(1234)
    context = openmm.Context(system, integrator)
    context.setPositions(c['positions'] * unit.nanometers)
    context.setVelocitiesToTemperature(30 * unit.kelvin, 1)
    temps = []
    for _ in range(12):
        integrator.step(50)
        ke = context.getState(getEnergy=True).getKineticEnergy()._value
        temps.append(2 * ke / (3 * len(c['mass']) * KB))
    assert temps[0] > 60 and abs(np.mean(temps[-4:]) - 300) < 25, temps
    pe = context.getState(getEnergy=True).getPotentialEnergy()._value
    assert np.isfinite(pe)


def test_bussi_thermostat_runs_and_holds_temperature(spcfw):
    """GlobalThermostatIntegrator(NVE, VelocityRescalingPropagator) (integrators.py:173-211, propagators.py:1156-1227):
    while-blocks on host-side random globals + a per-DOF expression with auxiliary definitions."""
    c = spcfw
    system = _water(c)
    dof = atomsmm.countDegreesOfFreedom(system)
    nve = atomsmm.UnconstrainedVelocityVerletPropagator()
    thermostat = atomsmm.VelocityRescalingPropagator(300 * unit.kelvin, dof, 0.05 * unit.picoseconds)
    integrator = atomsmm.GlobalThermostatIntegrator(1 * unit.femtoseconds, nve, thermostat)
    integrator.setRandomNumberSeed(1)
    context = openmm.Context(system, integrator)
    context.setPositions(c['positions'] * unit.nanometers)
    context.setVelocitiesToTemperature(300 * unit.kelvin, 1)
    integrator.step(200)
    ke = context.getState(getEnergy=True).getKineticEnergy()._value
    assert 200 < 2 * ke / (dof * KB) < 400


@pytest.mark.parametrize('kind', ['langevin', 'nhl'])
def test_respa_with_bath_equilibrates(spcfw, kind):
    """Langevin_R_Integrator / NHL_R_Integrator (integrators.py:272-323): RESPA [2,2,1] over RESPASystem with the bath in
    the 'middle' of the innermost loop.  The pair-force evaluations still go through amm_run_ops; the bath steps are EXPR ops of
    the same unrolled program.  A cold start reaches the bath temperature and the near / outer force caches stay consistent."""
    c = spcfw
    system = system_from_arrays(c, nonbondedMethod='CutoffPeriodic')
    respa = atomsmm.RESPASystem(system, 7 * unit.angstroms, 5 * unit.angstroms)
    nb = atomsmm.hijackForce(respa, atomsmm.findNonbondedForce(respa))
    outer = atomsmm.DampedSmoothedForce(0.29 / unit.angstroms, 10 * unit.angstroms, 9 * unit.angstroms).importFrom(nb)
    outer.setForceGroup(2)
    outer.addTo(respa)
    if kind == 'langevin':
        integrator = atomsmm.Langevin_R_Integrator(2 * unit.femtoseconds, [2, 2, 1], 300 * unit.kelvin, 20 / unit.picoseconds)
    else:
        integrator = atomsmm.NHL_R_Integrator(2 * unit.femtoseconds, [2, 2, 1], 300 * unit.kelvin, 10 * unit.femtoseconds,
                                              20 / unit.picoseconds)
    integrator.setRandomNumberSeed(77)
    context = openmm.Context(respa, integrator)
    context.setPositions(c['positions'] * unit.nanometers)
    context.setVelocitiesToTemperature(30 * unit.kelvin, 3)
    temps = []
    for _ in range(10):
        integrator.step(40)
        ke = context.getState(getEnergy=True).getKineticEnergy()._value
        temps.append(2 * ke / (3 * len(c['mass']) * KB))
    assert context._engine._interpreted is False      # the bath is an EXPR op inside the unrolled RESPA loop
    assert abs(np.mean(temps[-3:]) - 300) < 30, temps
    # the cached group forces the program leaves behind equal a fresh evaluation at the final positions
    eng = context._engine
    x = context.getState(getPositions=True).getPositions(asNumpy=True)._value
    dn = O.desc(O.NEAR_FSWITCH, rc=0.7, rc0=0.7, rs0=0.5)
    f1 = O.pair_eval(dn, x, c['box'], c['charge'], c['sigma'], c['epsilon'], c['exc_pairs'])[1]
    assert np.abs(eng._buffer('f1').cpu().numpy() - f1).max() <= 1e-9 * np.abs(f1).max()


@pytest.mark.parametrize('L,split', [(1, False), (2, False), (2, True)])
def test_sin_r_keeps_the_isokinetic_constraint(spcfw, L, split):
    """SIN_R_Integrator (integrators.py:358-416; SIN_R_Propagator / MassiveIsokineticPropagator, propagators.py:276-355,
    1045-1105): RESPA [2,2,1] whose kicks are the force-dependent isokinetic propagator, with the stochastic bath on the
    v2_i in the middle.  No reference literal exists; the size-independent invariant of the method is checked instead:
    EVERY degree of freedom obeys m v^2 + L/(L+1) Q1 sum_i v1_i^2 = L kT after every step, whatever the forces did."""
    c = spcfw
    system = system_from_arrays(c, nonbondedMethod='CutoffPeriodic')
    respa = atomsmm.RESPASystem(system, 7 * unit.angstroms, 5 * unit.angstroms)
    nb = atomsmm.hijackForce(respa, atomsmm.findNonbondedForce(respa))
    outer = atomsmm.DampedSmoothedForce(0.29 / unit.angstroms, 10 * unit.angstroms, 9 * unit.angstroms).importFrom(nb)
    outer.setForceGroup(2)
    outer.addTo(respa)
    integrator = atomsmm.SIN_R_Integrator(2 * unit.femtoseconds, [2, 2, 1], 300 * unit.kelvin, 10 * unit.femtoseconds,
                                          20 / unit.picoseconds, L=L, split=split)
    integrator.setRandomNumberSeed(5)
    context = openmm.Context(respa, integrator)
    context.setPositions(c['positions'] * unit.nanometers)
    context.setVelocitiesToTemperature(300 * unit.kelvin, 3)
    kT = integrator.getGlobalVariableByName('kT')
    Q1 = integrator.getGlobalVariableByName('Q1')
    assert kT == pytest.approx(300 * KB) and integrator.getGlobalVariableByName('LkT') == pytest.approx(L * kT)
    mass = np.asarray(c['mass'])[:, None]
    for nsteps in (1, 30):
        integrator.step(nsteps)
        v = context.getState(getVelocities=True).getVelocities(asNumpy=True)._value
        v1 = [np.array([list(row) for row in integrator.getPerDofVariableByName('v1_%d' % i)]) for i in range(L)]
        lhs = mass * v ** 2 + L / (L + 1) * Q1 * sum(a ** 2 for a in v1)
        assert np.abs(lhs / (L * kT) - 1.0).max() < 1e-12
    assert context._engine._interpreted is False      # isokinetic kicks and bath are EXPR ops of the unrolled RESPA program
    x = context.getState(getPositions=True).getPositions(asNumpy=True)._value
    assert np.isfinite(x).all() and np.abs(x - c['positions']).max() < 0.5
    # the thermostat velocities were drawn by initialize(): v2 ~ N(0, kT/Q2)
    v2 = np.array([list(row) for row in integrator.getPerDofVariableByName('v2_0')])
    assert 0.5 < v2.std() / np.sqrt(kT / integrator.getGlobalVariableByName('Q2')) < 2.0
    # the cached near force the program leaves behind equals a fresh evaluation at the final positions
    dn = O.desc(O.NEAR_FSWITCH, rc=0.7, rc0=0.7, rs0=0.5)
    f1 = O.pair_eval(dn, x, c['box'], c['charge'], c['sigma'], c['epsilon'], c['exc_pairs'])[1]
    assert np.abs(context._engine._buffer('f1').cpu().numpy() - f1).max() <= 1e-9 * np.abs(f1).max()


def test_bath_inside_the_inner_loop_kernel_is_bit_identical(spcfw):
    """Langevin_R: the inner-loop kernel runs kick ; move ; OU bath ; move ; forces ; kick for all n0 iterations in one
    launch, calling the same expression interpreter with the same random-stream counters as the separate EXPR launches
    do: the trajectory is bit-identical to the op-by-op execution."""
    c = spcfw
    out = []
    for fuse in (True, False):
        system = system_from_arrays(c, nonbondedMethod='CutoffPeriodic')
        respa = atomsmm.RESPASystem(system, 7 * unit.angstroms, 5 * unit.angstroms)
        nb = atomsmm.hijackForce(respa, atomsmm.findNonbondedForce(respa))
        outer = atomsmm.DampedSmoothedForce(0.29 / unit.angstroms, 10 * unit.angstroms, 9 * unit.angstroms).importFrom(nb)
        outer.setForceGroup(2)
        outer.addTo(respa)
        integrator = atomsmm.Langevin_R_Integrator(2 * unit.femtoseconds, [4, 2, 1], 300 * unit.kelvin, 5 / unit.picoseconds)
        integrator.setRandomNumberSeed(99)
        context = openmm.Context(respa, integrator)
        context._engine.ctx.set_fuse_inner(fuse)
        context.setPositions(c['positions'] * unit.nanometers)
        context.setVelocitiesToTemperature(300 * unit.kelvin, 3)
        integrator.step(12)
        st = context.getState(getPositions=True, getVelocities=True)
        out.append((st.getPositions(asNumpy=True)._value.copy(), st.getVelocities(asNumpy=True)._value.copy()))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    assert np.abs(out[0][0] - c['positions']).max() > 1e-3


def test_native_ou_bath_vs_numpy():
    """AMM_OP_BATH: v <- z v + sqrt(kT (1 - z^2)/m) gaussian, the Gaussian from the Philox stream of the EXPR/BATH ops."""
    rng = np.random.default_rng(5)
    n = 3000
    ctx = B.HipContext(n, np.array([3.0, 3.0, 3.0]))
    v0 = rng.normal(0, 0.5, (n, 3))
    mass = rng.choice([1.008, 15.9994], n)
    x, v = dev(rng.uniform(0, 3, (n, 3))), dev(v0)
    ctx.bind_state(x, v, dev(mass))
    z, kT = float(np.exp(-0.002 * 5.0)), 2.494
    bid = ctx.bath_define(z, kT)
    ctx.expr_seed(424242)
    ctx.run_ops([B.Op(B.OP_BATH, bid, B.SLOT_V, 0, 0.0)] * 2, 1)
    ctx.check()
    ref = v0.copy()
    for k in (1, 2):
        u1, u2 = XO.uniforms(3 * n, 0, 424242, (1 << 63) | k)
        g = (np.sqrt(-2.0 * np.log(u1)) * np.cos(6.283185307179586476925 * u2)).reshape(n, 3)
        ref = z * ref + np.sqrt(kT * (1.0 - z * z) / mass[:, None]) * g
    assert np.abs(v.cpu().numpy() - ref).max() < 1e-13
    ctx.close()


def _solvated(heaq):
    system = system_from_arrays(heaq, nonbondedMethod='PME', cutoff=1.0, switch=0.9)
    solute = set(int(i) for i in np.where(heaq['resname'] == 'aaa')[0])
    return atomsmm.SolvationSystem(system, solute)


def test_energy_parameter_derivative_of_softcore_force(heaq):
    """deriv(energy, lambda_vdw) (ExtendedSystemVariable.update_velocity, integrators.py:735-737): the pair kernel in
    derivative mode + the derivative of the long-range correction == central differences of the energy, and == the
    oracle's finite difference of the pinned softcore energy (G15 family)."""
    h = heaq
    solv = _solvated(h)
    context = openmm.Context(solv, openmm.VerletIntegrator(0.0))
    context.setPositions(h['positions'] * unit.nanometers)
    codes = np.where(h['resname'] == 'aaa', 1.0, 2.0)
    for lam in (0.5, 0.9, 0.15):
        context.setParameter('lambda_vdw', lam)
        d = context._engine.energy_derivative('lambda_vdw')
        eps_ = 1e-5
        e = []
        for sgn in (1, -1):
            context.setParameter('lambda_vdw', lam + sgn * eps_)
            e.append(context.getState(getEnergy=True).getPotentialEnergy()._value)
        assert d == pytest.approx((e[0] - e[1]) / (2 * eps_), rel=2e-6)
        ref = []
        for sgn in (1, -1):
            dd = O.desc(O.SOFTCORE, rc=1.0, rswitch=0.9, alpha=lam + sgn * eps_, flags=O.SWITCH, Kc=1.0)
            ref.append(O.pair_eval(dd, h['positions'], h['box'], codes, h['sigma'], h['epsilon'], h['exc_pairs'], want_forces=False)[0] +
                       O.softcore_lrc(h['sigma'], h['epsilon'], codes, h['box'], 1.0, 0.9, lam + sgn * eps_))
        assert d == pytest.approx((ref[0] - ref[1]) / (2 * eps_), rel=2e-6)
    # charge offsets: exact unit-step difference of a quadratic form (tests/test_gpu_api.py checks it against the oracle)
    context.setParameter('lambda_coul', 0.5)
    d = context._engine.energy_derivative('lambda_coul')
    e = []
    for point in (0.75, 0.25):
        context.setParameter('lambda_coul', point)
        e.append(context.getState(getEnergy=True).getPotentialEnergy()._value)
    assert d == pytest.approx((e[0] - e[1]) / 0.5, rel=1e-9)


def test_adiabatic_free_energy_dynamics_runs(heaq):
    """AFED (config 5 of BASELINE.json in miniature): RESPASystem over a SolvationSystem, RESPA [2,2,1] inside
    AdiabaticDynamicsIntegrator with lambda_vdw as extended variable.  lambda moves, stays inside its walls, the
    Context parameter follows it, and the dynamics stays finite."""
    h = heaq
    respa = atomsmm.RESPASystem(_solvated(h), 7 * unit.angstroms, 5 * unit.angstroms)
    inner = atomsmm.RespaPropagator([2, 2, 1]).integrator(1 * unit.femtoseconds)
    lam = atomsmm.ExtendedSystemVariable('lambda_vdw', 50, 2.5, 20 * unit.femtoseconds)
    integrator = atomsmm.AdiabaticDynamicsIntegrator(inner, 2, [lam])
    integrator.setRandomNumberSeed(11)
    context = openmm.Context(respa, integrator)
    context.setPositions(h['positions'] * unit.nanometers)
    context.setVelocitiesToTemperature(300 * unit.kelvin, 2)
    context.setParameter('lambda_vdw', 0.8)
    context.setParameter('lambda_coul', 0.0)
    seen = []
    for _ in range(10):
        integrator.step(3)
        seen.append(context.getParameter('lambda_vdw'))
    assert context._engine._interpreted is True
    assert all(0.0 <= v <= 1.0 for v in seen) and max(seen) - min(seen) > 1e-3, seen
    st = context.getState(getEnergy=True)
    assert np.isfinite(st.getPotentialEnergy()._value) and np.isfinite(st.getKineticEnergy()._value)
    assert integrator.getGlobalVariableByName('_v_lambda_vdw') != 0.0


def test_native_nhl_bath_vs_numpy():
    """AMM_OP_BATH of kind Nose-Hoover-Langevin (amm_bath_define_nhl): the three per-DOF steps of NHL_R_Integrator's bath block
    (integrators.py:272-330) -- v <- v exp(-h w); w <- z w + sqrt(kT (1 - z^2)/Q) gaussian + (m v^2 - kT)(1 - z)/(Q friction);
    v <- v exp(-h w) -- against numpy on the same Philox stream."""
    rng = np.random.default_rng(6)
    n = 3000
    ctx = B.HipContext(n, np.array([3.0, 3.0, 3.0]))
    v0, w0 = rng.normal(0, 0.5, (n, 3)), rng.normal(0, 50.0, (n, 3))
    mass = rng.choice([1.008, 15.9994], n)
    x, v, w = dev(rng.uniform(0, 3, (n, 3))), dev(v0), dev(w0)
    ctx.bind_state(x, v, dev(mass))
    ctx.bind_buffer(5, w)
    h, friction, kT, Q = 0.000125, 20.0, 2.494, 2.494e-4
    z = float(np.exp(-2 * h * friction))
    bid = ctx.bath_define_nhl(h, z, kT, Q, friction, 5)
    ctx.expr_seed(777)
    ctx.run_ops([B.Op(B.OP_BATH, bid, B.SLOT_V, 0, 0.0)] * 2, 1)
    ctx.check()
    rv, rw = v0.copy(), w0.copy()
    m = mass[:, None]
    for k in (1, 2):
        u1, u2 = XO.uniforms(3 * n, 0, 777, (1 << 63) | k)
        g = (np.sqrt(-2.0 * np.log(u1)) * np.cos(6.283185307179586476925 * u2)).reshape(n, 3)
        rv = rv * np.exp(-h * rw)
        rw = z * rw + np.sqrt(kT * (1.0 - z * z) / Q) * g + (m * rv * rv - kT) * (1.0 - z) / (Q * friction)
        rv = rv * np.exp(-h * rw)
    assert np.abs(v.cpu().numpy() - rv).max() < 1e-13 and np.abs(w.cpu().numpy() - rw).max() < 1e-10
    ctx.close()


def test_nhl_bath_inside_the_inner_loop_kernel_is_bit_identical(spcfw):
    """NHL_R: the inner-loop kernel carries the Nose-Hoover-Langevin block between its two half moves for all n0 iterations;
    the trajectory and the thermostat velocities are bit-identical to the op-by-op execution (same amm_nhl_step, same counters)."""
    c = spcfw
    out = []
    for fuse in (True, False):
        system = system_from_arrays(c, nonbondedMethod='CutoffPeriodic')
        respa = atomsmm.RESPASystem(system, 7 * unit.angstroms, 5 * unit.angstroms)
        nb = atomsmm.hijackForce(respa, atomsmm.findNonbondedForce(respa))
        outer = atomsmm.DampedSmoothedForce(0.29 / unit.angstroms, 10 * unit.angstroms, 9 * unit.angstroms).importFrom(nb)
        outer.setForceGroup(2)
        outer.addTo(respa)
        integrator = atomsmm.NHL_R_Integrator(2 * unit.femtoseconds, [4, 2, 1], 300 * unit.kelvin, 10 * unit.femtoseconds, 5 / unit.picoseconds)
        integrator.setRandomNumberSeed(99)
        context = openmm.Context(respa, integrator)
        context._engine.ctx.set_fuse_inner(fuse)
        context.setPositions(c['positions'] * unit.nanometers)
        context.setVelocitiesToTemperature(300 * unit.kelvin, 3)
        integrator.step(12)
        assert context._engine._interpreted is False
        st = context.getState(getPositions=True, getVelocities=True)
        v2 = np.array([list(row) for row in integrator.getPerDofVariableByName('v2')])
        out.append((st.getPositions(asNumpy=True)._value.copy(), st.getVelocities(asNumpy=True)._value.copy(), v2))
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)
    assert np.abs(out[0][0] - c['positions']).max() > 1e-3 and np.abs(out[0][2]).max() > 0


def test_native_isokinetic_kick_and_sin_bath_vs_numpy():
    """Isokinetic mode (amm_iso_define): AMM_OP_KICK is v <- v cosh(z) + sqrt(LkT/m) sinh(z), z = coef F/sqrt(m LkT), followed by
    the rescale H = sqrt(LkT/(m v^2 + Q1 v1^2/2)) of (v, v1); the stochastic-isokinetic bath op (amm_bath_define_sin) is
    v1 <- v1 exp(-h v2); rescale; v2 <- z v2 + sqrt(kT (1 - z^2)/Q2) gaussian + (Q1 v1^2 - kT)(1 - z)/(Q2 friction);
    v1 <- v1 exp(-h v2); rescale (SIN_R_Integrator with L = 1, propagators.py:276-355, 1045-1105) -- against numpy."""
    rng = np.random.default_rng(8)
    n = 2000
    ctx = B.HipContext(n, np.array([3.0, 3.0, 3.0]))
    mass = rng.choice([1.008, 15.9994], n)
    kT, Q1, Q2, friction, h = 2.494, 2.494e-4, 2.494e-4, 20.0, 0.000125
    LkT = kT
    v1_0 = rng.normal(0, np.sqrt(kT / Q1), (n, 3))
    v0 = rng.normal(0, 1.0, (n, 3))
    scale = np.sqrt(LkT / (mass[:, None] * v0 ** 2 + 0.5 * Q1 * v1_0 ** 2))          # start on the isokinetic surface
    v0, v1_0 = scale * v0, scale * v1_0
    v2_0 = rng.normal(0, np.sqrt(kT / Q2), (n, 3))
    force = rng.normal(0, 500.0, (n, 3))
    x, v, f, v1, v2 = dev(rng.uniform(0, 3, (n, 3))), dev(v0), dev(force), dev(v1_0), dev(v2_0)
    ctx.bind_state(x, v, dev(mass))
    ctx.bind_buffer(0, f)
    ctx.bind_buffer(4, v1)
    ctx.bind_buffer(5, v2)
    ctx.iso_define(True, LkT, Q1, 4)
    z = float(np.exp(-2 * h * friction))
    bid = ctx.bath_define_sin(h, z, kT, Q2, friction, 5)
    ctx.expr_seed(31337)
    coef = 0.00025
    ctx.run_ops([B.Op(B.OP_KICK, 0, -1, 0, coef), B.Op(B.OP_BATH, bid, B.SLOT_V, 0, 0.0), B.Op(B.OP_KICK, 0, -1, 0, coef)], 1)
    ctx.check()
    m = mass[:, None]

    def rescale(a, b):
        H = np.sqrt(LkT / (m * a * a + 0.5 * Q1 * b * b))
        return H * a, H * b

    def kick(a, b):
        zz = coef * force / np.sqrt(m * LkT)
        return rescale(a * np.cosh(zz) + np.sqrt(LkT / m) * np.sinh(zz), b)
    rv, r1 = kick(v0, v1_0)
    u1, u2 = XO.uniforms(3 * n, 0, 31337, (1 << 63) | 1)
    g = (np.sqrt(-2.0 * np.log(u1)) * np.cos(6.283185307179586476925 * u2)).reshape(n, 3)
    r1 = r1 * np.exp(-h * v2_0)
    rv, r1 = rescale(rv, r1)
    r2 = z * v2_0 + np.sqrt(kT * (1 - z * z) / Q2) * g + (Q1 * r1 * r1 - kT) * (1 - z) / (Q2 * friction)
    r1 = r1 * np.exp(-h * r2)
    rv, r1 = rescale(rv, r1)
    rv, r1 = kick(rv, r1)
    assert np.abs(v.cpu().numpy() - rv).max() < 1e-12 * np.abs(rv).max()
    assert np.abs(v1.cpu().numpy() - r1).max() < 1e-12 * np.abs(r1).max()
    assert np.abs(v2.cpu().numpy() - r2).max() < 1e-12 * np.abs(r2).max()
    lhs = m * v.cpu().numpy() ** 2 + 0.5 * Q1 * v1.cpu().numpy() ** 2
    assert np.abs(lhs / LkT - 1.0).max() < 1e-13
    ctx.iso_define(False)
    ctx.close()


def test_sin_r_inside_the_inner_loop_kernel_is_bit_identical(spcfw):
    """SIN_R (L = 1): isokinetic kicks and the stochastic-isokinetic bath carried by the inner-loop kernel (thermostat velocities
    v1, v2 in registers across the n0 iterations) give the op-by-op trajectory bit for bit."""
    c = spcfw
    out = []
    for fuse in (True, False):
        system = system_from_arrays(c, nonbondedMethod='CutoffPeriodic')
        respa = atomsmm.RESPASystem(system, 7 * unit.angstroms, 5 * unit.angstroms)
        nb = atomsmm.hijackForce(respa, atomsmm.findNonbondedForce(respa))
        outer = atomsmm.DampedSmoothedForce(0.29 / unit.angstroms, 10 * unit.angstroms, 9 * unit.angstroms).importFrom(nb)
        outer.setForceGroup(2)
        outer.addTo(respa)
        integrator = atomsmm.SIN_R_Integrator(2 * unit.femtoseconds, [4, 2, 1], 300 * unit.kelvin, 10 * unit.femtoseconds, 5 / unit.picoseconds)
        integrator.setRandomNumberSeed(99)
        context = openmm.Context(respa, integrator)
        context._engine.ctx.set_fuse_inner(fuse)
        context.setPositions(c['positions'] * unit.nanometers)
        context.setVelocitiesToTemperature(300 * unit.kelvin, 3)
        integrator.step(12)
        assert context._engine._interpreted is False
        st = context.getState(getPositions=True, getVelocities=True)
        extra = [np.array([list(row) for row in integrator.getPerDofVariableByName(name)]) for name in ('v1_0', 'v2_0')]
        out.append([st.getPositions(asNumpy=True)._value.copy(), st.getVelocities(asNumpy=True)._value.copy()] + extra)
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)
    assert np.abs(out[0][0] - c['positions']).max() > 1e-3
