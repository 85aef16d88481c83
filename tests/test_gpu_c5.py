"""Config C5 of BASELINE.json: a solvated bonded chain with 1-4 exceptions (NonbondedExceptionsForce in
group 0), RESPA near / outer split and one AFED extended variable lambda_vdw that couples a 30-atom solute through the
softcore force of SolvationSystem -- at the full ~250 000 atoms for the static checks, and at a 4 233-atom instance of the
same generator for the step-by-step comparison of the AFED program with its oracle-driven evaluation."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')

import atomsmm_amd as atomsmm  # noqa: E402
from atomsmm_amd import openmm, unit  # noqa: E402
from atomsmm_amd.testing import build_c5_system, solvated_chain  # noqa: E402
from oracle import oracle as O  # noqa: E402  (checker only)
from oracle.afed_cpu import AfedCPU  # noqa: E402


def group_forces(context, group):
    st = context.getState(getForces=True, getEnergy=True, groups={group})
    return st.getPotentialEnergy()._value, st.getForces(asNumpy=True)._value


def check_groups(case, context, ref, lam):
    e0, f0 = group_forces(context, 0)
    r0e, r0f = ref.group_energy_forces(0)
    lrc = O.softcore_lrc(case['sigma'], case['epsilon'], ref.codes, case['box'], 1.0, 0.9, lam)
    assert e0 == pytest.approx(r0e + lrc, rel=1e-9)
    assert np.abs(f0 - r0f).max() <= 1e-9 * np.abs(r0f).max()
    for g in (1, 2):
        e, f = group_forces(context, g)
        re_, rf = ref.group_energy_forces(g)
        assert e == pytest.approx(re_, rel=1e-10)
        assert np.abs(f - rf).max() <= 1e-9 * np.abs(rf).max()
        # without the energy the pair forces take the force-only traversal (the kernel the step program runs)
        f_only = context.getState(getForces=True, groups={g}).getForces(asNumpy=True)._value
        assert np.abs(f_only - rf).max() <= 1e-9 * np.abs(rf).max()
    f_only = context.getState(getForces=True, groups={0}).getForces(asNumpy=True)._value
    assert np.abs(f_only - r0f).max() <= 1e-9 * np.abs(r0f).max()
    d = context._engine.energy_derivative('lambda_vdw')
    assert d == pytest.approx(ref.dE_dlambda(), rel=2e-6)
    # which lists ran: the near and the outer force walk a HYBRID list (molecule rows for the water-water pairs, per-atom rows for
    # every pair with a chain or solute atom) -- a silent fall-back to per-atom rows for the whole box would stay green above
    eng = context._engine
    n_rest = len(case['chain']) + len(case['solute'])
    for g in (1, 2):
        for pid in eng.pair_force_ids(g):
            st = eng.ctx.pair_stats(pid)
            assert st['list_kind'] == 2 and st['n_rest_atoms'] == n_rest, st


def test_c5_groups_and_afed_steps_vs_oracle_small():
    """4 233 atoms: per-group energies and forces, deriv(energy, lambda_vdw), then 3 AFED steps (RESPA [2,2,1] inside,
    n = 2 sub-steps, Nose-Hoover bath on lambda) against the oracle-driven evaluation of the same program."""
    case = solvated_chain(nside=12, n_chain=300, n_solute=30)
    respa = build_c5_system(case)
    inner = atomsmm.RespaPropagator([2, 2, 1]).integrator(1 * unit.femtoseconds)
    var = atomsmm.ExtendedSystemVariable('lambda_vdw', 50, 2.5, 20 * unit.femtoseconds)
    integrator = atomsmm.AdiabaticDynamicsIntegrator(inner, 2, [var])
    context = openmm.Context(respa, integrator)
    context.setPositions(case['positions'] * unit.nanometers)
    context.setVelocities(case['velocities'])
    context.setParameter('lambda_vdw', 0.8)
    ref = AfedCPU(case, loops=(2, 2, 1), dt=0.001, nsteps=2, mass=50.0, kT=2.5, tau=0.02, lam=0.8, v_lam=0.0)
    check_groups(case, context, ref, 0.8)
    integrator.step(0)                                              # runs the (random) initialisation hook ...
    integrator.setGlobalVariableByName('_v_lambda_vdw', 0.05)       # ... which these known starting values replace
    integrator.setGlobalVariableByName('_v_eta_lambda_vdw', 0.0)
    ref.v_lam = 0.05
    for _ in range(3):
        integrator.step(1)
        ref.step(1)
        x = context.getState(getPositions=True).getPositions(asNumpy=True)._value
        assert np.abs(x - ref.x).max() < 1e-9
        assert context.getParameter('lambda_vdw') == pytest.approx(ref.lam, abs=1e-9)
        assert integrator.getGlobalVariableByName('_v_lambda_vdw') == pytest.approx(ref.v_lam, rel=1e-6, abs=1e-9)
    assert abs(ref.lam - 0.8) > 1e-4                                  # lambda did move


def test_c5_full_size_groups_and_afed():
    """~249 000 atoms (82 015 waters + the 3 000-atom chain + the solute): per-group energies and forces and
    deriv(energy, lambda_vdw) vs the oracle; then AFED steps: lambda moves and stays within its walls, the Context
    parameter follows the extended variable, nothing overflows (amm_check) and the energies stay finite."""
    case = solvated_chain()
    n = len(case['positions'])
    assert n > 245000 and len(case['chain']) == 3000 and len(case['solute']) == 30
    respa = build_c5_system(case)
    inner = atomsmm.RespaPropagator([4, 2, 1]).integrator(1 * unit.femtoseconds)
    var = atomsmm.ExtendedSystemVariable('lambda_vdw', 50, 2.5, 20 * unit.femtoseconds)
    integrator = atomsmm.AdiabaticDynamicsIntegrator(inner, 2, [var])
    integrator.setRandomNumberSeed(7)
    context = openmm.Context(respa, integrator)
    context.setPositions(case['positions'] * unit.nanometers)
    context.setVelocities(case['velocities'])
    context.setParameter('lambda_vdw', 0.6)
    ref = AfedCPU(case, lam=0.6)
    check_groups(case, context, ref, 0.6)
    seen = []
    for _ in range(4):
        integrator.step(2)
        seen.append(context.getParameter('lambda_vdw'))
    assert all(0.0 <= v <= 1.0 for v in seen) and max(seen) - min(seen) > 1e-5, seen
    st = context.getState(getEnergy=True)
    assert np.isfinite(st.getPotentialEnergy()._value) and np.isfinite(st.getKineticEnergy()._value)


def test_c5_fused_inner_iterations_bit_identical():
    """The inner RESPA iterations of a system that is not pure water (chain + solute + waters: the group of the innermost loop
    holds the bond lists AND the softcore pair force): softcore force without a list, terms, and ONE launch that gathers the
    terms' forces and applies the kicks and the move that follow (csrc/bonded.hip: k_terms_gather_kicks) -- against the same
    program with every fusion switched off (one launch per op), bit for bit, through list rebuilds."""
    case = solvated_chain(nside=12, n_chain=300, n_solute=30)

    def run(fuse):
        respa = build_c5_system(case)
        integrator = atomsmm.RespaPropagator([4, 2, 1]).integrator(2 * unit.femtoseconds)
        context = openmm.Context(respa, integrator)
        context._engine.ctx.set_fuse_inner(fuse)
        context._engine.ctx.set_option('terms_from', 1)        # term-parallel bond lists whatever the size (the full-size path)
        context.setPositions(case['positions'] * unit.nanometers)
        context.setVelocities(case['velocities'])
        context.setParameter('lambda_vdw', 0.7)
        integrator.step(12)
        st = context.getState(getPositions=True, getVelocities=True)
        builds = context._engine.ctx.pair_stats(context._engine.pair_force_ids(2)[0])['n_builds']
        soft = [context._engine.ctx.pair_stats(pid) for pid in context._engine.pair_force_ids(0)]
        soft = [s_ for s_ in soft if s_['list_kind'] == 3]
        assert len(soft) == 1
        return st.getPositions(asNumpy=True)._value, st.getVelocities(asNumpy=True)._value, builds, soft[0]

    x1, v1, b1, s1 = run(True)
    x0, v0, b0, s0 = run(False)
    assert b1 == b0 and b1 >= 2
    assert np.array_equal(x1, x0) and np.array_equal(v1, v0)
    assert np.isfinite(x1).all()
    # the fused iterations walked the atoms near the solute only (all but the evaluation after each build of the companion
    # list); one launch per op walks every atom every time
    n_eval = 12 * 8
    assert s1['n_candidate_walks'] >= n_eval - 3 * b1 - 4 and 30 < s1['n_candidates'] < len(x1) // 2, s1
    assert s0['n_candidate_walks'] == 0 and s0['n_candidates'] == 0, s0


def test_c5_candidate_walks_bit_identical():
    """Fused inner iterations with and without the candidate walk of the list-free softcore force (option group_candidates):
    the same trajectory bit for bit, through list rebuilds, after a jump of the positions set from the host, and with the
    buffer of the companion list nearly used up (a small skin: many rebuilds)."""
    case = solvated_chain(nside=12, n_chain=300, n_solute=30)

    def run(cand, skin):
        respa = build_c5_system(case)
        integrator = atomsmm.RespaPropagator([4, 2, 1]).integrator(2 * unit.femtoseconds)
        context = openmm.Context(respa, integrator, None, {'Skin': str(skin)})
        context._engine.ctx.set_option('group_candidates', cand)
        context._engine.ctx.set_option('terms_from', 1)
        context.setPositions(case['positions'] * unit.nanometers)
        context.setVelocities(case['velocities'])
        context.setParameter('lambda_vdw', 0.7)
        integrator.step(6)
        x = context.getState(getPositions=True).getPositions(asNumpy=True)._value
        shifted = x.copy()
        shifted[case['solute']] += np.array([0.31, -0.17, 0.23])        # the solute jumps: other waters are its neighbours now
        context.setPositions(shifted * unit.nanometers)
        integrator.step(6)
        st = context.getState(getPositions=True, getVelocities=True)
        # deriv(energy, lambda) after each of a few more steps: the energy-only launch walks the candidates too (its rows are
        # scratch) unless the step's last pair evaluation has just rebuilt the companion list (the list of candidates is then
        # renewed by the next launch that owns its rows, not by this one)
        def soft_stats():
            return [s_ for s_ in (context._engine.ctx.pair_stats(pid) for pid in context._engine.pair_force_ids(0)) if s_['list_kind'] == 3][0]
        derivs, walks = [], 0
        for _ in range(5):
            before = soft_stats()['n_candidate_walks']
            derivs.append(context._engine.energy_derivative('lambda_vdw'))
            walks += soft_stats()['n_candidate_walks'] - before
            integrator.step(1)
        st = context.getState(getPositions=True, getVelocities=True)
        return st.getPositions(asNumpy=True)._value, st.getVelocities(asNumpy=True)._value, soft_stats(), derivs, walks

    for skin in (0.1, 0.03):
        x1, v1, s1, d1, w1 = run(1, skin)
        x0, v0, s0, d0, w0 = run(0, skin)
        assert s1['n_candidate_walks'] > 20 and s0['n_candidate_walks'] == 0, (s1, s0)
        assert np.array_equal(x1, x0) and np.array_equal(v1, v0)
        assert np.isfinite(x1).all()
        assert w0 == 0 and (w1 >= 2 or skin < 0.1)      # derivative launches walked candidates (0.03 nm: every step ends in a rebuild)
        assert d1 == pytest.approx(d0, rel=1e-11)       # (block partial sums of the energy: other groupings, same pairs)


def test_c5_full_size_fused_inner_iterations_bit_identical():
    """The same comparison at the full 249 075 atoms (82 015 four-lane components + 3 030 atoms of big components, 258 000 terms of
    which 12 000 are parked; the softcore force without a list): 3 RESPA [4,2,1] steps, fused == unfused bit for bit."""
    case = solvated_chain()

    def run(fuse):
        respa = build_c5_system(case)
        integrator = atomsmm.RespaPropagator([4, 2, 1]).integrator(2 * unit.femtoseconds)
        context = openmm.Context(respa, integrator)
        context._engine.ctx.set_fuse_inner(fuse)
        context.setPositions(case['positions'] * unit.nanometers)
        context.setVelocities(case['velocities'])
        context.setParameter('lambda_vdw', 0.7)
        integrator.step(3)
        st = context.getState(getPositions=True, getVelocities=True)
        return st.getPositions(asNumpy=True)._value, st.getVelocities(asNumpy=True)._value

    x1, v1 = run(True)
    x0, v0 = run(False)
    assert np.array_equal(x1, x0) and np.array_equal(v1, v0)
    assert np.isfinite(x1).all() and np.abs(x1 - case['positions']).max() > 1e-4


def test_c5_full_size_afed_step_vs_oracle():
    """ONE AFED step of the program `bench.py --config c5` times -- 4 RESPA [4,2,1] steps of 2 fs around the lambda block -- at the full
    249 075 atoms against oracle/afed_cpu.py (the C oracle's OpenMP cell traversals driven by numpy): positions to 1e-9 nm, lambda and
    its velocity.  On the way the molecule rows of the hybrid list are rebuilt, the inner iterations are the two-launch form and the
    list-free softcore force walks its candidates: asserted, so that a silent fall-back cannot pass."""
    case = solvated_chain()
    respa = build_c5_system(case)
    inner = atomsmm.RespaPropagator([4, 2, 1]).integrator(2 * unit.femtoseconds)
    var = atomsmm.ExtendedSystemVariable('lambda_vdw', 50, 2.5, 20 * unit.femtoseconds)
    integrator = atomsmm.AdiabaticDynamicsIntegrator(inner, 2, [var])
    context = openmm.Context(respa, integrator)
    context.setPositions(case['positions'] * unit.nanometers)
    context.setVelocities(case['velocities'])
    context.setParameter('lambda_vdw', 0.6)
    integrator.step(0)                                              # the (random) initialisation hook ...
    integrator.setGlobalVariableByName('_v_lambda_vdw', 0.05)       # ... replaced by known starting values
    integrator.setGlobalVariableByName('_v_eta_lambda_vdw', 0.0)
    ref = AfedCPU(case, loops=(4, 2, 1), dt=0.002, nsteps=2, mass=50.0, kT=2.5, tau=0.02, lam=0.6, v_lam=0.05)
    integrator.step(1)
    ref.step(1)
    x = context.getState(getPositions=True).getPositions(asNumpy=True)._value
    assert np.abs(x - ref.x).max() < 1e-9
    assert np.abs(ref.x - case['positions']).max() > 1e-3                     # (the atoms did move)
    assert context.getParameter('lambda_vdw') == pytest.approx(ref.lam, abs=1e-9)
    assert integrator.getGlobalVariableByName('_v_lambda_vdw') == pytest.approx(ref.v_lam, rel=1e-6, abs=1e-9)
    eng = context._engine
    eng.ctx.check()
    for g in (1, 2):
        for pid in eng.pair_force_ids(g):
            st = eng.ctx.pair_stats(pid)
            assert st['list_kind'] == 2, st
    assert eng.ctx.pair_stats(eng.pair_force_ids(2)[0])['n_builds'] >= 2       # first build + at least one rebuild inside the step
    soft = [s_ for s_ in (eng.ctx.pair_stats(pid) for pid in eng.pair_force_ids(0)) if s_['list_kind'] == 3]
    assert len(soft) == 1 and soft[0]['n_candidate_walks'] > 0 and 30 < soft[0]['n_candidates'] < 20000, soft


def test_c5_full_size_candidate_walks_bit_identical():
    """scripts/check_c5_candidates.py as a test (shorter): the full-size AFED program with the candidate walk of the list-free softcore
    force against the run that walks every atom every time -- positions, velocities and lambda bit for bit over 6 AFED steps."""
    import bench

    def run(cand):
        sim, _case = bench.build_simulation_c5((4, 2, 1), 2.0)
        eng = sim.context._engine
        eng.ctx.set_option('group_candidates', cand)
        sim.step(6)
        eng._check()
        st = sim.context.getState(getPositions=True, getVelocities=True)
        soft = [s_ for s_ in (eng.ctx.pair_stats(p) for p in eng.pair_force_ids(0)) if s_['list_kind'] == 3][0]
        return st.getPositions(asNumpy=True)._value, st.getVelocities(asNumpy=True)._value, sim.context.getParameter('lambda_vdw'), soft

    x1, v1, l1, s1 = run(1)
    x0, v0, l0, s0 = run(0)
    assert s1['n_candidate_walks'] > 100 and s0['n_candidate_walks'] == 0, (s1, s0)
    assert np.array_equal(x1, x0) and np.array_equal(v1, v0) and l1 == l0


def test_far_flag_raised_by_the_kicks_and_move_launch():
    """the launch of plain kicks + move (csrc/integrate.hip: k_kicks_move_atoms) evaluates the lists' displacement triggers;
    it must raise BOTH flags -- "rebuild wanted" and "farther than the whole buffer" (AMM_FLAG_FAR) -- or the candidate walk of a
    list-free group force goes on trusting reference positions an atom has left.  A step program whose only move IS that launch
    (kick f0 ; kick f1 ; kick f2 + move), on the small C5 system; one water is shot 0.8 nm -- from beyond cutoff + twice the buffer of
    every solute atom to well inside the cutoff -- in a single step.  With and without the candidate walk: bit for bit."""
    case = solvated_chain(nside=12, n_chain=300, n_solute=30)
    box = np.asarray(case['box'])
    results = []
    for cand in (1, 0):
        respa = build_c5_system(case)
        integrator = openmm.CustomIntegrator(0.001)
        integrator.addComputePerDof('v', 'v+0.5*dt*f0/m')
        integrator.addComputePerDof('v', 'v+0.5*dt*f1/m')
        integrator.addComputePerDof('v', 'v+0.5*dt*f2/m')
        integrator.addComputePerDof('x', 'x+dt*v')
        context = openmm.Context(respa, integrator)
        eng = context._engine
        eng.ctx.set_option('group_candidates', cand)
        eng.ctx.set_option('terms_from', 1)
        context.setPositions(case['positions'] * unit.nanometers)
        context.setVelocities(case['velocities'])
        context.setParameter('lambda_vdw', 0.7)
        integrator.step(3)
        st = context.getState(getPositions=True, getVelocities=True)
        x, v = st.getPositions(asNumpy=True)._value, st.getVelocities(asNumpy=True)._value.copy()
        sol = x[case['solute']]
        oxy = 3 * np.arange(case['n_waters'])
        d = x[oxy][:, None, :] - sol[None, :, :]
        d -= box * np.rint(d / box)
        dist = np.linalg.norm(d, axis=2)
        nearest = dist.min(1)
        pick = int(np.argmin(np.abs(nearest - 1.5)))
        assert 1.3 < nearest[pick] < 1.7                      # outside cutoff (1.0) + twice the buffer (0.2) of every solute atom
        towards = -d[pick, int(np.argmin(dist[pick]))]
        towards /= np.linalg.norm(towards)
        v[oxy[pick]:oxy[pick] + 3] = 800.0 * towards          # 0.8 nm in the next 1 fs move
        context.setVelocities(v)
        integrator.step(1)                                     # the flight: the step's only move is the kicks + move launch
        # the next list-free evaluation is deriv(energy, lambda): it walks the candidates only while BOTH flags vouch for them
        # (no more steps: the water has landed on top of others)
        deriv = eng.energy_derivative('lambda_vdw')
        st = context.getState(getPositions=True, getVelocities=True)
        x1 = st.getPositions(asNumpy=True)._value
        d1 = x1[oxy[pick]][None, :] - x1[case['solute']]
        d1 -= box * np.rint(d1 / box)
        assert np.linalg.norm(d1, axis=1).min() < 0.95          # it did land inside the cutoff of a solute atom
        soft = [s_ for s_ in (eng.ctx.pair_stats(pid) for pid in eng.pair_force_ids(0)) if s_['list_kind'] == 3][0]
        results.append((x1, st.getVelocities(asNumpy=True)._value, soft, deriv))
    assert results[0][2]['n_candidate_walks'] > 0 and results[1][2]['n_candidate_walks'] == 0
    assert np.array_equal(results[0][0], results[1][0]) and np.array_equal(results[0][1], results[1][1])
    assert np.isfinite(results[0][0]).all()
    assert results[0][3] == pytest.approx(results[1][3], rel=1e-10) and results[0][3] != 0.0


def _afed_context(case, lam, v_lam, device_globals):
    respa = build_c5_system(case)
    inner = atomsmm.RespaPropagator([2, 2, 1]).integrator(1 * unit.femtoseconds)
    var = atomsmm.ExtendedSystemVariable('lambda_vdw', 50, 2.5, 20 * unit.femtoseconds)
    integrator = atomsmm.AdiabaticDynamicsIntegrator(inner, 2, [var])
    context = openmm.Context(respa, integrator)
    context._engine.device_globals = device_globals
    context.setPositions(case['positions'] * unit.nanometers)
    context.setVelocities(case['velocities'])
    context.setParameter('lambda_vdw', lam)
    integrator.step(0)
    integrator.setGlobalVariableByName('_v_lambda_vdw', v_lam)
    integrator.setGlobalVariableByName('_v_eta_lambda_vdw', 0.0)
    return context, integrator


@pytest.mark.parametrize('lam,v_lam', [(0.8, 0.05), (0.9985, 2.0)])
def test_afed_scalar_block_on_the_device(lam, v_lam):
    """AdiabaticDynamicsIntegrator's extended-variable block (integrators.py:701-737: lambda moves, reflects at the walls, its
    Nose-Hoover thermostat acts) runs on the device while its operands -- sums of deriv(energy, lambda) -- are still in flight
    (amm_expr_eval_scalar), the softcore kernels read lambda where it is (amm_pair_set_lambda_dev), and the host reads the scalars once
    per step() CALL instead of once per AFED step.  Six steps in one call against the waiting path (engine.device_globals = False) and
    against the CPU oracle; (0.9985, 2.0): lambda crosses 1 in the first step, the reflection (an if-block) is evaluated predicated."""
    case = solvated_chain(nside=12, n_chain=300, n_solute=30)
    ref = AfedCPU(case, loops=(2, 2, 1), dt=0.001, nsteps=2, mass=50.0, kT=2.5, tau=0.02, lam=lam, v_lam=v_lam)
    ref.step(6)
    out = {}
    for mode in (True, False):
        context, integrator = _afed_context(case, lam, v_lam, mode)
        eng = context._engine
        integrator.step(6)
        out[mode] = (context.getState(getPositions=True).getPositions(asNumpy=True)._value, context.getParameter('lambda_vdw'),
                     integrator.getGlobalVariableByName('_v_lambda_vdw'), integrator.getGlobalVariableByName('_v_eta_lambda_vdw'),
                     eng.n_scalar_evals, eng.n_settles, eng.n_scalar_launches)
        eng.ctx.check()
    dev, host = out[True], out[False]
    assert dev[4] >= 6 * 5 and host[4] == 0                 # the block's nonlinear steps (and the walls' test) ran on the device ...
    assert 6 <= dev[6] <= 6 * 6                              # ... as a few launches per AFED step (consecutive assignments share one) ...
    assert dev[5] <= 2 and host[5] >= 6                      # ... and the scalars were read once, at the end of the call
    assert np.abs(dev[0] - host[0]).max() < 1e-10
    for k in (1, 2, 3):
        assert dev[k] == pytest.approx(host[k], rel=1e-10, abs=1e-12)
    assert np.abs(dev[0] - ref.x).max() < 1e-9
    assert dev[1] == pytest.approx(ref.lam, abs=1e-9) and dev[2] == pytest.approx(ref.v_lam, rel=1e-6, abs=1e-9)
    if lam > 0.99:
        assert 0.9 < dev[1] < 1.0 and dev[2] < 1.0           # it bounced off the wall at 1 in the first half move (0.9985 + 0.002 x 2.0 > 1)


def test_afed_scalars_survive_a_full_buffer(monkeypatch):
    """The device scalars of a host-walked program are a finite buffer: when it runs full in the middle of a step() call the engine reads it
    (one blocking read), hands the library lambda's number back and starts over.  With room for 160 scalars instead of 2 048 that happens
    every few AFED steps: 20 steps in one call end where the waiting path ends."""
    from atomsmm_amd import engine as E
    case = solvated_chain(nside=12, n_chain=300, n_solute=30)
    out = {}
    for mode in (True, False):
        monkeypatch.setattr(E, '_SCALARS', 160 if mode else 2048)
        context, integrator = _afed_context(case, 0.7, -0.3, mode)
        eng = context._engine
        integrator.step(20)
        out[mode] = (context.getState(getPositions=True).getPositions(asNumpy=True)._value, context.getParameter('lambda_vdw'),
                     integrator.getGlobalVariableByName('_v_lambda_vdw'), eng.n_settles, eng.n_scalar_evals)
        eng.ctx.check()
    dev, host = out[True], out[False]
    assert dev[4] > 100 and 3 <= dev[3] < host[3]             # several reads forced by the full buffer, still fewer than one per step
    assert np.abs(dev[0] - host[0]).max() < 1e-9
    assert dev[1] == pytest.approx(host[1], abs=1e-10) and dev[2] == pytest.approx(host[2], rel=1e-8, abs=1e-10)
