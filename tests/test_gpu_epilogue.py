"""The inner RESPA loop as the epilogue of the molecule-row pair kernels (csrc/cluster.hip: cepi_rows; amm_run_ops plans it).

The reference runs `[kicks] ; n0 x { v += c1 f0/m ; x += d v ; f0 = bonded(x) ; v += c2 f0/m }` as CustomIntegrator steps
(propagators.py:933-973 unrolled).  Here, for a box of flexible three-site molecules, the wavefront that has just summed a
molecule's rows runs the loop for that molecule and writes the sorted copies of the next pair evaluation.  It must be the SAME
trajectory, bit for bit, as the launches of their own (option fuse_epilogue = 0) and as the plain one-op-per-launch sequence
(amm_set_fuse_inner(0)); against the CPU oracle the fused path is checked by tests/test_gpu_api.py and by smoke()."""
import numpy as np
import pytest

import atomsmm_amd as atomsmm
from atomsmm_amd import openmm, unit
from atomsmm_amd.testing import system_from_arrays, tip3p_box

pytestmark = pytest.mark.gpu


def _respa_system(c, outer='damped'):
    system = system_from_arrays(c, nonbondedMethod='CutoffPeriodic' if outer != 'pme' else 'PME')
    respa = atomsmm.RESPASystem(system, 0.7 * unit.nanometers, 0.5 * unit.nanometers)
    if outer == 'damped':
        nb = atomsmm.hijackForce(respa, atomsmm.findNonbondedForce(respa))
        f = atomsmm.DampedSmoothedForce(2.9 / unit.nanometers, 1.0 * unit.nanometers, 0.9 * unit.nanometers).importFrom(nb)
        f.setForceGroup(2)
        f.addTo(respa)
    return respa


def _run(c, loops, steps, mode, outer='damped', dt=4.0, chunks=1, options=()):
    """mode: 'epilogue' (default path), 'launches' (fuse_epilogue = 0), 'plain' (no fusion at all)."""
    respa = _respa_system(c, outer)
    integrator = atomsmm.RespaPropagator(loops).integrator(dt * unit.femtoseconds)
    context = openmm.Context(respa, integrator, openmm.Platform.getPlatformByName('HIP'))
    ctx = context._engine.ctx
    ctx.set_fuse_inner(mode != 'plain')
    ctx.set_option('fuse_epilogue', 1 if mode == 'epilogue' else 0)
    for name, value in options:
        ctx.set_option(name, value)
    context.setPositions(c['positions'] * unit.nanometers)
    context.setVelocities(c['velocities'])
    for _ in range(chunks):
        integrator.step(steps)
        st = context.getState(getPositions=True, getVelocities=True)        # (between the chunks too: the state must be whole)
    eng = context._engine
    out = dict(x=st.getPositions(asNumpy=True)._value.copy(), v=st.getVelocities(asNumpy=True)._value.copy(),
               stats=ctx.run_stats(), builds=ctx.pair_stats(eng.pair_force_ids(1)[0])['n_builds'],
               kind=ctx.pair_stats(eng.pair_force_ids(1)[0])['list_kind'],
               f1=eng._buffers['f1'].cpu().numpy().copy() if 'f1' in eng._buffers else None)
    ctx.check()
    return out


@pytest.mark.parametrize('loops,nside', [([4, 2, 1], 8), ([4, 2, 1], 12), ([3, 1, 1], 8), ([2, 3, 1], 8)])
def test_epilogue_trajectory_bit_identical(loops, nside):
    """RESPASystem(0.7, 0.5) + DampedSmoothedForce, flexible TIP3P: 24 outer steps in three calls -- the list is rebuilt several times,
    calls begin and end between two fused launches.  [3, 1, 1]: the near force is only ever evaluated together with the outer one, so
    the next evaluation reads the buffer this one reads (the second copy of the sorted positions); [4, 2, 1] alternates."""
    c = tip3p_box(nside)
    ref = _run(c, loops, 8, 'plain', chunks=3)
    mid = _run(c, loops, 8, 'launches', chunks=3)
    new = _run(c, loops, 8, 'epilogue', chunks=3)
    assert new['kind'] == 1
    assert new['stats']['epilogues'] > 0 and mid['stats']['epilogues'] == 0 and ref['stats']['epilogues'] == 0
    # every middle / boundary evaluation of a call but the last boundary carries the loop; their next evaluations need no gather
    # (the engine may run a context's first step as a call of its own: one more boundary without a successor)
    per_call = 8 * loops[1] - 1
    assert 3 * per_call - 1 <= new['stats']['epilogues'] <= 3 * per_call
    assert new['stats']['copies_current'] == new['stats']['epilogues']
    assert new['builds'] == ref['builds'] == mid['builds'] and new['builds'] >= 3
    for other in (mid, ref):
        assert np.array_equal(new['x'], other['x']) and np.array_equal(new['v'], other['v'])
    assert np.isfinite(new['x']).all()


def test_epilogue_with_more_lanes_per_row():
    """Rows shared out over 8 and 16 lanes (what slices and the remainder phases of big boxes use): the molecule's four lanes are the
    first four of its row's lanes."""
    c = tip3p_box(8)
    ref = _run(c, [4, 2, 1], 6, 'plain')
    for lanes in (8, 16):
        new = _run(c, [4, 2, 1], 6, 'epilogue', options=(('lanes_per_row', lanes),))
        old = _run(c, [4, 2, 1], 6, 'launches', options=(('lanes_per_row', lanes),))
        assert new['stats']['epilogues'] > 0
        # (the lanes per row change the order of summation of a row: compare like with like, bit for bit, and the 4-lane run to rounding)
        assert np.array_equal(new['x'], old['x']) and np.array_equal(new['v'], old['v'])
        assert np.abs(new['x'] - ref['x']).max() < 1e-9


def test_epilogue_stays_off_where_it_cannot_run():
    """PME outer force: group 2 holds the reciprocal space too, so the boundary pass cannot carry the kicks (they need the whole f2);
    the middle evaluations (the near force alone) still do.  Bit-identical either way."""
    c = tip3p_box(8)
    ref = _run(c, [4, 2, 1], 6, 'plain', outer='pme')
    new = _run(c, [4, 2, 1], 6, 'epilogue', outer='pme')
    assert new['stats']['epilogues'] == 6          # one middle evaluation per step
    assert np.array_equal(new['x'], ref['x']) and np.array_equal(new['v'], ref['v'])


def test_epilogue_off_without_site_tables():
    """Without site-site tables the kernels have no epilogue instantiation: the plan is declined, launches of their own run it."""
    c = tip3p_box(8)
    ref = _run(c, [4, 2, 1], 4, 'plain', options=(('site_tab', 0),))
    new = _run(c, [4, 2, 1], 4, 'epilogue', options=(('site_tab', 0),))
    assert new['stats']['epilogues'] == 0
    assert np.array_equal(new['x'], ref['x']) and np.array_equal(new['v'], ref['v'])


@pytest.mark.parametrize('loops,kwargs', [([1, 1, 1], {}), ([2, 2, 2], {}), ([4, 2, 1], {'has_memory': True}), ([3, 2], {}), ([4, 2, 1], {'blitz': True})])
def test_epilogue_other_step_programs(loops, kwargs):
    """Other shapes of the RESPA program (propagators.py:897-973): one iteration per level, an outermost loop of two, force memory,
    two levels only (group 2 is never integrated), `blitz` -- whatever the planner makes of them (fused where the ops behind an EVAL are
    kicks + the inner loop, declined elsewhere), the trajectory equals the plain op sequence bit for bit, in two calls of odd length."""
    c = tip3p_box(8)

    def run(mode):
        respa = _respa_system(c)
        integrator = atomsmm.RespaPropagator(loops, **kwargs).integrator(0.5 * float(np.prod(loops)) * unit.femtoseconds)      # (0.5 fs innermost)
        context = openmm.Context(respa, integrator, openmm.Platform.getPlatformByName('HIP'))
        ctx = context._engine.ctx
        ctx.set_fuse_inner(mode != 'plain')
        context.setPositions(c['positions'] * unit.nanometers)
        context.setVelocities(c['velocities'])
        integrator.step(5)
        integrator.step(4)
        st = context.getState(getPositions=True, getVelocities=True)
        ctx.check()
        return st.getPositions(asNumpy=True)._value, st.getVelocities(asNumpy=True)._value, ctx.run_stats()

    x1, v1, s1 = run('epilogue')
    x0, v0, s0 = run('plain')
    assert s0['epilogues'] == 0
    assert np.array_equal(x1, x0) and np.array_equal(v1, v0)
    assert np.isfinite(x1).all()
