"""Config C2 of BASELINE.json -- 32 768-atom Lennard-Jones fluid, NearNonbondedForce only (forces.py:655-670), fp64 -- as `bench.py
--config c2` times it: velocity Verlet through the AtomsMM-shaped API (UnconstrainedVelocityVerletPropagator, propagators.py:1136-1153),
whose closing half kick rides on the next step's kick + move launch (csrc/abi.hip: deferred kicks).  The timed program is pinned to
the CPU oracle here, at full size: the same steps as plain kick / move / pair_eval calls of oracle/oracle.py."""
import numpy as np
import pytest

import atomsmm_amd as atomsmm
from atomsmm_amd import openmm, unit
from atomsmm_amd.openmm import app
from atomsmm_amd.testing import lj_fluid, system_from_arrays
from oracle import oracle as O  # checker only

pytestmark = pytest.mark.gpu
KB = 0.0083144626181532


def _c2_simulation(case, rc, rs, dt_fs, options=None):
    n = len(case['positions'])
    system = system_from_arrays(case, nonbondedMethod='CutoffPeriodic', cutoff=rc)
    nb = atomsmm.hijackForce(system, atomsmm.findNonbondedForce(system))
    near = atomsmm.NearNonbondedForce(rc * unit.nanometers, rs * unit.nanometers, 'force-switch').importFrom(nb)
    near.addTo(system)
    integrator = atomsmm.UnconstrainedVelocityVerletPropagator().integrator(dt_fs * unit.femtoseconds)
    simulation = app.Simulation(app.Topology(n), system, integrator, openmm.Platform.getPlatformByName('HIP'), options)
    simulation.context.setPositions(case['positions'] * unit.nanometers)
    simulation.context.setVelocities(case['velocities'])
    return simulation


def test_c2_full_size_velocity_verlet_vs_oracle():
    """5 + 5 velocity-Verlet steps of 4 fs (two calls: the deferred closing kick is flushed at the end of a call and taken up again) of
    the full-size C2 system against the oracle-driven loop: positions to 1e-11 nm, velocities to 1e-9 nm/ps; the launch count of the
    timed path (chargeless kernel where the library has one) is read back so that a silent change of path shows."""
    case = lj_fluid(32)
    n = len(case['positions'])
    assert n == 32768
    sigma = float(case['sigma'][0])
    rc, rs, dt = 2.5 * sigma, 0.9 * 2.5 * sigma, 0.004
    rng = np.random.default_rng(7)
    case['velocities'] = rng.normal(size=(n, 3)) * np.sqrt(KB * 100.0 / case['mass'])[:, None]
    simulation = _c2_simulation(case, rc, rs, 4.0, {'Skin': '0.2'})          # (the Verlet buffer bench.py's C2 leg runs with: C2_SKIN_NM)
    d = O.desc(O.ADJ['force-switch'], rc=rc, rc0=rc, rs0=rs)
    force = lambda p: O.pair_eval(d, p, case['box'], case['charge'], case['sigma'], case['epsilon'], None, use_cells=True)[1]
    x, v, m = case['positions'].copy(), case['velocities'].copy(), case['mass']
    f = force(x)
    for chunk in range(2):
        simulation.step(5)
        for _ in range(5):
            O.kick(v, f, m, 0.5 * dt)
            O.move(x, v, dt)
            f = force(x)
            O.kick(v, f, m, 0.5 * dt)
        st = simulation.context.getState(getPositions=True, getVelocities=True)
        assert np.abs(st.getPositions(asNumpy=True)._value - x).max() < 1e-11
        assert np.abs(st.getVelocities(asNumpy=True)._value - v).max() < 1e-9
    assert np.abs(x - case['positions']).max() > 1e-3
    eng = simulation.context._engine
    st = eng.ctx.pair_stats(eng.pair_force_ids(0)[0])
    assert st['list_kind'] == 0 and st['has_table'] == 1          # per-atom rows, tabulated force-only kernel ...
    assert st['chargeless'] == 1                                   # ... in its instantiation without the Coulomb table (all q = 0)
    eng.ctx.check()
