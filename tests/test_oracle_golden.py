"""CPU tests: pin the oracle (oracle/amm_oracle.c) against the reference's own known-answer
energies (tests/golden/goldens.json, literals cited from /root/reference/tests) and check its
analytic forces against central differences of its energies.  Tolerance vs the reference literals:
rel 1e-6 (the reference's own pytest.approx default); observed agreement is 1e-12 or better except
G3 (7e-9, ill-conditioned b = 19) and G8 (4.6e-8, PME vs exact Ewald)."""
import itertools

import numpy as np
import pytest

from oracle import oracle as O
from helpers import solvation_respa_inputs

REL = 1e-6


def near(adj, rc, rs, **kw):
    return O.desc(O.ADJ[adj], rc=kw.pop('actual', rc), rc0=rc, rs0=rs, **kw)


@pytest.mark.parametrize('gid,adj', [('G1', None), ('G2', 'shift'), ('G3', 'force-switch')])
def test_near_goldens(spcfw, goldens, gid, adj):
    c = spcfw
    e, f, npairs = O.pair_eval(near(adj, 1.0, 0.95), c['positions'], c['box'], c['charge'], c['sigma'],
                               c['epsilon'], c['exc_pairs'])
    assert npairs == 314034                      # non-excluded pairs < 1.0 nm
    assert e == pytest.approx(goldens[gid]['value'], rel=REL)
    assert abs(f.sum(0)).max() < 1e-8            # Newton's third law


@pytest.mark.parametrize('gid,degree', [('G4', 1), ('G5', 2)])
def test_damped_goldens(spcfw, goldens, gid, degree):
    c = spcfw
    d = O.desc(O.DAMPED, rc=1.0, rswitch=0.95, alpha=2.9, degree=degree)
    e, _, _ = O.pair_eval(d, c['positions'], c['box'], c['charge'], c['sigma'], c['epsilon'], c['exc_pairs'],
                          want_forces=False)
    assert e == pytest.approx(goldens[gid]['value'], rel=1e-12)


def test_respa_near_golden_G6(spcfw, goldens):
    c = spcfw
    e, _, npairs = O.pair_eval(near('force-switch', 0.7, 0.5), c['positions'], c['box'], c['charge'],
                               c['sigma'], c['epsilon'], c['exc_pairs'], want_forces=False)
    assert npairs == 106161
    assert e == pytest.approx(goldens['G6']['value'], rel=1e-11)
    # group 31 = -step(rc0-r)*(near): same magnitude, opposite sign (tests/test_systems.py:145)
    em, _, _ = O.pair_eval(near('force-switch', 0.7, 0.5, sign=-1.0, flags=O.GUARD_RC0), c['positions'], c['box'],
                           c['charge'], c['sigma'], c['epsilon'], c['exc_pairs'], want_forces=False)
    assert em == pytest.approx(-goldens['G6']['value'], rel=1e-11)


def test_ewald_direct_golden_G7_and_reciprocal_G8(spcfw, goldens):
    c = spcfw
    alpha = np.sqrt(-np.log(2 * 5e-4)) / 1.0       # OpenMM: alpha = sqrt(-ln(2 tol))/rc
    d = O.desc(O.NONBONDED, rc=1.0, rswitch=0.9, alpha=alpha, flags=O.COULOMB_EWALD | O.SWITCH)
    e_dir, _, _ = O.pair_eval(d, c['positions'], c['box'], c['charge'], c['sigma'], c['epsilon'], c['exc_pairs'],
                              want_forces=False)
    e_exc, _ = O.ewald_exclusion(c['exc_pairs'], c['positions'], c['box'], c['charge'], alpha, want_forces=False)
    e_disp = O.dispersion_correction(c['sigma'], c['epsilon'], c['box'], 1.0, 0.9)
    assert e_dir + e_exc + e_disp == pytest.approx(goldens['G7']['value'], rel=1e-10)
    e_rec, _ = O.ewald_reciprocal(c['positions'], c['box'], c['charge'], alpha, 14)
    assert e_rec == pytest.approx(goldens['G8']['value'], rel=REL)


def test_reciprocal_golden_G14_heaq(heaq, goldens):
    """Second reciprocal-space literal: HEAQ in water with the solute charges scaled by lambda_coul = 0.5
    (tests/test_systems.py:84-101)."""
    h = heaq
    q = solvation_respa_inputs(h, 0.5)[0]
    alpha = np.sqrt(-np.log(2 * 5e-4)) / 1.0
    e_rec, _ = O.ewald_reciprocal(h['positions'], h['box'], q, alpha, 16)
    assert e_rec == pytest.approx(goldens['G14']['value'], rel=REL)


def test_reciprocal_of_a_charged_box_G18_and_torsions_G19(phenol, goldens):
    """Phenol in water (the water model of this force field file carries -0.02 e per molecule: Q = -9.98 e once the
    solute charges are zeroed): the reference literal is the plain Ewald sum, WITHOUT the neutralising-background term
    -pi Kc Q^2/(2 V alpha^2) = -213.66 kJ/mol."""
    c = phenol
    q = c['charge'].copy()
    q[c['resname'] == 'aaa'] = 0.0
    assert abs(q.sum() + 9.98) < 1e-9
    alpha = np.sqrt(-np.log(2 * 5e-4)) / 1.0
    e_rec, _ = O.ewald_reciprocal(c['positions'], c['box'], q, alpha, 16)
    assert e_rec == pytest.approx(goldens['G18']['value'], rel=REL)
    plasma = -np.pi * O.KC * q.sum() ** 2 / (2 * np.prod(c['box']) * alpha ** 2)
    assert abs(e_rec + plasma - goldens['G18']['value']) > 200
    e_tor = O.periodic_torsions(c['torsions'], c['torsion_n'], c['torsion_phase'], c['torsion_k'], c['positions'], c['box'],
                                want_forces=False)[0]
    assert e_tor == pytest.approx(goldens['G19']['value'], rel=1e-12)


def test_softcore_golden_G15(heaq, goldens):
    """SolvationSystem's softcore force (systems.py:266-272) on HEAQ at lambda_vdw = 0.5: pair sum over the
    (solute, solvent) interaction group with OpenMM's built-in switch + the CustomNonbondedForce long-range
    correction = tests/test_systems.py:39."""
    h = heaq
    codes = np.where(h['resname'] == 'aaa', 1.0, 2.0)
    d = O.desc(O.SOFTCORE, rc=1.0, rswitch=0.9, alpha=0.5, flags=O.SWITCH, Kc=1.0)
    e_pair, f, _ = O.pair_eval(d, h['positions'], h['box'], codes, h['sigma'], h['epsilon'], h['exc_pairs'])
    e_lrc = O.softcore_lrc(h['sigma'], h['epsilon'], codes, h['box'], 1.0, 0.9, 0.5)
    # 2.3e-8 observed: OpenMM integrates the correction with a 1e-5 relative stopping criterion
    assert e_pair + e_lrc == pytest.approx(goldens['G15']['value'], rel=2e-7)
    # forces of the pinned energy: central differences on a solute and a solvent atom
    x0, eps_ = h['positions'], 1e-5
    for atom in (int(np.where(codes == 1.0)[0][3]), int(np.where(codes == 2.0)[0][0])):
        for k in range(3):
            ep = []
            for sgn in (1, -1):
                x = x0.copy()
                x[atom, k] += sgn * eps_
                ep.append(O.pair_eval(d, x, h['box'], codes, h['sigma'], h['epsilon'], h['exc_pairs'], want_forces=False)[0])
            assert -(ep[0] - ep[1]) / (2 * eps_) == pytest.approx(f[atom, k], abs=2e-5)


def test_bonded_goldens(spcfw, heaq, goldens):
    c = spcfw
    eb, _ = O.harmonic_bonds(c['bonds'], c['bond_r0'], c['bond_k'], c['positions'], c['box'], want_forces=False)
    ea, _ = O.harmonic_angles(c['angles'], c['angle_theta0'], c['angle_k'], c['positions'], c['box'], want_forces=False)
    assert eb == pytest.approx(goldens['G_bonds']['value'], rel=REL)
    assert ea == pytest.approx(goldens['G_angles']['value'], rel=REL)
    ex, _ = O.ljc_bonds(c['exc_pairs'], c['exc_chargeprod'], c['exc_sigma'], c['exc_epsilon'], c['positions'],
                        c['box'], want_forces=False)
    assert ex == goldens['G_exc0']['value']
    h = heaq
    eb, _ = O.harmonic_bonds(h['bonds'], h['bond_r0'], h['bond_k'], h['positions'], h['box'], want_forces=False)
    ea, _ = O.harmonic_angles(h['angles'], h['angle_theta0'], h['angle_k'], h['positions'], h['box'], want_forces=False)
    et, _ = O.periodic_torsions(h['torsions'], h['torsion_n'], h['torsion_phase'], h['torsion_k'], h['positions'],
                                h['box'], want_forces=False)
    assert eb == pytest.approx(goldens['G_heaq_bonds']['value'], rel=REL)
    assert ea == pytest.approx(goldens['G_heaq_angles']['value'], rel=REL)
    assert et == pytest.approx(goldens['G_heaq_torsions']['value'], rel=REL)


def test_solvation_respa_goldens_G9_G10(heaq, goldens):
    h = heaq
    q, s, e, pairs, qq, sg, ep = solvation_respa_inputs(h, 0.5)
    en, _, _ = O.pair_eval(near('force-switch', 0.7, 0.5), h['positions'], h['box'], q, s, e, pairs, want_forces=False)
    assert en == pytest.approx(goldens['G9']['value'], rel=1e-10)
    ex, _ = O.ljc_bonds(pairs, qq, sg, ep, h['positions'], h['box'], periodic=True, want_forces=False)
    assert ex == pytest.approx(goldens['G10']['value'], rel=REL)


def test_exceptions_plus_bonded_golden_G11(emim, goldens):
    c = emim
    e = O.ljc_bonds(c['exc_pairs'], c['exc_chargeprod'], c['exc_sigma'], c['exc_epsilon'], c['positions'], c['box'],
                    periodic=True, want_forces=False)[0]
    e += O.harmonic_bonds(c['bonds'], c['bond_r0'], c['bond_k'], c['positions'], c['box'], want_forces=False)[0]
    e += O.harmonic_angles(c['angles'], c['angle_theta0'], c['angle_k'], c['positions'], c['box'], want_forces=False)[0]
    e += O.periodic_torsions(c['torsions'], c['torsion_n'], c['torsion_phase'], c['torsion_k'], c['positions'],
                             c['box'], want_forces=False)[0]
    assert e == pytest.approx(goldens['G11']['value'], rel=REL)


DESCS = {
    'near-none': near(None, 0.7, 0.5), 'near-shift': near('shift', 0.7, 0.5),
    'near-fswitch': near('force-switch', 0.7, 0.5),
    'near-fswitch-guarded-neg': near('force-switch', 0.7, 0.5, sign=-1.0, flags=O.GUARD_RC0, actual=1.0),
    'damped-1': O.desc(O.DAMPED, rc=1.0, rswitch=0.9, alpha=2.9, degree=1),
    'damped-2': O.desc(O.DAMPED, rc=1.0, rswitch=0.9, alpha=2.9, degree=2),
    'ewald': O.desc(O.NONBONDED, rc=1.0, rswitch=0.9, alpha=2.6, flags=O.COULOMB_EWALD | O.SWITCH),
    'rf': O.desc(O.NONBONDED, rc=1.0, rswitch=0.9, flags=O.COULOMB_RF | O.SWITCH,
                 krf=(78.3 - 1) / (2 * 78.3 + 1), crf=3 * 78.3 / (2 * 78.3 + 1)),
    'plain': O.desc(O.NONBONDED, rc=1.0),
}


@pytest.mark.parametrize('name', sorted(DESCS))
def test_pair_kernel_force_is_minus_gradient(name):
    """-dE/dr by central differences, incl. the force-switch identity V' = S V'_LJC (forces.py:628)."""
    d = DESCS[name]
    qq, sig, eps = -0.35, 0.3166, 0.65
    rmax = min(d.rc, d.rc0 if (d.flags & O.GUARD_RC0) else d.rc)
    for r in np.linspace(0.28, rmax - 1e-3, 57):
        h = 1e-6
        ep, _ = O.pair_kernel(d, (r + h) ** 2, qq, sig, eps)
        em, _ = O.pair_kernel(d, (r - h) ** 2, qq, sig, eps)
        _, fr = O.pair_kernel(d, r * r, qq, sig, eps)
        fd = -(ep - em) / (2 * h)
        assert fr * r == pytest.approx(fd, rel=1e-5, abs=2e-5), (name, r)


def test_near_continuity_at_cutoff():
    for adj in (None, 'shift', 'force-switch'):
        d = near(adj, 0.7, 0.5)
        e, fr = O.pair_kernel(d, (0.7 - 1e-9) ** 2, -0.35, 0.3166, 0.65)
        assert abs(e) < 1e-6 and abs(fr) < 1e-5


def _fd_forces(fn, pos, idxs, h=1e-6):
    out = []
    for (i, k) in idxs:
        p = pos.copy(); p[i, k] += h; ep = fn(p)
        p[i, k] -= 2 * h; em = fn(p)
        out.append(-(ep - em) / (2 * h))
    return np.array(out)


def test_bonded_forces_fd(heaq):
    h = heaq
    pos = h['positions']
    idxs = [(0, 0), (6, 1), (17, 2), (19, 0), (32, 1), (35, 2)]
    for fn in (
        lambda p, wf=False: O.harmonic_bonds(h['bonds'], h['bond_r0'], h['bond_k'], p, h['box'], want_forces=wf),
        lambda p, wf=False: O.harmonic_angles(h['angles'], h['angle_theta0'], h['angle_k'], p, h['box'], want_forces=wf),
        lambda p, wf=False: O.periodic_torsions(h['torsions'], h['torsion_n'], h['torsion_phase'], h['torsion_k'], p,
                                                h['box'], want_forces=wf),
        lambda p, wf=False: O.ljc_bonds(h['exc_pairs'], h['exc_chargeprod'], h['exc_sigma'], h['exc_epsilon'], p,
                                        h['box'], want_forces=wf),
    ):
        f = fn(pos, True)[1]
        fd = _fd_forces(lambda p: fn(p)[0], pos, idxs)
        an = np.array([f[i, k] for i, k in idxs])
        assert an == pytest.approx(fd, rel=1e-5, abs=1e-4)
        assert abs(f.sum(0)).max() < 1e-7


def test_pair_forces_fd_and_cells_match_n2(spcfw):
    c = spcfw
    args = (c['positions'], c['box'], c['charge'], c['sigma'], c['epsilon'], c['exc_pairs'])
    d = near('force-switch', 0.7, 0.5)
    e1, f1, n1 = O.pair_eval(d, *args)
    e2, f2, n2 = O.pair_eval(d, *args, use_cells=True)
    assert n1 == n2
    assert e2 == pytest.approx(e1, rel=1e-12)
    assert np.abs(f2 - f1).max() < 1e-9 * np.abs(f1).max()
    idxs = [(0, 0), (1, 1), (700, 2)]
    fd = _fd_forces(lambda p: O.pair_eval(d, p, *args[1:], want_forces=False)[0], c['positions'], idxs)
    assert np.array([f1[i, k] for i, k in idxs]) == pytest.approx(fd, rel=1e-5, abs=1e-3)


def test_ewald_exclusion_and_reciprocal_forces_fd(spcfw):
    c = spcfw
    n = 96
    pos, q, box = c['positions'][:n].copy(), c['charge'][:n], c['box']
    pairs = c['exc_pairs'][:n]
    idxs = [(0, 0), (4, 1), (50, 2)]
    f = O.ewald_exclusion(pairs, pos, box, q, 2.6)[1]
    fd = _fd_forces(lambda p: O.ewald_exclusion(pairs, p, box, q, 2.6, want_forces=False)[0], pos, idxs)
    assert np.array([f[i, k] for i, k in idxs]) == pytest.approx(fd, rel=1e-5, abs=1e-4)
    f = O.ewald_reciprocal(pos, box, q, 2.6, 6, want_forces=True)[1]
    fd = _fd_forces(lambda p: O.ewald_reciprocal(p, box, q, 2.6, 6)[0], pos, idxs)
    assert np.array([f[i, k] for i, k in idxs]) == pytest.approx(fd, rel=1e-5, abs=1e-4)


def test_step_primitives():
    rng = np.random.default_rng(1)
    n = 10
    x = rng.normal(size=(n, 3)); v = rng.normal(size=(n, 3)); f = rng.normal(size=(n, 3)); g = rng.normal(size=(n, 3))
    m = rng.uniform(1, 16, size=n)
    v0 = v.copy(); x0 = x.copy()
    O.kick(v, f, m, 0.25)
    assert np.array_equal(v, v0 + 0.25 * f / m[:, None])
    O.kick(v, f, m, 0.5, fsub=g)
    assert np.array_equal(v, (v0 + 0.25 * f / m[:, None]) + 0.5 * (f - g) / m[:, None])
    O.move(x, v, 0.125)
    assert np.array_equal(x, x0 + 0.125 * v)
    assert O.mvv(v, m) == pytest.approx((m[:, None] * v * v).sum(), rel=1e-14)


def test_verlet_list_port_equals_the_cell_walk():
    """The CPU-baseline mode of the oracle (Verlet lists, bench.py's cpu_baseline leg) gives the forces and the RESPA
    trajectory of the checker's 27-cell walk: same ammo_pair_kernel, same pairs."""
    from atomsmm_amd.testing import tip3p_box
    from oracle.respa_cpu import RespaCPU
    case = tip3p_box(12)
    walk = RespaCPU(case, dt=0.002)
    lists = RespaCPU(case, dt=0.002, verlet_skin=0.1)
    for g in (1, 2):
        assert np.abs(walk.f(g) - lists.f(g)).max() <= 1e-12 * np.abs(walk.f(g)).max()
    walk.step(2)
    lists.step(2)
    assert np.abs(walk.x - lists.x).max() < 1e-13
    assert lists.lists[2].builds >= 1


def test_constraint_oracle_routes_agree():
    """oracle/constraints_oracle.py: the closed form of SETTLE, Newton's method on the multipliers and the dense velocity solve are
    three derivations of the constraint equations' definition; they must agree with each other to round-off, satisfy the
    constraints, conserve the centre of mass and (velocities) leave nothing along the bonds."""
    from oracle import constraints_oracle as CO
    rng = np.random.default_rng(5)
    r_oh, r_hh = 0.09572, 0.15139
    m = np.array([15.9994, 1.008, 1.008])
    loc = [(0, 1), (0, 2), (1, 2)]
    for _ in range(100):
        o = rng.uniform(0, 5, 3)
        a = rng.normal(size=3); a /= np.linalg.norm(a)
        b = np.cross(a, rng.normal(size=3)); b /= np.linalg.norm(b)
        half = np.arcsin(0.5 * r_hh / r_oh)
        x0 = np.stack([o, o + r_oh * (np.cos(half) * a + np.sin(half) * b), o + r_oh * (np.cos(half) * a - np.sin(half) * b)])
        x1 = x0 + rng.normal(0, 0.005, (3, 3))
        ys = CO.shake_exact(x1, x0, m, loc, [r_oh, r_oh, r_hh])
        yt = CO.settle(x0, x1, m[0], m[1], r_oh, r_hh)
        assert np.abs(ys - yt).max() < 1e-13
        for (i, j), d in zip(loc, [r_oh, r_oh, r_hh]):
            assert abs(np.linalg.norm(yt[i] - yt[j]) - d) < 1e-14
        assert np.abs((m[:, None] * (yt - x1)).sum(0)).max() < 1e-12
        # the displacement is a mass-weighted combination of the REFERENCE bond vectors: no torque about them
        torque = sum(np.cross(x0[k], m[k] * (yt[k] - x1[k])) for k in range(3))
        assert np.abs(torque).max() < 1e-11
        v = rng.normal(0, 0.5, (3, 3))
        w = CO.rattle_exact(yt, v, m, loc)
        for i, j in loc:
            assert abs((w[i] - w[j]) @ (yt[i] - yt[j])) < 1e-14
        assert np.abs((m[:, None] * (w - v)).sum(0)).max() < 1e-12


def test_cpu_port_equals_the_oracle_driven_program():
    """oracle/cpu_port.c (bench.py's cpu_baseline, kind "port") runs the RESPA step program entirely in C; it must give what the
    oracle-driven program gives (oracle/respa_cpu.py over oracle/amm_oracle.c): same evaluation counts per group, positions and
    velocities to round-off after two steps -- on the small-box (all-pairs list) and on the cell-grid list build."""
    from atomsmm_amd.testing import tip3p_box
    from oracle import cpu_port, respa_cpu
    for nside, steps in ((8, 2), (20, 1)):
        c = tip3p_box(nside)
        a = respa_cpu.RespaCPU(c, dt=0.002, verlet_skin=0.1 if nside == 20 else None)
        a.step(steps)
        b = cpu_port.RespaPort(c, dt=0.002)
        b.step(steps)
        st = b.state()
        assert st['evals'] == (a.evals[0], a.evals[1], a.evals[2])
        assert np.abs(st['x'] - a.x).max() < 1e-13
        assert np.abs(st['v'] - a.v).max() < 1e-10
        b.close()
