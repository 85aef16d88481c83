"""GPU parity tests (run with -m gpu on an MI355X): HIP kernels called through the C-ABI
(include/atomsmm_hip.h via atomsmm_amd.backend) against the CPU oracle on the same inputs, and
against the reference's golden literals.

Stated fp64 tolerances: energies rel 1e-10 vs the oracle (1e-8 for the
ill-conditioned force-switch energy at b = 19), forces 1e-9*max|F| vs the oracle, energies rel 1e-6
vs the reference literals (the reference's own tolerance); kick/move: bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')

from oracle import oracle as O  # noqa: E402  (checker only)


def _backend():
    from atomsmm_amd import backend as B
    return B


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device='cuda')


def hip_pair(B, ctx, d, c, skin=-1.0, q=None, s=None, e=None, excl=None):
    q = c['charge'] if q is None else q
    s = c['sigma'] if s is None else s
    e = c['epsilon'] if e is None else e
    excl = c['exc_pairs'] if excl is None else excl
    desc = B.pair_desc(d.family, d.rc, rc0=d.rc0, rs0=d.rs0, rswitch=d.rswitch, alpha=d.alpha, degree=d.degree,
                       flags=d.flags, sign=d.sign, Kc=d.Kc, krf=d.krf, crf=d.crf)
    return ctx.pair_create(desc, q, s, e, excl, skin=skin)


def eval_force(ctx, fid, pos_t, n):
    f = torch.full((n, 3), float('nan'), dtype=torch.float64, device='cuda')
    en = torch.zeros(1, dtype=torch.float64, device='cuda')
    ctx.force_eval(fid, pos_t, f, accumulate=False, energy=en)
    ctx.check()
    return en.item(), f.cpu().numpy()


def near(adj, rc, rs, **kw):
    return O.desc(O.ADJ[adj], rc=kw.pop('actual', rc), rc0=rc, rs0=rs, **kw)


CASES = {
    'G1-near-none': (near(None, 1.0, 0.95), 'G1', 1e-10),
    'G2-near-shift': (near('shift', 1.0, 0.95), 'G2', 1e-10),
    'G3-near-fswitch-b19': (near('force-switch', 1.0, 0.95), 'G3', 1e-7),
    'G4-damped-1': (O.desc(O.DAMPED, rc=1.0, rswitch=0.95, alpha=2.9, degree=1), 'G4', 1e-10),
    'G5-damped-2': (O.desc(O.DAMPED, rc=1.0, rswitch=0.95, alpha=2.9, degree=2), 'G5', 1e-10),
    'G6-respa-near': (near('force-switch', 0.7, 0.5), 'G6', 1e-10),
    'G6-respa-minus-near': (near('force-switch', 0.7, 0.5, sign=-1.0, flags=O.GUARD_RC0), None, 1e-10),
    'discount-at-outer-cutoff': (near('force-switch', 0.7, 0.5, sign=-1.0, flags=O.GUARD_RC0, actual=1.0), None, 1e-10),
    'near-none-respa': (near(None, 0.7, 0.5), None, 1e-10),
    'near-shift-respa': (near('shift', 0.7, 0.5), None, 1e-10),
    'ewald-direct': (O.desc(O.NONBONDED, rc=1.0, rswitch=0.9, alpha=2.628260884878466,
                            flags=O.COULOMB_EWALD | O.SWITCH), None, 1e-10),
    'reaction-field': (O.desc(O.NONBONDED, rc=1.0, rswitch=0.9, flags=O.COULOMB_RF | O.SWITCH,
                              krf=(78.3 - 1) / (2 * 78.3 + 1), crf=3 * 78.3 / (2 * 78.3 + 1)), None, 1e-10),
    'plain-cutoff': (O.desc(O.NONBONDED, rc=1.0), None, 1e-10),
}


NARROW = {
    # switching windows far narrower than the tested 0.2 / 0.05 nm, and a DAMPED degree above 1: the radial table's refinement may
    # stop short of the arithmetic's accuracy: such a force must keep the analytic kernels, never a coarse table
    'damped-2-narrow': O.desc(O.DAMPED, rc=1.0, rswitch=0.98, alpha=2.9, degree=2),
    'damped-3': O.desc(O.DAMPED, rc=1.0, rswitch=0.9, alpha=2.9, degree=3),
    'near-fswitch-narrow': near('force-switch', 0.7, 0.68),
    'near-none-narrow': near(None, 0.7, 0.69),
}


@pytest.mark.parametrize('name', sorted(NARROW))
def test_narrow_switch_tables_meet_their_bound_or_are_not_used(spcfw, name):
    B = _backend()
    d = NARROW[name]
    c = spcfw
    n = len(c['positions'])
    e_ref, f_ref, _ = O.pair_eval(d, c['positions'], c['box'], c['charge'], c['sigma'], c['epsilon'], c['exc_pairs'])
    ctx = B.HipContext(n, c['box'])
    fid = hip_pair(B, ctx, d, c)
    f2 = torch.empty((n, 3), dtype=torch.float64, device='cuda')
    ctx.force_eval(fid, dev(c['positions']), f2)           # force-only: the tabulated kernel IF the table is good enough
    ctx.check()
    st = ctx.pair_stats(fid)
    assert (st['has_table'] == 1 and st['tab_error'] <= 1e-13) or st['has_table'] == 0
    assert np.abs(f2.cpu().numpy() - f_ref).max() <= 1e-9 * np.abs(f_ref).max()
    e, f = eval_force(ctx, fid, dev(c['positions']), n)
    # (the force-switch ENERGY is ill-conditioned in b = rs/(rc - rs) -- b = 34 here, cf. G3 with b = 19 --, its force is not)
    assert e == pytest.approx(e_ref, rel=1e-6 if name == 'near-fswitch-narrow' else 1e-9)
    assert np.abs(f2.cpu().numpy() - f).max() <= 1e-11 * np.abs(f).max()
    ctx.close()


def test_molecule_rows_edge_cases(spcfw):
    """Molecule rows (csrc/cluster.hip) on the reference's small water box: the box is too small for one periodic image per
    molecule pair (the per-atom-pair image path), rows are rebuilt when whole molecules move, a molecule whose atoms drift
    apart WITHIN the extent bound stays exact, and one stretched BEYOND the bound the cells were sized for is reported by
    amm_check instead of giving silently incomplete forces."""
    B = _backend()
    c = spcfw
    n = len(c['positions'])
    d = near('force-switch', 0.7, 0.5)
    ctx = B.HipContext(n, c['box'])
    fid = hip_pair(B, ctx, d, c)
    f = torch.empty((n, 3), dtype=torch.float64, device='cuda')

    def check_against_oracle(pos):
        ctx.force_eval(fid, dev(pos), f)
        ctx.check()
        ref = O.pair_eval(d, pos, c['box'], c['charge'], c['sigma'], c['epsilon'], c['exc_pairs'])[1]
        assert np.abs(f.cpu().numpy() - ref).max() <= 1e-9 * np.abs(ref).max()

    check_against_oracle(c['positions'])
    st = ctx.pair_stats(fid)
    assert st['list_kind'] == 1 and st['n_builds'] == 1
    slots, entries = ctx.pair_row_padding(fid)            # lane-trips executed vs entries held (measurement helper)
    assert 9 * entries == st['n_list_pairs'] and entries <= slots <= 4 * entries
    rng = np.random.default_rng(7)
    # whole molecules move by up to 0.12 nm (beyond skin / 2: rebuild), every atom a little on top, across the box faces
    pos = c['positions'].reshape(-1, 3, 3) + rng.uniform(-0.12, 0.12, (n // 3, 1, 3)) + rng.normal(0.0, 0.004, (n // 3, 3, 3))
    pos = pos.reshape(n, 3) + np.array([0.4, -0.7, 1.3]) * c['box']
    check_against_oracle(pos)
    assert ctx.pair_stats(fid)['n_builds'] == 2
    # a hydrogen 0.03 nm farther from its oxygen: inside the bound (1.5 x the first extent)
    pos2 = pos.copy()
    pos2[1] += 0.03 * (pos2[1] - pos2[0]) / np.linalg.norm(pos2[1] - pos2[0])
    check_against_oracle(pos2)
    # ... and 0.6 nm away: beyond it -- amm_check must say so
    pos3 = pos.copy()
    pos3[4] += np.array([0.6, 0.0, 0.0])
    ctx.force_eval(fid, dev(pos3), f)
    with pytest.raises(RuntimeError, match='stretched beyond'):
        ctx.check()
    ctx.close()


@pytest.mark.parametrize('family', ['near-fswitch', 'damped', 'ewald-direct', 'near-shift'])
def test_site_site_tables(spcfw, family):
    """Molecule rows: when every Lennard-Jones site of a force has the same sigma, eps and charge (water: the oxygens) a pair of
    two sites reads ONE radial table that holds Coulomb + Lennard-Jones (csrc/pair_tab.h: SiteTable) and the kernels carry no
    Lennard-Jones arithmetic.  Checked: the table exists and meets its bound; forces equal the oracle's and the analytic-LJ
    kernels' (option site_tab = 0); two oxygens pushed closer than the table reaches (0.7 sigma) take the analytic path; sites
    with unequal charges get no table and the same kernels' analytic branch."""
    B = _backend()
    c = spcfw
    n = len(c['positions'])
    d = {'near-fswitch': near('force-switch', 0.7, 0.5), 'near-shift': near('shift', 0.7, 0.5),
         'damped': O.desc(O.DAMPED, rc=1.0, rswitch=0.9, alpha=2.9, degree=1),
         'ewald-direct': O.desc(O.NONBONDED, rc=1.0, rswitch=0.9, alpha=2.628260884878466, flags=O.COULOMB_EWALD | O.SWITCH)}[family]

    def forces(pos, q, site_tab):
        ctx = B.HipContext(n, c['box'])
        ctx.set_option('site_tab', site_tab)
        fid = hip_pair(B, ctx, d, c, q=q)
        f = torch.empty((n, 3), dtype=torch.float64, device='cuda')
        ctx.force_eval(fid, dev(pos), f)
        ctx.check()
        st = ctx.pair_stats(fid)
        ctx.close()
        return f.cpu().numpy(), st

    pos = c['positions']
    ref = O.pair_eval(d, pos, c['box'], c['charge'], c['sigma'], c['epsilon'], c['exc_pairs'])[1]
    f1, st1 = forces(pos, c['charge'], 1)
    f0, st0 = forces(pos, c['charge'], 0)
    assert st1['list_kind'] == 1 and st1['has_site_table'] == 1 and 0.0 < st1['site_tab_error'] <= 3e-13
    assert st0['has_site_table'] == 0
    scale = np.abs(ref).max()
    assert np.abs(f1 - ref).max() <= 1e-9 * scale and np.abs(f0 - ref).max() <= 1e-9 * scale
    assert np.abs(f1 - f0).max() <= 1e-11 * scale
    # two water molecules with their oxygens 0.2 nm apart (0.63 sigma: below the site-site table, above the Coulomb table's end)
    close = pos.copy()
    shift = close[0] + np.array([0.2, 0.0, 0.0]) - close[3]
    close[3:6] += shift
    ref_c = O.pair_eval(d, close, c['box'], c['charge'], c['sigma'], c['epsilon'], c['exc_pairs'])[1]
    fc, _ = forces(close, c['charge'], 1)
    assert np.abs(fc - ref_c).max() <= 1e-9 * np.abs(ref_c).max()
    # parameter updates through amm_pair_set_params (what parameter offsets do): charges scaled (another QQ), then other sites
    # (another sigma / eps), then back to a state without a table and with one again -- the tables follow every time
    ctx = B.HipContext(n, c['box'])
    fid = hip_pair(B, ctx, d, c)
    f = torch.empty((n, 3), dtype=torch.float64, device='cuda')
    ctx.force_eval(fid, dev(pos), f)
    sites = c['epsilon'] != 0.0
    uneven = c['charge'].copy()
    uneven[np.where(sites)[0][0]] *= 1.1
    for q_, s_, e_, table in ((0.9 * c['charge'], c['sigma'], c['epsilon'], 1),
                              (0.9 * c['charge'], np.where(sites, 0.29, c['sigma']), 1.7 * c['epsilon'], 1),
                              (uneven, c['sigma'], c['epsilon'], 0),
                              (c['charge'], c['sigma'], c['epsilon'], 1)):
        ctx.pair_set_params(fid, q_, s_, e_)
        ctx.force_eval(fid, dev(pos), f)
        ctx.check()
        ref_p = O.pair_eval(d, pos, c['box'], q_, s_, e_, c['exc_pairs'])[1]
        assert ctx.pair_stats(fid)['has_site_table'] == table
        assert np.abs(f.cpu().numpy() - ref_p).max() <= 1e-9 * np.abs(ref_p).max()
    assert np.array_equal(f.cpu().numpy(), f1)          # the same tables as a fresh context builds
    ctx.close()
    # unequal site charges: no site-site table (the pair's qq is no longer one number), same forces as the oracle
    q2 = c['charge'].copy()
    q2[0] *= 1.25
    q2[1] -= 0.125 * c['charge'][0]
    q2[2] -= 0.125 * c['charge'][0]
    ref_q = O.pair_eval(d, pos, c['box'], q2, c['sigma'], c['epsilon'], c['exc_pairs'])[1]
    fq, stq = forces(pos, q2, 1)
    assert stq['has_site_table'] == 0
    assert np.abs(fq - ref_q).max() <= 1e-9 * np.abs(ref_q).max()


@pytest.mark.parametrize('name', sorted(CASES))
def test_pair_families_vs_oracle_and_goldens(spcfw, goldens, name):
    B = _backend()
    d, gid, etol = CASES[name]
    c = spcfw
    n = len(c['positions'])
    e_ref, f_ref, _ = O.pair_eval(d, c['positions'], c['box'], c['charge'], c['sigma'], c['epsilon'], c['exc_pairs'])
    ctx = B.HipContext(n, c['box'])
    fid = hip_pair(B, ctx, d, c)
    e, f = eval_force(ctx, fid, dev(c['positions']), n)
    assert e == pytest.approx(e_ref, rel=etol)
    assert np.abs(f - f_ref).max() <= 1e-9 * np.abs(f_ref).max()
    if gid:
        assert e == pytest.approx(goldens[gid]['value'], rel=1e-6)
    # the force-only launch (no energy) is a different kernel -- its Coulomb part comes from the radial table of
    # csrc/pair_tab.h (relative interpolation error < 1e-14) -- and must give the same forces, also against the oracle
    f2 = torch.empty((n, 3), dtype=torch.float64, device='cuda')
    ctx.force_eval(fid, dev(c['positions']), f2)
    assert np.abs(f2.cpu().numpy() - f).max() <= 1e-12 * np.abs(f).max()
    assert np.abs(f2.cpu().numpy() - f_ref).max() <= 1e-9 * np.abs(f_ref).max()
    ctx.close()


def test_ewald_direct_golden_G7(spcfw, goldens):
    """Group-2 NonbondedForce direct space = pair(erfc, switched LJ) + exclusion erf term + dispersion constant."""
    B = _backend()
    c = spcfw
    n = len(c['positions'])
    alpha = np.sqrt(-np.log(2 * 5e-4)) / 1.0
    ctx = B.HipContext(n, c['box'])
    d = O.desc(O.NONBONDED, rc=1.0, rswitch=0.9, alpha=alpha, flags=O.COULOMB_EWALD | O.SWITCH)
    fid = hip_pair(B, ctx, d, c)
    bid = ctx.bonded_create()
    qq = c['charge'][c['exc_pairs'][:, 0]] * c['charge'][c['exc_pairs'][:, 1]]
    ctx.bonded_add_terms(bid, B.BOND_EWALD_EXCL, c['exc_pairs'], qq, periodic=True,
                         desc=B.pair_desc(B.NONBONDED, 1.0, alpha=alpha))
    ctx.bonded_finalize(bid)
    pos = dev(c['positions'])
    e1, f1 = eval_force(ctx, fid, pos, n)
    e2, f2 = eval_force(ctx, bid, pos, n)
    e_exc, f_exc = O.ewald_exclusion(c['exc_pairs'], c['positions'], c['box'], c['charge'], alpha)
    assert e2 == pytest.approx(e_exc, rel=1e-11)
    assert np.abs(f2 - f_exc).max() <= 1e-9 * np.abs(f_exc).max()
    e_disp = O.dispersion_correction(c['sigma'], c['epsilon'], c['box'], 1.0, 0.9)   # constant, host side
    assert e1 + e2 + e_disp == pytest.approx(goldens['G7']['value'], rel=1e-6)
    ctx.close()


@pytest.mark.parametrize('case_name', ['spcfw', 'heaq'])
def test_pme_reciprocal_vs_exact_ewald_and_goldens(spcfw, heaq, goldens, case_name):
    """amm_pme_create / amm_force_eval: smooth PME (order 5) against the oracle's explicit Ewald k-sum and the
    reference's two reciprocal-space literals (G8, G14).  On OpenMM's default mesh (ceil(2 alpha L / (3 tol^0.2)))
    the energy reproduces the literals to rel 1e-10 (measured 1.3e-13 and 7.9e-13: same mesh, same splines, same
    moduli as the PME that wrote them; the explicit k-sum differs from them by 4.6e-8 / 1.4e-7); on a 3x finer mesh PME converges
    to the explicit sum with the expected order (measured on q-SPC-FW: energy rel 4.6e-8 -> 8.6e-11, max force error
    8.3e-4 -> 3.7e-6 of max|F|, a factor 226 ~ 3^5).  Fixed-point spread => bit-reproducible."""
    B = _backend()
    from helpers import solvation_respa_inputs
    if case_name == 'spcfw':
        c, q, gold, kmax = spcfw, spcfw['charge'], goldens['G8']['value'], 14
    else:
        c, q, gold, kmax = heaq, solvation_respa_inputs(heaq, 0.5)[0], goldens['G14']['value'], 16
    n = len(c['positions'])
    alpha = np.sqrt(-np.log(2 * 5e-4)) / 1.0
    e_ref, f_ref = O.ewald_reciprocal(c['positions'], c['box'], q, alpha, kmax, want_forces=True)
    ctx = B.HipContext(n, c['box'])
    pos = dev(c['positions'])
    grid = [int(np.ceil(2 * alpha * L / (3 * 5e-4 ** 0.2))) for L in c['box']]
    fid = ctx.pme_create(alpha, grid, q)
    e1, f1 = eval_force(ctx, fid, pos, n)
    assert e1 == pytest.approx(gold, rel=1e-10)
    assert e1 == pytest.approx(e_ref, rel=1e-6)
    fmax = np.abs(f_ref).max()
    assert np.abs(f1 - f_ref).max() < 3e-3 * fmax            # OpenMM's "tolerance 5e-4" mesh: interpolation error
    e1b, f1b = eval_force(ctx, fid, pos, n)
    assert e1b == e1 and np.array_equal(f1, f1b)
    fine = ctx.pme_create(alpha, [3 * k for k in grid], q)
    e2, f2 = eval_force(ctx, fine, pos, n)
    print(case_name, grid, 'vs literal', abs(e1 - gold) / abs(gold), 'dE/E default', abs(e1 - e_ref) / abs(e_ref), 'fine', abs(e2 - e_ref) / abs(e_ref),
          'dF default', np.abs(f1 - f_ref).max() / fmax, 'fine', np.abs(f2 - f_ref).max() / fmax)
    assert e2 == pytest.approx(e_ref, rel=2e-9)
    assert np.abs(f2 - f_ref).max() < 1e-5 * fmax
    # accumulate adds on top of an existing buffer
    acc = torch.ones((n, 3), dtype=torch.float64, device='cuda')
    ctx.force_eval(fine, pos, acc, accumulate=True, energy=None)
    assert np.allclose(acc.cpu().numpy() - 1.0, f2, rtol=0, atol=1e-9 * fmax)
    ctx.close()


def test_pme_small_mesh_atomic_path_is_consistent(spcfw):
    """Meshes narrower than one spread tile + halo (K < 12) take the plain fixed-point-atomics spread: forces are the
    gradient of the mesh energy (central differences) and repeated evaluations are bit-identical."""
    B = _backend()
    c = spcfw
    n = len(c['positions'])
    ctx = B.HipContext(n, c['box'])
    fid = ctx.pme_create(1.2, [10, 11, 9], c['charge'])
    x0 = c['positions'].copy()
    e0, f0 = eval_force(ctx, fid, dev(x0), n)
    e0b, f0b = eval_force(ctx, fid, dev(x0), n)
    assert e0 == e0b and np.array_equal(f0, f0b)
    h = 1e-5
    for atom, k in ((0, 0), (5, 1), (1000, 2)):
        e = []
        for sgn in (+1, -1):
            x = x0.copy()
            x[atom, k] += sgn * h
            e.append(eval_force(ctx, fid, dev(x), n)[0])
        assert -(e[0] - e[1]) / (2 * h) == pytest.approx(f0[atom, k], abs=1e-6 * np.abs(f0).max())
    ctx.close()


@pytest.mark.parametrize('small_group', [1, 0])
def test_softcore_interaction_group_vs_oracle_and_G15(heaq, goldens, small_group):
    """AMM_SOFTCORE (SolvationSystem's solute-solvent softcore LJ, systems.py:266-272) through the C-ABI: energy
    and forces vs the oracle at three lambdas (amm_pair_set_lambda), and with the long-range correction vs the
    reference literal tests/test_systems.py:39.  Both evaluation paths: without a neighbour list (the solute is a small set:
    csrc/group.hip, list_kind 3) and through the filtered list (option small_group = 0)."""
    B = _backend()
    h = heaq
    n = len(h['positions'])
    codes = np.where(h['resname'] == 'aaa', 1.0, 2.0)
    ctx = B.HipContext(n, h['box'])
    ctx.set_option('small_group', small_group)
    desc = B.pair_desc(B.SOFTCORE, 1.0, rswitch=0.9, alpha=0.5, flags=B.SWITCH, Kc=1.0)
    fid = ctx.pair_create(desc, codes, h['sigma'], h['epsilon'], h['exc_pairs'])
    pos = dev(h['positions'])
    for lam in (0.5, 1.0, 0.1):
        ctx.pair_set_lambda(fid, lam)
        e, f = eval_force(ctx, fid, pos, n)
        assert ctx.pair_stats(fid)['list_kind'] == (3 if small_group else 0)
        # force only, added to a buffer: the rows of both sets and of neither
        g = torch.full((n, 3), 0.25, dtype=torch.float64, device='cuda')
        ctx.force_eval(fid, pos, g, accumulate=True)
        assert np.abs(g.cpu().numpy() - 0.25 - f).max() <= 1e-12 * np.abs(f).max()
        d = O.desc(O.SOFTCORE, rc=1.0, rswitch=0.9, alpha=lam, flags=O.SWITCH, Kc=1.0)
        e_ref, f_ref, _ = O.pair_eval(d, h['positions'], h['box'], codes, h['sigma'], h['epsilon'], h['exc_pairs'])
        assert e == pytest.approx(e_ref, rel=1e-10)
        assert np.abs(f - f_ref).max() <= 1e-9 * np.abs(f_ref).max()
        if lam == 0.5:
            e_lrc = O.softcore_lrc(h['sigma'], h['epsilon'], codes, h['box'], 1.0, 0.9, 0.5)
            assert e + e_lrc == pytest.approx(goldens['G15']['value'], rel=2e-7)
    ctx.close()


@pytest.mark.parametrize('small_group', [1, 0])
def test_group_flags_vs_oracle(heaq, small_group):
    """AMM_GROUP_LJ (Lennard-Jones over a (set 1, set 2) interaction group: the charge slot carries the set codes) and
    AMM_GROUP_Q (Coulomb only: the sigma slot carries twice the set code; the force-switched electrostatics of Coulomb
    scaling, systems.py:848-856) through the C-ABI vs the oracle, and the group energy as a difference of plain sums:
    E(group) = E(all) - E(set 1 alone) - E(set 2 alone) for the Coulomb-only force."""
    B = _backend()
    h = heaq
    n = len(h['positions'])
    codes = np.where(h['resname'] == 'aaa', 1.0, 2.0)
    pos = dev(h['positions'])
    zero = np.zeros(n)
    ctx = B.HipContext(n, h['box'])
    ctx.set_option('small_group', small_group)
    cases = [(B.GROUP_LJ | B.NO_SHIFT, O.GROUP_LJ | O.NO_SHIFT, codes, h['sigma'], h['epsilon'], 1.0),
             (B.GROUP_Q | B.NO_SHIFT, O.GROUP_Q | O.NO_SHIFT, h['charge'], 2.0 * codes, zero, 138.935456637)]
    for bflags, oflags, q, sigma, eps, Kc in cases:
        fid = ctx.pair_create(B.pair_desc(B.NEAR_FSWITCH, 0.7, rc0=0.7, rs0=0.5, flags=bflags, Kc=Kc), q, sigma, eps, h['exc_pairs'])
        e, f = eval_force(ctx, fid, pos, n)
        d = O.desc(O.NEAR_FSWITCH, rc=0.7, rc0=0.7, rs0=0.5, flags=oflags, Kc=Kc)
        e_ref, f_ref, _ = O.pair_eval(d, h['positions'], h['box'], q, sigma, eps, h['exc_pairs'])
        assert abs(e_ref) > 1.0
        assert e == pytest.approx(e_ref, rel=1e-10)
        assert np.abs(f - f_ref).max() <= 1e-9 * np.abs(f_ref).max()
    # Coulomb-only group energy from three plain (ungrouped) evaluations with masked charges
    d = O.desc(O.NEAR_FSWITCH, rc=0.7, rc0=0.7, rs0=0.5, flags=O.NO_SHIFT, Kc=138.935456637)
    one = np.full(n, 1.0)
    parts = [O.pair_eval(d, h['positions'], h['box'], qq, one, zero, h['exc_pairs'])[0]
             for qq in (h['charge'], np.where(codes == 1.0, h['charge'], 0.0), np.where(codes == 2.0, h['charge'], 0.0))]
    assert e == pytest.approx(parts[0] - parts[1] - parts[2], rel=1e-9)
    ctx.close()


def test_filtered_list_follows_a_moving_solute_and_reports_overflow():
    """Interaction-group forces walk only the rows that hold entries (collected by the list build, a whole wavefront per row):
    (i) after the solute has moved through the solvent -- lists rebuilt, another set of rows -- energy and forces still equal
    the oracle's; (ii) the grid of the pair kernel covers twice the rows of the first build (+ 64): a solute that starts
    beyond the list radius of every solvent atom (no row at all) and then lands among them exceeds that, and amm_check says
    so instead of dropping rows."""
    B = _backend()
    from atomsmm_amd.testing import tip3p_box
    c = tip3p_box(8)                       # 1536 atoms, L = 2.48 nm
    n = len(c['positions'])
    rng = np.random.default_rng(5)
    codes = np.full(n, 2.0)
    codes[:3] = 1.0                        # one water is the solute
    sigma = np.where(c['sigma'] > 0, c['sigma'], 0.1)
    eps = np.full(n, 0.5)
    desc = B.pair_desc(B.SOFTCORE, 0.9, rswitch=0.8, alpha=0.7, flags=B.SWITCH, Kc=1.0)
    d = O.desc(O.SOFTCORE, rc=0.9, rswitch=0.8, alpha=0.7, flags=O.SWITCH, Kc=1.0)
    for small_group in (0, 1):         # the filtered list; and no list at all (csrc/group.hip): the same hops
        ctx = B.HipContext(n, c['box'])
        ctx.set_option('small_group', small_group)
        fid = ctx.pair_create(desc, codes, sigma, eps, c['exc_pairs'])
        x = c['positions'].copy()
        for hop in range(4):
            pos = dev(x)
            e, f = eval_force(ctx, fid, pos, n)
            e_ref, f_ref, _ = O.pair_eval(d, x, c['box'], codes, sigma, eps, c['exc_pairs'])
            assert e == pytest.approx(e_ref, rel=1e-10), hop
            assert np.abs(f - f_ref).max() <= 1e-9 * np.abs(f_ref).max(), hop
            assert np.count_nonzero(np.abs(f).sum(axis=1)) < n // 2          # most rows are empty and were never visited
            x[:3] += rng.uniform(0.3, 0.9, 3)                                    # the solute hops (beyond the Verlet buffer)
        ctx.close()
    # overflow of the active-row grid: solvent = the atoms of one corner only, the solute starts far from it
    corner = np.all(c['positions'] < 0.9, axis=1)
    codes = np.where(corner, 2.0, 0.0)
    codes[n - 3:] = 1.0
    assert not corner[n - 3:].any() and corner.sum() > 70
    x = c['positions'].copy()
    x[n - 3:] += np.array([1.69, 1.69, 1.69]) - x[n - 3]                     # the point farthest from the cube and its images: 1.37 nm
    far = np.linalg.norm((x[corner][:, None, :] - x[None, n - 3:, :] + 0.5 * c['box']) % c['box'] - 0.5 * c['box'], axis=2).min()
    assert far > 1.05
    ctx = B.HipContext(n, c['box'])
    ctx.set_option('small_group', 0)
    fid = ctx.pair_create(desc, codes, sigma, eps, c['exc_pairs'], skin=0.1)
    e, f = eval_force(ctx, fid, dev(x), n)
    assert e == 0.0 and not f.any()
    x[n - 3:] += np.array([0.45, 0.45, 0.45]) - x[n - 3]                     # into the corner
    out = torch.zeros((n, 3), dtype=torch.float64, device='cuda')
    ctx.force_eval(fid, dev(x), out)
    with pytest.raises(B.HipError):
        ctx.check()
    ctx.close()


def test_bonded_terms_vs_oracle(heaq, goldens):
    B = _backend()
    h = heaq
    n = len(h['positions'])
    ctx = B.HipContext(n, h['box'])
    pos = dev(h['positions'])
    sets = {
        'bonds': (B.BOND_HARMONIC, h['bonds'], np.stack([h['bond_r0'], h['bond_k']], 1), False,
                  O.harmonic_bonds(h['bonds'], h['bond_r0'], h['bond_k'], h['positions'], h['box']), 'G_heaq_bonds'),
        'angles': (B.ANGLE_HARMONIC, h['angles'], np.stack([h['angle_theta0'], h['angle_k']], 1), False,
                   O.harmonic_angles(h['angles'], h['angle_theta0'], h['angle_k'], h['positions'], h['box']),
                   'G_heaq_angles'),
        'torsions': (B.TORSION_PERIODIC, h['torsions'],
                     np.stack([h['torsion_n'].astype(float), h['torsion_phase'], h['torsion_k']], 1), False,
                     O.periodic_torsions(h['torsions'], h['torsion_n'], h['torsion_phase'], h['torsion_k'],
                                         h['positions'], h['box']), 'G_heaq_torsions'),
        'ljc': (B.BOND_LJC, h['exc_pairs'], np.stack([h['exc_chargeprod'], h['exc_sigma'], h['exc_epsilon']], 1), True,
                O.ljc_bonds(h['exc_pairs'], h['exc_chargeprod'], h['exc_sigma'], h['exc_epsilon'], h['positions'],
                            h['box'], periodic=True), None),
    }
    ftot_ref = np.zeros((n, 3))
    allid = ctx.bonded_create()
    for name, (kind, idx, par, periodic, (e_ref, f_ref), gid) in sets.items():
        bid = ctx.bonded_create()
        ctx.bonded_add_terms(bid, kind, idx, par, periodic=periodic)
        ctx.bonded_finalize(bid)
        ctx.bonded_add_terms(allid, kind, idx, par, periodic=periodic)
        e, f = eval_force(ctx, bid, pos, n)
        assert e == pytest.approx(e_ref, rel=1e-11, abs=1e-10), name
        assert np.abs(f - f_ref).max() <= 1e-10 * max(1.0, np.abs(f_ref).max()), name
        if gid:
            assert e == pytest.approx(goldens[gid]['value'], rel=1e-6)
        ftot_ref += f_ref
    ctx.bonded_finalize(allid)
    _, f = eval_force(ctx, allid, pos, n)
    assert np.abs(f - ftot_ref).max() <= 1e-10 * np.abs(ftot_ref).max()
    # near-guarded exception bonds (NearExceptionForce, forces.py:673-680)
    d = near('force-switch', 0.7, 0.5, flags=O.GUARD_RC0)
    e_ref, f_ref = O.near_bonds(d, h['exc_pairs'], h['exc_chargeprod'], h['exc_sigma'], h['exc_epsilon'],
                                h['positions'], h['box'], periodic=True)
    bid = ctx.bonded_create()
    ctx.bonded_add_terms(bid, B.BOND_NEAR, h['exc_pairs'],
                         np.stack([h['exc_chargeprod'], h['exc_sigma'], h['exc_epsilon']], 1), periodic=True,
                         desc=B.pair_desc(B.NEAR_FSWITCH, 0.7, rc0=0.7, rs0=0.5, flags=B.GUARD_RC0))
    ctx.bonded_finalize(bid)
    e, f = eval_force(ctx, bid, pos, n)
    assert e == pytest.approx(e_ref, rel=1e-10)
    assert np.abs(f - f_ref).max() <= 1e-10 * np.abs(f_ref).max()
    ctx.close()


def test_term_parallel_bond_lists_equal_the_record_walk_bit_for_bit(heaq):
    """Sets with many terms are evaluated term by term (k_terms_eval + k_terms_gather) when only forces are asked for, and
    by the per-atom record walk (k_bonded) when the energy is wanted too: same bonded_term_forces, same order of an atom's
    records => the same bits.  Every kind of term at once (harmonic bonds and angles, periodic torsions, periodic LJC
    exception bonds), each four times so that the set is past the switch-over (8192 terms)."""
    B = _backend()
    h = heaq
    n = len(h['positions'])
    ctx = B.HipContext(n, h['box'])
    pos = dev(h['positions'])
    kinds = [(B.BOND_HARMONIC, h['bonds'], np.stack([h['bond_r0'], h['bond_k']], 1), False),
             (B.ANGLE_HARMONIC, h['angles'], np.stack([h['angle_theta0'], h['angle_k']], 1), False),
             (B.TORSION_PERIODIC, h['torsions'], np.stack([h['torsion_n'].astype(float), h['torsion_phase'], h['torsion_k']], 1), False),
             (B.BOND_LJC, h['exc_pairs'], np.stack([h['exc_chargeprod'], h['exc_sigma'], h['exc_epsilon']], 1), True)]
    copies = 1 + 8192 // sum(len(k[1]) for k in kinds)
    bid = ctx.bonded_create()
    for kind, idx, par, periodic in kinds:
        ctx.bonded_add_terms(bid, kind, np.tile(idx, (copies, 1)), np.tile(par, (copies, 1)), periodic=periodic)
    ctx.bonded_finalize(bid)
    assert copies * sum(len(k[1]) for k in kinds) >= 8192
    walk_e, walk = eval_force(ctx, bid, pos, n)                         # with the energy: k_bonded
    by_term = torch.full((n, 3), float('nan'), dtype=torch.float64, device='cuda')
    ctx.force_eval(bid, pos, by_term, accumulate=False)
    ctx.check()
    assert np.array_equal(by_term.cpu().numpy(), walk)
    by_term.fill_(1.0)
    ctx.force_eval(bid, pos, by_term, accumulate=True)                  # accumulating form
    assert np.array_equal(by_term.cpu().numpy(), 1.0 + walk)
    ref = (O.harmonic_bonds(h['bonds'], h['bond_r0'], h['bond_k'], h['positions'], h['box'])[1] +
           O.harmonic_angles(h['angles'], h['angle_theta0'], h['angle_k'], h['positions'], h['box'])[1] +
           O.periodic_torsions(h['torsions'], h['torsion_n'], h['torsion_phase'], h['torsion_k'], h['positions'], h['box'])[1] +
           O.ljc_bonds(h['exc_pairs'], h['exc_chargeprod'], h['exc_sigma'], h['exc_epsilon'], h['positions'], h['box'], periodic=True)[1])
    assert np.abs(walk - copies * ref).max() <= 1e-10 * np.abs(copies * ref).max()
    ctx.close()


def test_solvation_respa_golden_G9(heaq, goldens):
    from helpers import solvation_respa_inputs
    B = _backend()
    h = heaq
    n = len(h['positions'])
    q, s, e, pairs, _, _, _ = solvation_respa_inputs(h, 0.5)
    ctx = B.HipContext(n, h['box'])
    d = near('force-switch', 0.7, 0.5)
    fid = hip_pair(B, ctx, d, h, q=h['charge'], s=s, e=e, excl=pairs)
    ctx.pair_set_params(fid, q, s, e)       # lambda_coul = 0.5 applied as a parameter update
    en, _ = eval_force(ctx, fid, dev(h['positions']), n)
    assert en == pytest.approx(goldens['G9']['value'], rel=1e-6)
    ctx.close()


def test_kick_move_bit_exact():
    B = _backend()
    rng = np.random.default_rng(7)
    n = 1000
    x = rng.normal(size=(n, 3)); v = rng.normal(size=(n, 3)); f = rng.normal(size=(n, 3)) * 100
    g = rng.normal(size=(n, 3)) * 100
    m = rng.uniform(1, 16, size=n)
    ctx = B.HipContext(n, [3.0, 3.0, 3.0])
    xd, vd, fd, gd, md = dev(x), dev(v), dev(f), dev(g), dev(m)
    ctx.kick(vd, fd, md, 0.0625 * 0.004)
    O.kick(v, f, m, 0.0625 * 0.004)
    ctx.kick(vd, fd, md, 0.5 * 0.004, fsub=gd)
    O.kick(v, f, m, 0.5 * 0.004, fsub=g)
    ctx.move(xd, vd, 0.125 * 0.004)
    O.move(x, v, 0.125 * 0.004)
    ctx.synchronize()
    assert np.array_equal(vd.cpu().numpy(), v)
    assert np.array_equal(xd.cpu().numpy(), x)
    out = torch.zeros(1, dtype=torch.float64, device='cuda')
    ctx.mvv(vd, md, out)
    assert out.item() == pytest.approx(O.mvv(v, m), rel=1e-13)
    ctx.close()


@pytest.mark.parametrize('outer_skin', [None, 0.3])
def test_neighbour_list_rebuild_and_reuse(spcfw, outer_skin):
    """Positions drift: the Verlet list is reused while max displacement < skin/2 and rebuilt after;
    forces match the oracle at every stage.  outer_skin = 0.3: dual list (outer cell-built list pruned to the
    traversed one); None: single list."""
    B = _backend()
    c = spcfw
    n = len(c['positions'])
    d = near('force-switch', 0.7, 0.5)
    ctx = B.HipContext(n, c['box'])
    if outer_skin:
        ctx.set_outer_skin(outer_skin)
    fid = hip_pair(B, ctx, d, c, skin=0.1)
    rng = np.random.default_rng(3)
    pos = c['positions'].copy()
    builds = []
    for step in range(6):
        e_ref, f_ref, _ = O.pair_eval(d, pos, c['box'], c['charge'], c['sigma'], c['epsilon'], c['exc_pairs'])
        e, f = eval_force(ctx, fid, dev(pos), n)
        assert e == pytest.approx(e_ref, rel=1e-10)
        assert np.abs(f - f_ref).max() <= 1e-9 * np.abs(f_ref).max()
        builds.append(ctx.pair_stats(fid)['n_builds'])
        pos = pos + rng.normal(scale=0.012, size=pos.shape)     # random walk, ~0.02 nm per step
    assert builds[0] == 1 and builds[1] == 1          # reused
    assert builds[-1] > 1                              # rebuilt once the skin was consumed
    st = ctx.pair_stats(fid)
    if outer_skin:
        assert st['rlist_outer'] == pytest.approx(1.0) and 1 <= st['n_outer_builds'] < st['n_builds']
        assert st['n_outer_pairs'] > st['n_list_pairs']
    # a big jump (atoms leave the box) forces a rebuild and wrapping
    pos = pos + np.array([3.1, -2.7, 5.0])
    e_ref, f_ref, _ = O.pair_eval(d, pos, c['box'], c['charge'], c['sigma'], c['epsilon'], c['exc_pairs'])
    e, f = eval_force(ctx, fid, dev(pos), n)
    assert e == pytest.approx(e_ref, rel=1e-10)
    assert np.abs(f - f_ref).max() <= 1e-9 * np.abs(f_ref).max()
    ctx.close()


def test_positions_private_contract(spcfw):
    """Option positions_private: amm_run_ops no longer assumes that the caller moved the atoms between two calls -- it keeps the
    displacement checks its own launches made -- and the caller says amm_positions_changed when it did write the bound position
    buffer.  Checked: (a) evaluations through amm_run_ops after the caller rewrote the positions + amm_positions_changed meet the
    oracle (a jump far beyond the Verlet buffer: the list must be rebuilt); (b) moves made by the library's own ops are followed
    without any notice (a MOVE of 0.3 nm); (c) the default (option off) needs no notice at all."""
    B = _backend()
    c = spcfw
    n = len(c['positions'])
    d = near('force-switch', 0.7, 0.5)
    for private in (1, 0):
        ctx = B.HipContext(n, c['box'])
        ctx.set_option('positions_private', private)
        fid = hip_pair(B, ctx, d, c, skin=0.1)
        x, v, m = dev(c['positions']), dev(np.zeros((n, 3))), dev(c['mass'])
        f = torch.zeros((n, 3), dtype=torch.float64, device='cuda')
        ctx.bind_state(x, v, m)
        ctx.bind_buffer(1, f)
        ctx.group_define(1, 1, [fid])
        ev = [B.Op(B.OP_EVAL, 1, 0, 0, 0.0)]

        def check(pos):
            f_ref = O.pair_eval(d, pos, c['box'], c['charge'], c['sigma'], c['epsilon'], c['exc_pairs'])[1]
            assert np.abs(f.cpu().numpy() - f_ref).max() <= 1e-9 * np.abs(f_ref).max()

        ctx.run_ops(ev, 1)
        ctx.check()
        check(c['positions'])
        builds0 = ctx.pair_stats(fid)['n_builds']
        # (a) the caller rewrites the bound buffer
        rng = np.random.default_rng(5)
        pos = c['positions'] + np.repeat(rng.normal(scale=0.2, size=(n // 3, 3)), 3, axis=0)       # (whole molecules)
        x.copy_(dev(pos))
        if private:
            ctx.positions_changed()
        ctx.run_ops(ev, 1)
        ctx.check()
        check(pos)
        assert ctx.pair_stats(fid)['n_builds'] > builds0
        # (b) the library moves the atoms itself: v = const, MOVE by 0.3 nm, then the evaluation -- in ONE call and in two
        v.copy_(dev(np.tile(np.array([1.0, -0.5, 0.25]), (n, 1))))
        step = 0.3 / np.sqrt(1.0 + 0.25 + 0.0625)
        ctx.run_ops([B.Op(B.OP_MOVE, 0, 0, 0, step)] + ev, 1)
        ctx.check()
        pos = pos + step * np.array([1.0, -0.5, 0.25])
        check(pos)
        ctx.run_ops([B.Op(B.OP_MOVE, 0, 0, 0, step)], 1)
        ctx.run_ops(ev, 1)
        ctx.check()
        pos = pos + step * np.array([1.0, -0.5, 0.25])
        check(pos)
        ctx.close()


@pytest.mark.parametrize('seed', [1, 2, 3, 4, 5, 6, 7, 8, 9])
def test_random_molecule_boxes_vs_oracle(seed):
    random_molecule_case(seed, seed - 1)


def random_molecule_case(seed, pattern_index):
    """Randomised boxes of three-atom molecules for the molecule-row path (csrc/cluster.hip): non-cubic boxes from the size that
    needs the image per atom pair up to one with interior rows, molecules of random shape and orientation (extent 0.07-0.14 nm),
    and the site patterns that select the kernel variants -- one site class on the first atom (site-site tables, SMASK 1), the
    same class on any atom (SMASK 7), two classes or unequal site charges (no table: analytic Lennard-Jones part), no site at
    all.  The near force alone and the fused pass (near force as guest of a damped or Ewald-direct outer force) against the
    oracle, before and after whole molecules move far enough to rebuild the rows."""
    B = _backend()
    rng = np.random.default_rng(100 + seed)
    box = rng.uniform(2.2, 3.0, 3) if seed % 3 else rng.uniform(3.6, 4.4, 3)
    nm = int(rng.integers(250, 420) * np.prod(box) / 15.0)
    m = int(np.ceil(nm ** (1 / 3)))
    grid = np.stack(np.meshgrid(*[np.arange(m)] * 3, indexing='ij'), -1).reshape(-1, 3)[rng.permutation(m ** 3)[:nm]]
    centre = (grid + 0.5 + rng.uniform(-0.2, 0.2, (nm, 3))) / m * box
    arms = rng.normal(size=(nm, 2, 3))
    arms *= (rng.uniform(0.07, 0.14, (nm, 2)) / np.linalg.norm(arms, axis=2))[:, :, None]
    pos = np.concatenate([centre[:, None, :], centre[:, None, :] + arms], axis=1).reshape(-1, 3)
    n = 3 * nm
    pattern = ['first', 'any', 'two-classes', 'unequal-charges', 'first', 'none', 'any', 'first', 'two-classes'][pattern_index % 9]
    q = np.tile([-0.8, 0.4, 0.4], nm) * rng.uniform(0.9, 1.1)
    sigma = np.full(n, 0.1)
    eps = np.zeros(n)
    first = np.arange(0, n, 3)
    if pattern in ('first', 'two-classes', 'unequal-charges'):
        sigma[first], eps[first] = 0.31, 0.65
    if pattern == 'two-classes':
        some = first[rng.random(nm) < 0.3]
        sigma[some], eps[some] = 0.27, 0.4
    if pattern == 'unequal-charges':
        some = rng.random(nm) < 0.5
        q[first[some]] *= 1.2
        q[first[some] + 1] -= 0.1 * q[first[some]] / 1.2
        q[first[some] + 2] -= 0.1 * q[first[some]] / 1.2
    if pattern == 'any':
        q[:] = np.tile([0.3, -0.6, 0.3], nm)
        who = rng.integers(0, 3, nm)
        sigma[first + who], eps[first + who] = 0.30, 0.5
        q[first + who] = -0.6                       # every site carries the same charge
        rest = np.ones(n, bool)
        rest[first + who] = False
        q[rest] = 0.3
    excl = np.array([(3 * k + a, 3 * k + b) for k in range(nm) for a, b in ((0, 1), (0, 2), (1, 2))])
    dn = near('force-switch', 0.7, 0.5)
    dd = (O.desc(O.DAMPED, rc=1.0, rswitch=0.9, alpha=2.9, degree=1) if seed % 2 else
          O.desc(O.NONBONDED, rc=1.0, rswitch=0.9, alpha=2.628260884878466, flags=O.COULOMB_EWALD | O.SWITCH))
    ctx = B.HipContext(n, box)
    case = dict(charge=q, sigma=sigma, epsilon=eps, exc_pairs=excl)
    fn = hip_pair(B, ctx, dn, case, skin=0.1)
    ff = hip_pair(B, ctx, dd, case, skin=0.1)
    ctx.pair_share_list(fn, ff)
    x, v, mass = dev(pos), dev(np.zeros((n, 3))), dev(np.ones(n))
    f = [torch.zeros((n, 3), dtype=torch.float64, device='cuda') for _ in range(3)]
    ctx.bind_state(x, v, mass)
    for slot, buf in enumerate(f):
        ctx.bind_buffer(slot, buf)
    ctx.group_define(1, 1, [fn])
    ctx.group_define(2, 2, [ff])
    for stage in range(2):
        x.copy_(dev(pos))
        refs = {1: O.pair_eval(dn, pos, box, q, sigma, eps, excl)[1], 2: O.pair_eval(dd, pos, box, q, sigma, eps, excl)[1]}
        fa = torch.empty((n, 3), dtype=torch.float64, device='cuda')
        ctx.force_eval(fn, x, fa)                                        # the near force alone
        ctx.check()
        assert np.abs(fa.cpu().numpy() - refs[1]).max() <= 1e-9 * np.abs(refs[1]).max()
        ctx.run_ops([B.Op(B.OP_EVAL, 1, 0, 0, 0.0), B.Op(B.OP_EVAL, 2, 0, 0, 0.0)], 1)       # the fused pass
        ctx.check()
        for slot in (1, 2):
            assert np.abs(f[slot].cpu().numpy() - refs[slot]).max() <= 1e-9 * np.abs(refs[slot]).max(), (pattern, slot)
        st = ctx.pair_stats(fn)
        assert st['list_kind'] == 1 and st['n_builds'] == stage + 1 and st['rode_along'] == 1
        assert st['has_site_table'] == (1 if pattern in ('first', 'any') else 0)
        # whole molecules move (many beyond skin / 2), across the faces of the box
        pos = (pos.reshape(nm, 3, 3) + rng.uniform(-0.09, 0.09, (nm, 1, 3))).reshape(n, 3) + rng.normal(0.0, 0.003, (n, 3))
    ctx.close()


@pytest.mark.parametrize('seed', [11, 12, 13, 14, 15, 16, 21, 22, 23, 31, 32])
def test_random_boxes_and_site_mixes_vs_oracle(seed):
    """Randomised configurations for the list build and both traversals: non-cubic boxes down to the size where a
    dimension holds fewer cells than the stencil is wide (minimum-image build), fractions of atoms with a Lennard-Jones site
    from none to all (the build walks sites and the rest as two streams; the force-only kernel cuts its tasks by them),
    random exclusions between near neighbours, the near force alone and as the guest of a damped outer force's pass.  Both
    the energy-carrying and the force-only (tabulated) kernel against the oracle, the in-cutoff pair count against numpy,
    before and after a random displacement that triggers a rebuild."""
    B = _backend()
    rng = np.random.default_rng(seed)
    box = rng.uniform(2.3, 4.6, 3)
    if seed % 3 == 0:
        box[rng.integers(3)] = rng.uniform(2.05, 2.25)               # fewer than five cells along one axis
    n = int(rng.integers(700, 2600))
    # jittered lattice: liquid-like density without overlaps
    m = int(np.ceil(n ** (1 / 3)))
    grid = np.stack(np.meshgrid(*[np.arange(m)] * 3, indexing='ij'), -1).reshape(-1, 3)[rng.permutation(m ** 3)[:n]]
    pos = (grid + 0.5 + rng.uniform(-0.28, 0.28, (n, 3))) / m * box
    site_fraction = [0.0, 0.1, 1 / 3, 0.5, 0.8, 1.0][seed % 6]
    has_site = rng.random(n) < site_fraction
    if seed > 30:
        # a tight ball of atoms with a site + a few isolated ones in a gas of atoms without: a wavefront then holds a row whose
        # partners with a site all sit in a long front part and none in the back, next to rows with short front parts and a
        # couple of such partners in the back (the two stretches with Lennard-Jones arithmetic overlap)
        has_site = np.zeros(n, bool)
        ball = rng.choice(n, 70, replace=False)
        centre = pos[ball[0]].copy()
        placed = []
        while len(placed) < len(ball):
            trial = centre + rng.normal(0.0, 0.2, 3)
            if np.linalg.norm(trial - centre) < 0.36 and all(np.linalg.norm(trial - p_) > 0.11 for p_ in placed):
                placed.append(trial)
        pos[ball] = np.array(placed)
        has_site[ball] = True
        far_away = np.linalg.norm((pos - centre + 0.5 * box) % box - 0.5 * box, axis=1) > 1.25
        has_site[rng.choice(np.where(far_away)[0], 40, replace=False)] = True
    elif seed > 20:
        # gradients:    elif seed > 20:
        # gradients: the density falls by a factor of ~6 along x and the atoms with a site crowd at one end of y, so that the
        # rows that share a wavefront differ widely in length and in their numbers of partners with a site
        u = pos[:, 0] / box[0]
        pos[:, 0] = box[0] * (0.35 * u + 0.65 * u ** 3)
        has_site = rng.random(n) < np.clip(1.2 - 1.6 * pos[:, 1] / box[1], 0.02, 0.98)
    q = rng.normal(0.0, 0.4, n)
    q -= q.mean()
    sigma = np.where(has_site, rng.uniform(0.25, 0.34, n), 0.1)
    eps = np.where(has_site, rng.uniform(0.2, 0.9, n), 0.0)
    # exclusions: some of the closest pairs
    d2 = ((pos[:, None, :] - pos[None, :, :] + 0.5 * box) % box - 0.5 * box)
    d2 = (d2 ** 2).sum(-1)
    close = np.argwhere(np.triu(d2 < 0.2 ** 2, 1))
    excl = close[rng.random(len(close)) < 0.5]
    dn = near('force-switch', 0.7, 0.5)
    dd = O.desc(O.DAMPED, rc=1.0, rswitch=0.9, alpha=2.9, degree=1)
    ctx = B.HipContext(n, box)
    case = dict(charge=q, sigma=sigma, epsilon=eps, exc_pairs=excl)
    fn = hip_pair(B, ctx, dn, case, skin=0.1)
    ff = hip_pair(B, ctx, dd, case, skin=0.1)
    ctx.pair_share_list(fn, ff)
    for stage in range(2):
        x = dev(pos)
        refs = {fid: O.pair_eval(d, pos, box, q, sigma, eps, excl) for fid, d in ((fn, dn), (ff, dd))}
        for fid in (fn, ff):
            e, f = eval_force(ctx, fid, x, n)                                   # energy-carrying kernel
            e_ref, f_ref, _ = refs[fid]
            scale = max(np.abs(f_ref).max(), 1.0)
            assert e == pytest.approx(e_ref, rel=1e-10, abs=1e-9)
            assert np.abs(f - f_ref).max() <= 1e-9 * scale
            only = torch.full((n, 3), float('nan'), dtype=torch.float64, device='cuda')
            ctx.force_eval(fid, x, only, accumulate=False)                      # force-only kernel (tabulated)
            ctx.check()
            assert np.abs(only.cpu().numpy() - f_ref).max() <= 1e-9 * scale
        # dual pass through the op interface: both forces in one traversal
        bufs = [torch.zeros((n, 3), dtype=torch.float64, device='cuda') for _ in range(3)]
        ctx.bind_state(x, dev(np.zeros((n, 3))), dev(np.ones(n)))
        for slot, buf in enumerate(bufs):
            ctx.bind_buffer(slot, buf)
        ctx.group_define(1, 1, [fn])
        ctx.group_define(2, 2, [ff])
        ctx.run_ops([B.Op(B.OP_EVAL, 1, 0, 0, 0.0), B.Op(B.OP_EVAL, 2, 0, 0, 0.0)], 1)
        ctx.check()
        for slot, fid in ((1, fn), (2, ff)):
            f_ref = refs[fid][1]
            assert np.abs(bufs[slot].cpu().numpy() - f_ref).max() <= 1e-9 * max(np.abs(f_ref).max(), 1.0)
        within = ctx.pair_count_within(ff, x, 1.0)
        dist2 = ((pos[:, None, :] - pos[None, :, :] + 0.5 * box) % box - 0.5 * box)
        dist2 = (dist2 ** 2).sum(-1)
        mask = dist2 < 1.0
        np.fill_diagonal(mask, False)
        mask[excl[:, 0], excl[:, 1]] = False
        mask[excl[:, 1], excl[:, 0]] = False
        assert within == int(mask.sum())
        assert ctx.pair_stats(ff)['n_builds'] == stage + 1
        pos = pos + rng.normal(0.0, 0.05, pos.shape)                            # beyond skin / 2 for many atoms
    ctx.close()


def test_site_pattern_change_rebuilds_the_rows(spcfw):
    """Parameter offsets on epsilon can give an atom a Lennard-Jones site it did not have (SolvationSystem with
    use_softcore=False at lambda_vdw = 0 -> 0.5, systems.py:309-312).  The rows' order and site counts were made for the old
    pattern: amm_pair_set_params asks for a rebuild, and a guest whose sites differ from its list owner's does not use the
    owner's counts.  Force-only evaluations (the kernel that cuts its walk by those counts) against the oracle, at fixed
    positions, before and after the change."""
    B = _backend()
    c = spcfw
    n = len(c['positions'])
    x = dev(c['positions'])
    rng = np.random.default_rng(8)
    hydrogens = np.where(c['epsilon'] == 0.0)[0]
    assert len(hydrogens) > n // 2
    dn = near('force-switch', 0.7, 0.5)
    dd = O.desc(O.DAMPED, rc=1.0, rswitch=0.9, alpha=2.9, degree=1)
    ctx = B.HipContext(n, c['box'])
    fn = hip_pair(B, ctx, dn, c, skin=0.1)
    ff = hip_pair(B, ctx, dd, c, skin=0.1)
    ctx.pair_share_list(fn, ff)

    def force_only(fid):
        out = torch.full((n, 3), float('nan'), dtype=torch.float64, device='cuda')
        ctx.force_eval(fid, x, out, accumulate=False)
        ctx.check()
        return out.cpu().numpy()

    def check(fid, d, sigma, eps):
        f_ref = O.pair_eval(d, c['positions'], c['box'], c['charge'], sigma, eps, c['exc_pairs'])[1]
        assert np.abs(force_only(fid) - f_ref).max() <= 1e-9 * np.abs(f_ref).max()

    check(ff, dd, c['sigma'], c['epsilon'])
    check(fn, dn, c['sigma'], c['epsilon'])
    builds = ctx.pair_stats(ff)['n_builds']
    # half of the hydrogens get a (small) site on BOTH forces: same positions, another site pattern
    chosen = rng.choice(hydrogens, len(hydrogens) // 2, replace=False)
    sigma2, eps2 = c['sigma'].copy(), c['epsilon'].copy()
    sigma2[chosen], eps2[chosen] = 0.12, 0.08
    ctx.pair_set_params(ff, c['charge'], sigma2, eps2)
    ctx.pair_set_params(fn, c['charge'], sigma2, eps2)
    check(ff, dd, sigma2, eps2)
    check(fn, dn, sigma2, eps2)
    assert ctx.pair_stats(ff)['n_builds'] == builds + 1
    # now only the guest changes back: its sites differ from the owner's, whose counts it must not use
    ctx.pair_set_params(fn, c['charge'], c['sigma'], c['epsilon'])
    check(fn, dn, c['sigma'], c['epsilon'])
    check(ff, dd, sigma2, eps2)
    assert ctx.pair_stats(ff)['n_builds'] == builds + 1
    ctx.close()


def test_atom_decomposition_slices_sum_to_full(spcfw):
    """amm_set_slice: two 'ranks' (two contexts on one GPU) each compute their i-slice; the sum of the
    two buffers (what the RCCL all-reduce does) equals the single-rank forces bit for bit."""
    B = _backend()
    c = spcfw
    n = len(c['positions'])
    d = near('force-switch', 0.7, 0.5)
    pos = dev(c['positions'])
    ctx = B.HipContext(n, c['box'])
    # the same Verlet buffer on both sides (its default grows with the number of ranks): same rows, same summation order
    e_full, f_full = eval_force(ctx, hip_pair(B, ctx, d, c, skin=0.1), pos, n)
    parts, es = [], []
    for r in range(2):
        cr = B.HipContext(n, c['box'], rank=r, world=2)
        e, f = eval_force(cr, hip_pair(B, cr, d, c, skin=0.1), pos, n)
        parts.append(f); es.append(e)
        assert cr.pair_stats(0)['n_slice_atoms'] == n // 2
        cr.close()
    assert np.array_equal(parts[0] + parts[1], f_full)
    assert ((parts[0] != 0).any(1) & (parts[1] != 0).any(1)).sum() == 0      # disjoint rows
    assert es[0] + es[1] == pytest.approx(e_full, rel=1e-12)
    ctx.close()


def test_shared_neighbour_list(spcfw):
    """amm_pair_share_list: the near force (rc 0.7) traverses the front part of the damped force's (rc 1.0) rows;
    both reproduce the oracle, also after drifting positions force a rebuild of the shared list."""
    B = _backend()
    c = spcfw
    n = len(c['positions'])
    dn = near('force-switch', 0.7, 0.5)
    dd = O.desc(O.DAMPED, rc=1.0, rswitch=0.9, alpha=2.9, degree=1)
    ctx = B.HipContext(n, c['box'])
    far_id = hip_pair(B, ctx, dd, c, skin=0.1)
    near_id = hip_pair(B, ctx, dn, c, skin=0.1)
    ctx.pair_share_list(near_id, far_id)
    with pytest.raises(B.HipError):
        ctx.pair_share_list(far_id, near_id)
    rng = np.random.default_rng(5)
    pos = c['positions'].copy()
    for step in range(5):
        for fid, d in ((near_id, dn), (far_id, dd)) if step % 2 == 0 else ((far_id, dd), (near_id, dn)):
            e_ref, f_ref, _ = O.pair_eval(d, pos, c['box'], c['charge'], c['sigma'], c['epsilon'], c['exc_pairs'])
            e, f = eval_force(ctx, fid, dev(pos), n)
            assert e == pytest.approx(e_ref, rel=1e-10)
            assert np.abs(f - f_ref).max() <= 1e-9 * np.abs(f_ref).max()
        pos = pos + rng.normal(scale=0.015, size=pos.shape)
    sn, sf = ctx.pair_stats(near_id), ctx.pair_stats(far_id)
    assert sn['shares_list'] == 1 and sf['shares_list'] == 0
    assert sn['n_builds'] == sf['n_builds'] > 1 and sn['capacity'] == sf['capacity']
    ctx.close()


def lj_fluid(ncell, seed=20240521):
    """C2-style synthetic LJ fluid: simple-cubic lattice + jitter, rho*sigma^3 = 0.8."""
    rng = np.random.default_rng(seed)
    sigma, eps = 0.34, 0.996
    a = sigma / 0.8 ** (1 / 3)
    g = np.arange(ncell)
    pos = np.stack(np.meshgrid(g, g, g, indexing='ij'), -1).reshape(-1, 3) * a
    pos = pos + rng.normal(scale=0.05 * sigma, size=pos.shape)
    n = len(pos)
    return pos, np.full(3, ncell * a), np.zeros(n), np.full(n, sigma), np.full(n, eps)


@pytest.mark.parametrize('adj', [None, 'shift', 'force-switch'])
def test_lj_fluid_vs_oracle_cells(adj):
    B = _backend()
    pos, box, q, s, e = lj_fluid(16)      # 4096 atoms (the 32k config C2 runs in bench/test_gpu_full)
    n = len(pos)
    d = near(adj, 0.85, 0.765)
    e_ref, f_ref, npairs = O.pair_eval(d, pos, box, q, s, e, None, use_cells=True)
    ctx = B.HipContext(n, box)
    desc = B.pair_desc(d.family, d.rc, rc0=d.rc0, rs0=d.rs0)
    fid = ctx.pair_create(desc, q, s, e, None)
    en, f = eval_force(ctx, fid, dev(pos), n)
    assert en == pytest.approx(e_ref, rel=1e-10)
    assert np.abs(f - f_ref).max() <= 1e-9 * np.abs(f_ref).max()
    st = ctx.pair_stats(fid)
    assert st['n_list_pairs'] >= 2 * npairs
    ctx.close()


def test_velocity_verlet_steps_deferred_kick_is_bit_identical():
    """A velocity-Verlet step KICK ; MOVE ; EVAL(pair force) ; KICK repeated by amm_run_ops: the kick that closes a repetition is
    deferred into the kick + move launch that opens the next (one launch less per step, config C2).  Same bits as the same steps
    issued one amm_run_ops call each (nothing to defer into), as every fusion off, and with a program of two kicks per half step
    (three leading kicks in one launch); list rebuilds on the way."""
    B = _backend()
    pos, box, q, s, e = lj_fluid(12)
    n = len(pos)
    rng = np.random.default_rng(4)
    mass = np.full(n, 39.9)
    v0 = rng.normal(size=(n, 3)) * np.sqrt(2.494 / mass)[:, None]
    d = near('force-switch', 0.85, 0.765)
    dt = 0.004

    def run(fuse, split, double_kick):
        ctx = B.HipContext(n, box)
        ctx.set_fuse_inner(fuse)
        fid = ctx.pair_create(B.pair_desc(d.family, d.rc, rc0=d.rc0, rs0=d.rs0), q, s, e, None)
        x, v, m = dev(pos), dev(v0), dev(mass)
        f = torch.zeros((n, 3), dtype=torch.float64, device='cuda')
        ctx.bind_state(x, v, m)
        ctx.bind_buffer(1, f)
        ctx.group_define(1, 1, [fid])
        K, M, E = B.OP_KICK, B.OP_MOVE, B.OP_EVAL
        half = [B.Op(K, 1, -1, 0, 0.25 * dt), B.Op(K, 1, -1, 0, 0.25 * dt)] if double_kick else [B.Op(K, 1, -1, 0, 0.5 * dt)]
        step = half + [B.Op(M, 0, 0, 0, dt), B.Op(E, 1, 0, 0, 0.0)] + half
        ctx.run_ops([B.Op(E, 1, 0, 0, 0.0)], 1)
        if split:
            for _ in range(40):
                ctx.run_ops(step, 1)
        else:
            ctx.run_ops(step, 40)
        ctx.check()
        out = x.cpu().numpy().copy(), v.cpu().numpy().copy(), ctx.pair_stats(fid)['n_builds']
        ctx.close()
        return out

    for double_kick in (False, True):
        ref = run(False, True, double_kick)
        assert ref[2] >= 2 and np.abs(ref[0] - pos).max() > 1e-3
        for fuse, split in ((True, False), (True, True), (False, False)):
            got = run(fuse, split, double_kick)
            assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]) and got[2] == ref[2]


def test_fused_inner_iteration_is_bit_identical(spcfw):
    """amm_run_ops fuses KICK;MOVE;EVAL(bond lists);KICK into one launch: same bits as the four separate launches,
    for an even and an odd number of fused iterations (the odd case copies the ping-pong state back)."""
    B = _backend()
    c = spcfw
    n = len(c['positions'])
    rng = np.random.default_rng(2)
    v0 = rng.normal(size=(n, 3)) * np.sqrt(2.494 / c['mass'])[:, None]
    results = []
    for fuse in (True, False):
        for niter in (4, 3):
            ctx = B.HipContext(n, c['box'])
            ctx.set_fuse_inner(fuse)
            bid = ctx.bonded_create()
            ctx.bonded_add_terms(bid, B.BOND_HARMONIC, c['bonds'], np.stack([c['bond_r0'], c['bond_k']], 1))
            ctx.bonded_add_terms(bid, B.ANGLE_HARMONIC, c['angles'], np.stack([c['angle_theta0'], c['angle_k']], 1))
            ctx.bonded_finalize(bid)
            x, v, m = dev(c['positions']), dev(v0), dev(c['mass'])
            f0 = torch.zeros((n, 3), dtype=torch.float64, device='cuda')
            ctx.bind_state(x, v, m)
            ctx.bind_buffer(0, f0)
            ctx.group_define(0, 0, [bid])
            ops = [B.Op(B.OP_EVAL, 0, 0, 0, 0.0)]
            for _ in range(niter):
                ops += [B.Op(B.OP_KICK, 0, -1, 0, 0.25e-3), B.Op(B.OP_MOVE, 0, 0, 0, 0.5e-3), B.Op(B.OP_EVAL, 0, 0, 0, 0.0),
                        B.Op(B.OP_KICK, 0, -1, 0, 0.25e-3)]
            ctx.run_ops(ops, repeat=3)
            ctx.check()
            results.append((fuse, niter, x.cpu().numpy(), v.cpu().numpy(), f0.cpu().numpy()))
            ctx.close()
    for niter in (4, 3):
        a = [r for r in results if r[0] and r[1] == niter][0]
        b = [r for r in results if not r[0] and r[1] == niter][0]
        assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])


def test_respa_ops_vs_oracle_trajectory(spcfw, goldens):
    """RespaPropagator([4,2,1]) op list on flexible q-SPC-FW for 2 outer steps:
    groups 0 = bonds+angles, 1 = near force-switch(0.7,0.5), 2 = DampedSmoothed(2.9,1.0,0.9); the GPU
    trajectory matches the same program driven on the oracle (positions to 1e-11 nm)."""
    B = _backend()
    c = spcfw
    n = len(c['positions'])
    dt = 0.004
    rng = np.random.default_rng(11)
    x = c['positions'].copy()
    m = c['mass'].copy()
    v = rng.normal(size=(n, 3)) * np.sqrt(2.494 / m)[:, None]
    dn = near('force-switch', 0.7, 0.5)
    dd = O.desc(O.DAMPED, rc=1.0, rswitch=0.9, alpha=2.9, degree=1)

    def f0(p):
        return (O.harmonic_bonds(c['bonds'], c['bond_r0'], c['bond_k'], p, c['box'])[1] +
                O.harmonic_angles(c['angles'], c['angle_theta0'], c['angle_k'], p, c['box'])[1])

    def f1(p):
        return O.pair_eval(dn, p, c['box'], c['charge'], c['sigma'], c['epsilon'], c['exc_pairs'])[1]

    def f2(p):
        return O.pair_eval(dd, p, c['box'], c['charge'], c['sigma'], c['epsilon'], c['exc_pairs'])[1]

    # --- oracle-driven program (exact op order of SURVEY.md 3.2)
    xo, vo = x.copy(), v.copy()
    nsteps = 2
    for _ in range(nsteps):
        F2 = f2(xo); F1 = f1(xo)
        O.kick(vo, F2, m, 0.5 * dt, fsub=F1)
        for _n1 in range(2):
            O.kick(vo, F1, m, 0.25 * dt)
            F0 = f0(xo)
            for _n0 in range(4):
                O.kick(vo, F0, m, 0.0625 * dt)
                O.move(xo, vo, 0.125 * dt)
                F0 = f0(xo)
                O.kick(vo, F0, m, 0.0625 * dt)
            F1 = f1(xo)
            O.kick(vo, F1, m, 0.25 * dt)
        F2 = f2(xo)
        O.kick(vo, F2, m, 0.5 * dt, fsub=F1)
    # --- HIP
    ctx = B.HipContext(n, c['box'])
    bid = ctx.bonded_create()
    ctx.bonded_add_terms(bid, B.BOND_HARMONIC, c['bonds'], np.stack([c['bond_r0'], c['bond_k']], 1))
    ctx.bonded_add_terms(bid, B.ANGLE_HARMONIC, c['angles'], np.stack([c['angle_theta0'], c['angle_k']], 1))
    ctx.bonded_finalize(bid)
    nid = hip_pair(B, ctx, dn, c)
    did = hip_pair(B, ctx, dd, c)
    xd, vd, md = dev(x), dev(v), dev(m)
    bufs = [torch.zeros((n, 3), dtype=torch.float64, device='cuda') for _ in range(4)]
    ctx.bind_state(xd, vd, md)
    for k, b in enumerate(bufs):
        ctx.bind_buffer(k, b)
    ctx.group_define(0, 0, [bid]); ctx.group_define(1, 1, [nid]); ctx.group_define(2, 2, [did])
    E, K, M, CP = B.OP_EVAL, B.OP_KICK, B.OP_MOVE, B.OP_COPY
    ops = [B.Op(E, 2, 0, 0, 0.0), B.Op(E, 1, 0, 0, 0.0), B.Op(CP, 3, 2, 0, 0.0), B.Op(K, 3, 1, 0, 0.5 * dt)]
    for _n1 in range(2):
        ops += [B.Op(K, 1, -1, 0, 0.25 * dt), B.Op(E, 0, 0, 0, 0.0)]
        for _n0 in range(4):
            ops += [B.Op(K, 0, -1, 0, 0.0625 * dt), B.Op(M, 0, 0, 0, 0.125 * dt), B.Op(E, 0, 0, 0, 0.0),
                    B.Op(K, 0, -1, 0, 0.0625 * dt)]
        ops += [B.Op(E, 1, 0, 0, 0.0), B.Op(K, 1, -1, 0, 0.25 * dt)]
    ops += [B.Op(E, 2, 0, 0, 0.0), B.Op(CP, 3, 2, 0, 0.0), B.Op(K, 3, 1, 0, 0.5 * dt)]
    ctx.run_ops(ops, repeat=nsteps)
    ctx.check()
    assert np.abs(xd.cpu().numpy() - xo).max() < 1e-11
    assert np.abs(vd.cpu().numpy() - vo).max() < 1e-9
    ctx.close()


@pytest.mark.parametrize('adj', ['force-switch'])
def test_c2_full_size_lj_fluid(adj):
    """Config C2 of BASELINE.json at full size: 32 768-atom LJ fluid, NearNonbondedForce only, fp64 -- energy and
    forces vs the oracle's OpenMP cell-list traversal; plus size-independent checks (Newton's third law, and
    invariance under a rigid translation that pushes every atom across the periodic boundary)."""
    B = _backend()
    from atomsmm_amd.testing import lj_fluid as lj32
    c = lj32(32)
    n = len(c['positions'])
    assert n == 32768
    d = near(adj, 0.85, 0.765)
    e_ref, f_ref, npairs = O.pair_eval(d, c['positions'], c['box'], c['charge'], c['sigma'], c['epsilon'], None, use_cells=True)
    ctx = B.HipContext(n, c['box'])
    fid = ctx.pair_create(B.pair_desc(d.family, d.rc, rc0=d.rc0, rs0=d.rs0), c['charge'], c['sigma'], c['epsilon'], None)
    e, f = eval_force(ctx, fid, dev(c['positions']), n)
    assert e == pytest.approx(e_ref, rel=1e-10)
    assert np.abs(f - f_ref).max() <= 1e-9 * np.abs(f_ref).max()
    assert np.abs(f.sum(0)).max() <= 1e-8 * np.abs(f).max()
    # the force-only launch is the kernel bench.py times (tabulated Coulomb, persistent grid): at full size too it must meet the
    # ORACLE's forces, not only its energy-carrying sibling
    f_only = torch.full((n, 3), float('nan'), dtype=torch.float64, device='cuda')
    ctx.force_eval(fid, dev(c['positions']), f_only)
    ctx.check()
    assert np.abs(f_only.cpu().numpy() - f_ref).max() <= 1e-9 * np.abs(f_ref).max()
    shifted = c['positions'] + np.array([0.61, -7.3, 23.9]) * c['box']
    e2, f2 = eval_force(ctx, fid, dev(shifted), n)
    assert e2 == pytest.approx(e, rel=1e-11)
    assert np.abs(f2 - f).max() <= 1e-8 * np.abs(f).max()
    ctx.close()


def test_c3_full_size_tip3p_respa():
    """Configs C3/C4 of BASELINE.json at full size (98 304-atom flexible TIP3P box, near 0.7/0.5 force-switch + outer
    DampedSmoothedForce sharing one neighbour list, RESPA [4,2,1] op list through amm_run_ops).  Checked: the near and
    outer forces against the oracle's OpenMP cell-list traversal; Newton's third law; the dual (one-pass) evaluation
    == the two separate evaluations bit for bit; 10 RESPA steps with list rebuilds on the way == the same ops with
    every fusion switched off, bit for bit; time reversal returns to the start."""
    B = _backend()
    from atomsmm_amd.testing import tip3p_box
    c = tip3p_box(32)
    n = len(c['positions'])
    assert n == 98304
    dn = near('force-switch', 0.7, 0.5)
    dd = O.desc(O.DAMPED, rc=1.0, rswitch=0.9, alpha=2.9, degree=1)
    ref = {}
    for key, d in (('near', dn), ('far', dd)):
        ref[key] = O.pair_eval(d, c['positions'], c['box'], c['charge'], c['sigma'], c['epsilon'], c['exc_pairs'], use_cells=True)

    def make(fuse):
        ctx = B.HipContext(n, c['box'])
        ctx.set_fuse_inner(fuse)
        fn = hip_pair(B, ctx, dn, c)
        ff = hip_pair(B, ctx, dd, c)
        ctx.pair_share_list(fn, ff)
        bid = ctx.bonded_create()
        ctx.bonded_add_terms(bid, B.BOND_HARMONIC, c['bonds'], np.stack([c['bond_r0'], c['bond_k']], 1))
        ctx.bonded_add_terms(bid, B.ANGLE_HARMONIC, c['angles'], np.stack([c['angle_theta0'], c['angle_k']], 1))
        ctx.bonded_finalize(bid)
        x, v, m = dev(c['positions']), dev(c['velocities']), dev(c['mass'])
        f = [torch.zeros((n, 3), dtype=torch.float64, device='cuda') for _ in range(4)]
        ctx.bind_state(x, v, m)
        for slot, buf in enumerate(f):
            ctx.bind_buffer(slot, buf)
        ctx.group_define(0, 0, [bid])
        ctx.group_define(1, 1, [fn])
        ctx.group_define(2, 2, [ff])
        return ctx, fn, ff, x, v, f

    ctx, fn, ff, x, v, f = make(True)
    pos = dev(c['positions'])
    for key, fid in (('near', fn), ('far', ff)):
        e, fo = eval_force(ctx, fid, pos, n)
        e_ref, f_ref, _ = ref[key]
        assert e == pytest.approx(e_ref, rel=1e-10)
        assert np.abs(fo - f_ref).max() <= 1e-9 * np.abs(f_ref).max()
        assert np.abs(fo.sum(0)).max() <= 1e-8 * np.abs(fo).max()
        # force-only launch of the same force = the traversal kernel bench.py times, against the oracle at full size
        f_only = torch.full((n, 3), float('nan'), dtype=torch.float64, device='cuda')
        ctx.force_eval(fid, pos, f_only)
        ctx.check()
        assert np.abs(f_only.cpu().numpy() - f_ref).max() <= 1e-9 * np.abs(f_ref).max()
    # the dual pass (outer + near force of the shared list in ONE traversal: the dominant kernel of the bench) through the op
    # list, against the oracle's forces of both
    x.copy_(pos)
    ctx.run_ops([B.Op(B.OP_EVAL, 1, 0, 0, 0.0), B.Op(B.OP_EVAL, 2, 0, 0, 0.0)], 1)
    ctx.check()
    for key, slot in (('near', 1), ('far', 2)):
        f_ref = ref[key][1]
        assert np.abs(f[slot].cpu().numpy() - f_ref).max() <= 1e-9 * np.abs(f_ref).max()
    # ... and WHICH kernel produced them: molecule rows, the near force inside the outer force's launch, site-site tables on both
    # (a silent fall-back to per-atom rows or to analytic Lennard-Jones would meet the oracle too, while bench.py times this one)
    st_f, st_n = ctx.pair_stats(ff), ctx.pair_stats(fn)
    assert st_f['list_kind'] == 1 and st_n['list_kind'] == 1 and st_n['shares_list'] == 1
    assert st_n['rode_along'] == 1 and st_f['rode_along'] == 0
    assert st_f['has_site_table'] == 1 and st_n['has_site_table'] == 1 and st_f['has_table'] == 1 and st_n['has_table'] == 1
    assert st_f['lanes_per_atom'] == 4                 # the product's choice for a whole box of this size
    # one outer step as RespaPropagator([4,2,1]) emits it at 2 fs, steady-state form
    dt = 0.002
    E, K, M, C_ = B.OP_EVAL, B.OP_KICK, B.OP_MOVE, B.OP_COPY
    inner = [(K, 0, -1, 0, 0.0625 * dt), (M, 0, 0, 0, 0.125 * dt), (E, 0, 0, 0, 0.0), (K, 0, -1, 0, 0.0625 * dt)] * 4
    step = [(C_, 3, 2, 0, 0.0), (K, 3, 1, 0, 0.5 * dt)]
    for _ in range(2):
        step += [(K, 1, -1, 0, 0.25 * dt)] + inner + [(E, 1, 0, 0, 0.0), (K, 1, -1, 0, 0.25 * dt)]
    ops_first = [(E, 0, 0, 0, 0.0), (E, 1, 0, 0, 0.0), (E, 2, 0, 0, 0.0)]
    # the engine pairs the last near evaluation with the outer one (atomsmm_amd/engine.py:_pair_up_evals)
    step_paired = step[:-2] + [(E, 1, 0, 0, 0.0), (E, 2, 0, 0, 0.0), (K, 1, -1, 0, 0.25 * dt), (C_, 3, 2, 0, 0.0), (K, 3, 1, 0, 0.5 * dt)]
    step_paired = [op for op in step_paired]

    def run(ctx_, nsteps):
        ctx_.run_ops([B.Op(*op) for op in ops_first], 1)
        ctx_.run_ops([B.Op(*op) for op in step_paired], nsteps)
        ctx_.check()

    run(ctx, 10)
    st = ctx.pair_stats(ff)
    assert st['n_builds'] >= 3                      # the list was rebuilt on the way
    x1, v1 = x.cpu().numpy().copy(), v.cpu().numpy().copy()
    ctx2, fn2, ff2, x2, v2, f2 = make(False)
    run(ctx2, 10)
    assert np.array_equal(x2.cpu().numpy(), x1) and np.array_equal(v2.cpu().numpy(), v1)
    assert np.array_equal(f2[1].cpu().numpy(), f[1].cpu().numpy()) and np.array_equal(f2[2].cpu().numpy(), f[2].cpu().numpy())
    ctx2.close()
    # time reversal
    v.mul_(-1.0)
    ctx.run_ops([B.Op(*op) for op in step_paired], 10)
    ctx.check()
    assert np.abs(x.cpu().numpy() - c['positions']).max() < 1e-9
    ctx.close()


@pytest.mark.parametrize('nside', [24, 27])
def test_row_phases_walk_every_row_once(nside):
    """Molecule rows whose number is no multiple of what the resident wavefronts take per round (csrc/cluster.hip: cpair_plan --
    whole rounds of the biggest tasks, the remainder in tasks of more lanes per row): nside 24 = 1 728 rows per XCD = one round
    of 4-row tasks + one of 2-row tasks + 192 single rows; 27 = 2 461 rows per XCD.  Forces of the near force, the outer force
    and the fused pass against the oracle, and against the same launch with one task size (option row_phases = 0): the same rows,
    a different order of summation inside a row (lanes per row) -- agreement to rounding, not bit for bit."""
    B = _backend()
    from atomsmm_amd.testing import tip3p_box
    c = tip3p_box(nside)
    c['positions'] = c['positions'] + np.random.default_rng(nside).normal(0.0, 0.02, c['positions'].shape)
    n = len(c['positions'])
    dn = near('force-switch', 0.7, 0.5)
    dd = O.desc(O.DAMPED, rc=1.0, rswitch=0.9, alpha=2.9, degree=1)
    ref = {key: O.pair_eval(d, c['positions'], c['box'], c['charge'], c['sigma'], c['epsilon'], c['exc_pairs'], use_cells=True)[1]
           for key, d in (('near', dn), ('far', dd))}
    out = {}
    for phases in (1, 0):
        ctx = B.HipContext(n, c['box'])
        ctx.set_option('row_phases', phases)
        fn = hip_pair(B, ctx, dn, c)
        ff = hip_pair(B, ctx, dd, c)
        ctx.pair_share_list(fn, ff)
        x, v, m = dev(c['positions']), dev(np.zeros((n, 3))), dev(c['mass'])
        f = [torch.full((n, 3), float('nan'), dtype=torch.float64, device='cuda') for _ in range(3)]
        ctx.bind_state(x, v, m)
        for slot, buf in enumerate(f):
            ctx.bind_buffer(slot, buf)
        ctx.group_define(1, 1, [fn])
        ctx.group_define(2, 2, [ff])
        ctx.run_ops([B.Op(B.OP_EVAL, 1, 0, 0, 0.0), B.Op(B.OP_EVAL, 2, 0, 0, 0.0)], 1)        # the fused pass
        ctx.check()
        fused = (f[1].cpu().numpy().copy(), f[2].cpu().numpy().copy())
        assert ctx.pair_stats(fn)['rode_along'] == 1 and ctx.pair_stats(ff)['list_kind'] == 1
        alone = torch.full((n, 3), float('nan'), dtype=torch.float64, device='cuda')
        ctx.force_eval(fn, x, alone)                                                         # the near force's own launch
        ctx.check()
        out[phases] = fused + (alone.cpu().numpy().copy(),)
        ctx.close()
    for phases in (1, 0):
        for got, key in zip(out[phases], ('near', 'far', 'near')):
            assert np.isfinite(got).all()
            assert np.abs(got - ref[key]).max() <= 1e-9 * np.abs(ref[key]).max()
    for a, b, key in zip(out[1], out[0], ('near', 'far', 'near')):
        assert np.abs(a - b).max() <= 1e-12 * np.abs(ref[key]).max()


def test_c3_full_size_ewald_direct_fused_pass():
    """The step-boundary pass of the PME variant that bench.py times under detail.pme_outer -- Ewald direct space (erfc) as list
    owner, force-switched near force as guest, ONE walk of the molecule rows -- at the full 98 304 atoms against the oracle
    (direct space only; reciprocal space keeps its small-box pins), with the kernel that ran asserted."""
    B = _backend()
    from atomsmm_amd.testing import tip3p_box
    c = tip3p_box(32)
    n = len(c['positions'])
    dn = near('force-switch', 0.7, 0.5)
    de = O.desc(O.NONBONDED, rc=1.0, rswitch=0.9, alpha=2.628260884878466, flags=O.COULOMB_EWALD | O.SWITCH)
    ctx = B.HipContext(n, c['box'])
    fn, fe = hip_pair(B, ctx, dn, c), hip_pair(B, ctx, de, c)
    ctx.pair_share_list(fn, fe)
    x, v, m = dev(c['positions']), dev(c['velocities']), dev(c['mass'])
    bufs = [torch.full((n, 3), float('nan'), dtype=torch.float64, device='cuda') for _ in range(2)]
    ctx.bind_state(x, v, m)
    ctx.bind_buffer(1, bufs[0])
    ctx.bind_buffer(2, bufs[1])
    ctx.group_define(1, 1, [fn])
    ctx.group_define(2, 2, [fe])
    ctx.run_ops([B.Op(B.OP_EVAL, 1, 0, 0, 0.0), B.Op(B.OP_EVAL, 2, 0, 0, 0.0)], 1)
    ctx.check()
    for d, buf in ((dn, bufs[0]), (de, bufs[1])):
        f_ref = O.pair_eval(d, c['positions'], c['box'], c['charge'], c['sigma'], c['epsilon'], c['exc_pairs'], use_cells=True)[1]
        assert np.abs(buf.cpu().numpy() - f_ref).max() <= 1e-9 * np.abs(f_ref).max()
    st_e, st_n = ctx.pair_stats(fe), ctx.pair_stats(fn)
    assert st_e['list_kind'] == 1 and st_n['rode_along'] == 1 and st_e['has_site_table'] == 1 and st_n['has_site_table'] == 1
    # the fused pass == the two stand-alone launches, bit for bit
    f = torch.empty((n, 3), dtype=torch.float64, device='cuda')
    ctx.force_eval(fn, x, f)
    assert torch.equal(f, bufs[0])
    ctx.force_eval(fe, x, f)
    assert torch.equal(f, bufs[1])
    ctx.close()


def test_c3_full_size_one_slice_of_eight_with_the_products_lanes():
    """Rank 3 of a world of 8 at the full 98 304 atoms, exactly as the product walks its slice (lanes per row chosen by the library for
    4 096 rows: 32), fused pass, against the ORACLE's rows of that slice -- the bit-for-bit multi-rank tests pin the lanes on both
    sides, so this is the only place the 32-lane walk meets the oracle at full size."""
    B = _backend()
    from atomsmm_amd.testing import tip3p_box
    c = tip3p_box(32)
    n = len(c['positions'])
    dn = near('force-switch', 0.7, 0.5)
    dd = O.desc(O.DAMPED, rc=1.0, rswitch=0.9, alpha=2.9, degree=1)
    ctx = B.HipContext(n, c['box'], rank=3, world=8)
    fn, fd = hip_pair(B, ctx, dn, c), hip_pair(B, ctx, dd, c)
    ctx.pair_share_list(fn, fd)
    x, v, m = dev(c['positions']), dev(c['velocities']), dev(c['mass'])
    bufs = [torch.full((n, 3), float('nan'), dtype=torch.float64, device='cuda') for _ in range(2)]
    ctx.bind_state(x, v, m)
    ctx.bind_buffer(1, bufs[0])
    ctx.bind_buffer(2, bufs[1])
    ctx.group_define(1, 1, [fn])
    ctx.group_define(2, 2, [fd])
    ctx.run_ops([B.Op(B.OP_EVAL, 1, 0, 0, 0.0), B.Op(B.OP_EVAL, 2, 0, 0, 0.0)], 1)       # (no exchange mode: rows outside the slice are zero)
    ctx.check()
    st = ctx.pair_stats(fd)
    assert st['list_kind'] == 1 and st['lanes_per_atom'] == 32 and st['n_slice_atoms'] == n // 8
    assert ctx.pair_stats(fn)['rode_along'] == 1
    for d, buf in ((dn, bufs[0]), (dd, bufs[1])):
        f_ref = O.pair_eval(d, c['positions'], c['box'], c['charge'], c['sigma'], c['epsilon'], c['exc_pairs'], use_cells=True)[1]
        got = buf.cpu().numpy()
        mine = np.abs(got).sum(axis=1) > 0.0
        assert mine.sum() == n // 8 and (mine.reshape(-1, 3).all(axis=1) == mine.reshape(-1, 3).any(axis=1)).all()      # whole molecules
        assert np.abs(got[mine] - f_ref[mine]).max() <= 1e-9 * np.abs(f_ref).max()
    ctx.close()


@pytest.mark.parametrize('nside,world,rank', [(32, 8, 3), (16, 4, 1), (8, 2, 1)])
def test_split_stream_slice_build_writes_the_same_rows(nside, world, rank):
    """The list build of a rank's slice (k_cbuild<.., SPLIT>: a block per (cell, part), the cell's candidate stream split over the block's
    four wavefronts, partial rows joined in wavefront order = stream order) must write the rows of the one-wavefront build entry for
    entry: the forces of the slice -- sums in row order -- agree BIT FOR BIT with option build_split = 0, at the first build and after a
    rebuild at moved positions; nside 8 is a box of fewer than five cells per edge (the per-pair-image variant of the kernel)."""
    B = _backend()
    from atomsmm_amd.testing import tip3p_box
    c = tip3p_box(nside)
    n = len(c['positions'])
    dn = near('force-switch', 0.7, 0.5)
    dd = O.desc(O.DAMPED, rc=1.0, rswitch=0.9, alpha=2.9, degree=1)
    rng = np.random.default_rng(5)
    moved = c['positions'] + rng.normal(0.0, 0.05, (n // 3, 3)).repeat(3, axis=0)          # (whole molecules, beyond half the Verlet buffer: a rebuild)
    results = {}
    for split in (0, -1, 2):
        ctx = B.HipContext(n, c['box'], rank=rank, world=world)
        ctx.set_option('build_split', split)
        fn, fd = hip_pair(B, ctx, dn, c), hip_pair(B, ctx, dd, c)
        ctx.pair_share_list(fn, fd)
        out = []
        for pos in (c['positions'], moved):
            x = dev(pos)
            fa = torch.empty((n, 3), dtype=torch.float64, device='cuda')
            fb = torch.empty((n, 3), dtype=torch.float64, device='cuda')
            ctx.force_eval(fd, x, fa)
            ctx.force_eval(fn, x, fb)
            out += [fa.cpu().numpy().copy(), fb.cpu().numpy().copy()]
        ctx.check()
        st = ctx.pair_stats(fd)
        assert st['list_kind'] == 1 and st['n_builds'] == 2
        assert (st['build_split'] > 0) == (split != 0) and (split <= 0 or st['build_split'] == split)
        results[split] = (out, st['n_list_pairs'])
        ctx.close()
    for split in (-1, 2):
        assert results[split][1] == results[0][1]
        for a, b in zip(results[split][0], results[0][0]):
            assert np.array_equal(a, b)
    assert np.abs(results[0][0][0]).max() > 1.0


def test_list_free_group_force_reports_overflow(heaq):
    """The list-free interaction-group path (csrc/group.hip) sums the forces on the small set in 64-bit fixed point (+-8.4e6 kJ/mol/nm at
    2^-40): a solute atom pushed INTO a solvent atom (r = 0.005 nm, Lennard-Jones force ~1e20) must be reported by amm_check, not
    wrapped around silently."""
    B = _backend()
    h = heaq
    n = len(h['positions'])
    codes = np.where(h['resname'] == 'aaa', 1.0, 2.0)
    ctx = B.HipContext(n, h['box'])
    fid = ctx.pair_create(B.pair_desc(B.NEAR_FSWITCH, 0.7, rc0=0.7, rs0=0.5, flags=B.GROUP_LJ | B.NO_SHIFT, Kc=1.0), codes, h['sigma'],
                          h['epsilon'], h['exc_pairs'])
    f = torch.empty((n, 3), dtype=torch.float64, device='cuda')
    ctx.force_eval(fid, dev(h['positions']), f)
    ctx.check()
    assert ctx.pair_stats(fid)['list_kind'] == 3
    x = h['positions'].copy()
    solute = np.where((codes == 1.0) & (h['epsilon'] > 0))[0][0]          # two Lennard-Jones sites
    solvent = np.where((codes == 2.0) & (h['epsilon'] > 0))[0][0]
    x[solute] = x[solvent] + np.array([0.005, 0.0, 0.0])
    ctx.force_eval(fid, dev(x), f)
    with pytest.raises(B.HipError, match='fixed-point'):
        ctx.check()
    ctx.close()
