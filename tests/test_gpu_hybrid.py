"""Hybrid neighbour lists (csrc/cluster.hip + csrc/pair.hip): three-site molecules that share the box with other atoms.

The reference copies ANY particle list into its CustomNonbondedForce and turns every exception into an exclusion
(/root/reference/src/atomsmm/forces.py:299-312); RESPASystem splits any System (systems.py:62-95).  The force-only hot path
therefore must not depend on the box being pure water: pairs of two three-site molecules walk molecule rows, every pair with an
atom outside them (ions, a bonded chain) walks per-atom rows kept by the force's hidden child, into the same force rows.  Checked
here against the C oracle on the same inputs (forces 1e-9 max|F|), for rest atoms behind, before and between the waters."""
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


def water_box_with_rest(nside, layout, n_ions=20, n_chain=6, seed=5):
    """tip3p_box(nside) with `n_ions` waters replaced by monatomic ions (+-1 e, a Lennard-Jones site each) and `n_chain`
    neighbouring lattice sites by a bonded chain (1-2 and 1-3 pairs excluded) -- rest atoms at the 'tail', at the 'head', or
    'between' the molecules (each one behind a randomly chosen water), so that a molecule's first atom is no multiple of 3."""
    from atomsmm_amd.testing import tip3p_box
    w = tip3p_box(nside)
    rng = np.random.default_rng(seed)
    nmol = nside ** 3
    chain_mols = np.arange(n_chain) * nside * nside + 1            # neighbours along x (the lattice is 'ij'-ordered: x slowest)
    others = np.setdiff1d(np.arange(nmol), chain_mols)
    ion_mols = rng.choice(others, n_ions, replace=False)
    gone = np.zeros(nmol, bool)
    gone[chain_mols] = gone[ion_mols] = True
    keep = np.nonzero(~gone)[0]
    opos = w['positions'].reshape(nmol, 3, 3)
    rest_pos = np.concatenate([opos[ion_mols, 0], opos[chain_mols, 0]])
    rest_q = np.concatenate([np.tile([1.0, -1.0], (n_ions + 1) // 2)[:n_ions], np.tile([0.3, -0.3], (n_chain + 1) // 2)[:n_chain]])
    rest_q[-1] -= rest_q[n_ions:].sum()
    rest_s = np.concatenate([np.tile([0.25, 0.44], (n_ions + 1) // 2)[:n_ions], np.full(n_chain, 0.34)])
    rest_e = np.concatenate([np.tile([0.2, 0.4], (n_ions + 1) // 2)[:n_ions], np.full(n_chain, 0.3)])
    nrest = n_ions + n_chain
    # order of the records: ('w', molecule) or ('r', rest atom)
    if layout == 'tail':
        order = [('w', m) for m in keep] + [('r', r) for r in range(nrest)]
    elif layout == 'head':
        order = [('r', r) for r in range(nrest)] + [('w', m) for m in keep]
    else:
        after = {}
        for r, m in enumerate(rng.choice(keep, nrest, replace=False)):
            after[m] = r
        order = []
        for m in keep:
            order.append(('w', m))
            if m in after:
                order.append(('r', after[m]))
    pos, q, s, e, exc = [], [], [], [], []
    rest_index = np.empty(nrest, dtype=np.int64)
    for kind, k in order:
        i = len(q)
        if kind == 'w':
            pos.extend(opos[k])
            q.extend([-0.834, 0.417, 0.417])
            s.extend([0.315075, 1.0, 1.0])
            e.extend([0.635968, 0.0, 0.0])
            exc.extend([(i, i + 1), (i, i + 2), (i + 1, i + 2)])
        else:
            rest_index[k] = i
            pos.append(rest_pos[k])
            q.append(rest_q[k])
            s.append(rest_s[k])
            e.append(rest_e[k])
    ci = rest_index[n_ions:]
    for gap in (1, 2):
        exc.extend(zip(ci[:-gap], ci[gap:]))
    return dict(positions=np.array(pos), box=w['box'], charge=np.array(q), sigma=np.array(s), epsilon=np.array(e),
                exc_pairs=np.array(exc, dtype=np.int32), rest=np.sort(rest_index), n_waters=len(keep))


def near(rc, rs, **kw):
    return O.desc(O.ADJ['force-switch'], rc=rc, rc0=rc, rs0=rs, **kw)


DN = near(0.7, 0.5)
DD = O.desc(O.DAMPED, rc=1.0, rswitch=0.9, alpha=2.9, degree=1)
DE = O.desc(O.NONBONDED, rc=1.0, rswitch=0.9, alpha=2.628260884878466, flags=O.COULOMB_EWALD | O.SWITCH)


def create(B, ctx, d, c, skin=-1.0):
    desc = B.pair_desc(d.family, d.rc, rc0=d.rc0, rs0=d.rs0, rswitch=d.rswitch, alpha=d.alpha, degree=d.degree, flags=d.flags,
                       sign=d.sign, Kc=d.Kc, krf=d.krf, crf=d.crf)
    return ctx.pair_create(desc, c['charge'], c['sigma'], c['epsilon'], c['exc_pairs'], skin=skin)


def oracle_forces(d, c, pos=None):
    return O.pair_eval(d, c['positions'] if pos is None else pos, c['box'], c['charge'], c['sigma'], c['epsilon'], c['exc_pairs'])


@pytest.mark.parametrize('layout', ['tail', 'head', 'between'])
@pytest.mark.parametrize('nside', [8, 12])
def test_water_box_with_ions_and_a_chain(layout, nside):
    """Near force, outer force and the fused step-boundary pass of a water box with 20 ions and a 6-atom chain, against the oracle;
    the list is hybrid (list_kind 2); energies (per-atom rows of the parent) agree too; a rebuild after the atoms moved; a
    force-only evaluation that ADDS to a buffer; and the same forces with hybrid lists switched off (per-atom rows only)."""
    B = _backend()
    c = water_box_with_rest(nside, layout)
    n = len(c['charge'])
    assert n == 3 * c['n_waters'] + 26
    ctx = B.HipContext(n, c['box'])
    fn, fd = create(B, ctx, DN, c), create(B, ctx, DD, c)
    ctx.pair_share_list(fn, fd)
    pos = dev(c['positions'])
    refs = {fn: oracle_forces(DN, c), fd: oracle_forces(DD, c)}
    f = torch.empty((n, 3), dtype=torch.float64, device='cuda')
    for fid in (fn, fd):
        f.fill_(float('nan'))
        ctx.force_eval(fid, pos, f)
        ctx.check()
        e_ref, f_ref, _ = refs[fid]
        assert np.abs(f.cpu().numpy() - f_ref).max() <= 1e-9 * np.abs(f_ref).max()
        st = ctx.pair_stats(fid)
        assert st['list_kind'] == 2 and st['n_rest_atoms'] == 26
        # adding to a buffer
        g = torch.full((n, 3), 1.5, dtype=torch.float64, device='cuda')
        ctx.force_eval(fid, pos, g, accumulate=True)
        assert np.abs(g.cpu().numpy() - 1.5 - f_ref).max() <= 1e-9 * np.abs(f_ref).max()
        # with the energy: per-atom rows of the parent itself
        en = torch.zeros(1, dtype=torch.float64, device='cuda')
        ctx.force_eval(fid, pos, f, accumulate=False, energy=en)
        ctx.check()
        assert en.item() == pytest.approx(e_ref, rel=1e-10)
        assert np.abs(f.cpu().numpy() - f_ref).max() <= 1e-9 * np.abs(f_ref).max()
    # the step-boundary pass through the op list: both forces in one walk of the molecule rows + one of the per-atom part
    x, v, m = dev(c['positions']), torch.zeros((n, 3), dtype=torch.float64, device='cuda'), torch.ones(n, dtype=torch.float64, device='cuda')
    bufs = [torch.full((n, 3), float('nan'), dtype=torch.float64, device='cuda') for _ in range(2)]
    ctx.bind_state(x, v, m)
    ctx.bind_buffer(1, bufs[0])
    ctx.bind_buffer(2, bufs[1])
    ctx.group_define(1, 1, [fn])
    ctx.group_define(2, 2, [fd])
    ctx.run_ops([B.Op(B.OP_EVAL, 1, 0, 0, 0.0), B.Op(B.OP_EVAL, 2, 0, 0, 0.0)], 1)
    ctx.check()
    for fid, buf in ((fn, bufs[0]), (fd, bufs[1])):
        f_ref = refs[fid][1]
        assert np.abs(buf.cpu().numpy() - f_ref).max() <= 1e-9 * np.abs(f_ref).max()
    assert ctx.pair_stats(fn)['rode_along'] == 1
    # the fused pass == the two stand-alone launches, bit for bit (the guest's rounding is its own launch's)
    ctx.force_eval(fn, pos, f)
    assert np.array_equal(f.cpu().numpy(), bufs[0].cpu().numpy())
    ctx.force_eval(fd, pos, f)
    assert np.array_equal(f.cpu().numpy(), bufs[1].cpu().numpy())
    # atoms move (whole molecules beyond skin / 2, every atom a little, across the faces): both lists are rebuilt
    rng = np.random.default_rng(11)
    moved = c['positions'] + rng.normal(0.0, 0.004, (n, 3))
    first = np.setdiff1d(np.arange(n), c['rest'])[::3]
    shift = rng.uniform(-0.12, 0.12, (len(first), 3))
    for a in range(3):
        moved[first + a] += shift
    moved[c['rest']] += rng.uniform(-0.1, 0.1, (26, 3))
    moved += np.array([0.4, -0.7, 1.3]) * c['box']
    b0 = ctx.pair_stats(fd)['n_builds']
    for fid, d in ((fn, DN), (fd, DD)):
        ctx.force_eval(fid, dev(moved), f)
        ctx.check()
        f_ref = oracle_forces(d, c, moved)[1]
        assert np.abs(f.cpu().numpy() - f_ref).max() <= 1e-9 * np.abs(f_ref).max()
    assert ctx.pair_stats(fd)['n_builds'] == b0 + 1
    ctx.close()
    # hybrid lists off: per-atom rows for everything, the same forces to rounding
    ctx = B.HipContext(n, c['box'])
    ctx.set_option('hybrid', 0)
    fd2 = create(B, ctx, DD, c)
    ctx.force_eval(fd2, pos, f)
    ctx.check()
    assert ctx.pair_stats(fd2)['list_kind'] == 0
    assert np.abs(f.cpu().numpy() - refs[fd][1]).max() <= 1e-9 * np.abs(refs[fd][1]).max()
    ctx.close()


def test_ewald_direct_host_with_ions():
    """The PME outer force's direct space (Ewald erfc) as list owner of a hybrid list, near force as guest."""
    B = _backend()
    c = water_box_with_rest(12, 'between')
    n = len(c['charge'])
    ctx = B.HipContext(n, c['box'])
    fn, fe = create(B, ctx, DN, c), create(B, ctx, DE, c)
    ctx.pair_share_list(fn, fe)
    x, v, m = dev(c['positions']), torch.zeros((n, 3), dtype=torch.float64, device='cuda'), torch.ones(n, dtype=torch.float64, device='cuda')
    bufs = [torch.full((n, 3), float('nan'), dtype=torch.float64, device='cuda') for _ in range(2)]
    ctx.bind_state(x, v, m)
    ctx.bind_buffer(1, bufs[0])
    ctx.bind_buffer(2, bufs[1])
    ctx.group_define(1, 1, [fn])
    ctx.group_define(2, 2, [fe])
    ctx.run_ops([B.Op(B.OP_EVAL, 1, 0, 0, 0.0), B.Op(B.OP_EVAL, 2, 0, 0, 0.0)], 1)
    ctx.check()
    for d, buf in ((DN, bufs[0]), (DE, bufs[1])):
        f_ref = oracle_forces(d, c)[1]
        assert np.abs(buf.cpu().numpy() - f_ref).max() <= 1e-9 * np.abs(f_ref).max()
    assert ctx.pair_stats(fe)['list_kind'] == 2
    ctx.close()


def test_atom_wrapped_waters():
    """A configuration whose atoms were wrapped into the box one by one (a PDB written by another program): molecules that straddle a
    face arrive in two or three pieces.  The molecule rows keep every molecule whole by the minimum image relative to its first
    atom: same forces as the unwrapped configuration's oracle."""
    B = _backend()
    from atomsmm_amd.testing import tip3p_box
    c = tip3p_box(12)
    n = len(c['positions'])
    L = c['box']
    wrapped = (c['positions'] + 0.5 * (L / 12)) % L            # lattice shifted so that a layer of molecules straddles each face
    assert (np.abs(wrapped.reshape(-1, 3, 3)[:, 1:] - wrapped.reshape(-1, 3, 3)[:, :1]).max(axis=(1, 2)) > 1.0).sum() > 50
    f_ref = O.pair_eval(DD, wrapped, L, c['charge'], c['sigma'], c['epsilon'], c['exc_pairs'])[1]
    ctx = B.HipContext(n, L)
    fd = create(B, ctx, DD, c)
    f = torch.empty((n, 3), dtype=torch.float64, device='cuda')
    ctx.force_eval(fd, dev(wrapped), f)
    ctx.check()
    assert ctx.pair_stats(fd)['list_kind'] == 1
    assert np.abs(f.cpu().numpy() - f_ref).max() <= 1e-9 * np.abs(f_ref).max()
    ctx.close()


@pytest.mark.parametrize('seed', [1, 2, 3, 4])
def test_random_mixtures(seed):
    """Random mixtures in random index order -- three-site waters, monatomic ions, diatomics (own pair excluded), four-site
    molecules (all six pairs excluded: NOT three-site molecules, whatever their first three atoms look like) -- at a random density
    in a random orthorhombic box: the hybrid list (or, below half the atoms in molecules, per-atom rows) against the oracle."""
    B = _backend()
    rng = np.random.default_rng(100 + seed)
    L = rng.uniform(2.6, 3.4, 3)
    n_w, n_ion, n_di, n_four = int(rng.integers(150, 500)), int(rng.integers(0, 40)), int(rng.integers(0, 30)), int(rng.integers(0, 20))
    kinds = ['w'] * n_w + ['i'] * n_ion + ['d'] * n_di + ['f'] * n_four
    rng.shuffle(kinds)
    pos, q, s, e, exc = [], [], [], [], []
    # molecules on a jittered lattice (no overlaps), random orientation
    m = len(kinds)
    g = int(np.ceil(m ** (1 / 3)))
    cells = rng.permutation(g ** 3)[:m]
    for kind, cell in zip(kinds, cells):
        c = (np.array([cell // (g * g), (cell // g) % g, cell % g]) + 0.5 + rng.uniform(-0.1, 0.1, 3)) / g * L
        rot = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        i = len(q)
        if kind == 'w':
            local = np.array([[0, 0, 0], [0.0757, 0.0586, 0], [-0.0757, 0.0586, 0]])
            pos.extend(c + local @ rot.T)
            q.extend([-0.834, 0.417, 0.417]); s.extend([0.315075, 1.0, 1.0]); e.extend([0.635968, 0.0, 0.0])
            exc.extend([(i, i + 1), (i, i + 2), (i + 1, i + 2)])
        elif kind == 'i':
            pos.append(c)
            q.append(float(rng.choice([-1.0, 1.0]))); s.append(0.3); e.append(0.3)
        elif kind == 'd':
            pos.extend(c + np.array([[0.05, 0, 0], [-0.05, 0, 0]]) @ rot.T)
            q.extend([0.3, -0.3]); s.extend([0.3, 0.28]); e.extend([0.4, 0.2])
            exc.append((i, i + 1))
        else:
            local = np.array([[0, 0, 0], [0.0757, 0.0586, 0], [-0.0757, 0.0586, 0], [0, 0.015, 0]])
            pos.extend(c + local @ rot.T)
            q.extend([0.0, 0.52, 0.52, -1.04]); s.extend([0.3154, 1.0, 1.0, 1.0]); e.extend([0.65, 0.0, 0.0, 0.0])
            exc.extend([(i + a, i + b) for a in range(4) for b in range(a + 1, 4)])
    c = dict(positions=np.array(pos), box=L, charge=np.array(q), sigma=np.array(s), epsilon=np.array(e),
             exc_pairs=np.array(exc, dtype=np.int32).reshape(-1, 2))
    n = len(q)
    dn_, dd_ = near(0.6, 0.45), O.desc(O.DAMPED, rc=0.9, rswitch=0.8, alpha=2.9, degree=1)
    ctx = B.HipContext(n, L)
    fn, fd = create(B, ctx, dn_, c), create(B, ctx, dd_, c)
    ctx.pair_share_list(fn, fd)
    x, v, mass = dev(c['positions']), torch.zeros((n, 3), dtype=torch.float64, device='cuda'), torch.ones(n, dtype=torch.float64, device='cuda')
    bufs = [torch.full((n, 3), float('nan'), dtype=torch.float64, device='cuda') for _ in range(2)]
    ctx.bind_state(x, v, mass)
    ctx.bind_buffer(1, bufs[0])
    ctx.bind_buffer(2, bufs[1])
    ctx.group_define(1, 1, [fn])
    ctx.group_define(2, 2, [fd])
    ctx.run_ops([B.Op(B.OP_EVAL, 1, 0, 0, 0.0), B.Op(B.OP_EVAL, 2, 0, 0, 0.0)], 1)
    ctx.check()
    for d, buf in ((dn_, bufs[0]), (dd_, bufs[1])):
        f_ref = oracle_forces(d, c)[1]
        assert np.abs(buf.cpu().numpy() - f_ref).max() <= 1e-9 * np.abs(f_ref).max()
    st = ctx.pair_stats(fd)
    n_rest = n_ion + 2 * n_di + 4 * n_four
    if n_rest == 0:
        assert st['list_kind'] == 1
    elif 2 * 3 * n_w >= n:
        assert st['list_kind'] == 2 and st['n_rest_atoms'] == n_rest
    else:
        assert st['list_kind'] == 0
    ctx.close()
