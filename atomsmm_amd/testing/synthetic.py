"""Seeded synthetic inputs for configs C2/C3 of SURVEY.md section 8(d) (numpy only, no GPU)."""
import numpy as np

SEED = 20240521


def lj_fluid(ncell=32, seed=SEED):
    """C2: N = ncell^3 LJ atoms on a simple-cubic lattice + N(0,(0.05 sigma)^2) jitter; sigma = 0.34 nm,
    eps = 0.996 kJ/mol, q = 0, m = 39.948; rho*sigma^3 = 0.8."""
    rng = np.random.default_rng(seed)
    sigma, eps = 0.34, 0.996
    a = sigma / 0.8 ** (1.0 / 3.0)
    g = np.arange(ncell)
    pos = np.stack(np.meshgrid(g, g, g, indexing='ij'), -1).reshape(-1, 3) * a
    pos = pos + rng.normal(scale=0.05 * sigma, size=pos.shape)
    n = len(pos)
    return dict(positions=pos, box=np.full(3, ncell * a), charge=np.zeros(n), sigma=np.full(n, sigma),
                epsilon=np.full(n, eps), mass=np.full(n, 39.948), exc_pairs=np.zeros((0, 2), np.int32))


def tip3p_box(nside=32, seed=SEED, density=33.368):
    """C3: nside^3 flexible TIP3P waters on a lattice with random orientations (N = 3*nside^3 atoms;
    nside = 32 -> 98 304 atoms, L = 9.939 nm).  q_O = -0.834, q_H = 0.417, sigma_O = 0.315075,
    eps_O = 0.635968, H: sigma = 1, eps = 0; harmonic O-H r0 = 0.09572 k = 462750.4, H-O-H theta0 = 1.82421813
    k = 836.8.  Atom order per molecule: O, H1, H2.  Velocities ~ N(0, kT/m) at 300 K are returned too."""
    rng = np.random.default_rng(seed)
    nmol = nside ** 3
    L = (nmol / density) ** (1.0 / 3.0)
    a = L / nside
    g = (np.arange(nside) + 0.5) * a
    centers = np.stack(np.meshgrid(g, g, g, indexing='ij'), -1).reshape(-1, 3)
    r0, th0 = 0.09572, 1.82421813
    # molecule frame: O at origin, H's in the xy plane
    h1 = np.array([r0 * np.sin(th0 / 2), r0 * np.cos(th0 / 2), 0.0])
    h2 = np.array([-r0 * np.sin(th0 / 2), r0 * np.cos(th0 / 2), 0.0])
    # random rotations from unit quaternions
    qn = rng.normal(size=(nmol, 4))
    qn /= np.linalg.norm(qn, axis=1)[:, None]
    w, x, y, z = qn.T
    R = np.stack([
        np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
        np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
        np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], 1)
    pos = np.empty((nmol, 3, 3))
    pos[:, 0] = centers
    pos[:, 1] = centers + R @ h1
    pos[:, 2] = centers + R @ h2
    pos = pos.reshape(-1, 3)
    n = 3 * nmol
    charge = np.tile([-0.834, 0.417, 0.417], nmol)
    sigma = np.tile([0.315075, 1.0, 1.0], nmol)
    epsilon = np.tile([0.635968, 0.0, 0.0], nmol)
    mass = np.tile([15.9994, 1.008, 1.008], nmol)
    o = 3 * np.arange(nmol, dtype=np.int32)
    bonds = np.concatenate([np.stack([o, o + 1], 1), np.stack([o, o + 2], 1)]).astype(np.int32)
    angles = np.stack([o + 1, o, o + 2], 1).astype(np.int32)
    exc = np.concatenate([np.stack([o, o + 1], 1), np.stack([o, o + 2], 1), np.stack([o + 1, o + 2], 1)]).astype(np.int32)
    kT = 0.0083144626181532 * 300.0
    vel = rng.normal(size=(n, 3)) * np.sqrt(kT / mass)[:, None]
    return dict(positions=pos, box=np.full(3, L), charge=charge, sigma=sigma, epsilon=epsilon, mass=mass,
                bonds=bonds, bond_r0=np.full(len(bonds), r0), bond_k=np.full(len(bonds), 462750.4),
                angles=angles, angle_theta0=np.full(nmol, th0), angle_k=np.full(nmol, 836.8),
                exc_pairs=exc, velocities=vel)


def system_from_arrays(c, nonbondedMethod='CutoffPeriodic', cutoff=1.0, switch=None, flexible=True,
                       ewaldTolerance=5e-4, dispersionCorrection=True, constraints=None, rigidWater=False):
    """Build the System `app.ForceField.createSystem(topology, ...)` would produce for a case given as arrays
    (positions, box, charge, sigma, epsilon, mass, bonds/angles/torsions, exception list): the reference's
    tests start from exactly such a System (tests/test_respa_forces.py:14-17, tests/test_systems.py:12-24).
    flexible=False drops the harmonic bond/angle terms altogether (what the static energy tests need: constraints carry
    no energy).  constraints='HBonds' turns every bond to a hydrogen (mass < 1.5) into a distance
    constraint and drops its harmonic term; rigidWater=True does the same for the three-site waters (residues whose
    three atoms are O, H, H bonded O-H twice) and also fixes their H-H distance, dropping the H-O-H angle term -- the
    combination the reference's dynamic tests use (tests/test_propagators.py:11-18)."""
    from .. import openmm
    system = openmm.System()
    for m in c['mass']:
        system.addParticle(float(m))
    L = c['box']
    system.setDefaultPeriodicBoxVectors((float(L[0]), 0, 0), (0, float(L[1]), 0), (0, 0, float(L[2])))
    mass = np.asarray(c['mass'])
    is_h = mass < 1.5
    constrained_bonds, dropped_angles = set(), set()
    if rigidWater and 'angles' in c:
        r0_of = {(int(i), int(j)): float(r0) for (i, j), r0 in zip(c['bonds'], c['bond_r0'])}
        r0_of.update({(j, i): r for (i, j), r in list(r0_of.items())})
        for idx, ((i, j, k_), t0) in enumerate(zip(c['angles'], c['angle_theta0'])):
            i, j, k_ = int(i), int(j), int(k_)
            if is_h[i] and is_h[k_] and not is_h[j] and (i, j) in r0_of and (k_, j) in r0_of and c['residue'][i] == c['residue'][k_] \
                    and int(np.sum(c['residue'] == c['residue'][j])) == 3:
                r1, r2 = r0_of[(i, j)], r0_of[(k_, j)]
                system.addConstraint(i, j, r1)
                system.addConstraint(k_, j, r2)
                system.addConstraint(i, k_, float(np.sqrt(r1 * r1 + r2 * r2 - 2 * r1 * r2 * np.cos(float(t0)))))
                constrained_bonds.update({(i, j), (j, i), (k_, j), (j, k_)})
                dropped_angles.add(idx)
    if constraints == 'HBonds' and 'bonds' in c:
        for (i, j), r0 in zip(c['bonds'], c['bond_r0']):
            i, j = int(i), int(j)
            if (is_h[i] or is_h[j]) and (i, j) not in constrained_bonds:
                system.addConstraint(i, j, float(r0))
                constrained_bonds.update({(i, j), (j, i)})
    elif constraints not in (None, 'HBonds'):
        raise ValueError("constraints must be None or 'HBonds'")
    if flexible and 'bonds' in c and len(c['bonds']):
        f = openmm.HarmonicBondForce()
        for (i, j), r0, k in zip(c['bonds'], c['bond_r0'], c['bond_k']):
            if (int(i), int(j)) not in constrained_bonds:
                f.addBond(int(i), int(j), float(r0), float(k))
        system.addForce(f)
    if flexible and 'angles' in c and len(c['angles']):
        f = openmm.HarmonicAngleForce()
        for idx, ((i, j, k_), t0, k) in enumerate(zip(c['angles'], c['angle_theta0'], c['angle_k'])):
            if idx not in dropped_angles:
                f.addAngle(int(i), int(j), int(k_), float(t0), float(k))
        system.addForce(f)
    if 'torsions' in c and len(c['torsions']):
        f = openmm.PeriodicTorsionForce()
        for (a, b, d, e), n, ph, k in zip(c['torsions'], c['torsion_n'], c['torsion_phase'], c['torsion_k']):
            f.addTorsion(int(a), int(b), int(d), int(e), int(n), float(ph), float(k))
        system.addForce(f)
    nb = openmm.NonbondedForce()
    for q, s, e in zip(c['charge'], c['sigma'], c['epsilon']):
        nb.addParticle(float(q), float(s), float(e))
    if 'exc_chargeprod' in c:
        for (i, j), qq, s, e in zip(c['exc_pairs'], c['exc_chargeprod'], c['exc_sigma'], c['exc_epsilon']):
            nb.addException(int(i), int(j), float(qq), float(s), float(e))
    else:
        for (i, j) in c['exc_pairs']:
            nb.addException(int(i), int(j), 0.0, 1.0, 0.0)
    nb.setNonbondedMethod(getattr(openmm.NonbondedForce, nonbondedMethod))
    nb.setCutoffDistance(cutoff)
    if switch is not None:
        nb.setUseSwitchingFunction(True)
        nb.setSwitchingDistance(switch)
    nb.setEwaldErrorTolerance(ewaldTolerance)
    nb.setUseDispersionCorrection(dispersionCorrection)
    system.addForce(nb)
    return system


def _zigzag_block(n_atoms, per_row, origin, dx=0.124, dy=0.042, row_gap=0.45):
    """n_atoms on a serpentine of zig-zag rows (bond 0.15 nm inside a row, rows and layers row_gap apart): a compact,
    self-avoiding bonded chain whose non-bonded neighbours are never closer than row_gap."""
    rows = (n_atoms + per_row - 1) // per_row
    side = int(np.ceil(np.sqrt(rows)))
    pos = np.empty((n_atoms, 3))
    for k in range(n_atoms):
        row, col = divmod(k, per_row)
        layer, line = divmod(row, side)
        if layer % 2:
            line = side - 1 - line                  # layers snake back so that consecutive rows stay neighbours
        c = col if row % 2 == 0 else per_row - 1 - col
        pos[k] = (c * dx, line * row_gap + (dy if c % 2 else -dy), layer * row_gap)
    return pos + np.asarray(origin)


def solvated_chain(nside=44, n_chain=3000, n_solute=30, seed=SEED, density=33.368):
    """C5 of BASELINE.json: a box of flexible TIP3P waters (nside^3 lattice sites; 44 -> 13.67 nm) holding a
    bonded `n_chain`-atom chain -- harmonic bonds and angles, periodic torsions, 1-2 / 1-3 exclusions and scaled 1-4
    exceptions (coulomb14 0.8333, lj14 0.5), alternating charges -- and an uncharged `n_solute`-atom solute meant to be
    coupled through lambda_vdw.  Waters that overlap either are removed (about 83 000 remain at nside = 44, ~252 000 atoms).
    Bond lengths and angles of the chain are at their equilibrium values in the generated geometry.  Atom order: waters
    (O, H, H), chain, solute.  Keys as tip3p_box plus torsions, the exception parameters, 'chain' and 'solute' index arrays."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(seed)
    w = tip3p_box(nside, seed, density)
    L = float(w['box'][0])
    per_row = 30 if n_chain >= 900 else 10
    rows = (n_chain + per_row - 1) // per_row
    side = int(np.ceil(np.sqrt(rows)))
    block = np.array([per_row * 0.124, side * 0.45, max(1, (rows + side - 1) // side) * 0.45])
    chain = _zigzag_block(n_chain, per_row, 0.5 * (L - block) + np.array([-0.2 * L, 0.0, 0.0]))
    solute = _zigzag_block(n_solute, 10, np.array([0.72 * L, 0.5 * L, 0.5 * L]))
    guests = np.concatenate([chain, solute]) % L
    # drop the waters that overlap the chain or the solute
    tree = cKDTree(guests, boxsize=L)
    wpos = w['positions'] % L
    near = tree.query_ball_point(wpos, r=0.30, return_length=True) > 0
    keep_mol = ~near.reshape(-1, 3).any(axis=1)
    nmol = int(keep_mol.sum())
    keep_atoms = np.repeat(keep_mol, 3)
    nw = 3 * nmol
    pos = np.concatenate([w['positions'][keep_atoms], chain, solute])
    n = len(pos)
    ic = nw + np.arange(n_chain)
    isol = nw + n_chain + np.arange(n_solute)
    charge = np.concatenate([np.tile([-0.834, 0.417, 0.417], nmol), np.tile([0.2, -0.2], (n_chain + 1) // 2)[:n_chain], np.zeros(n_solute)])
    charge[ic[-1]] -= charge[ic].sum()                      # neutral chain whatever its length
    sigma = np.concatenate([np.tile([0.315075, 1.0, 1.0], nmol), np.tile([0.33, 0.30], (n_chain + 1) // 2)[:n_chain], np.full(n_solute, 0.34)])
    epsilon = np.concatenate([np.tile([0.635968, 0.0, 0.0], nmol), np.tile([0.40, 0.25], (n_chain + 1) // 2)[:n_chain], np.full(n_solute, 0.36)])
    mass = np.concatenate([np.tile([15.9994, 1.008, 1.008], nmol), np.full(n_chain + n_solute, 12.011)])
    o = 3 * np.arange(nmol, dtype=np.int64)
    bonds = [np.stack([o, o + 1], 1), np.stack([o, o + 2], 1)]
    bond_r0 = [np.full(2 * nmol, 0.09572)]
    bond_k = [np.full(2 * nmol, 462750.4)]
    angles = [np.stack([o + 1, o, o + 2], 1)]
    angle_t0 = [np.full(nmol, 1.82421813)]
    angle_k = [np.full(nmol, 836.8)]
    torsions = []
    exc = [np.stack([o, o + 1], 1), np.stack([o, o + 2], 1), np.stack([o + 1, o + 2], 1)]
    exc_qq = [np.zeros(3 * nmol)]
    exc_sig = [np.concatenate([np.full(2 * nmol, 0.5 * (0.315075 + 1.0)), np.full(nmol, 1.0)])]
    exc_eps = [np.zeros(3 * nmol)]
    for idx in (ic, isol):
        m = len(idx)
        p = pos[idx]
        b = np.stack([idx[:-1], idx[1:]], 1)
        bonds.append(b)
        bond_r0.append(np.linalg.norm(p[1:] - p[:-1], axis=1))
        bond_k.append(np.full(m - 1, 250000.0))
        u, v = p[:-2] - p[1:-1], p[2:] - p[1:-1]
        cosang = (u * v).sum(1) / (np.linalg.norm(u, axis=1) * np.linalg.norm(v, axis=1))
        angles.append(np.stack([idx[:-2], idx[1:-1], idx[2:]], 1))
        angle_t0.append(np.arccos(np.clip(cosang, -1.0, 1.0)))
        angle_k.append(np.full(m - 2, 400.0))
        torsions.append(np.stack([idx[:-3], idx[1:-2], idx[2:-1], idx[3:]], 1))
        for gap, scale_q, scale_e in ((1, 0.0, 0.0), (2, 0.0, 0.0), (3, 0.8333, 0.5)):       # 1-2, 1-3 excluded; 1-4 scaled
            i, j = idx[:-gap], idx[gap:]
            exc.append(np.stack([i, j], 1))
            exc_qq.append(scale_q * charge[i] * charge[j])
            exc_sig.append(0.5 * (sigma[i] + sigma[j]))
            exc_eps.append(scale_e * np.sqrt(epsilon[i] * epsilon[j]))
    tors = np.concatenate(torsions).astype(np.int32)
    kT = 0.0083144626181532 * 300.0
    vel = rng.normal(size=(n, 3)) * np.sqrt(kT / mass)[:, None]
    residue = np.concatenate([np.repeat(np.arange(nmol), 3), np.full(n_chain, nmol), np.full(n_solute, nmol + 1)]).astype(np.int32)
    return dict(positions=pos, box=np.full(3, L), charge=charge, sigma=sigma, epsilon=epsilon, mass=mass,
                bonds=np.concatenate(bonds).astype(np.int32), bond_r0=np.concatenate(bond_r0), bond_k=np.concatenate(bond_k),
                angles=np.concatenate(angles).astype(np.int32), angle_theta0=np.concatenate(angle_t0), angle_k=np.concatenate(angle_k),
                torsions=tors, torsion_n=np.full(len(tors), 3, dtype=np.int32), torsion_phase=np.zeros(len(tors)),
                torsion_k=np.full(len(tors), 1.0),
                exc_pairs=np.concatenate(exc).astype(np.int32), exc_chargeprod=np.concatenate(exc_qq),
                exc_sigma=np.concatenate(exc_sig), exc_epsilon=np.concatenate(exc_eps), velocities=vel, residue=residue,
                chain=ic.astype(np.int32), solute=isol.astype(np.int32), n_waters=nmol)


def build_c5_system(case, outer='damped'):
    """The System of config C5 from a solvated_chain() case, through the AtomsMM-shaped API only:
    SolvationSystem (softcore solute-solvent force with the global parameter lambda_vdw, systems.py:240-315) ->
    RESPASystem(0.7, 0.5) (near force in group 1, exceptions as NonbondedExceptionsForce-type bonds in group 0,
    systems.py:62-95) -> the outer NonbondedForce replaced by DampedSmoothedForce(2.9/nm, 1.0, 0.9) in group 2 (the
    composition recipe of SURVEY.md 8d C1/C3) unless outer == 'pme'."""
    import atomsmm_amd as atomsmm
    from atomsmm_amd import unit
    method = 'PME' if outer == 'pme' else 'CutoffPeriodic'
    system = system_from_arrays(case, nonbondedMethod=method, cutoff=1.0, switch=0.9)
    solvated = atomsmm.SolvationSystem(system, set(int(i) for i in case['solute']))
    respa = atomsmm.RESPASystem(solvated, 0.7 * unit.nanometers, 0.5 * unit.nanometers)
    if outer != 'pme':
        nb = atomsmm.hijackForce(respa, atomsmm.findNonbondedForce(respa))
        force = atomsmm.DampedSmoothedForce(2.9 / unit.nanometers, 1.0 * unit.nanometers, 0.9 * unit.nanometers).importFrom(nb)
        force.setForceGroup(2)
        force.addTo(respa)
    return respa
