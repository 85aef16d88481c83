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
