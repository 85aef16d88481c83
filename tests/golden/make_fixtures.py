#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ (run in the BUILD container only).

Input  : the reference's own test DATA files  /root/reference/tests/data/<case>.{pdb,xml}
         (data, not source: coordinates + force-field tables the reference's tests load with
         app.PDBFile / app.ForceField, e.g. tests/test_respa_forces.py:14-17).
Output : tests/golden/<case>.npz  -- plain arrays in OpenMM's unit system (nm, e, kJ/mol, dalton, rad)
         that describe what `ForceField.createSystem(topology, ...)` would have produced:
         per-particle (charge, sigma, epsilon, mass), harmonic bonds/angles, periodic torsions and the
         NonbondedForce exception list (1-2/1-3 exclusions, scaled 1-4 pairs).

The expected values the fixtures are checked against are the literals in the reference's tests; they
live in tests/golden/goldens.json with file:line citations.  Nothing of /root/reference is read at
test time: only the .npz/.json files travel.

Usage:  python tests/golden/make_fixtures.py [/root/reference/tests/data]
"""
import os
import sys
import xml.etree.ElementTree as ET

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = ['q-SPC-FW', 'hydroxyethylaminoanthraquinone-in-water', 'emim_BCN4_Jiung2014',
         'phenol-in-water', 'methane-in-water']


def read_pdb(path):
    box = None
    names, resnames, resids, xyz = [], [], [], []
    chain_res = []
    for line in open(path):
        rec = line[:6]
        if rec == 'CRYST1':
            box = [float(line[6:15]), float(line[15:24]), float(line[24:33])]
        elif rec in ('ATOM  ', 'HETATM'):
            names.append(line[12:16].strip())
            resnames.append(line[17:20].strip())
            chain_res.append((line[21], line[22:26]))
            xyz.append([float(line[30:38]), float(line[38:46]), float(line[46:54])])
        elif rec == 'ENDMDL':
            break
    # residue index = run of identical (chain, resSeq, resName)
    resid = []
    last = None
    k = -1
    for cr, rn in zip(chain_res, resnames):
        key = (cr, rn)
        if key != last:
            k += 1
            last = key
        resid.append(k)
    return (np.array(xyz) * 0.1, np.array(box) * 0.1, names, resnames, np.array(resid, dtype=np.int32))


def read_ff(path):
    root = ET.parse(path).getroot()
    types = {}
    for t in root.find('AtomTypes'):
        types[t.get('name')] = dict(cls=t.get('class'), mass=float(t.get('mass')))
    residues = {}
    for r in root.find('Residues'):
        atoms = [(a.get('name'), a.get('type'), float(a.get('charge'))) for a in r.findall('Atom')]
        bonds = [(b.get('atomName1'), b.get('atomName2')) for b in r.findall('Bond')]
        residues[r.get('name')] = dict(atoms=atoms, bonds=bonds)
    hb = root.find('HarmonicBondForce')
    bonds = [] if hb is None else [((b.get('type1'), b.get('type2')), float(b.get('length')), float(b.get('k')))
                                   for b in hb.findall('Bond')]
    ha = root.find('HarmonicAngleForce')
    angles = [] if ha is None else [((a.get('type1'), a.get('type2'), a.get('type3')),
                                     float(a.get('angle')), float(a.get('k'))) for a in ha.findall('Angle')]
    pt = root.find('PeriodicTorsionForce')
    propers, impropers = [], []
    if pt is not None:
        for tag, out in (('Proper', propers), ('Improper', impropers)):
            for t in pt.findall(tag):
                terms = []
                n = 1
                while t.get('periodicity%d' % n) is not None:
                    terms.append((int(t.get('periodicity%d' % n)), float(t.get('phase%d' % n)), float(t.get('k%d' % n))))
                    n += 1
                out.append((tuple(t.get('type%d' % i) for i in (1, 2, 3, 4)), terms))
    nb = root.find('NonbondedForce')
    lj = {a.get('type'): (float(a.get('sigma')), float(a.get('epsilon'))) for a in nb.findall('Atom')}
    scales = (float(nb.get('coulomb14scale')), float(nb.get('lj14scale')))
    return dict(types=types, residues=residues, bonds=bonds, angles=angles, propers=propers,
                impropers=impropers, lj=lj, scales=scales)


def build(case, datadir):
    pos, box, names, resnames, resid = read_pdb(os.path.join(datadir, case + '.pdb'))
    ff = read_ff(os.path.join(datadir, case + '.xml'))
    n = len(names)
    # --- template matching by residue name + atom name (all cases use unique names per residue)
    atype = [None] * n
    charge = np.zeros(n)
    bonds = []
    start = 0
    while start < n:
        end = start
        while end < n and resid[end] == resid[start]:
            end += 1
        tmpl = ff['residues'][resnames[start]]
        local = {names[i]: i for i in range(start, end)}
        assert len(local) == end - start == len(tmpl['atoms']), (case, resnames[start], start)
        for (an, at, q) in tmpl['atoms']:
            i = local[an]
            atype[i] = at
            charge[i] = q
        for (a, b) in tmpl['bonds']:
            bonds.append((local[a], local[b]))
        start = end
    cls = [ff['types'][t]['cls'] for t in atype]
    mass = np.array([ff['types'][t]['mass'] for t in atype])
    sigma = np.array([ff['lj'][t][0] for t in atype])
    epsilon = np.array([ff['lj'][t][1] for t in atype])
    bonds = np.array(bonds, dtype=np.int32).reshape(-1, 2)

    def lookup(table, key):
        """Exact match first; then patterns with wildcards ('' matches any type: OpenMM ForceField semantics)."""
        for (k, *vals) in table:
            if tuple(k) == tuple(key) or tuple(k) == tuple(reversed(key)):
                return vals
        for (k, *vals) in table:
            if '' in k:
                for cand in (tuple(key), tuple(reversed(key))):
                    if all(a == '' or a == b for a, b in zip(k, cand)):
                        return vals
        return None

    # --- harmonic bonds
    b_r0, b_k, b_keep = [], [], []
    for (i, j) in bonds:
        v = lookup(ff['bonds'], (cls[i], cls[j]))
        if v is not None:
            b_keep.append((i, j)); b_r0.append(v[0]); b_k.append(v[1])
    # --- angles from the bond graph (i-j-k, j central), unique
    nbrs = [[] for _ in range(n)]
    for (i, j) in bonds:
        nbrs[i].append(int(j)); nbrs[j].append(int(i))
    angles, a_t0, a_k = [], [], []
    for j in range(n):
        nb = nbrs[j]
        for a in range(len(nb)):
            for b in range(a + 1, len(nb)):
                i, k = nb[a], nb[b]
                v = lookup(ff['angles'], (cls[i], cls[j], cls[k]))
                if v is not None:
                    angles.append((i, j, k)); a_t0.append(v[0]); a_k.append(v[1])
    # --- proper torsions i-j-k-l
    tors, t_n, t_ph, t_k = [], [], [], []
    for (j, k) in bonds:
        for i in nbrs[j]:
            if i == k:
                continue
            for l in nbrs[k]:
                if l == j or l == i:
                    continue
                v = lookup(ff['propers'], (cls[i], cls[j], cls[k], cls[l]))
                if v is not None:
                    for (per, ph, kk) in v[0]:
                        tors.append((i, j, k, l)); t_n.append(per); t_ph.append(ph); t_k.append(kk)
    # --- exceptions: 1-2 and 1-3 excluded, 1-4 scaled (OpenMM createExceptionsFromBonds semantics)
    c14, l14 = ff['scales']
    excl = set()
    for (i, j) in bonds:
        excl.add((min(i, j), max(i, j)))
    for j in range(n):
        nb = nbrs[j]
        for a in range(len(nb)):
            for b in range(a + 1, len(nb)):
                excl.add((min(nb[a], nb[b]), max(nb[a], nb[b])))
    p14 = set()
    for (j, k) in bonds:
        for i in nbrs[j]:
            if i == k:
                continue
            for l in nbrs[k]:
                if l == j or l == i:
                    continue
                key = (min(i, l), max(i, l))
                if key not in excl:
                    p14.add(key)
    exc_pairs, exc_qq, exc_sig, exc_eps = [], [], [], []
    for (i, j) in sorted(excl):
        exc_pairs.append((i, j)); exc_qq.append(0.0); exc_sig.append(0.5 * (sigma[i] + sigma[j])); exc_eps.append(0.0)
    for (i, j) in sorted(p14):
        exc_pairs.append((i, j)); exc_qq.append(c14 * charge[i] * charge[j])
        exc_sig.append(0.5 * (sigma[i] + sigma[j])); exc_eps.append(l14 * np.sqrt(epsilon[i] * epsilon[j]))
    out = dict(
        positions=pos, box=box, charge=charge, sigma=sigma, epsilon=epsilon, mass=mass,
        residue=resid, resname=np.array(resnames), atomname=np.array(names),
        bonds=np.array(b_keep, dtype=np.int32).reshape(-1, 2), bond_r0=np.array(b_r0), bond_k=np.array(b_k),
        angles=np.array(angles, dtype=np.int32).reshape(-1, 3), angle_theta0=np.array(a_t0), angle_k=np.array(a_k),
        torsions=np.array(tors, dtype=np.int32).reshape(-1, 4), torsion_n=np.array(t_n, dtype=np.int32),
        torsion_phase=np.array(t_ph), torsion_k=np.array(t_k),
        exc_pairs=np.array(exc_pairs, dtype=np.int32).reshape(-1, 2), exc_chargeprod=np.array(exc_qq),
        exc_sigma=np.array(exc_sig), exc_epsilon=np.array(exc_eps),
    )
    np.savez_compressed(os.path.join(HERE, case + '.npz'), **out)
    print('%-45s N=%5d bonds=%5d angles=%5d tors=%4d exceptions=%5d (1-4: %d) box=%s' % (
        case, n, len(b_keep), len(angles), len(tors), len(exc_pairs), len(p14), box))


if __name__ == '__main__':
    datadir = sys.argv[1] if len(sys.argv) > 1 else '/root/reference/tests/data'
    for case in CASES:
        build(case, datadir)
