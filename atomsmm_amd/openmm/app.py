"""`simtk.openmm.app`-shaped entry points that AtomsMM scripts start from.

Every test and example of the reference opens with the same three calls (tests/test_respa_forces.py:14-17,
tests/test_systems.py:12-19, tests/test_propagators.py:12-17):

    pdb = app.PDBFile(case + '.pdb')
    forcefield = app.ForceField(case + '.xml')
    system = forcefield.createSystem(pdb.topology, nonbondedMethod=app.PME, nonbondedCutoff=..., rigidWater=...,
                                     constraints=..., removeCMMotion=...)

This module provides them for the subset of the two file formats the reference's data uses: PDB `CRYST1` /
`ATOM` / `HETATM` / `CONECT` records with an orthorhombic cell, and force-field XML with `AtomTypes`, `Residues`
(templates matched by residue name and atom names), `HarmonicBondForce`, `HarmonicAngleForce`,
`PeriodicTorsionForce` (`Proper` / `Improper`, wildcard types) and `NonbondedForce` (`coulomb14scale`,
`lj14scale`, charges from the residue templates or from the `Atom` entries).  No arithmetic happens here: the result is
a `System` of the object model in `atomsmm_amd.openmm`, evaluated by the HIP path once a `Context` is created.
"""
import itertools
import os
import xml.etree.ElementTree as ET

import numpy as np

from . import (CMMotionRemover, Context, HarmonicAngleForce, HarmonicBondForce, NonbondedForce,  # noqa: F401
               PeriodicTorsionForce, Platform, System, Vec3)
from .. import unit as _unit
from ..unit import Quantity, md_value

# nonbonded methods and constraint choices, as exposed by openmm.app
NoCutoff = NonbondedForce.NoCutoff
CutoffNonPeriodic = NonbondedForce.CutoffNonPeriodic
CutoffPeriodic = NonbondedForce.CutoffPeriodic
Ewald = NonbondedForce.Ewald
PME = NonbondedForce.PME


class _ConstraintChoice:
    def __init__(self, name):
        self._name = name

    def __repr__(self):
        return self._name


HBonds = _ConstraintChoice('HBonds')
AllBonds = _ConstraintChoice('AllBonds')
HAngles = _ConstraintChoice('HAngles')


# ------------------------------------------------------------------------------------------------ topology
class Element:
    def __init__(self, symbol, mass):
        self.symbol, self.mass = symbol, mass

    def __repr__(self):
        return '<Element %s>' % self.symbol


_ELEMENT_MASS = {'H': 1.007947, 'B': 10.811, 'C': 12.01078, 'N': 14.00672, 'O': 15.99943, 'F': 18.99840325,
                 'NA': 22.98976928, 'MG': 24.3050, 'P': 30.9737622, 'S': 32.0655, 'CL': 35.4532, 'K': 39.09831}
_elements = {}


def _element(symbol):
    s = (symbol or '').strip().upper()
    if not s:
        return None
    if s not in _elements:
        _elements[s] = Element(s.capitalize(), _ELEMENT_MASS.get(s))
    return _elements[s]


class Chain:
    def __init__(self, index, id):
        self.index, self.id, self._residues = index, id, []

    def residues(self):
        return iter(self._residues)


class Residue:
    def __init__(self, name, index, chain, id=''):
        self.name, self.index, self.chain, self.id, self._atoms = name, index, chain, id, []

    def atoms(self):
        return iter(self._atoms)

    def __len__(self):
        return len(self._atoms)


class Atom:
    def __init__(self, name, element, index, residue, id=''):
        self.name, self.element, self.index, self.residue, self.id = name, element, index, residue, id

    def __repr__(self):
        return '<Atom %d (%s) of residue %d (%s)>' % (self.index, self.name, self.residue.index, self.residue.name)


class Topology:
    """The part of `openmm.app.Topology` AtomsMM touches: the atom count (utils.py:155), `atoms()` with `.name`,
    `.residue.name` (systems.py:149-150, tests/test_systems.py:23), residues, bonds and the unit cell.
    `Topology(n)` makes an anonymous topology of n atoms; `Topology.from_arrays(...)` one that carries a fixture's names."""

    def __init__(self, n_atoms=0):
        self._chains, self._residues, self._atoms, self._bonds = [], [], [], []
        self._box = None
        self._anonymous = n_atoms

    @classmethod
    def from_arrays(cls, atom_names, residue_names, residue_index=None):
        top = cls()
        chain = top.addChain()
        current, last = None, None
        for i, (an, rn) in enumerate(zip(atom_names, residue_names)):
            key = int(residue_index[i]) if residue_index is not None else i
            if key != last:
                current = top.addResidue(str(rn), chain)
                last = key
            top.addAtom(str(an), None, current)
        return top

    def addChain(self, id=None):
        chain = Chain(len(self._chains), id if id is not None else str(len(self._chains) + 1))
        self._chains.append(chain)
        return chain

    def addResidue(self, name, chain, id=None):
        res = Residue(name, len(self._residues), chain, id if id is not None else str(len(self._residues) + 1))
        self._residues.append(res)
        chain._residues.append(res)
        return res

    def addAtom(self, name, element, residue, id=None):
        atom = Atom(name, element, len(self._atoms), residue, id if id is not None else str(len(self._atoms) + 1))
        self._atoms.append(atom)
        residue._atoms.append(atom)
        return atom

    def addBond(self, atom1, atom2):
        self._bonds.append((atom1, atom2))

    def getNumAtoms(self):
        return len(self._atoms) if self._atoms else self._anonymous

    def getNumResidues(self):
        return len(self._residues)

    def getNumChains(self):
        return len(self._chains)

    def chains(self):
        return iter(self._chains)

    def residues(self):
        return iter(self._residues)

    def atoms(self):
        if not self._atoms and self._anonymous:
            raise ValueError('this Topology carries no atom names: read it with PDBFile or build it with Topology.from_arrays')
        return iter(self._atoms)

    def bonds(self):
        return iter(self._bonds)

    def setPeriodicBoxVectors(self, vectors):
        self._box = None if vectors is None else [Vec3(*[float(x) for x in md_value(v)]) for v in vectors]

    def getPeriodicBoxVectors(self):
        return None if self._box is None else Quantity(list(self._box), _unit.nanometer)

    def setUnitCellDimensions(self, dimensions):
        if dimensions is None:           # clears the cell
            self._box = None
            return
        d = md_value(dimensions)
        self._box = [Vec3(float(d[0]), 0, 0), Vec3(0, float(d[1]), 0), Vec3(0, 0, float(d[2]))]

    def getUnitCellDimensions(self):
        return None if self._box is None else Quantity(Vec3(self._box[0][0], self._box[1][1], self._box[2][2]), _unit.nanometer)


# ------------------------------------------------------------------------------------------------ PDB
class PDBFile:
    """`app.PDBFile(path)`: `.topology`, `.positions` (first model; Quantity of Vec3 in nm).  Fixed-column records;
    residues are runs of identical (chain id, residue number, insertion code, residue name); TER starts a new chain."""

    def __init__(self, file):
        own = isinstance(file, (str, os.PathLike))
        handle = open(file) if own else file
        try:
            self._parse(handle)
        finally:
            if own:
                handle.close()

    def _parse(self, handle):
        top = Topology()
        xyz, serial_to_atom = [], {}
        chain, residue, res_key, chain_id, new_chain = None, None, None, None, True
        conect = []
        done = False
        for line in handle:
            rec = line[:6]
            if rec == 'CRYST1':
                a, b, c = float(line[6:15]), float(line[15:24]), float(line[24:33])
                angles = [float(line[33:40] or 90), float(line[40:47] or 90), float(line[47:54] or 90)]
                if any(abs(x - 90.0) > 1e-6 for x in angles):
                    raise ValueError('PDBFile: only orthorhombic cells are supported by the HIP path')
                top.setUnitCellDimensions((0.1 * a, 0.1 * b, 0.1 * c))
            elif rec in ('ATOM  ', 'HETATM') and not done:
                name, resname = line[12:16].strip(), line[17:20].strip()
                key = (line[21], line[22:27], resname)
                if new_chain or line[21] != chain_id:
                    chain = top.addChain(line[21].strip() or None)
                    chain_id, new_chain, res_key = line[21], False, None
                if key != res_key:
                    residue = top.addResidue(resname, chain, line[22:26].strip())
                    res_key = key
                symbol = line[76:78].strip() if len(line) >= 78 else ''
                atom = top.addAtom(name, _element(symbol), residue, line[6:11].strip())
                serial_to_atom[line[6:11].strip()] = atom
                xyz.append(Vec3(0.1 * float(line[30:38]), 0.1 * float(line[38:46]), 0.1 * float(line[46:54])))
            elif rec[:3] == 'TER':
                new_chain = True
            elif rec == 'ENDMDL':
                done = True
            elif rec == 'CONECT':
                fields = [line[k:k + 5].strip() for k in range(6, min(len(line.rstrip('\n')), 31), 5)]
                conect.append([f for f in fields if f])
        seen = set()
        for fields in conect:
            a = serial_to_atom.get(fields[0])
            for f in fields[1:]:
                b = serial_to_atom.get(f)
                if a is not None and b is not None and (b.index, a.index) not in seen and (a.index, b.index) not in seen:
                    seen.add((a.index, b.index))
                    top.addBond(a, b)
        self.topology = top
        self.positions = Quantity(xyz, _unit.nanometer)

    def getTopology(self):
        return self.topology

    def getPositions(self, asNumpy=False, frame=0):
        if asNumpy:
            return Quantity(np.array([list(v) for v in self.positions._value]), _unit.nanometer)
        return self.positions

    def getNumFrames(self):
        return 1


# ------------------------------------------------------------------------------------------------ force field
def _match(pattern, key):
    """Type patterns of the XML tables: '' matches anything; a pattern applies forwards or backwards."""
    return any(all(p == '' or p == k for p, k in zip(pattern, cand)) for cand in (key, key[::-1]))


class ForceField:
    """`app.ForceField(*xml files)` with `createSystem(topology, ...)`.

    Templates are matched by residue name and by the set of atom names (every data file of the reference names its atoms
    uniquely per residue); table look-ups try exact type/class matches before wildcard patterns, as OpenMM does; impropers are
    not reordered (none of the reference's files defines any)."""

    def __init__(self, *files):
        self._types, self._templates = {}, {}
        self._bonds, self._angles, self._propers, self._impropers = [], [], [], []
        self._lj, self._charge_by_type = {}, {}
        self._scales = (1.0, 1.0)
        self._has_nonbonded = False
        self._charge_from_residue = False
        for f in files:
            self.loadFile(f)

    def loadFile(self, file):
        root = ET.parse(file).getroot()
        for t in root.findall('./AtomTypes/Type'):
            self._types[t.get('name')] = dict(cls=t.get('class'), mass=float(t.get('mass')), element=t.get('element'))
        for r in root.findall('./Residues/Residue'):
            atoms = [(a.get('name'), a.get('type'), float(a.get('charge', 0.0))) for a in r.findall('Atom')]
            bonds = []
            for b in r.findall('Bond'):
                if b.get('atomName1') is not None:
                    bonds.append((b.get('atomName1'), b.get('atomName2')))
                else:
                    bonds.append((atoms[int(b.get('from'))][0], atoms[int(b.get('to'))][0]))
            self._templates[r.get('name')] = dict(atoms=atoms, bonds=bonds)

        def types_of(node, n):
            """type1..n or class1..n attributes -> tuple of (kind, value)"""
            if node.get('type1') is not None:
                return ('type', tuple(node.get('type%d' % i) for i in range(1, n + 1)))
            return ('class', tuple(node.get('class%d' % i) for i in range(1, n + 1)))

        for b in root.findall('./HarmonicBondForce/Bond'):
            self._bonds.append((types_of(b, 2), float(b.get('length')), float(b.get('k'))))
        for a in root.findall('./HarmonicAngleForce/Angle'):
            self._angles.append((types_of(a, 3), float(a.get('angle')), float(a.get('k'))))
        for tag, table in (('Proper', self._propers), ('Improper', self._impropers)):
            for t in root.findall('./PeriodicTorsionForce/' + tag):
                terms = []
                for n in itertools.count(1):
                    if t.get('periodicity%d' % n) is None:
                        break
                    terms.append((int(t.get('periodicity%d' % n)), float(t.get('phase%d' % n)), float(t.get('k%d' % n))))
                table.append((types_of(t, 4), terms))
        nb = root.find('NonbondedForce')
        if nb is not None:
            self._has_nonbonded = True
            self._scales = (float(nb.get('coulomb14scale')), float(nb.get('lj14scale')))
            self._charge_from_residue = any(u.get('name') == 'charge' for u in nb.findall('UseAttributeFromResidue'))
            for a in nb.findall('Atom'):
                key = a.get('type') if a.get('type') is not None else ('class', a.get('class'))
                self._lj[key] = (float(a.get('sigma')), float(a.get('epsilon')))
                if a.get('charge') is not None:
                    self._charge_by_type[key] = float(a.get('charge'))

    # -- look-ups
    def _lookup(self, table, types, classes):
        for wildcard_pass in (False, True):
            for (kind, pattern), *values in table:
                if ('' in pattern) != wildcard_pass:
                    continue
                if _match(pattern, types if kind == 'type' else classes):
                    return values
        return None

    def _nonbonded_of(self, type_name):
        if type_name in self._lj:
            return type_name
        key = ('class', self._types[type_name]['cls'])
        if key in self._lj:
            return key
        raise ValueError('No nonbonded parameters defined for atom type %s' % type_name)

    def describe(self, topology):
        """Match templates and tables; returns plain arrays (MD units) describing the system createSystem would build:
        type, charge, sigma, epsilon, mass per atom; bonds, angles, proper torsions with parameters; the exception list
        (1-2 and 1-3 pairs excluded, 1-4 pairs scaled: OpenMM's createExceptionsFromBonds)."""
        atoms = list(topology.atoms())
        n = len(atoms)
        atype, charge, bonds = [None] * n, np.zeros(n), []
        for res in topology.residues():
            tmpl = self._templates.get(res.name)
            members = {a.name: a.index for a in res.atoms()}
            if tmpl is None or len(members) != len(res) or set(members) != set(a[0] for a in tmpl['atoms']):
                raise ValueError('No template found for residue %d (%s)' % (res.index + 1, res.name))
            for (name, type_name, q) in tmpl['atoms']:
                atype[members[name]] = type_name
                charge[members[name]] = q
            bonds += [(members[a], members[b]) for (a, b) in tmpl['bonds']]
        if not self._charge_from_residue:
            for i in range(n):
                charge[i] = self._charge_by_type.get(self._nonbonded_of(atype[i]), charge[i]) if self._has_nonbonded else charge[i]
        cls = [self._types[t]['cls'] for t in atype]
        mass = np.array([self._types[t]['mass'] for t in atype])
        lj = [self._lj[self._nonbonded_of(t)] if self._has_nonbonded else (1.0, 0.0) for t in atype]
        partners = [[] for _ in range(n)]
        for (i, j) in bonds:
            partners[i].append(j)
            partners[j].append(i)
        out = dict(type=np.array(atype), charge=charge, mass=mass, sigma=np.array([s for s, _ in lj]),
                   epsilon=np.array([e for _, e in lj]), all_bonds=np.array(bonds, dtype=np.int32).reshape(-1, 2))
        # harmonic bonds
        rows = []
        for (i, j) in bonds:
            v = self._lookup(self._bonds, (atype[i], atype[j]), (cls[i], cls[j]))
            if v is not None:
                rows.append((i, j, v[0], v[1]))
        out['bonds'] = np.array([r[:2] for r in rows], dtype=np.int32).reshape(-1, 2)
        out['bond_r0'] = np.array([r[2] for r in rows])
        out['bond_k'] = np.array([r[3] for r in rows])
        # angles i-j-k around every central atom j
        rows = []
        for j in range(n):
            for i, k in itertools.combinations(partners[j], 2):
                v = self._lookup(self._angles, (atype[i], atype[j], atype[k]), (cls[i], cls[j], cls[k]))
                if v is not None:
                    rows.append((i, j, k, v[0], v[1]))
        out['angles'] = np.array([r[:3] for r in rows], dtype=np.int32).reshape(-1, 3)
        out['angle_theta0'] = np.array([r[3] for r in rows])
        out['angle_k'] = np.array([r[4] for r in rows])
        # proper torsions i-j-k-l over every bond j-k; the same walk yields the 1-4 pairs
        rows, one_four = [], set()
        excluded = set((min(i, j), max(i, j)) for (i, j) in bonds)
        for j in range(n):
            excluded.update((min(a, b), max(a, b)) for a, b in itertools.combinations(partners[j], 2))
        for (j, k) in bonds:
            for i in partners[j]:
                if i == k:
                    continue
                for l in partners[k]:
                    if l == j or l == i:
                        continue
                    pair = (min(i, l), max(i, l))
                    if pair not in excluded:
                        one_four.add(pair)
                    v = self._lookup(self._propers, (atype[i], atype[j], atype[k], atype[l]), (cls[i], cls[j], cls[k], cls[l]))
                    if v is not None:
                        rows += [(i, j, k, l, per, phase, kk) for (per, phase, kk) in v[0]]
        out['torsions'] = np.array([r[:4] for r in rows], dtype=np.int32).reshape(-1, 4)
        out['torsion_n'] = np.array([r[4] for r in rows], dtype=np.int32)
        out['torsion_phase'] = np.array([r[5] for r in rows])
        out['torsion_k'] = np.array([r[6] for r in rows])
        c14, l14 = self._scales
        sig, eps = out['sigma'], out['epsilon']
        exc = [(i, j, 0.0, 0.5 * (sig[i] + sig[j]), 0.0) for (i, j) in sorted(excluded)]
        exc += [(i, j, c14 * charge[i] * charge[j], 0.5 * (sig[i] + sig[j]), l14 * np.sqrt(eps[i] * eps[j])) for (i, j) in sorted(one_four)]
        out['exc_pairs'] = np.array([e[:2] for e in exc], dtype=np.int32).reshape(-1, 2)
        out['exc_chargeprod'] = np.array([e[2] for e in exc])
        out['exc_sigma'] = np.array([e[3] for e in exc])
        out['exc_epsilon'] = np.array([e[4] for e in exc])
        out['n_one_four'] = len(one_four)
        return out

    def createSystem(self, topology, nonbondedMethod=NoCutoff, nonbondedCutoff=1.0 * _unit.nanometer, constraints=None,
                     rigidWater=True, removeCMMotion=True, hydrogenMass=None, residueTemplates=None, ignoreExternalBonds=False,
                     switchDistance=None, flexibleConstraints=False, ewaldErrorTolerance=0.0005, useDispersionCorrection=True):
        if hydrogenMass is not None or residueTemplates:
            raise NotImplementedError('createSystem: hydrogenMass / residueTemplates are not supported')
        if constraints not in (None, HBonds):
            raise NotImplementedError('createSystem: constraints must be None or HBonds')
        d = self.describe(topology)
        n = len(d['mass'])
        system = System()
        for m in d['mass']:
            system.addParticle(float(m))
        box = topology.getPeriodicBoxVectors()
        if box is not None:
            system.setDefaultPeriodicBoxVectors(*[tuple(v) for v in box._value])
        # constraints: rigid three-site waters (both bonds + the H-H distance), then bonds to hydrogens
        is_h = d['mass'] < 1.5
        r0_of = {}
        for (i, j), r0 in zip(d['bonds'], d['bond_r0']):
            r0_of[(int(i), int(j))] = r0_of[(int(j), int(i))] = float(r0)
        constrained, dropped_angles = set(), set()
        if rigidWater:
            size_of = {res.index: len(res) for res in topology.residues()}
            res_of = [a.residue.index for a in topology.atoms()]
            for idx, ((i, j, k), theta0) in enumerate(zip(d['angles'].tolist(), d['angle_theta0'])):
                water = is_h[i] and is_h[k] and not is_h[j] and res_of[i] == res_of[j] == res_of[k] and size_of[res_of[j]] == 3
                if water and (i, j) in r0_of and (k, j) in r0_of:
                    r1, r2 = r0_of[(i, j)], r0_of[(k, j)]
                    system.addConstraint(i, j, r1)
                    system.addConstraint(k, j, r2)
                    system.addConstraint(i, k, float(np.sqrt(r1 * r1 + r2 * r2 - 2.0 * r1 * r2 * np.cos(float(theta0)))))
                    constrained.update({(i, j), (j, i), (k, j), (j, k)})
                    dropped_angles.add(idx)
        if constraints is HBonds:
            for (i, j), r0 in zip(d['bonds'].tolist(), d['bond_r0']):
                if (is_h[i] or is_h[j]) and (i, j) not in constrained:
                    system.addConstraint(i, j, float(r0))
                    constrained.update({(i, j), (j, i)})
        if len(d['bonds']):
            force = HarmonicBondForce()
            for (i, j), r0, k in zip(d['bonds'].tolist(), d['bond_r0'], d['bond_k']):
                if flexibleConstraints or (i, j) not in constrained:
                    force.addBond(i, j, float(r0), float(k))
            system.addForce(force)
        if len(d['angles']):
            force = HarmonicAngleForce()
            for idx, ((i, j, k), theta0, kk) in enumerate(zip(d['angles'].tolist(), d['angle_theta0'], d['angle_k'])):
                if flexibleConstraints or idx not in dropped_angles:
                    force.addAngle(i, j, k, float(theta0), float(kk))
            system.addForce(force)
        if len(d['torsions']):
            force = PeriodicTorsionForce()
            for (a, b, c, e), per, phase, kk in zip(d['torsions'].tolist(), d['torsion_n'], d['torsion_phase'], d['torsion_k']):
                force.addTorsion(a, b, c, e, int(per), float(phase), float(kk))
            system.addForce(force)
        if self._has_nonbonded:
            nb = NonbondedForce()
            for q, s, e in zip(d['charge'], d['sigma'], d['epsilon']):
                nb.addParticle(float(q), float(s), float(e))
            for (i, j), qq, s, e in zip(d['exc_pairs'].tolist(), d['exc_chargeprod'], d['exc_sigma'], d['exc_epsilon']):
                nb.addException(i, j, float(qq), float(s), float(e))
            if nonbondedMethod not in (NoCutoff, CutoffNonPeriodic, CutoffPeriodic, Ewald, PME):
                raise ValueError('Illegal nonbonded method for NonbondedForce')
            nb.setNonbondedMethod(nonbondedMethod)
            nb.setCutoffDistance(nonbondedCutoff)
            if switchDistance is not None:
                nb.setUseSwitchingFunction(True)
                nb.setSwitchingDistance(switchDistance)
            nb.setEwaldErrorTolerance(ewaldErrorTolerance)
            nb.setUseDispersionCorrection(bool(useDispersionCorrection))
            system.addForce(nb)
        if removeCMMotion:
            system.addForce(CMMotionRemover())
        assert system.getNumParticles() == n
        return system


# ------------------------------------------------------------------------------------------------ simulation
class Simulation:
    """app.Simulation(topology, system, integrator, platform=None): owns a Context."""

    def __init__(self, topology, system, integrator, platform=None, platformProperties=None, state=None):
        self.topology = topology
        self.system = system
        self.integrator = integrator
        self.context = Context(system, integrator, platform, platformProperties)
        self.currentStep = 0
        self.reporters = []

    def step(self, steps):
        """Advance by `steps`, serving `self.reporters` with OpenMM's reporter protocol: `describeNextReport(simulation)` ->
        (steps until the next report, positions?, velocities?, forces?, energies?[, wrap?]) and `report(simulation, state)`."""
        remaining = int(steps)
        while remaining > 0:
            plans = [(reporter, reporter.describeNextReport(self)) for reporter in self.reporters]
            due = [plan[0] for _, plan in plans if 0 < plan[0] <= remaining]
            stride = min(due) if due else remaining
            self.integrator.step(stride)
            self.currentStep += stride
            remaining -= stride
            reporting = [(reporter, plan) for reporter, plan in plans if plan[0] == stride]
            if reporting:
                wants = [any(plan[k] for _, plan in reporting) for k in range(1, 5)]
                state = self.context.getState(getPositions=wants[0], getVelocities=wants[1], getForces=wants[2],
                                              getEnergy=wants[3], getParameters=True)
                for reporter, _ in reporting:
                    reporter.report(self, state)


class StateDataReporter:
    """app.StateDataReporter(file, reportInterval, step=..., time=..., potentialEnergy=..., kineticEnergy=..., totalEnergy=...,
    temperature=..., volume=..., density=..., speed=..., separator=','): the columns of OpenMM's reporter that a script of
    the reference typically asks for, written as separated text (units: ps, kJ/mol, K, nm^3, g/mL, ns/day)."""

    def __init__(self, file, reportInterval, step=False, time=False, potentialEnergy=False, kineticEnergy=False,
                 totalEnergy=False, temperature=False, volume=False, density=False, speed=False, separator=',', **ignored):
        self._out = open(file, 'w') if isinstance(file, str) else file
        self._interval = int(reportInterval)
        self._columns = [name for name, on in (('Step', step), ('Time (ps)', time), ('Potential Energy (kJ/mole)', potentialEnergy),
                                               ('Kinetic Energy (kJ/mole)', kineticEnergy), ('Total Energy (kJ/mole)', totalEnergy),
                                               ('Temperature (K)', temperature), ('Box Volume (nm^3)', volume),
                                               ('Density (g/mL)', density), ('Speed (ns/day)', speed)) if on]
        self._separator = separator
        self._started = None

    def describeNextReport(self, simulation):
        steps = self._interval - simulation.currentStep % self._interval
        return (steps, False, False, False, True)

    def report(self, simulation, state):
        import time as _time
        system = simulation.system
        if self._started is None:
            print('#"' + ('"' + self._separator + '"').join(self._columns) + '"', file=self._out)
            self._started = (_time.time(), state.getTime().value_in_unit(_unit.picoseconds))
            masses = [system.getParticleMass(i).value_in_unit(_unit.dalton) for i in range(system.getNumParticles())]
            self._mass = sum(masses)
            self._dof = 3 * sum(1 for m in masses if m > 0) - system.getNumConstraints()
            if any(type(force).__name__ == 'CMMotionRemover' for force in system.getForces()):
                self._dof -= 3
        pe = state.getPotentialEnergy().value_in_unit(_unit.kilojoules_per_mole)
        ke = state.getKineticEnergy().value_in_unit(_unit.kilojoules_per_mole)
        box = state.getPeriodicBoxVectors()
        volume = (box[0][0] * box[1][1] * box[2][2]).value_in_unit(_unit.nanometers ** 3)
        now, t_ps = _time.time(), state.getTime().value_in_unit(_unit.picoseconds)
        elapsed = now - self._started[0]
        values = {'Step': simulation.currentStep, 'Time (ps)': t_ps, 'Potential Energy (kJ/mole)': pe,
                  'Kinetic Energy (kJ/mole)': ke, 'Total Energy (kJ/mole)': pe + ke,
                  'Temperature (K)': 2.0 * ke / (self._dof * 8.3144626e-3),
                  'Box Volume (nm^3)': volume, 'Density (g/mL)': self._mass / volume / 602.214076,
                  'Speed (ns/day)': (t_ps - self._started[1]) * 86.4 / elapsed if elapsed > 0 else 0.0}
        print(self._separator.join(str(values[name]) for name in self._columns), file=self._out)
        if hasattr(self._out, 'flush'):
            self._out.flush()
