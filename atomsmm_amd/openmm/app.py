"""`simtk.openmm.app`-shaped helpers: Simulation and a bare Topology.

PDB / ForceField XML readers are not part of the hot path ('Hard parts'):
systems are built from arrays -- see `atomsmm_amd.testing.system_from_arrays`, which mirrors what
`ForceField.createSystem` produces for the reference's test cases.
"""
from . import Context, NonbondedForce, Platform  # noqa: F401

# nonbonded method constants as exposed by openmm.app
NoCutoff = NonbondedForce.NoCutoff
CutoffNonPeriodic = NonbondedForce.CutoffNonPeriodic
CutoffPeriodic = NonbondedForce.CutoffPeriodic
Ewald = NonbondedForce.Ewald
PME = NonbondedForce.PME


class _Residue:
    def __init__(self, name, index):
        self.name, self.index = name, index


class _Atom:
    def __init__(self, name, index, residue):
        self.name, self.index, self.residue = name, index, residue


class Topology:
    """What the reference needs from `pdb.topology`: the number of atoms and, for `redefine_bond/angle`
    (systems.py:149-150), `atoms()` with `.name` and `.residue.name`.  `Topology(n)` is anonymous;
    `Topology.from_arrays(atom_names, residue_names[, residue_index])` carries the names of a fixture."""

    def __init__(self, n_atoms=0):
        self._n = n_atoms
        self._atoms = None

    @classmethod
    def from_arrays(cls, atom_names, residue_names, residue_index=None):
        top = cls(len(atom_names))
        residues = {}
        top._atoms = []
        for i, (an, rn) in enumerate(zip(atom_names, residue_names)):
            key = int(residue_index[i]) if residue_index is not None else i
            res = residues.setdefault(key, _Residue(str(rn), key))
            top._atoms.append(_Atom(str(an), i, res))
        return top

    def getNumAtoms(self):
        return self._n

    def atoms(self):
        if self._atoms is None:
            raise ValueError('this Topology carries no atom names: build it with Topology.from_arrays')
        return iter(self._atoms)


class Simulation:
    """app.Simulation(topology, system, integrator, platform=None): owns a Context."""

    def __init__(self, topology, system, integrator, platform=None, platformProperties=None):
        self.topology = topology
        self.system = system
        self.integrator = integrator
        self.context = Context(system, integrator, platform, platformProperties)
        self.currentStep = 0
        self.reporters = []

    def step(self, steps):
        self.integrator.step(steps)
        self.currentStep += steps
