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


class Topology:
    """Placeholder accepted wherever the reference passes `pdb.topology` (e.g. utils.splitPotentialEnergy)."""

    def __init__(self, n_atoms=0):
        self._n = n_atoms

    def getNumAtoms(self):
        return self._n


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
