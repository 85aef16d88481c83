"""`RESPASystem` with the constructor signature of `atomsmm.systems.RESPASystem`
(reference: src/atomsmm/systems.py:34-119).

Given a System, it re-groups the nonbonded interactions for RESPA integration:

    group 0   everything bonded + (fastExceptions) the NonbondedForce exceptions as a CustomBondForce
    group 1   near pair force  (CustomNonbondedForce, cutoff rcutIn, `adjustment` = 'force-switch' by default)
    group 2   the original NonbondedForce, direct + reciprocal space
    group 31  -step(rc0-r)*(near)   bookkeeping copy so that all groups still sum to the original energy

The integrator forms the slow force as f2 - f1 (propagators.py:917-919).  Each custom force built
here also carries the structured descriptor the HIP engine consumes (see atomsmm_amd.forces).
Alchemical inputs (CustomNonbondedForces named U_linear/U_spline/U_art/U_general, systems.py:83-95)
are outside this round's scope.
"""
import copy
import itertools
import math
import re

from . import forces, openmm, utils
from .unit import md_value


class RESPASystem(openmm.System):
    def __init__(self, system, rcutIn, rswitchIn, **kwargs):
        openmm.System.__init__(self)
        self._copy_from(system)
        adjustment = kwargs.pop('adjustment', 'force-switch')
        fastExceptions = kwargs.get('fastExceptions', True)
        ljc_potential = ['4*epsilon*x*(x-1) + Kc*chargeprod/r', 'x=(sigma/r)^6', 'Kc=138.935456']
        for force in self.getForces():
            if isinstance(force, openmm.NonbondedForce):
                near = forces.nearForceExpressions(rcutIn, rswitchIn, adjustment)
                minus_near = list(near)
                minus_near[0] = '-step(rc0-r)*({})'.format(near[0])
                force.setForceGroup(2)
                force.setReciprocalSpaceForceGroup(2)
                desc = forces.NearForce._descriptor(rcutIn, rswitchIn, adjustment, False, False)
                minus = dict(desc, sign=-1.0, guard=True)
                self._addCustomNonbondedForce(near, rcutIn, 1, force, desc)
                self._addCustomNonbondedForce(minus_near, rcutIn, 31, force, minus)
                if fastExceptions:
                    self._addCustomBondForce(ljc_potential, 0, force, extract=True,
                                             descriptor=dict(family='ljc', sign=1.0, guard=False, Kc=forces.KC))
                else:
                    self._addCustomBondForce(near, 1, force, descriptor=dict(desc, guard=True))
                    self._addCustomBondForce(minus_near, 31, force, descriptor=minus)
            elif isinstance(force, openmm.CustomNonbondedForce):
                head = force.getEnergyFunction().split(';')[0]
                if head in ('U_linear', 'U_spline', 'U_art', 'U_general'):
                    raise NotImplementedError('alchemical CustomNonbondedForces are outside the HIP hot path (SURVEY.md 8a-7)')

    # The equilibrium value of a bond / an angle is changed for integration at the fastest time scale; the
    # difference between the original and the redefined harmonic potentials goes to another force group as a
    # CustomBondForce / CustomAngleForce (systems.py:121-237).
    def _matcher(self, topology, residue, atoms):
        resname = [atom.residue.name for atom in topology.atoms()]
        name = [atom.name for atom in topology.atoms()]
        r_regex = re.compile(residue)
        a_regex = [re.compile(a) for a in atoms]

        def match(*idx):
            if not all(r_regex.match(resname[j]) for j in idx):
                return False
            forward = all(a_regex[k].match(name[j]) for k, j in enumerate(idx))
            backward = all(a_regex[k].match(name[j]) for k, j in enumerate(reversed(idx)))
            return forward or backward
        return match

    def redefine_bond(self, topology, residue, atom1, atom2, length, K=None, group=1):
        match = self._matcher(topology, residue, [atom1, atom2])
        changed = []
        for force in self.getForces():
            if isinstance(force, openmm.HarmonicBondForce):
                for index in range(force.getNumBonds()):
                    i, j, r0, K0 = force.getBondParameters(index)
                    if match(i, j):
                        force.setBondParameters(index, i, j, length, K0 if K is None else K)
                        changed.append((i, j, r0, K0))
        if changed and getattr(self, '_special_bond_force', None) is None:
            new_force = openmm.CustomBondForce('0.5*(K0*(r - r0)^2 - Kn*(r - rn)^2)')
            for name in ('r0', 'K0', 'rn', 'Kn'):
                new_force.addPerBondParameter(name)
            new_force.setForceGroup(group)
            self.addForce(new_force)
            self._special_bond_force = new_force
        for (i, j, r0, K0) in changed:
            self._special_bond_force.addBond(i, j, (r0, K0, length, K0 if K is None else K))

    def redefine_angle(self, topology, residue, atom1, atom2, atom3, angle, K=None, group=1):
        match = self._matcher(topology, residue, [atom1, atom2, atom3])
        changed = []
        for force in self.getForces():
            if isinstance(force, openmm.HarmonicAngleForce):
                for index in range(force.getNumAngles()):
                    i, j, k, theta0, K0 = force.getAngleParameters(index)
                    if match(i, j, k):
                        force.setAngleParameters(index, i, j, k, angle, K0 if K is None else K)
                        changed.append((i, j, k, theta0, K0))
        if changed and getattr(self, '_special_angle_force', None) is None:
            new_force = openmm.CustomAngleForce('0.5*(K0*(theta - t0)^2 - Kn*(theta - tn)^2)')
            for name in ('t0', 'K0', 'tn', 'Kn'):
                new_force.addPerAngleParameter(name)
            new_force.setForceGroup(group)
            self.addForce(new_force)
            self._special_angle_force = new_force
        for (i, j, k, theta0, K0) in changed:
            self._special_angle_force.addAngle(i, j, k, (theta0, K0, angle, K0 if K is None else K))

    def _addCustomNonbondedForce(self, expressions, rcut, group, source, descriptor):
        force = forces._AtomsMM_CustomNonbondedForce(';'.join(expressions), rcut, use_switching_function=False,
                                                     use_dispersion_correction=False)
        force.importFrom(source)
        force._amm = descriptor
        force.setForceGroup(group)
        self.addForce(force)

    def _addCustomBondForce(self, expressions, group, nonbonded, extract=False, descriptor=None):
        force = forces._AtomsMM_CustomBondForce(';'.join(expressions))
        force.importFrom(nonbonded, extract)
        force._amm = descriptor
        if force.getNumBonds() > 0:
            force.setForceGroup(group)
            self.addForce(force)


class SolvationSystem(openmm.System):
    """`atomsmm.systems.SolvationSystem(system, solute_atoms, use_softcore=True, softcore_group=0,
    split_exceptions=False)` (reference: src/atomsmm/systems.py:240-313): a System prepared for solvation
    free-energy calculations.

    * solute-solvent Lennard-Jones either as a softcore CustomNonbondedForce over the interaction group
      (solute, solvent) with global parameter `lambda_vdw` (systems.py:266-272), or -- `use_softcore=False` -- by
      `lambda_vdw` parameter offsets on sigma and epsilon of the solute atoms (systems.py:309-312);
    * every solute-solute pair becomes an exception of the NonbondedForce (systems.py:274-287);
    * solute LJ parameters are zeroed and the solute charges become `lambda_coul` offsets (systems.py:289-308).
    """

    def __init__(self, system, solute_atoms, use_softcore=True, softcore_group=0, split_exceptions=False):
        openmm.System.__init__(self)
        self._copy_from(system)
        nonbonded = self.getForce(utils.findNonbondedForce(self))
        solute_atoms = set(int(i) for i in solute_atoms)
        solvent_atoms = set(range(nonbonded.getNumParticles())) - solute_atoms
        if split_exceptions:
            exceptions = forces._AtomsMM_CustomBondForce('4*epsilon*x*(x-1) + Kc*chargeprod/r; x=(sigma/r)^6; Kc=138.935456')
            exceptions.importFrom(nonbonded, extract=True)
            exceptions._amm = dict(family='ljc', sign=1.0, guard=False, Kc=forces.KC)
            if exceptions.getNumBonds() > 0:
                self.addForce(exceptions)
        softcore = None
        if use_softcore:
            softcore = forces._AtomsMM_CustomNonbondedForce(
                '4*lambda_vdw*epsilon*(1-x)/x^2; x=(r/sigma)^6+0.5*(1-lambda_vdw)', lambda_vdw=1)
            softcore.importFrom(nonbonded)
            softcore.addInteractionGroup(solute_atoms, solvent_atoms)
            softcore.setForceGroup(softcore_group)
            self.addForce(softcore)
        have = set()
        for index in range(nonbonded.getNumExceptions()):
            i, j = nonbonded.getExceptionParameters(index)[:2]
            if i in solute_atoms and j in solute_atoms:
                have.add(frozenset((i, j)))
        for i, j in itertools.combinations(sorted(solute_atoms), 2):
            if frozenset((i, j)) not in have:
                q1, sig1, eps1 = nonbonded.getParticleParameters(i)
                q2, sig2, eps2 = nonbonded.getParticleParameters(j)
                nonbonded.addException(i, j, q1 * q2, (sig1 + sig2) / 2, (eps1 * eps2).sqrt())
                if softcore is not None:
                    softcore.addExclusion(i, j)
        charges, lj_parameters = {}, {}
        for index in sorted(solute_atoms):
            charge, sigma, epsilon = nonbonded.getParticleParameters(index)
            nonbonded.setParticleParameters(index, 0.0, 0.0, 0.0)
            if md_value(charge) != 0.0:
                charges[index] = charge
            if md_value(epsilon) != 0.0:
                lj_parameters[index] = (sigma, epsilon)
        if charges:
            nonbonded.addGlobalParameter('lambda_coul', 1.0)
            for index, charge in charges.items():
                nonbonded.addParticleParameterOffset('lambda_coul', index, charge, 0.0, 0.0)
        if lj_parameters and not use_softcore:
            nonbonded.addGlobalParameter('lambda_vdw', 1.0)
            for index, (sigma, epsilon) in lj_parameters.items():
                nonbonded.addParticleParameterOffset('lambda_vdw', index, 0.0, sigma, epsilon)


class _AtomsMM_System(openmm.System):
    """A copy of a System, optionally without its forces (systems.py:26-31)."""

    def __init__(self, system, copyForces=True):
        openmm.System.__init__(self)
        self._copy_from(system)
        if not copyForces:
            for index in reversed(range(self.getNumForces())):
                self.removeForce(index)


class ComputingSystem(_AtomsMM_System):
    """A System whose "potential energies" are the Coulomb energy and the pieces of the internal atomic virial of the
    original one (systems.py:867-934): group 0 the dispersion virial 24 eps (2 (sigma/r)^12 - (sigma/r)^6) of pairs and
    exceptions, group 1 the bond-stretching virial -K r (r - r0), group 2 the NonbondedForce with the Lennard-Jones
    parameters switched off (W_Coulomb = E_Coulomb).  Custom bond forces with other expressions (the reference
    differentiates them with sympy, systems.py:936-947) are not supported."""

    def __init__(self, system):
        super().__init__(system, copyForces=False)
        dispersionGroup, bondedGroup, coulombGroup = 0, 1, 2
        self._dispersion, self._bonded, self._coulomb = 2 ** dispersionGroup, 2 ** bondedGroup, 2 ** coulombGroup
        expression = '24*epsilon*(2*(sigma/r)^12-(sigma/r)^6)'
        for force in system.getForces():
            if isinstance(force, openmm.NonbondedForce) and force.getNumParticles() > 0:
                nonbonded = copy.deepcopy(force)
                virial = forces._AtomsMM_CustomNonbondedForce(expression)
                virial.importFrom(nonbonded)
                virial.setForceGroup(dispersionGroup)
                self.addForce(virial)
                exceptions = forces._AtomsMM_CustomBondForce(expression)
                exceptions.importFrom(nonbonded, extract=False)
                if exceptions.getNumBonds() > 0:
                    exceptions.setForceGroup(dispersionGroup)
                    self.addForce(exceptions)
                for index in range(nonbonded.getNumParticles()):
                    charge = nonbonded.getParticleParameters(index)[0]
                    nonbonded.setParticleParameters(index, charge, 1.0, 0.0)
                for index in range(nonbonded.getNumExceptions()):
                    i, j, chargeprod = nonbonded.getExceptionParameters(index)[:3]
                    nonbonded.setExceptionParameters(index, i, j, chargeprod, 1.0, 0.0)
                nonbonded.setForceGroup(coulombGroup)
                nonbonded.setReciprocalSpaceForceGroup(coulombGroup)
                self.addForce(nonbonded)
            elif isinstance(force, openmm.HarmonicBondForce) and force.getNumBonds() > 0:
                bondforce = openmm.CustomBondForce('-K*r*(r-r0)')
                bondforce.addPerBondParameter('r0')
                bondforce.addPerBondParameter('K')
                for index in range(force.getNumBonds()):
                    i, j, r0, K = force.getBondParameters(index)
                    bondforce.addBond(i, j, [r0, K])
                bondforce.setForceGroup(bondedGroup)
                self.addForce(bondforce)
            elif isinstance(force, openmm.CustomBondForce) and force.getNumBonds() > 0:
                raise NotImplementedError('ComputingSystem: virial of a user CustomBondForce is not supported')
