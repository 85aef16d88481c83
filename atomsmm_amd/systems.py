"""`RESPASystem` with the constructor signature of `atomsmm.systems.RESPASystem`
(reference: src/atomsmm/systems.py:34-119).

Given a System, it re-groups the nonbonded interactions for RESPA integration:

    group 0   everything bonded + (fastExceptions) the NonbondedForce exceptions as a CustomBondForce
    group 1   near pair force  (CustomNonbondedForce, cutoff rcutIn, `adjustment` = 'force-switch' by default)
    group 2   the original NonbondedForce, direct + reciprocal space
    group 31  -step(rc0-r)*(near)   bookkeeping copy so that all groups still sum to the original energy

The integrator forms the slow force as f2 - f1 (propagators.py:917-919).  Each custom force built
here also carries the structured descriptor the HIP engine consumes (see atomsmm_amd.forces).
RESPA splitting of AlchemicalSystem's coupling force (CustomNonbondedForces named U_linear/U_spline/U_art/U_general,
systems.py:83-95) is not built: the constructor refuses such inputs.
"""
import copy
import itertools
import math
import re

from . import forces, openmm, unit, utils
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
        # the solute's own parameters become offsets scaled by lambda_coul / lambda_vdw (systems.py:289-312); the particle itself
        # keeps (0, 0, 0).  Offsets are added charge-first, in atom order: the order OpenMM evaluates them in is contract.
        scaled = {'lambda_coul': [], 'lambda_vdw': []}
        for atom in sorted(solute_atoms):
            q, sig, eps = nonbonded.getParticleParameters(atom)
            nonbonded.setParticleParameters(atom, 0.0, 0.0, 0.0)
            if md_value(q) != 0.0:
                scaled['lambda_coul'].append((atom, q, 0.0, 0.0))
            if md_value(eps) != 0.0 and not use_softcore:
                scaled['lambda_vdw'].append((atom, 0.0, sig, eps))
        for parameter in ('lambda_coul', 'lambda_vdw'):
            if scaled[parameter]:
                nonbonded.addGlobalParameter(parameter, 1.0)
                for atom, q, sig, eps in scaled[parameter]:
                    nonbonded.addParticleParameterOffset(parameter, atom, q, sig, eps)


class AlchemicalSystem(openmm.System):
    """`atomsmm.systems.AlchemicalSystem(system, atoms, coupling='softcore', group=0, use_lrc=False)` (reference:
    src/atomsmm/systems.py:318-410): a System prepared for solvation free-energy calculations with one coupling
    parameter, `lambda_vdw`.

    * the Lennard-Jones interactions between the solute `atoms` and everything else move to a CustomNonbondedForce over
      that interaction group (force group `group`, per-particle parameters `sigma` and `epsilon`, lambda_vdw registered as
      an energy parameter derivative).  `coupling` selects its energy: `softcore` (Beutler et al. 1994), or
      Lennard-Jones times ((gt0-gt1)*S + gt1) with S(lambda_vdw) = `linear` / `art` (Abrams, Rosso & Tuckerman 2006) /
      `spline` / any other text, taken as the function itself.  The emitted strings are the reference's, including its
      `linear` variant that names `two_pi` without defining it (a Context then refuses it, as OpenMM does);
    * the solute atoms lose charge and epsilon in the NonbondedForce (charge 0, sigma 1, epsilon 0);
    * every solute-solute pair that is not an exception yet becomes one, with the combined original parameters, and is
      excluded from the coupling force.
    """

    _COUPLINGS = {'linear': 'lambda_vdw - sin(two_pi*lambda_vdw)/two_pi',
                  'spline': 'lambda_vdw^3*(10 - 15*lambda_vdw + 6*lambda_vdw^2)',
                  'art': 'lambda_vdw - sin(two_pi*lambda_vdw)/two_pi; two_pi = 6.28318530717958'}

    @classmethod
    def _energy_text(cls, coupling):
        mixing = '; sigma = 0.5*(sigma1 + sigma2); epsilon = sqrt(epsilon1*epsilon2)'
        if coupling == 'softcore':
            return ('U_softcore; U_softcore = 4*lambda_vdw*epsilon*(1 - x)/x^2'
                    '; x = (r/sigma)^6 + 0.5*(1 - lambda_vdw)') + mixing
        label = 'U_{}'.format(coupling) if coupling in cls._COUPLINGS else 'U_general'
        return ('{0}; {0} = 4*((gt0-gt1)*S + gt1)*epsilon*x*(x - 1); x = (sigma/r)^6; gt0 = step(lambda_vdw)'
                '; gt1 = step(lambda_vdw-1); S = {1}').format(label, cls._COUPLINGS.get(coupling, coupling)) + mixing

    def __init__(self, system, atoms, coupling='softcore', group=0, use_lrc=False):
        openmm.System.__init__(self)
        self._copy_from(system)
        nonbonded = self.getForce(utils.findNonbondedForce(self))
        solute = sorted(set(int(i) for i in atoms))
        everyone = range(nonbonded.getNumParticles())
        pair_force = openmm.CustomNonbondedForce(self._energy_text(coupling))
        free_space = nonbonded.getNonbondedMethod() == openmm.NonbondedForce.NoCutoff
        pair_force.setNonbondedMethod(pair_force.NoCutoff if free_space else pair_force.CutoffPeriodic)
        pair_force.setCutoffDistance(nonbonded.getCutoffDistance())
        pair_force.setUseSwitchingFunction(nonbonded.getUseSwitchingFunction())
        pair_force.setSwitchingDistance(nonbonded.getSwitchingDistance())
        pair_force.setUseLongRangeCorrection(use_lrc)
        pair_force.addGlobalParameter('lambda_vdw', 1.0)
        for name in ('sigma', 'epsilon'):
            pair_force.addPerParticleParameter(name)
        original = {}
        for index in everyone:
            charge, sigma, epsilon = nonbonded.getParticleParameters(index)
            pair_force.addParticle([sigma, epsilon])
            original[index] = (charge, sigma, epsilon)
        known = set()
        for index in range(nonbonded.getNumExceptions()):
            i, j = nonbonded.getExceptionParameters(index)[:2]
            pair_force.addExclusion(i, j)
            known.add(frozenset((i, j)))
        pair_force.addInteractionGroup(set(solute), set(everyone) - set(solute))
        pair_force.setForceGroup(group)
        pair_force.addEnergyParameterDerivative('lambda_vdw')
        self.addForce(pair_force)
        for index in solute:
            nonbonded.setParticleParameters(index, 0.0, 1.0, 0.0)
        for i, j in itertools.combinations(solute, 2):
            if frozenset((i, j)) in known:
                continue
            (q1, sig1, eps1), (q2, sig2, eps2) = original[i], original[j]
            nonbonded.addException(i, j, q1 * q2, (sig1 + sig2) / 2, (eps1 * eps2).sqrt())
            pair_force.addExclusion(i, j)          # both forces keep the same number of excluded pairs


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
        # force groups: 0 dispersion virial, 1 bond-stretching virial, 2 Coulomb (whose virial is its energy)
        group = dict(dispersion=0, bonded=1, coulomb=2)
        self._dispersion, self._bonded, self._coulomb = (1 << group[k] for k in ('dispersion', 'bonded', 'coulomb'))
        lj_virial = '24*epsilon*(2*(sigma/r)^12-(sigma/r)^6)'

        def lj_virial_forces(source):
            """W_LJ of the pair interactions and of the exceptions, as energies of two custom forces (systems.py:894-905)."""
            pairs = forces._AtomsMM_CustomNonbondedForce(lj_virial).importFrom(source)
            bonds = forces._AtomsMM_CustomBondForce(lj_virial).importFrom(source, extract=False)
            return [pairs] + ([bonds] if bonds.getNumBonds() > 0 else [])

        def coulomb_only(source):
            """The source force with every Lennard-Jones interaction switched off (sigma 1, epsilon 0)."""
            for k in range(source.getNumParticles()):
                source.setParticleParameters(k, source.getParticleParameters(k)[0], 1.0, 0.0)
            for k in range(source.getNumExceptions()):
                a, b, qq = source.getExceptionParameters(k)[:3]
                source.setExceptionParameters(k, a, b, qq, 1.0, 0.0)
            source.setReciprocalSpaceForceGroup(group['coulomb'])
            return source

        def stretching_virial(harmonic):
            """-r dE/dr = -K r (r - r0) of every harmonic bond (systems.py:914)."""
            out = openmm.CustomBondForce('-K*r*(r-r0)')
            for name in ('r0', 'K'):
                out.addPerBondParameter(name)
            for k in range(harmonic.getNumBonds()):
                a, b, r0, K = harmonic.getBondParameters(k)
                out.addBond(a, b, [r0, K])
            return out

        for force in system.getForces():
            if isinstance(force, openmm.NonbondedForce) and force.getNumParticles() > 0:
                source = copy.deepcopy(force)
                members = [(f, 'dispersion') for f in lj_virial_forces(source)] + [(coulomb_only(source), 'coulomb')]
            elif isinstance(force, openmm.HarmonicBondForce) and force.getNumBonds() > 0:
                members = [(stretching_virial(force), 'bonded')]
            elif isinstance(force, openmm.CustomBondForce) and force.getNumBonds() > 0:
                raise NotImplementedError('ComputingSystem: virial of a user CustomBondForce is not supported')
            else:
                members = []
            for member, kind in members:
                member.setForceGroup(group[kind])
                self.addForce(member)


class AlchemicalCoulombCVForce(object):
    """`atomsmm.systems.AlchemicalCoulombCVForce(alchemical_system)` (systems.py:471-489): the solute-solvent Coulomb
    energy as the difference of the outer group's energy at lambda_coul = 1 and 0."""

    def __init__(self, alchemical_system):
        self._system = alchemical_system

    def getNumCollectiveVariables(self):
        return 1

    def getCollectiveVariableName(self, index):
        return 'alchemical_coulomb_energy'

    def getCollectiveVariableValues(self, context):
        lambda_coul = self._system._coulomb_factor
        group = 2 if self._system._middle_scale else 1
        self._system.reset_coulomb_scaling_factor(0.0, context)
        E0 = context.getState(getEnergy=True, groups=2 ** group).getPotentialEnergy()
        self._system.reset_coulomb_scaling_factor(1.0, context)
        E1 = context.getState(getEnergy=True, groups=2 ** group).getPotentialEnergy()
        self._system.reset_coulomb_scaling_factor(lambda_coul, context)
        return [(E1 - E0).value_in_unit(unit.kilojoules_per_mole)]


class AlchemicalRespaSystem(openmm.System):
    """`atomsmm.systems.AlchemicalRespaSystem(system, rcutIn, rswitchIn, alchemical_atoms, coupling_parameter='lambda',
    coupling_function='lambda', middle_scale=True, coulomb_scaling=False, lambda_coul=0, use_softcore=False,
    split_alchemical=True)` (reference: src/atomsmm/systems.py:492-783): RESPA splitting plus alchemical coupling of
    the solute-solvent van der Waals interactions.

    * the NonbondedForce keeps the solvent-solvent interactions only (solute parameters (0, 1, 0), every solute-solute
      pair an exclusion), group 2 (1 without a middle scale);
    * with a middle scale, group 1 gets a force-switched short-ranged copy, `respa_switch*(V_LJC + step(r-rs)*(...))`
      -- no constant shift -- as CustomNonbondedForce, and the non-zero exceptions as `step(rc-r)*U` bonds;
    * the solute-solute interactions become cutoff-less LJC bonds (group 2) plus their short-ranged copy (group 1);
    * the solute-solvent Lennard-Jones energy is a collective variable multiplied by ((gt0-gt1)*S(lambda)+gt1) in a
      CustomCVForce (group 2; its force-switched short-ranged copy in group 1), or a softcore force.
    * with `coulomb_scaling`, the solute charges come back into the NonbondedForce scaled by `lambda_coul` (solute-solute
      pairs stay excluded) and, with a middle scale, a force-switched electrostatic potential over the (solute, solvent)
      interaction group joins group 1 (systems.py:686-708, 794-815, 848-856)."""

    Kc = 138.935456637          # systems.py:572 (the other classes use 138.935456)

    _MIXING = '; chargeprod = charge1*charge2; sigma = 0.5*(sigma1 + sigma2); epsilon = sqrt(epsilon1*epsilon2)'
    _PAIR_PARAMETERS = ('chargeprod', 'sigma', 'epsilon')

    def __init__(self, system, rcutIn, rswitchIn, alchemical_atoms=[], coupling_parameter='lambda',
                 coupling_function='lambda', middle_scale=True, coulomb_scaling=False, lambda_coul=0,
                 use_softcore=False, split_alchemical=True):
        openmm.System.__init__(self)
        self._copy_from(system)
        self._parameter, self._middle_scale, self._use_softcore = coupling_parameter, middle_scale, use_softcore
        self._coulomb_scaling = coulomb_scaling
        self._original_solute_charge = {}
        self._coulomb_factor = 0
        self._inner_cutoff = rcutIn
        solute = set(int(i) for i in alchemical_atoms)
        solvent = set(range(self.getNumParticles())) - solute
        rc, rs = md_value(rcutIn), md_value(rswitchIn)
        outer = 2 if middle_scale else 1
        switched = self._force_switched_potential(rc, rs, self.Kc)
        guarded = 'step({}-r)*U; U = {}'.format(rc, switched)

        # 1. the NonbondedForce keeps the solvent; everything else goes to group 0
        original = None
        for force in self.getForces():
            if not isinstance(force, openmm.NonbondedForce):
                force.setForceGroup(0)
                continue
            original = copy.deepcopy(force)
            self._strip_solute(force, original, solute, outer)
            if middle_scale:
                self.addForce(self._switched_pair_force(switched, force, group=1))
                self._add_if_any(self._pair_bonds(guarded, 1, ((i, j, q, s_, e) for i, j, q, s_, e in self._exceptions(force)
                                                                    if md_value(q) != 0.0 or md_value(e) != 0.0)))
            self._nonbonded_force = force
        if not solute or original is None:
            return

        # 2. solute-solute pairs: cutoff-less LJC bonds, and their short-ranged copy
        inside = [(i, j, q, s_, e) for i, j, q, s_, e in self._exceptions(original) if i in solute and j in solute]
        ljc = '4*epsilon*x*(x - 1) + {}*chargeprod/r; x = (sigma/r)^6'.format(self.Kc)
        self.addForce(self._pair_bonds(ljc, outer, inside, switchable=False))
        if middle_scale:
            self.addForce(self._pair_bonds(guarded, 1, inside))

        # 3. short-ranged solute-solvent electrostatics (particles keep the ORIGINAL charges until
        #    reset_coulomb_scaling_factor rescales the solute's, systems.py:698-708)
        if coulomb_scaling and middle_scale:
            text = self._force_switched_eletrostatic_potential(rc, rs, self.Kc)
            self._switched_coulomb_force = self._switched_pair_force(text, original, group=1, sets=(solute, solvent))
            self.addForce(self._switched_coulomb_force)

        # 4. solute-solvent Lennard-Jones: softcore, or a collective variable times the coupling function
        if use_softcore:
            ljsoft = '4*{0}*epsilon*x*(x - 1); x = 1/((r/sigma)^6 + 0.5*(1-{0}))'.format(coupling_parameter)
            softcore = self._outer_pair_force(ljsoft, original, (solute, solvent))
            softcore.addGlobalParameter(coupling_parameter, 1.0)
            softcore.addEnergyParameterDerivative(coupling_parameter)
            softcore.setForceGroup(outer)
            self.addForce(softcore)
            self._alchemical_vdw_force = softcore
            if middle_scale:
                self.addForce(self._switched_copy(softcore, ljsoft))
        else:
            coupling = ('((gt0-gt1)*S + gt1)*alchemical_vdw_energy; gt0 = step({0}); gt1 = step({0}-1); S = {1}'
                        .format(coupling_parameter, coupling_function))
            lj = '4*epsilon*x*(x - 1); x = (sigma/r)^6'
            energy = self._outer_pair_force(lj, original, (solute, solvent))
            self._alchemical_vdw_force = self._coupled(coupling, coupling_parameter, energy, outer)
            self.addForce(self._alchemical_vdw_force)
            if middle_scale and split_alchemical:
                near_energy = self._switched_pair_force(self._force_switched_potential(rc, rs, 0.0), original, group=None,
                                                        sets=(solute, solvent))
                self.addForce(self._coupled(coupling, coupling_parameter, near_energy, 1))
            elif middle_scale:
                self.addForce(self._switched_copy(energy, lj))

        # stored as zero and reset only if a different value was passed (systems.py:781-783): with the default the
        # force-switched electrostatic force keeps the solute's full charges, as in the reference
        self._coulomb_factor = 0
        self.reset_coulomb_scaling_factor(lambda_coul)

    # ---- pieces of the constructor -------------------------------------------------------------------------------
    @staticmethod
    def _exceptions(force):
        return [force.getExceptionParameters(k) for k in range(force.getNumExceptions())]

    def _strip_solute(self, force, original, solute, group):
        """The solute leaves the NonbondedForce: parameters (0, 1, 0), every solute-solute pair an exclusion; `original`
        (a copy made before) receives the missing solute-solute pairs as exceptions with the combined parameters."""
        force.setForceGroup(group)
        force.setReciprocalSpaceForceGroup(group)
        for i in solute:
            self._original_solute_charge[i] = force.getParticleParameters(i)[0]
            force.setParticleParameters(i, 0.0, 1.0, 0.0)
        listed = set()
        for index, (i, j) in enumerate(e[:2] for e in self._exceptions(original)):
            if i in solute and j in solute:
                listed.add(frozenset((i, j)))
                force.setExceptionParameters(index, i, j, 0.0, 1.0, 0.0)
        for i, j in itertools.combinations(sorted(solute), 2):
            if frozenset((i, j)) in listed:
                continue
            force.addException(i, j, 0.0, 1.0, 0.0)
            (q1, sig1, eps1), (q2, sig2, eps2) = original.getParticleParameters(i), original.getParticleParameters(j)
            original.addException(i, j, q1 * q2, (sig1 + sig2) / 2, (eps1 * eps2).sqrt())

    def _switched_pair_force(self, expression, source, group, sets=None):
        """CustomNonbondedForce of the middle time scale: cutoff rcutIn, no built-in switch or correction, gated by the
        global parameter respa_switch; optionally restricted to an interaction group."""
        force = openmm.CustomNonbondedForce(expression + self._MIXING)
        self._import_from_nonbonded(force, source)
        force.setCutoffDistance(self._inner_cutoff)
        force.setUseSwitchingFunction(False)
        force.setUseLongRangeCorrection(False)
        force.addGlobalParameter('respa_switch', 0)
        if group is not None:
            force.setForceGroup(group)
        if sets is not None:
            force.addInteractionGroup(*sets)
        return force

    def _outer_pair_force(self, expression, source, sets):
        """CustomNonbondedForce over the (solute, solvent) group with the cutoff, switch and correction of the NonbondedForce."""
        force = openmm.CustomNonbondedForce(expression + self._MIXING)
        self._import_from_nonbonded(force, source, import_globals=True)
        force.addInteractionGroup(*sets)
        return force

    def _switched_copy(self, force, expression):
        """Copy of an outer pair force for group 1: the same particles and settings, its energy times respa_switch."""
        twin = copy.deepcopy(force)
        twin.setEnergyFunction('respa_switch*{}'.format(expression) + self._MIXING)
        twin.addGlobalParameter('respa_switch', 0)
        twin.setForceGroup(1)
        return twin

    def _pair_bonds(self, expression, group, pairs, switchable=True):
        """CustomBondForce with (chargeprod, sigma, epsilon) per bond over `pairs` = (i, j, chargeprod, sigma, epsilon)."""
        force = openmm.CustomBondForce(expression)
        if switchable:
            force.addGlobalParameter('respa_switch', 0)
        for name in self._PAIR_PARAMETERS:
            force.addPerBondParameter(name)
        for i, j, chargeprod, sigma, epsilon in pairs:
            force.addBond(i, j, (chargeprod, sigma, epsilon))
        force.setForceGroup(group)
        return force

    def _add_if_any(self, bond_force):
        if bond_force.getNumBonds() > 0:
            self.addForce(bond_force)

    @staticmethod
    def _coupled(text, parameter, energy_force, group):
        """CustomCVForce `coupling(parameter) * alchemical_vdw_energy` over one pair force."""
        force = openmm.CustomCVForce(text)
        force.addGlobalParameter(parameter, 1.0)
        force.addEnergyParameterDerivative(parameter)
        force.addCollectiveVariable('alchemical_vdw_energy', energy_force)
        force.setForceGroup(group)
        return force

    def get_alchemical_vdw_force(self, parameter_values=[1]):
        if self._use_softcore:
            return AlchemicalSoftcoreCVForce(self, parameter_values)
        return self._alchemical_vdw_force

    def get_alchemical_coul_force(self):
        return AlchemicalCoulombCVForce(self)

    def reset_coulomb_scaling_factor(self, lambda_coul, context=None):
        """Scaling factor of the solute-solvent electrostatics (interface of systems.py:794-815).  Every force that carries the
        solute's charges gets them again as factor x original charge -- the NonbondedForce always, the force-switched
        solute-solvent force of the middle time scale when there is one -- and, given a Context, uploads them."""
        factor = md_value(lambda_coul)
        if not self._coulomb_scaling or factor == self._coulomb_factor:
            return
        carriers = [self._nonbonded_force]
        if self._middle_scale:
            carriers.append(self._switched_coulomb_force)
        for force in carriers:
            packed = force is not self._nonbonded_force          # a CustomNonbondedForce takes its parameters as one sequence
            for index, original in self._original_solute_charge.items():
                values = (factor * original, 1.0, 0.0)
                if packed:
                    force.setParticleParameters(index, values)
                else:
                    force.setParticleParameters(index, *values)
            if context is not None:
                force.updateParametersInContext(context)
        self._coulomb_factor = factor

    @staticmethod
    def _force_switched_eletrostatic_potential(rc, rs, Kc):
        """Expression text of systems.py:848-856: Kc*chargeprod/r times (1 + step(r-rs)*f1), times respa_switch."""
        b = rs / (rc - rs)
        a1 = 5 * (b + 1) ** 2
        f1 = '{}*({}*R*log(R)-{}*u-{}*u^2+u^3)-{}*u^4+{}*u^5'.format(a1, 6 * b ** 3, 6 * b ** 2, 3 * b, 5 * (b / 2 + 1), 3 / 2)
        fsep = 'respa_switch*(1 + step(r-{})*f1)*{}*chargeprod/r'.format(rs, Kc)
        fsep += '; f1 = {}'.format(f1)
        fsep += '; R = {}*u + 1'.format(1 / b)
        fsep += '; u = {}*r - {}'.format(b / rs, b)
        return fsep

    @staticmethod
    def _force_switched_potential(rc, rs, Kc):
        """The expression text of systems.py:823-846: V_LJC plus, beyond rs, the force-switch perturbation, times the
        global parameter respa_switch (no constant shift)."""
        b = rs / (rc - rs)
        a12 = (6 * b ** 2 - 21 * b + 28) / 462
        a6 = 6 * b ** 2 - 3 * b + 1
        f = {}
        f[12] = '{}*({}*(R^12-1)-{}*u-{}*u^2-220*u^3)+({})*u^4-{}*u^5'.format(a12, b ** 3, 12 * b ** 2, 66 * b, 45 * (7 - 2 * b) / 14, 72 / 7)
        f[6] = '{}*({}*(R^6-1)-{}*u-{}*u^2-20*u^3)+({})*u^4-36*u^5'.format(a6, b ** 3, 6 * b ** 2, 15 * b, 45 * (1 - 2 * b))
        if Kc == 0.0:
            fsp = 'respa_switch*(4*epsilon*x*(x-1) + step(r-{})*perturbation)'.format(rs)
            fsp += '; perturbation = 4*epsilon*x*(f12*x-f6)'
        else:
            a1 = 5 * (b + 1) ** 2
            f[1] = '{}*({}*R*log(R)-{}*u-{}*u^2+u^3)-{}*u^4+{}*u^5'.format(a1, 6 * b ** 3, 6 * b ** 2, 3 * b, 5 * (b / 2 + 1), 3 / 2)
            fsp = 'respa_switch*(4*epsilon*x*(x-1) + {}*chargeprod/r + step(r-{})*perturbation)'.format(Kc, rs)
            fsp += '; perturbation = 4*epsilon*x*(f12*x-f6) + {}*f1*chargeprod/r'.format(Kc)
        fsp += '; x = (sigma/r)^6'
        for variable, expression in f.items():
            fsp += '; f{} = {}'.format(variable, expression)
        fsp += '; R = {}*u + 1'.format(1 / b)
        fsp += '; u = {}*r - {}'.format(b / rs, b)
        return fsp

    @staticmethod
    def _import_from_nonbonded(force, nonbonded, import_globals=False):
        """systems.py:856-875: method, particles, exceptions -> exclusions; optionally cutoff / switch / correction."""
        if nonbonded.getNonbondedMethod() == openmm.NonbondedForce.NoCutoff:
            force.setNonbondedMethod(openmm.CustomNonbondedForce.NoCutoff)
        else:
            force.setNonbondedMethod(openmm.CustomNonbondedForce.CutoffPeriodic)
        for parameter in ['charge', 'sigma', 'epsilon']:
            force.addPerParticleParameter(parameter)
        for i in range(nonbonded.getNumParticles()):
            force.addParticle(nonbonded.getParticleParameters(i))
        for index in range(nonbonded.getNumExceptions()):
            i, j = nonbonded.getExceptionParameters(index)[:2]
            force.addExclusion(i, j)
        if import_globals:
            force.setCutoffDistance(nonbonded.getCutoffDistance())
            force.setUseSwitchingFunction(nonbonded.getUseSwitchingFunction())
            force.setSwitchingDistance(nonbonded.getSwitchingDistance())
            force.setUseLongRangeCorrection(nonbonded.getUseDispersionCorrection())


class AlchemicalSoftcoreCVForce(object):
    """Collective variables E0, E1, ... for reweighting: the softcore solute-solvent energy at every value of a grid of
    coupling parameters (interface of systems.py:412-470).  One private System holds a clone of the alchemical softcore
    force per grid point, with the coupling parameter frozen into its expression, each clone in the force group that
    carries its index; a private Context on the caller's platform evaluates them group by group."""

    _EXPRESSION = ('4*lambda*epsilon*x*(x - 1); x = 1/((r/sigma)^6 + 0.5*(1-lambda)); lambda = {}; '
                   'sigma = 0.5*(sigma1 + sigma2); epsilon = sqrt(epsilon1*epsilon2)')

    def __init__(self, alchemical_system, grid):
        template = alchemical_system._alchemical_vdw_force
        self._grid = list(grid)
        self._context = None
        self._system = openmm.System()
        for mass in (alchemical_system.getParticleMass(k) for k in range(alchemical_system.getNumParticles())):
            self._system.addParticle(mass)
        self._system.setDefaultPeriodicBoxVectors(*alchemical_system.getDefaultPeriodicBoxVectors())
        lj = [template.getParticleParameters(k)[1:] for k in range(template.getNumParticles())]
        exclusions = [template.getExclusionParticles(k) for k in range(template.getNumExclusions())]
        groups = [template.getInteractionGroupParameters(k) for k in range(template.getNumInteractionGroups())]
        for group, coupling in enumerate(self._grid):
            self._system.addForce(self._clone(template, coupling, group, lj, exclusions, groups))

    def _clone(self, template, coupling, group, lj, exclusions, interaction_groups):
        force = openmm.CustomNonbondedForce(self._EXPRESSION.format(coupling))
        force.addPerParticleParameter('sigma')
        force.addPerParticleParameter('epsilon')
        for sigma_epsilon in lj:
            force.addParticle(tuple(sigma_epsilon))
        for pair in exclusions:
            force.addExclusion(*pair)
        for sets in interaction_groups:
            force.addInteractionGroup(*sets)
        force.setNonbondedMethod(template.getNonbondedMethod())
        force.setCutoffDistance(template.getCutoffDistance())
        force.setUseSwitchingFunction(template.getUseSwitchingFunction())
        force.setSwitchingDistance(template.getSwitchingDistance())
        # a fully decoupled solute has no dispersion tail to correct
        force.setUseLongRangeCorrection(template.getUseLongRangeCorrection() and coupling != 0.0)
        force.setForceGroup(group)
        return force

    def getNumCollectiveVariables(self):
        return len(self._grid)

    def getCollectiveVariableName(self, index):
        return 'E%d' % index

    def getCollectiveVariableValues(self, context):
        import numpy as np
        if self._context is None:
            self._context = openmm.Context(self._system, openmm.CustomIntegrator(0), context.getPlatform())
        self._context.setState(context.getState(getPositions=True))
        return np.array([md_value(self._context.getState(getEnergy=True, groups={g}).getPotentialEnergy())
                         for g in range(len(self._grid))])
