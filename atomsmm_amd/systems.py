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
and the special bond/angle redefinitions (systems.py:121-237) are outside this round's scope.
"""
import copy

from . import forces, openmm
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
