"""Helpers with the names and behaviour of `atomsmm.utils` (reference: src/atomsmm/utils.py)."""
from collections import OrderedDict
from copy import deepcopy

from . import unit

kB = unit.BOLTZMANN_CONSTANT_kB * unit.AVOGADRO_CONSTANT_NA     # utils.py:16


class InputError(Exception):
    """User error; the message is wrapped in ANSI bold red exactly like the reference's (utils.py:19-21)."""

    def __init__(self, msg):
        super(InputError, self).__init__('\033[1;31m' + msg + '\033[0m')


def _openmm():
    from . import openmm
    return openmm


def countDegreesOfFreedom(system):
    """3*(particles with mass > 0) - 3 - constraints  (utils.py:24-40)."""
    n = system.getNumParticles()
    moving = sum(1 for i in range(n) if system.getParticleMass(i) / unit.dalton > 0)
    return 3 * moving - 3 - system.getNumConstraints()


def findNonbondedForce(system, position=0):
    """Index of the position-th NonbondedForce of a system (utils.py:43-62)."""
    nb = _openmm().NonbondedForce
    hits = [i for i in range(system.getNumForces()) if isinstance(system.getForce(i), nb)]
    return hits[position]


def hijackForce(system, index):
    """Deep-copy force `index`, remove it from the system and return the copy (utils.py:65-89)."""
    force = deepcopy(system.getForce(index))
    system.removeForce(index)
    return force


def globalParameters(force):
    return {force.getGlobalParameterName(i): force.getGlobalParameterDefaultValue(i)
            for i in range(force.getNumGlobalParameters())}


def _offset_parameters(force, count, getter):
    defaults = globalParameters(force)
    seen = []
    for index in range(count):
        name = getter(index)[0]
        if name not in seen:
            seen.append(name)
    return OrderedDict((name, defaults[name]) for name in seen)


def particleOffsetParameters(force):
    """Global parameters used by particle parameter offsets, with default values (utils.py:100-106).
    (First-use order; the reference iterates a Python set, whose order is arbitrary.)"""
    return _offset_parameters(force, force.getNumParticleParameterOffsets(), force.getParticleParameterOffset)


def exceptionOffsetParameters(force):
    return _offset_parameters(force, force.getNumExceptionParameterOffsets(), force.getExceptionParameterOffset)


_REPORTED_AS = ('NonbondedForce', 'CustomNonbondedForce', 'CustomBondForce', 'HarmonicBondForce', 'HarmonicAngleForce',
                'PeriodicTorsionForce')


def _energy_label(force, openmm):
    """OpenMM hands back plain base-class proxies from System.getForces(), so AtomsMM's subclasses are reported under the
    name of their OpenMM base class (tests/test_systems.py:63-79); the direct space of a NonbondedForce is 'Real-Space'."""
    label = type(force).__name__
    for base_name in _REPORTED_AS:
        if isinstance(force, getattr(openmm, base_name)):
            label = base_name
    return 'Real-Space' if label == 'NonbondedForce' else label


def splitPotentialEnergy(system, topology, positions, **globals):
    """Potential energy per Force object (interface of utils.py:118-186): in a deep copy of the system every force gets a
    force group of its own -- the reciprocal space of a NonbondedForce one more -- and each group is evaluated by itself.
    Keys: 'ClassName' for the first force of a kind and 'ClassName(k)' for the k-th further one, 'Real-Space' /
    'Reciprocal-Space' for a NonbondedForce, and 'Total'.  Keyword arguments set global Context parameters first."""
    openmm = _openmm()
    work = deepcopy(system)
    plan = []                                   # (label, group) in evaluation order
    repeats = {}
    next_group = 0
    for force in work.getForces():
        label = _energy_label(force, openmm)
        k = repeats[label] = repeats.get(label, -1) + 1
        suffix = '' if k == 0 else '(%d)' % k
        force.setForceGroup(next_group)
        plan.append((label + suffix, next_group))
        next_group += 1
        if isinstance(force, openmm.NonbondedForce):
            force.setReciprocalSpaceForceGroup(next_group)
            plan.append(('Reciprocal-Space' + suffix, next_group))
            next_group += 1
    simulation = openmm.app.Simulation(topology, work, openmm.VerletIntegrator(0.0), openmm.Platform.getPlatformByName('HIP'))
    simulation.context.setPositions(positions)
    for name, value in globals.items():
        simulation.context.setParameter(name, value)
    energy = {label: simulation.context.getState(getEnergy=True, groups={group}).getPotentialEnergy() for label, group in plan}
    energy['Total'] = sum(energy.values(), 0.0 * unit.kilojoules_per_mole)
    return energy


def evaluateForce(force, positions, boxVectors=None):
    """Energy of one Force object for given coordinates (utils.py:189-228)."""
    openmm = _openmm()
    system = openmm.System()
    for _ in range(len(positions)):
        system.addParticle(0)
    if boxVectors is not None:
        system.setDefaultPeriodicBoxVectors(*boxVectors)
    system.addForce(deepcopy(force))
    context = openmm.Context(system, openmm.CustomIntegrator(0), openmm.Platform.getPlatformByName('HIP'))
    context.setPositions(positions)
    return context.getState(getEnergy=True).getPotentialEnergy()
