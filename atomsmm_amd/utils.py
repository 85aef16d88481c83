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


def splitPotentialEnergy(system, topology, positions, **globals):
    """Potential energy split per Force object (utils.py:118-186): every force of a deep copy of the
    system gets its own group (a NonbondedForce's reciprocal space a separate one), one getState per
    group, keys 'ClassName', 'ClassName(k)', 'Real-Space', 'Reciprocal-Space', 'Total'."""
    openmm = _openmm()
    syscopy = deepcopy(system)
    forces = syscopy.getForces()
    index = 0
    for force in forces:
        force.setForceGroup(index)
        index += 1
        if isinstance(force, openmm.NonbondedForce):
            force.setReciprocalSpaceForceGroup(index)
            index += 1
    platform = openmm.Platform.getPlatformByName('HIP')
    integrator = openmm.VerletIntegrator(0.0)
    simulation = openmm.app.Simulation(topology, syscopy, integrator, platform)
    simulation.context.setPositions(positions)
    for parameter, value in globals.items():
        simulation.context.setParameter(parameter, value)
    seen = dict()
    energy = dict()
    index = 0
    for force in forces:
        state = simulation.context.getState(getEnergy=True, groups=set([index]))
        # OpenMM hands back plain base-class proxies from System.getForces(), so AtomsMM subclasses are
        # reported under their OpenMM base-class name (tests/test_systems.py:63-79)
        name = force.__class__.__name__
        for base in (openmm.NonbondedForce, openmm.CustomNonbondedForce, openmm.CustomBondForce,
                     openmm.HarmonicBondForce, openmm.HarmonicAngleForce, openmm.PeriodicTorsionForce):
            if isinstance(force, base):
                name = base.__name__
        if name == 'NonbondedForce':
            name = 'Real-Space'
        new = name not in seen
        if new:
            seen[name] = 0
            energy[name] = state.getPotentialEnergy()
        else:
            seen[name] += 1
            energy['%s(%d)' % (name, seen[name])] = state.getPotentialEnergy()
        index += 1
        if isinstance(force, openmm.NonbondedForce):
            state = simulation.context.getState(getEnergy=True, groups=set([index]))
            if new:
                energy['Reciprocal-Space'] = state.getPotentialEnergy()
            else:
                energy['%s(%d)' % ('Reciprocal-Space', seen[name])] = state.getPotentialEnergy()
            index += 1
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
