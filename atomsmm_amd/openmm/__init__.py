"""Minimal OpenMM-shaped object model for the AtomsMM hot path (host side, pure Python).

AtomsMM's classes subclass OpenMM's SWIG classes (forces.py:134,193,326; systems.py:34;
integrators.py:26).  OpenMM is absent here, and north_star forbids building an OpenMM platform, so
this module supplies just the container/accessor API that AtomsMM and its tests call
(`addParticle`, `getExceptionParameters`, `addComputePerDof`, `getState(groups=...)`, ...), with
the same method names and argument meaning.  It holds *descriptions only*; all arithmetic happens in
the HIP library through `atomsmm_amd.engine` when a `Context` is created.

Usage in a script written for the reference:

    from atomsmm_amd import openmm, unit          # instead of: from simtk import openmm, unit
    from atomsmm_amd.openmm import app
    import atomsmm_amd as atomsmm
"""
import copy
import math

import numpy as np

from .. import unit as _unit
from ..unit import Quantity, md_value

nm = _unit.nanometer
kjmol = _unit.kilojoule_per_mole


class OpenMMException(Exception):
    pass


class Vec3(tuple):
    def __new__(cls, x, y, z):
        return tuple.__new__(cls, (x, y, z))

    x = property(lambda s: s[0])
    y = property(lambda s: s[1])
    z = property(lambda s: s[2])

    def __add__(self, o):
        return Vec3(self[0] + o[0], self[1] + o[1], self[2] + o[2])

    def __sub__(self, o):
        return Vec3(self[0] - o[0], self[1] - o[1], self[2] - o[2])

    def __mul__(self, s):
        if isinstance(s, (_unit.Unit, Quantity)):
            return Quantity(self, s) if isinstance(s, _unit.Unit) else NotImplemented
        return Vec3(self[0] * s, self[1] * s, self[2] * s)

    __rmul__ = __mul__


# ------------------------------------------------------------------------------------------------ forces
class Force:
    def __init__(self):
        self._group = 0
        self._name = self.__class__.__name__

    def getForceGroup(self):
        return self._group

    def setForceGroup(self, group):
        if not 0 <= int(group) <= 31:
            raise OpenMMException('Force group must be between 0 and 31')
        self._group = int(group)

    def usesPeriodicBoundaryConditions(self):
        return False

    def getName(self):
        return self._name

    def setName(self, name):
        self._name = name


class _GlobalParams:
    """Mixin: global parameters with default values."""

    def _init_globals(self):
        self._gnames = []
        self._gvalues = []

    def addGlobalParameter(self, name, defaultValue):
        self._gnames.append(name)
        self._gvalues.append(float(md_value(defaultValue)))
        return len(self._gnames) - 1

    def getNumGlobalParameters(self):
        return len(self._gnames)

    def getGlobalParameterName(self, index):
        return self._gnames[index]

    def getGlobalParameterDefaultValue(self, index):
        return self._gvalues[index]

    def setGlobalParameterDefaultValue(self, index, value):
        self._gvalues[index] = float(md_value(value))


class NonbondedForce(Force, _GlobalParams):
    NoCutoff, CutoffNonPeriodic, CutoffPeriodic, Ewald, PME, LJPME = range(6)

    def __init__(self):
        Force.__init__(self)
        self._init_globals()
        self._particles = []       # [q, sigma, eps]
        self._exceptions = []      # [i, j, qq, sigma, eps]
        self._exc_index = {}
        self._method = self.NoCutoff
        self._cutoff = 1.0
        self._use_switch = False
        self._switch = -1.0
        self._ewald_tol = 5e-4
        self._pme = (0.0, 0, 0, 0)
        self._dispersion = True
        self._recip_group = -1
        self._rf_dielectric = 78.3
        self._particle_offsets = []    # [param, particle, qs, ss, es]
        self._exception_offsets = []   # [param, exception, qqs, ss, es]

    def usesPeriodicBoundaryConditions(self):
        return self._method in (self.CutoffPeriodic, self.Ewald, self.PME, self.LJPME)

    def getNumParticles(self):
        return len(self._particles)

    def addParticle(self, charge, sigma, epsilon):
        self._particles.append([float(md_value(charge)), float(md_value(sigma)), float(md_value(epsilon))])
        return len(self._particles) - 1

    def getParticleParameters(self, index):
        q, s, e = self._particles[index]
        return [Quantity(q, _unit.elementary_charge), Quantity(s, nm), Quantity(e, kjmol)]

    def setParticleParameters(self, index, charge, sigma, epsilon):
        self._particles[index] = [float(md_value(charge)), float(md_value(sigma)), float(md_value(epsilon))]

    def updateParametersInContext(self, context):
        """Upload the particle and exception parameters again (systems.py:811-812)."""
        context._engine.update_force_parameters(self)

    def getNumExceptions(self):
        return len(self._exceptions)

    def addException(self, particle1, particle2, chargeProd, sigma, epsilon, replace=False):
        key = (min(particle1, particle2), max(particle1, particle2))
        rec = [int(particle1), int(particle2), float(md_value(chargeProd)), float(md_value(sigma)), float(md_value(epsilon))]
        if key in self._exc_index:
            if not replace:
                raise OpenMMException('NonbondedForce: There is already an exception for particles %d and %d' % key)
            self._exceptions[self._exc_index[key]] = rec
            return self._exc_index[key]
        self._exceptions.append(rec)
        self._exc_index[key] = len(self._exceptions) - 1
        return len(self._exceptions) - 1

    def getExceptionParameters(self, index):
        i, j, qq, s, e = self._exceptions[index]
        return [i, j, Quantity(qq, _unit.elementary_charge ** 2), Quantity(s, nm), Quantity(e, kjmol)]

    def setExceptionParameters(self, index, particle1, particle2, chargeProd, sigma, epsilon):
        self._exceptions[index] = [int(particle1), int(particle2), float(md_value(chargeProd)), float(md_value(sigma)),
                                   float(md_value(epsilon))]

    def createExceptionsFromBonds(self, bonds, coulomb14Scale, lj14Scale):
        n = self.getNumParticles()
        nbrs = [set() for _ in range(n)]
        for i, j in bonds:
            nbrs[i].add(j); nbrs[j].add(i)
        excl, p14 = set(), set()
        for i, j in bonds:
            excl.add((min(i, j), max(i, j)))
        for j in range(n):
            nb = sorted(nbrs[j])
            for a in range(len(nb)):
                for b in range(a + 1, len(nb)):
                    excl.add((nb[a], nb[b]))
        for j, k in bonds:
            for i in nbrs[j]:
                for l in nbrs[k]:
                    if i != k and l != j and i != l:
                        key = (min(i, l), max(i, l))
                        if key not in excl:
                            p14.add(key)
        for i, j in sorted(excl):
            self.addException(i, j, 0.0, 0.5 * (self._particles[i][1] + self._particles[j][1]), 0.0, True)
        for i, j in sorted(p14):
            qi, si, ei = self._particles[i]
            qj, sj, ej = self._particles[j]
            self.addException(i, j, coulomb14Scale * qi * qj, 0.5 * (si + sj), lj14Scale * math.sqrt(ei * ej), True)

    def getNonbondedMethod(self):
        return self._method

    def setNonbondedMethod(self, method):
        self._method = int(method)

    def getCutoffDistance(self):
        return Quantity(self._cutoff, nm)

    def setCutoffDistance(self, distance):
        self._cutoff = float(md_value(distance))

    def getUseSwitchingFunction(self):
        return self._use_switch

    def setUseSwitchingFunction(self, use):
        self._use_switch = bool(use)

    def getSwitchingDistance(self):
        return Quantity(self._switch, nm)

    def setSwitchingDistance(self, distance):
        self._switch = float(md_value(distance))

    def getEwaldErrorTolerance(self):
        return self._ewald_tol

    def setEwaldErrorTolerance(self, tol):
        self._ewald_tol = float(tol)

    def getPMEParameters(self):
        a, nx, ny, nz = self._pme
        return [Quantity(a, nm ** -1), nx, ny, nz]

    def setPMEParameters(self, alpha, nx, ny, nz):
        self._pme = (float(md_value(alpha)), int(nx), int(ny), int(nz))

    def getUseDispersionCorrection(self):
        return self._dispersion

    def setUseDispersionCorrection(self, use):
        self._dispersion = bool(use)

    def getReciprocalSpaceForceGroup(self):
        return self._recip_group

    def setReciprocalSpaceForceGroup(self, group):
        self._recip_group = int(group)

    def getReactionFieldDielectric(self):
        return self._rf_dielectric

    def setReactionFieldDielectric(self, d):
        self._rf_dielectric = float(d)

    def addParticleParameterOffset(self, parameter, particleIndex, chargeScale, sigmaScale, epsilonScale):
        self._particle_offsets.append([parameter, int(particleIndex), float(md_value(chargeScale)),
                                       float(md_value(sigmaScale)), float(md_value(epsilonScale))])
        return len(self._particle_offsets) - 1

    def getNumParticleParameterOffsets(self):
        return len(self._particle_offsets)

    def getParticleParameterOffset(self, index):
        return list(self._particle_offsets[index])

    def addExceptionParameterOffset(self, parameter, exceptionIndex, chargeProdScale, sigmaScale, epsilonScale):
        self._exception_offsets.append([parameter, int(exceptionIndex), float(md_value(chargeProdScale)),
                                        float(md_value(sigmaScale)), float(md_value(epsilonScale))])
        return len(self._exception_offsets) - 1

    def getNumExceptionParameterOffsets(self):
        return len(self._exception_offsets)

    def getExceptionParameterOffset(self, index):
        return list(self._exception_offsets[index])


class CustomNonbondedForce(Force, _GlobalParams):
    NoCutoff, CutoffNonPeriodic, CutoffPeriodic = range(3)

    def __init__(self, energy):
        Force.__init__(self)
        self._init_globals()
        self._energy = energy
        self._pnames = []
        self._particles = []
        self._exclusions = []
        self._method = self.NoCutoff
        self._cutoff = 1.0
        self._use_switch = False
        self._switch = -1.0
        self._lrc = False
        self._groups = []
        self._derivs = []

    def usesPeriodicBoundaryConditions(self):
        return self._method == self.CutoffPeriodic

    def getEnergyFunction(self):
        return self._energy

    def setEnergyFunction(self, energy):
        self._energy = energy

    def addPerParticleParameter(self, name):
        self._pnames.append(name)
        return len(self._pnames) - 1

    def getNumPerParticleParameters(self):
        return len(self._pnames)

    def getPerParticleParameterName(self, index):
        return self._pnames[index]

    def addParticle(self, parameters=()):
        self._particles.append([float(md_value(p)) for p in parameters])
        return len(self._particles) - 1

    def getNumParticles(self):
        return len(self._particles)

    def getParticleParameters(self, index):
        return tuple(self._particles[index])

    def setParticleParameters(self, index, parameters):
        self._particles[index] = [float(md_value(p)) for p in parameters]

    def updateParametersInContext(self, context):
        """Upload the per-particle parameters again (systems.py:813-814)."""
        context._engine.update_force_parameters(self)

    def addExclusion(self, particle1, particle2):
        self._exclusions.append((int(particle1), int(particle2)))
        return len(self._exclusions) - 1

    def getNumExclusions(self):
        return len(self._exclusions)

    def getExclusionParticles(self, index):
        return list(self._exclusions[index])

    def getNonbondedMethod(self):
        return self._method

    def setNonbondedMethod(self, method):
        self._method = int(method)

    def getCutoffDistance(self):
        return Quantity(self._cutoff, nm)

    def setCutoffDistance(self, distance):
        self._cutoff = float(md_value(distance))

    def getUseSwitchingFunction(self):
        return self._use_switch

    def setUseSwitchingFunction(self, use):
        self._use_switch = bool(use)

    def getSwitchingDistance(self):
        return Quantity(self._switch, nm)

    def setSwitchingDistance(self, distance):
        self._switch = float(md_value(distance))

    def getUseLongRangeCorrection(self):
        return self._lrc

    def setUseLongRangeCorrection(self, use):
        self._lrc = bool(use)

    def addInteractionGroup(self, set1, set2):
        self._groups.append((set(set1), set(set2)))
        return len(self._groups) - 1

    def getNumInteractionGroups(self):
        return len(self._groups)

    def getInteractionGroupParameters(self, index):
        return [set(self._groups[index][0]), set(self._groups[index][1])]

    def addEnergyParameterDerivative(self, name):
        self._derivs.append(name)


class CustomCVForce(Force, _GlobalParams):
    """Energy = function of collective variables, each the energy of an inner Force object."""

    def __init__(self, energy):
        Force.__init__(self)
        self._init_globals()
        self._energy = energy
        self._cvs = []
        self._derivs = []

    def getEnergyFunction(self):
        return self._energy

    def addCollectiveVariable(self, name, force):
        self._cvs.append((name, copy.deepcopy(force)))
        return len(self._cvs) - 1

    def getNumCollectiveVariables(self):
        return len(self._cvs)

    def getCollectiveVariableName(self, index):
        return self._cvs[index][0]

    def getCollectiveVariable(self, index):
        return self._cvs[index][1]

    def addEnergyParameterDerivative(self, name):
        self._derivs.append(name)


class CustomBondForce(Force, _GlobalParams):
    def __init__(self, energy):
        Force.__init__(self)
        self._init_globals()
        self._energy = energy
        self._bnames = []
        self._bonds = []
        self._periodic = False

    def getEnergyFunction(self):
        return self._energy

    def setEnergyFunction(self, energy):
        self._energy = energy

    def addPerBondParameter(self, name):
        self._bnames.append(name)
        return len(self._bnames) - 1

    def getNumPerBondParameters(self):
        return len(self._bnames)

    def getPerBondParameterName(self, index):
        return self._bnames[index]

    def addBond(self, particle1, particle2, parameters=()):
        self._bonds.append([int(particle1), int(particle2), [float(md_value(p)) for p in parameters]])
        return len(self._bonds) - 1

    def getNumBonds(self):
        return len(self._bonds)

    def getBondParameters(self, index):
        i, j, p = self._bonds[index]
        return [i, j, tuple(p)]

    def setBondParameters(self, index, particle1, particle2, parameters=()):
        self._bonds[index] = [int(particle1), int(particle2), [float(md_value(p)) for p in parameters]]

    def setUsesPeriodicBoundaryConditions(self, periodic):
        self._periodic = bool(periodic)

    def usesPeriodicBoundaryConditions(self):
        return self._periodic


class HarmonicBondForce(Force):
    def __init__(self):
        Force.__init__(self)
        self._bonds = []
        self._periodic = False

    def addBond(self, particle1, particle2, length, k):
        self._bonds.append([int(particle1), int(particle2), float(md_value(length)), float(md_value(k))])
        return len(self._bonds) - 1

    def getNumBonds(self):
        return len(self._bonds)

    def getBondParameters(self, index):
        i, j, r0, k = self._bonds[index]
        return [i, j, Quantity(r0, nm), Quantity(k, kjmol / nm ** 2)]

    def setBondParameters(self, index, particle1, particle2, length, k):
        self._bonds[index] = [int(particle1), int(particle2), float(md_value(length)), float(md_value(k))]

    def setUsesPeriodicBoundaryConditions(self, periodic):
        self._periodic = bool(periodic)

    def usesPeriodicBoundaryConditions(self):
        return self._periodic


class CustomAngleForce(Force, _GlobalParams):
    def __init__(self, energy):
        Force.__init__(self)
        self._init_globals()
        self._energy = energy
        self._anames = []
        self._angles = []
        self._periodic = False

    def getEnergyFunction(self):
        return self._energy

    def addPerAngleParameter(self, name):
        self._anames.append(name)
        return len(self._anames) - 1

    def getNumPerAngleParameters(self):
        return len(self._anames)

    def getPerAngleParameterName(self, index):
        return self._anames[index]

    def addAngle(self, particle1, particle2, particle3, parameters=()):
        self._angles.append([int(particle1), int(particle2), int(particle3), [float(md_value(p)) for p in parameters]])
        return len(self._angles) - 1

    def getNumAngles(self):
        return len(self._angles)

    def getAngleParameters(self, index):
        i, j, k, p = self._angles[index]
        return [i, j, k, tuple(p)]

    def setUsesPeriodicBoundaryConditions(self, periodic):
        self._periodic = bool(periodic)

    def usesPeriodicBoundaryConditions(self):
        return self._periodic


class HarmonicAngleForce(Force):
    def __init__(self):
        Force.__init__(self)
        self._angles = []
        self._periodic = False

    def addAngle(self, particle1, particle2, particle3, angle, k):
        self._angles.append([int(particle1), int(particle2), int(particle3), float(md_value(angle)), float(md_value(k))])
        return len(self._angles) - 1

    def getNumAngles(self):
        return len(self._angles)

    def getAngleParameters(self, index):
        i, j, k_, t0, k = self._angles[index]
        return [i, j, k_, Quantity(t0, _unit.radian), Quantity(k, kjmol / _unit.radian ** 2)]

    def setAngleParameters(self, index, particle1, particle2, particle3, angle, k):
        self._angles[index] = [int(particle1), int(particle2), int(particle3), float(md_value(angle)), float(md_value(k))]

    def setUsesPeriodicBoundaryConditions(self, periodic):
        self._periodic = bool(periodic)

    def usesPeriodicBoundaryConditions(self):
        return self._periodic


class PeriodicTorsionForce(Force):
    def __init__(self):
        Force.__init__(self)
        self._torsions = []
        self._periodic = False

    def addTorsion(self, p1, p2, p3, p4, periodicity, phase, k):
        self._torsions.append([int(p1), int(p2), int(p3), int(p4), int(periodicity), float(md_value(phase)),
                               float(md_value(k))])
        return len(self._torsions) - 1

    def getNumTorsions(self):
        return len(self._torsions)

    def getTorsionParameters(self, index):
        a, b, c, d, n, ph, k = self._torsions[index]
        return [a, b, c, d, n, Quantity(ph, _unit.radian), Quantity(k, kjmol)]

    def setUsesPeriodicBoundaryConditions(self, periodic):
        self._periodic = bool(periodic)

    def usesPeriodicBoundaryConditions(self):
        return self._periodic


class CMMotionRemover(Force):
    """Accepted and ignored (removes centre-of-mass motion in OpenMM; no energy)."""

    def __init__(self, frequency=1):
        Force.__init__(self)
        self._frequency = frequency


# ------------------------------------------------------------------------------------------------ system
class System:
    def __init__(self):
        self._masses = []
        self._forces = []
        self._box = None
        self._constraints = []

    def addParticle(self, mass):
        self._masses.append(float(md_value(mass)))
        return len(self._masses) - 1

    def getNumParticles(self):
        return len(self._masses)

    def getParticleMass(self, index):
        return Quantity(self._masses[index], _unit.dalton)

    def setParticleMass(self, index, mass):
        self._masses[index] = float(md_value(mass))

    def addForce(self, force):
        self._forces.append(force)
        return len(self._forces) - 1

    def getNumForces(self):
        return len(self._forces)

    def getForce(self, index):
        return self._forces[index]

    def getForces(self):
        return list(self._forces)

    def removeForce(self, index):
        del self._forces[index]

    def getNumConstraints(self):
        return len(self._constraints)

    def addConstraint(self, p1, p2, distance):
        self._constraints.append((int(p1), int(p2), float(md_value(distance))))
        return len(self._constraints) - 1

    def setDefaultPeriodicBoxVectors(self, a, b, c):
        vecs = []
        for v in (a, b, c):
            v = md_value(v)
            vecs.append(tuple(float(md_value(x)) for x in v))
        self._box = tuple(vecs)

    def getDefaultPeriodicBoxVectors(self):
        if self._box is None:
            raise OpenMMException('System has no periodic box')
        return [Quantity(Vec3(*v), nm) for v in self._box]

    def usesPeriodicBoundaryConditions(self):
        return any(f.usesPeriodicBoundaryConditions() for f in self._forces)

    def _copy_from(self, other):
        clone = copy.deepcopy(other)
        self.__dict__.update(clone.__dict__)


# ------------------------------------------------------------------------------------------------ integrators
class Integrator:
    def __init__(self, stepSize):
        self._dt = float(md_value(stepSize))
        self._context = None
        self._seed = 0

    def getConstraintTolerance(self):
        return getattr(self, '_ctol', 1e-5)

    def setConstraintTolerance(self, tol):
        self._ctol = float(tol)

    def getStepSize(self):
        return Quantity(self._dt, _unit.picosecond)

    def setStepSize(self, stepSize):
        self._dt = float(md_value(stepSize))
        if self._context is not None:
            self._context._engine.invalidate_program()

    def setRandomNumberSeed(self, seed):
        self._seed = int(seed)

    def getRandomNumberSeed(self):
        return self._seed

    def step(self, steps):
        if self._context is None:
            raise OpenMMException('This Integrator is not bound to a context!')
        self._context._engine.step(int(steps))


class VerletIntegrator(Integrator):
    """Leapfrog Verlet in OpenMM; here it only serves the reference's static energy checks (step size 0)."""


class CustomIntegrator(Integrator):
    ComputeGlobal, ComputePerDof, ComputeSum, ConstrainPositions, ConstrainVelocities, UpdateContextState, \
        IfBlock, WhileBlock, EndBlock = range(9)

    def __init__(self, stepSize):
        Integrator.__init__(self, stepSize)
        self._gnames, self._gvalues = [], []
        self._pnames, self._pvalues = [], []
        self._steps = []

    # variables
    def addGlobalVariable(self, name, initialValue):
        self._gnames.append(name)
        self._gvalues.append(float(md_value(initialValue)))
        return len(self._gnames) - 1

    def getNumGlobalVariables(self):
        return len(self._gnames)

    def getGlobalVariableName(self, index):
        return self._gnames[index]

    def getGlobalVariable(self, index):
        return self._gvalues[index]

    def getGlobalVariableByName(self, name):
        return self._gvalues[self._gnames.index(name)]

    def setGlobalVariable(self, index, value):
        self._gvalues[index] = float(md_value(value))
        if self._context is not None:
            self._context._engine.invalidate_program()

    def setGlobalVariableByName(self, name, value):
        self.setGlobalVariable(self._gnames.index(name), value)

    def addPerDofVariable(self, name, initialValue):
        self._pnames.append(name)
        self._pvalues.append(float(md_value(initialValue)))
        return len(self._pnames) - 1

    def getNumPerDofVariables(self):
        return len(self._pnames)

    def getPerDofVariableName(self, index):
        return self._pnames[index]

    def getPerDofVariableByName(self, name):
        idx = self._pnames.index(name)
        if self._context is not None:
            return self._context._engine.get_per_dof(name)
        n = getattr(self, '_n_hint', 0)
        return [Vec3(self._pvalues[idx], self._pvalues[idx], self._pvalues[idx]) for _ in range(n)]

    def getPerDofVariable(self, index):
        return self.getPerDofVariableByName(self._pnames[index])

    def setPerDofVariableByName(self, name, values):
        if self._context is not None:
            self._context._engine.set_per_dof(name, values)

    # program
    def _add(self, kind, target='', expr=''):
        self._steps.append((kind, target, expr))
        return len(self._steps) - 1

    def addComputeGlobal(self, variable, expression):
        return self._add(self.ComputeGlobal, variable, expression)

    def addComputePerDof(self, variable, expression):
        return self._add(self.ComputePerDof, variable, expression)

    def addComputeSum(self, variable, expression):
        return self._add(self.ComputeSum, variable, expression)

    def addConstrainPositions(self):
        return self._add(self.ConstrainPositions)

    def addConstrainVelocities(self):
        return self._add(self.ConstrainVelocities)

    def addUpdateContextState(self):
        return self._add(self.UpdateContextState)

    def beginIfBlock(self, condition):
        return self._add(self.IfBlock, '', condition)

    def beginWhileBlock(self, condition):
        return self._add(self.WhileBlock, '', condition)

    def endBlock(self):
        return self._add(self.EndBlock)

    def getNumComputations(self):
        return len(self._steps)

    def getComputationStep(self, index):
        return list(self._steps[index])


# ------------------------------------------------------------------------------------------------ platform / context
class Platform:
    _warned = set()

    def __init__(self, name='HIP'):
        self._name = name
        self._props = {}

    @staticmethod
    def getPlatformByName(name):
        """Only one platform exists here: the hand-written HIP path on MI355X.  Scripts written for
        the reference ask for 'Reference'/'CPU'/'CUDA'/'OpenCL'; they get the HIP platform (there is no
        CPU fallback) and `getName()` says so."""
        return Platform('HIP')

    @staticmethod
    def getNumPlatforms():
        return 1

    @staticmethod
    def getPlatform(index):
        return Platform('HIP')

    def getName(self):
        return self._name

    def setPropertyDefaultValue(self, name, value):
        self._props[name] = value

    def getPropertyDefaultValue(self, name):
        return self._props.get(name, '')


class State:
    def __init__(self, energy=None, kinetic=None, forces=None, positions=None, velocities=None, box=None, time=0.0):
        self._e, self._k, self._f, self._x, self._v, self._box, self._t = energy, kinetic, forces, positions, velocities, box, time

    def getPotentialEnergy(self):
        if self._e is None:
            raise OpenMMException('Invoked getPotentialEnergy() on a State which does not contain energies.')
        return Quantity(self._e, kjmol)

    def getKineticEnergy(self):
        if self._k is None:
            raise OpenMMException('Invoked getKineticEnergy() on a State which does not contain energies.')
        return Quantity(self._k, kjmol)

    def _vecs(self, arr, u, asNumpy):
        if arr is None:
            raise OpenMMException('State does not contain the requested data.')
        return Quantity(arr if asNumpy else [Vec3(*r) for r in arr.tolist()], u)

    def getForces(self, asNumpy=False):
        return self._vecs(self._f, kjmol / nm, asNumpy)

    def getPositions(self, asNumpy=False):
        return self._vecs(self._x, nm, asNumpy)

    def getVelocities(self, asNumpy=False):
        return self._vecs(self._v, nm / _unit.picosecond, asNumpy)

    def getPeriodicBoxVectors(self):
        return [Quantity(Vec3(*v), nm) for v in self._box]

    def getTime(self):
        return Quantity(self._t, _unit.picosecond)

    def getEnergyParameterDerivatives(self):
        if getattr(self, '_derivatives', None) is None:
            raise OpenMMException('Invoked getEnergyParameterDerivatives() on a State which does not contain parameter derivatives.')
        return dict(self._derivatives)


def _as_array(values, n, what):
    v = md_value(values)
    if isinstance(v, (list, tuple)) and len(v) and isinstance(v[0], Quantity):
        v = [md_value(r) for r in v]
    arr = np.array(v, dtype=np.float64)
    if arr.shape != (n, 3):
        raise OpenMMException('Called %s() on a Context with the wrong number of %s' % (what, what[3:].lower()))
    return arr


class Context:
    def __init__(self, system, integrator, platform=None, properties=None):
        from ..engine import Engine
        if integrator._context is not None:
            raise OpenMMException('This Integrator is already bound to a context')
        self._system = system
        self._integrator = integrator
        self._platform = platform or Platform('HIP')
        self._engine = Engine(system, integrator, properties or {})
        integrator._context = self

    def getSystem(self):
        return self._system

    def getIntegrator(self):
        return self._integrator

    def getPlatform(self):
        return self._platform

    def setPositions(self, positions):
        self._engine.set_positions(_as_array(positions, self._system.getNumParticles(), 'setPositions'))

    def setVelocities(self, velocities):
        self._engine.set_velocities(_as_array(velocities, self._system.getNumParticles(), 'setVelocities'))

    def setVelocitiesToTemperature(self, temperature, randomSeed=None):
        """Maxwell-Boltzmann velocities from numpy's generator (OpenMM's own RNG stream is not
        reproduced: SURVEY.md 8c G13 'parity unpinned')."""
        T = float(md_value(temperature))
        rng = np.random.default_rng(randomSeed)
        m = np.array(self._system._masses)
        kT = _unit.BOLTZMANN_CONSTANT_kB._value * T
        v = rng.normal(size=(len(m), 3)) * np.sqrt(kT / np.where(m > 0, m, 1.0))[:, None]
        v[m <= 0] = 0.0
        # multi-rank runs integrate every atom on every rank: all ranks must hold the SAME velocities, also when no seed
        # was given (each process would draw its own from OS entropy) -- rank 0's draw is broadcast
        v = self._engine.broadcast_from_rank0(v)
        self._engine.set_velocities(v)
        self._engine.apply_velocity_constraints()     # as OpenMM does after drawing the velocities

    def applyConstraints(self, tol=None):
        """Move the positions onto the constraint surface (reference = the current positions)."""
        self._engine.apply_constraints()

    def applyVelocityConstraints(self, tol=None):
        self._engine.apply_velocity_constraints()

    def setParameter(self, name, value):
        self._engine.set_parameter(name, float(md_value(value)))

    def getParameter(self, name):
        return self._engine.get_parameter(name)

    def getParameters(self):
        return dict(self._engine.parameters)

    def setPeriodicBoxVectors(self, a, b, c):
        new = [float(md_value(a[0])), float(md_value(b[1])), float(md_value(c[2]))]
        if not np.allclose(new, self._engine.box, rtol=0, atol=1e-12):
            raise OpenMMException('changing the box of a live Context is not supported by the HIP path')

    def setState(self, state):
        """Positions and velocities (those the State carries) of another Context's State."""
        if state._x is not None:
            self._engine.set_positions(state._x)
        if state._v is not None:
            self._engine.set_velocities(state._v)

    def getMolecules(self):
        """Connected components of the bond graph (bonds of the bonded forces + constraints), as OpenMM's
        Context.getMolecules(): a list of lists of atom indices, each in increasing order."""
        n = self._system.getNumParticles()
        parent = list(range(n))

        def find(a):
            while parent[a] != a:
                parent[a] = parent[parent[a]]
                a = parent[a]
            return a
        pairs = [(c[0], c[1]) for c in self._system._constraints]
        for force in self._system.getForces():
            if isinstance(force, (HarmonicBondForce, CustomBondForce)):
                pairs += [(b[0], b[1]) for b in force._bonds]
            elif isinstance(force, (HarmonicAngleForce, CustomAngleForce)):
                pairs += [(r[0], r[1]) for r in force._angles] + [(r[1], r[2]) for r in force._angles]
            elif isinstance(force, PeriodicTorsionForce):
                pairs += [(r[0], r[1]) for r in force._torsions] + [(r[1], r[2]) for r in force._torsions] + \
                         [(r[2], r[3]) for r in force._torsions]
        for i, j in pairs:
            a, b = find(int(i)), find(int(j))
            if a != b:
                parent[max(a, b)] = min(a, b)
        groups = {}
        for i in range(n):
            groups.setdefault(find(i), []).append(i)
        return [groups[r] for r in sorted(groups)]

    def getState(self, getPositions=False, getVelocities=False, getForces=False, getEnergy=False,
                 getParameters=False, enforcePeriodicBox=False, groups=-1, getParameterDerivatives=False):
        if isinstance(groups, (set, list, tuple, frozenset)):
            mask = 0
            for g in groups:
                mask |= 1 << int(g)
        else:
            mask = int(groups) & 0xFFFFFFFF
        state = self._engine.get_state(getPositions, getVelocities, getForces, getEnergy, mask)
        if getParameterDerivatives:
            # the parameters some force asked for with addEnergyParameterDerivative (e.g. AlchemicalSystem, systems.py:390)
            names = []
            for force in self._system.getForces():
                for name in getattr(force, '_derivs', ()):
                    if name not in names:
                        names.append(name)
            state._derivatives = {name: self._engine.energy_derivative(name) for name in names}
        return state

    def reinitialize(self, preserveState=False):
        self._engine.reinitialize(preserveState)


from . import app  # noqa: E402,F401
