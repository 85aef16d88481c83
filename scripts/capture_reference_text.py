#!/usr/bin/env python3
"""Capture what the reference's PYTHON layer emits -- CustomIntegrator step programs and energy strings -- into
tests/golden/programs.json.  BUILD CONTAINER ONLY (it reads /root/reference; nothing of the reference travels to the GPU box: the
fixture holds inputs -- constructor expressions -- and expected outputs -- names, program lines, strings -- as data).

    python scripts/capture_reference_text.py [--reference /root/reference/src] [--out tests/golden/programs.json]

OpenMM is not installed here, and every module of the reference imports `simtk` at the top.  The reference's own
arithmetic lives in OpenMM; its Python layer only COMPOSES text, so it runs under the small recording stand-in for
simtk.{openmm, unit} below: a CustomIntegrator that records the add* calls, force classes that keep their energy expression and
parameters, and a unit module whose quantities are floats in OpenMM's unit system (nm, ps, dalton, kJ/mol, K, e).  The
stand-in shares no code with atomsmm_amd.  tests/test_host_api.py evaluates the same constructor expressions against
atomsmm_amd and compares.  Numbers inside energy strings come through this stand-in's float conversions and may differ from real
simtk.unit in the last digit: the test compares strings after normalising numeric literals to 12 digits.
"""
import argparse
import json
import math
import os
import sys
import types

# --------------------------------------------------------------------------------------------- simtk.unit stand-in


class Q(float):
    """A quantity reduced to a float in OpenMM's unit system; arithmetic keeps the type so that .value_in_unit survives it."""

    def _w(self, v):
        return Q(v)

    def __mul__(self, o): return self._w(float(self) * float(o)) if isinstance(o, (int, float)) else NotImplemented
    __rmul__ = __mul__
    def __truediv__(self, o): return self._w(float(self) / float(o)) if isinstance(o, (int, float)) else NotImplemented
    def __rtruediv__(self, o): return self._w(float(o) / float(self))
    def __add__(self, o): return self._w(float(self) + float(o))
    __radd__ = __add__
    def __sub__(self, o): return self._w(float(self) - float(o))
    def __rsub__(self, o): return self._w(float(o) - float(self))
    def __neg__(self): return self._w(-float(self))
    def __pow__(self, p): return self._w(float(self) ** p)
    def value_in_unit(self, unit): return float(self) / float(unit)
    def value_in_unit_system(self, system): return float(self)
    def in_units_of(self, unit): return self
    def sqrt(self): return self._w(math.sqrt(float(self)))
    @property
    def _value(self): return float(self)
    @property
    def unit(self): return Q(1.0)


def make_unit_module():
    u = types.ModuleType('simtk.unit')
    u.Quantity = Q
    u.Unit = Q
    u.is_quantity = lambda x: isinstance(x, Q)
    u.md_unit_system = 'md'
    u.sqrt = lambda x: Q(math.sqrt(float(x)))
    for name, value in dict(nanometer=1.0, nanometers=1.0, angstrom=0.1, angstroms=0.1, picosecond=1.0, picoseconds=1.0,
                            femtosecond=1e-3, femtoseconds=1e-3, nanosecond=1e3, nanoseconds=1e3, kelvin=1.0, dalton=1.0, daltons=1.0,
                            amu=1.0, kilojoule_per_mole=1.0, kilojoules_per_mole=1.0, kilocalorie_per_mole=4.184,
                            kilocalories_per_mole=4.184, elementary_charge=1.0, elementary_charges=1.0, radian=1.0, radians=1.0,
                            degree=math.pi / 180, degrees=math.pi / 180, dimensionless=1.0, atmospheres=1.01325, bar=1.0, bars=1.0,
                            mole=1.0, moles=1.0, item=1.0, liter=1e24, gram=6.02214179e23, grams=6.02214179e23, joule=6.02214179e20,
                            # CODATA 2006, the values simtk.unit carries (tests/test_computers.py of the reference pins them)
                            BOLTZMANN_CONSTANT_kB=1.3806504e-23 * 1e-3, AVOGADRO_CONSTANT_NA=6.02214179e23,
                            MOLAR_GAS_CONSTANT_R=1.3806504e-23 * 1e-3 * 6.02214179e23).items():
        setattr(u, name, Q(value))
    return u


# --------------------------------------------------------------------------------------------- simtk.openmm stand-in


class Recorder:
    """Fallback for everything the captured code paths construct but never read back."""

    def __init__(self, *args, **kwargs):
        self.args, self.kwargs, self.calls = args, kwargs, []

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)

        def method(*args, **kwargs):
            self.calls.append((name, args, kwargs))
            return 0
        return method


class CustomIntegrator:
    ComputeGlobal, ComputePerDof, ComputeSum, ConstrainPositions, ConstrainVelocities, UpdateContextState, IfBlock, WhileBlock, \
        EndBlock = range(9)
    BlockEnd, IfBlockStart, WhileBlockStart = EndBlock, IfBlock, WhileBlock        # OpenMM's own names

    def __init__(self, stepSize):
        self._dt = float(stepSize)
        self.globals_, self.perdof, self.steps = [], [], []
        self.seed = None

    def getStepSize(self): return Q(self._dt)
    def setStepSize(self, v): self._dt = float(v)
    def addGlobalVariable(self, name, value): self.globals_.append([name, float(value)]); return len(self.globals_) - 1
    def addPerDofVariable(self, name, value): self.perdof.append(name); return len(self.perdof) - 1
    def getNumGlobalVariables(self): return len(self.globals_)
    def getNumPerDofVariables(self): return len(self.perdof)
    def getGlobalVariableName(self, i): return self.globals_[i][0]
    def getPerDofVariableName(self, i): return self.perdof[i]
    def getGlobalVariable(self, i): return self.globals_[i][1]
    def getGlobalVariableByName(self, name): return dict(self.globals_)[name]

    def setGlobalVariableByName(self, name, value):
        for g in self.globals_:
            if g[0] == name:
                g[1] = float(value)

    def setPerDofVariableByName(self, name, values): pass
    def getPerDofVariableByName(self, name): return []
    def _add(self, kind, target='', expr=''): self.steps.append((kind, target, expr)); return len(self.steps) - 1
    def addComputeGlobal(self, v, e): return self._add(self.ComputeGlobal, v, e)
    def addComputePerDof(self, v, e): return self._add(self.ComputePerDof, v, e)
    def addComputeSum(self, v, e): return self._add(self.ComputeSum, v, e)
    def addConstrainPositions(self): return self._add(self.ConstrainPositions)
    def addConstrainVelocities(self): return self._add(self.ConstrainVelocities)
    def addUpdateContextState(self): return self._add(self.UpdateContextState)
    def beginIfBlock(self, c): return self._add(self.IfBlock, '', c)
    def beginWhileBlock(self, c): return self._add(self.WhileBlock, '', c)
    def endBlock(self): return self._add(self.EndBlock)
    def getNumComputations(self): return len(self.steps)
    def getComputationStep(self, i): return list(self.steps[i])
    def setRandomNumberSeed(self, seed): self.seed = seed
    def getRandomNumberSeed(self): return self.seed or 0
    def setConstraintTolerance(self, tol): pass
    def getConstraintTolerance(self): return 1e-5
    def setKineticEnergyExpression(self, e): pass
    def step(self, n): pass


class _Force:
    def __init__(self):
        self.group = 0

    def setForceGroup(self, g): self.group = g
    def getForceGroup(self): return self.group
    def usesPeriodicBoundaryConditions(self): return True


class NonbondedForce(_Force):
    NoCutoff, CutoffNonPeriodic, CutoffPeriodic, Ewald, PME, LJPME = range(6)

    def __init__(self):
        super().__init__()
        self.particles, self.exceptions, self.offsets, self.exc_offsets, self.globals_ = [], [], [], [], []
        self.method, self.cutoff, self.use_switch, self.switch, self.lrc = self.PME, Q(1.0), False, Q(0.9), True
        self.tol, self.recip_group = 5e-4, -1

    def addParticle(self, q, s, e): self.particles.append((q, s, e)); return len(self.particles) - 1
    def getNumParticles(self): return len(self.particles)
    def getParticleParameters(self, i): return [Q(v) for v in self.particles[i]]
    def setParticleParameters(self, i, q, s, e): self.particles[i] = (q, s, e)
    def addException(self, i, j, qq, s, e, replace=False): self.exceptions.append((i, j, qq, s, e)); return len(self.exceptions) - 1
    def getNumExceptions(self): return len(self.exceptions)
    def getExceptionParameters(self, k): e = self.exceptions[k]; return [e[0], e[1]] + [Q(v) for v in e[2:]]
    def setExceptionParameters(self, k, i, j, qq, s, e): self.exceptions[k] = (i, j, qq, s, e)
    def getNonbondedMethod(self): return self.method
    def setNonbondedMethod(self, m): self.method = m
    def getCutoffDistance(self): return self.cutoff
    def setCutoffDistance(self, c): self.cutoff = Q(float(c))
    def getUseSwitchingFunction(self): return self.use_switch
    def setUseSwitchingFunction(self, u): self.use_switch = u
    def getSwitchingDistance(self): return self.switch
    def setSwitchingDistance(self, d): self.switch = Q(float(d))
    def getUseDispersionCorrection(self): return self.lrc
    def setUseDispersionCorrection(self, u): self.lrc = u
    def getEwaldErrorTolerance(self): return self.tol
    def setEwaldErrorTolerance(self, t): self.tol = t
    def getPMEParameters(self): return [Q(0.0), 0, 0, 0]
    def setPMEParameters(self, *a): pass
    def getReactionFieldDielectric(self): return 78.3
    def setReactionFieldDielectric(self, d): pass
    def getReciprocalSpaceForceGroup(self): return self.recip_group
    def setReciprocalSpaceForceGroup(self, g): self.recip_group = g
    def getNumParticleParameterOffsets(self): return len(self.offsets)
    def getNumExceptionParameterOffsets(self): return len(self.exc_offsets)
    def getParticleParameterOffset(self, k): return self.offsets[k]
    def getExceptionParameterOffset(self, k): return self.exc_offsets[k]
    def addParticleParameterOffset(self, *a): self.offsets.append(a); return len(self.offsets) - 1
    def addExceptionParameterOffset(self, *a): self.exc_offsets.append(a); return len(self.exc_offsets) - 1
    def getNumGlobalParameters(self): return len(self.globals_)
    def getGlobalParameterName(self, k): return self.globals_[k][0]
    def getGlobalParameterDefaultValue(self, k): return self.globals_[k][1]
    def addGlobalParameter(self, n, v): self.globals_.append((n, v)); return len(self.globals_) - 1


class _CustomForce(_Force):
    def __init__(self, energy=''):
        super().__init__()
        self.energy, self.globals_, self.per, self.items, self.exclusions = energy, [], [], [], []
        self.method, self.cutoff, self.use_switch, self.switch, self.lrc = 0, Q(1.0), False, Q(0.0), False
        self.derivs = []

    def getEnergyFunction(self): return self.energy
    def setEnergyFunction(self, e): self.energy = e
    def addGlobalParameter(self, n, v): self.globals_.append((n, float(v))); return len(self.globals_) - 1
    def getNumGlobalParameters(self): return len(self.globals_)
    def getGlobalParameterName(self, k): return self.globals_[k][0]
    def getGlobalParameterDefaultValue(self, k): return self.globals_[k][1]
    def addEnergyParameterDerivative(self, n): self.derivs.append(n)
    def setNonbondedMethod(self, m): self.method = m
    def getNonbondedMethod(self): return self.method
    def setCutoffDistance(self, c): self.cutoff = Q(float(c))
    def getCutoffDistance(self): return self.cutoff
    def setUseSwitchingFunction(self, u): self.use_switch = u
    def getUseSwitchingFunction(self): return self.use_switch
    def setSwitchingDistance(self, d): self.switch = Q(float(d))
    def getSwitchingDistance(self): return self.switch
    def setUseLongRangeCorrection(self, u): self.lrc = u
    def getUseLongRangeCorrection(self): return self.lrc
    def setUsesPeriodicBoundaryConditions(self, p): pass


class CustomNonbondedForce(_CustomForce):
    NoCutoff, CutoffNonPeriodic, CutoffPeriodic = range(3)
    def addPerParticleParameter(self, n): self.per.append(n); return len(self.per) - 1
    def getNumPerParticleParameters(self): return len(self.per)
    def getPerParticleParameterName(self, k): return self.per[k]
    def addParticle(self, p=()): self.items.append(tuple(p)); return len(self.items) - 1
    def getNumParticles(self): return len(self.items)
    def getParticleParameters(self, i): return self.items[i]
    def setParticleParameters(self, i, p): self.items[i] = tuple(p)
    def addExclusion(self, i, j): self.exclusions.append((i, j)); return len(self.exclusions) - 1
    def getNumExclusions(self): return len(self.exclusions)
    def addInteractionGroup(self, a, b): return 0


class CustomBondForce(_CustomForce):
    def addPerBondParameter(self, n): self.per.append(n); return len(self.per) - 1
    def getNumPerBondParameters(self): return len(self.per)
    def addBond(self, i, j, p=()): self.items.append((i, j, tuple(p))); return len(self.items) - 1
    def getNumBonds(self): return len(self.items)


class System(Recorder):
    def __init__(self):
        super().__init__()
        self.forces = []

    def addForce(self, f): self.forces.append(f); return len(self.forces) - 1
    def getForces(self): return list(self.forces)
    def getNumForces(self): return len(self.forces)
    def getForce(self, i): return self.forces[i]


def make_openmm_module(unit):
    mm = types.ModuleType('simtk.openmm')
    for cls in (CustomIntegrator, NonbondedForce, CustomNonbondedForce, CustomBondForce, System):
        setattr(mm, cls.__name__, cls)
    for name in ('CustomCVForce', 'CustomAngleForce', 'HarmonicBondForce', 'HarmonicAngleForce', 'PeriodicTorsionForce', 'Context',
                 'Platform', 'Force', 'CMMotionRemover', 'LocalEnergyMinimizer', 'State', 'Vec3', 'VerletIntegrator',
                 'CustomExternalForce', 'CustomCompoundBondForce', 'CustomTorsionForce'):
        setattr(mm, name, type(name, (Recorder,), {}))
    mm.OpenMMException = type('OpenMMException', (Exception,), {})
    app = types.ModuleType('simtk.openmm.app')
    for name in ('StateDataReporter', 'Simulation', 'Topology', 'PDBFile', 'ForceField', 'Element'):
        setattr(app, name, type(name, (Recorder,), {}))
    for name in ('PME', 'Ewald', 'CutoffPeriodic', 'NoCutoff', 'HBonds', 'AllBonds', 'HAngles'):
        setattr(app, name, name)
    mm.app = app
    mm.unit = unit
    return mm, app


def install_stand_in():
    unit = make_unit_module()
    mm, app = make_openmm_module(unit)
    simtk = types.ModuleType('simtk')
    simtk.openmm, simtk.unit = mm, unit
    sys.modules.update({'simtk': simtk, 'simtk.openmm': mm, 'simtk.unit': unit, 'simtk.openmm.app': app})
    return mm, unit


# --------------------------------------------------------------------------------------------- what is captured

FMT = ['{target} <- {expr}', '{target} <- {expr}', '{target} <- sum({expr})', 'constrain positions', 'constrain velocities',
       'allow forces to update the context state', 'if ({expr}):', 'while ({expr}):', 'end']


def pretty_steps(integrator):
    out, depth = [], 0
    for kind, target, expr in integrator.steps:
        if kind == CustomIntegrator.EndBlock:
            depth -= 1
        out.append('   ' * depth + FMT[kind].format(target=target, expr=expr))
        if kind in (CustomIntegrator.IfBlock, CustomIntegrator.WhileBlock):
            depth += 1
    return out


T, TAU, GAMMA = '300*unit.kelvin', '10*unit.femtoseconds', '10/unit.picoseconds'
P = 'atomsmm.propagators.'
PROGRAMS = {
    # a-8 / a-9: base and composition classes (propagators.py:24-273)
    'translation_unconstrained': P + 'TranslationPropagator(constrained=False).integrator(1*unit.femtoseconds)',
    'translation_constrained': P + 'TranslationPropagator().integrator(1*unit.femtoseconds)',
    'boost_unconstrained': P + 'VelocityBoostPropagator(constrained=False).integrator(1*unit.femtoseconds)',
    'boost_constrained': P + 'VelocityBoostPropagator().integrator(1*unit.femtoseconds)',
    'chained': 'atomsmm.ChainedPropagator([atomsmm.VelocityBoostPropagator(False), atomsmm.TranslationPropagator(False)]).integrator(2*unit.femtoseconds)',
    'split_3': 'atomsmm.SplitPropagator(atomsmm.TranslationPropagator(False), 3).integrator(2*unit.femtoseconds)',
    'trotter_suzuki': 'atomsmm.TrotterSuzukiPropagator(atomsmm.TranslationPropagator(False), atomsmm.VelocityBoostPropagator(False)).integrator(2*unit.femtoseconds)',
    'suzuki_yoshida_3': 'atomsmm.SuzukiYoshidaPropagator(atomsmm.TranslationPropagator(False), 3).integrator(2*unit.femtoseconds)',
    'suzuki_yoshida_7': 'atomsmm.SuzukiYoshidaPropagator(atomsmm.TranslationPropagator(False), 7).integrator(2*unit.femtoseconds)',
    'suzuki_yoshida_15': 'atomsmm.SuzukiYoshidaPropagator(atomsmm.TranslationPropagator(False), 15).integrator(2*unit.femtoseconds)',
    # a-10: RESPA and its schemes (propagators.py:830-1042)
    'respa_4_2_1': 'atomsmm.RespaPropagator([4, 2, 1]).integrator(4*unit.femtoseconds)',
    'respa_1_1': 'atomsmm.RespaPropagator([1, 1]).integrator(1*unit.femtoseconds)',
    'respa_4_1_constrained': 'atomsmm.RespaPropagator([4, 1], boost=atomsmm.VelocityBoostPropagator(constrained=True), move=atomsmm.TranslationPropagator(constrained=True)).integrator(1*unit.femtoseconds)',
    'respa_2_1_memory': 'atomsmm.RespaPropagator([2, 1], has_memory=True).integrator(1*unit.femtoseconds)',
    'respa_3_2_1_memory': 'atomsmm.RespaPropagator([3, 2, 1], has_memory=True).integrator(3*unit.femtoseconds)',
    'respa_2_2_switch': 'atomsmm.RespaPropagator([2, 2], use_respa_switch=True).integrator(2*unit.femtoseconds)',
    'respa_2_2_1_blitz': 'atomsmm.RespaPropagator([2, 2, 1], blitz=True).integrator(2*unit.femtoseconds)',
    'respa_core_ou': 'atomsmm.RespaPropagator([2, 2, 1], core=atomsmm.OrnsteinUhlenbeckPropagator(%s, %s)).integrator(2*unit.femtoseconds)' % (T, GAMMA),
    'respa_shell_ou': 'atomsmm.RespaPropagator([2, 2, 1], shell={1: atomsmm.OrnsteinUhlenbeckPropagator(%s, %s)}).integrator(2*unit.femtoseconds)' % (T, GAMMA),
    **{'mts_%s' % scheme.replace('-', '_'): P + 'MultipleTimeScalePropagator([2, 2, 1], bath=atomsmm.OrnsteinUhlenbeckPropagator(%s, %s), scheme=%r).integrator(2*unit.femtoseconds)'
       % (T, GAMMA, scheme) for scheme in ('middle', 'blitz', 'xi-respa', 'xo-respa', 'side')},
    'mts_middle_nres_2_nsy_3': P + 'MultipleTimeScalePropagator([2, 1], bath=atomsmm.OrnsteinUhlenbeckPropagator(%s, %s), scheme="middle", nres=2, nsy=3).integrator(2*unit.femtoseconds)' % (T, GAMMA),
    # f-2: velocity Verlet and the thermostat propagators (propagators.py:276-355, 685-827, 1045-1536)
    'velocity_verlet': 'atomsmm.GlobalThermostatIntegrator(1*unit.femtoseconds, atomsmm.VelocityVerletPropagator())',
    'unconstrained_velocity_verlet': 'atomsmm.GlobalThermostatIntegrator(1*unit.femtoseconds, ' + P + 'UnconstrainedVelocityVerletPropagator())',
    'nvt_velocity_rescaling': 'atomsmm.GlobalThermostatIntegrator(1*unit.femtoseconds, atomsmm.VelocityVerletPropagator(), atomsmm.VelocityRescalingPropagator(%s, 1000, %s))' % (T, TAU),
    'nvt_nose_hoover': 'atomsmm.GlobalThermostatIntegrator(1*unit.femtoseconds, atomsmm.VelocityVerletPropagator(), atomsmm.NoseHooverPropagator(%s, 1000, %s))' % (T, TAU),
    'nvt_nose_hoover_nloops_2': 'atomsmm.GlobalThermostatIntegrator(1*unit.femtoseconds, atomsmm.VelocityVerletPropagator(), atomsmm.NoseHooverPropagator(%s, 1000, %s, 2))' % (T, TAU),
    'nvt_nose_hoover_chain': 'atomsmm.GlobalThermostatIntegrator(1*unit.femtoseconds, atomsmm.VelocityVerletPropagator(), ' + P + 'NoseHooverChainPropagator(%s, 1000, %s))' % (T, TAU),
    'nvt_nose_hoover_chain_friction': 'atomsmm.GlobalThermostatIntegrator(1*unit.femtoseconds, atomsmm.VelocityVerletPropagator(), ' + P + 'NoseHooverChainPropagator(%s, 1000, %s, %s))' % (T, TAU, GAMMA),
    'nvt_nose_hoover_langevin': 'atomsmm.GlobalThermostatIntegrator(1*unit.femtoseconds, atomsmm.VelocityVerletPropagator(), atomsmm.NoseHooverLangevinPropagator(%s, 1000, %s))' % (T, TAU),
    'nvt_nose_hoover_langevin_friction': 'atomsmm.GlobalThermostatIntegrator(1*unit.femtoseconds, atomsmm.VelocityVerletPropagator(), atomsmm.NoseHooverLangevinPropagator(%s, 1000, %s, %s))' % (T, TAU, GAMMA),
    'massive_nose_hoover': P + 'MassiveNoseHooverPropagator(%s, %s).integrator(1*unit.femtoseconds)' % (T, TAU),
    'massive_nose_hoover_nloops_3': P + 'MassiveNoseHooverPropagator(%s, %s, 3).integrator(1*unit.femtoseconds)' % (T, TAU),
    'massive_ggm': P + 'MassiveGeneralizedGaussianMomentPropagator(%s, %s).integrator(1*unit.femtoseconds)' % (T, TAU),
    'ornstein_uhlenbeck': 'atomsmm.OrnsteinUhlenbeckPropagator(%s, %s).integrator(1*unit.femtoseconds)' % (T, GAMMA),
    'ornstein_uhlenbeck_force': 'atomsmm.OrnsteinUhlenbeckPropagator(%s, %s, "v1", "Q1", force="G").integrator(1*unit.femtoseconds)' % (T, GAMMA),
    'generic_boost': 'atomsmm.GenericBoostPropagator().integrator(1*unit.femtoseconds)',
    'generic_boost_global': 'atomsmm.GenericBoostPropagator("v_eta", "Q_eta", "(mvv - LkT)", perDof=False, LkT=2.5).integrator(1*unit.femtoseconds)',
    'generic_scaling': P + 'GenericScalingPropagator("v1", "v2").integrator(1*unit.femtoseconds)',
    'generic_scaling_global': P + 'GenericScalingPropagator("v", "v_eta", perDof=False).integrator(1*unit.femtoseconds)',
    'massive_isokinetic_force': 'atomsmm.MassiveIsokineticPropagator(%s, %s, 1, forceDependent=True).integrator(1*unit.femtoseconds)' % (T, TAU),
    'massive_isokinetic_bath': 'atomsmm.MassiveIsokineticPropagator(%s, %s, 1, forceDependent=False).integrator(1*unit.femtoseconds)' % (T, TAU),
    'massive_isokinetic_L4': 'atomsmm.MassiveIsokineticPropagator(%s, %s, 4, forceDependent=True).integrator(1*unit.femtoseconds)' % (T, TAU),
    'sin_r_propagator': P + 'SIN_R_Propagator([2, 2, 1], %s, %s, %s).integrator(2*unit.femtoseconds)' % (T, TAU, GAMMA),
    # a-11 / f-2: integrators (integrators.py:173-417)
    'mts_integrator': 'atomsmm.integrators.MultipleTimeScaleIntegrator(2*unit.femtoseconds, [2, 2, 1])',
    'mts_integrator_bath_side': 'atomsmm.integrators.MultipleTimeScaleIntegrator(2*unit.femtoseconds, [2, 2, 1], bath=atomsmm.OrnsteinUhlenbeckPropagator(%s, %s), scheme="side")' % (T, GAMMA),
    'langevin_r': 'atomsmm.integrators.Langevin_R_Integrator(4*unit.femtoseconds, [4, 2, 1], %s, %s)' % (T, GAMMA),
    'langevin_r_xo': 'atomsmm.integrators.Langevin_R_Integrator(4*unit.femtoseconds, [4, 2, 1], %s, %s, scheme="xo-respa")' % (T, GAMMA),
    'nhl_r': 'atomsmm.NHL_R_Integrator(4*unit.femtoseconds, [4, 2, 1], %s, %s, %s)' % (T, TAU, GAMMA),
    'nhl_r_side': 'atomsmm.NHL_R_Integrator(4*unit.femtoseconds, [2, 1], %s, %s, %s, scheme="side")' % (T, TAU, GAMMA),
    'sin_r': 'atomsmm.SIN_R_Integrator(4*unit.femtoseconds, [4, 2, 1], %s, %s, %s)' % (T, TAU, GAMMA),
    # a-12: AFED (integrators.py:642-860)
    'afed_respa_2_1_nsteps_2': 'atomsmm.AdiabaticDynamicsIntegrator(atomsmm.RespaPropagator([2, 1]).integrator(1*unit.femtoseconds), 2, [atomsmm.ExtendedSystemVariable("lambda_vdw", 1000, 5, 40*unit.femtoseconds)])',
    'afed_periodic_langevin': 'atomsmm.AdiabaticDynamicsIntegrator(atomsmm.RespaPropagator([2, 1]).integrator(1*unit.femtoseconds), 1, [atomsmm.ExtendedSystemVariable("phi", 100, 2.5, 40*unit.femtoseconds, -3.14, 3.14, periodic=True, thermostat="langevin")])',
}

# a-2 ... a-6: energy strings after importFrom(nonbonded) of a two-particle CutoffPeriodic force
FORCES = {
    'near_none': 'atomsmm.NearNonbondedForce(0.7*unit.nanometers, 0.5*unit.nanometers, None)',
    'near_shift': 'atomsmm.NearNonbondedForce(0.7*unit.nanometers, 0.5*unit.nanometers, "shift")',
    'near_force_switch': 'atomsmm.NearNonbondedForce(0.7*unit.nanometers, 0.5*unit.nanometers, "force-switch")',
    'near_force_switch_subtract': 'atomsmm.NearNonbondedForce(0.7*unit.nanometers, 0.5*unit.nanometers, "force-switch", subtract=True)',
    'near_force_switch_actual_cutoff': 'atomsmm.NearNonbondedForce(0.7*unit.nanometers, 0.5*unit.nanometers, "force-switch", actual_cutoff=1.0*unit.nanometers)',
    'damped_degree_1': 'atomsmm.DampedSmoothedForce(2.9/unit.nanometers, 1.0*unit.nanometers, 0.9*unit.nanometers)',
    'damped_degree_2': 'atomsmm.DampedSmoothedForce(2.9/unit.nanometers, 1.0*unit.nanometers, 0.9*unit.nanometers, degree=2)',
    'exceptions': 'atomsmm.NonbondedExceptionsForce()',
    'near_exception': 'atomsmm.NearExceptionForce(0.7*unit.nanometers, 0.5*unit.nanometers, "force-switch")',
    'softcore_lj': 'atomsmm.SoftcoreLennardJonesForce(parameter="lambda_vdw")',
    'softcore': 'atomsmm.SoftcoreForce(1.0*unit.nanometers, 0.9*unit.nanometers)',
}


def patch_sympy():
    """The reference scans every expression with sympy's parse_expr (integrators.py:91-104).  From sympy 1.x on, a bare `Q` parses
    to sympy.Q (the assumptions object), so the reference's own variable names Q, Q1, Q2 ... raise inside sympy.  The reference was
    written against an older sympy; here every identifier of the expression is declared a Symbol first.  An environment shim:
    no line of the reference is changed."""
    import re
    import sympy
    from sympy.parsing import sympy_parser
    original = sympy_parser.parse_expr

    def parse_expr(text, local_dict=None, **kwargs):
        names = dict(local_dict or {})
        for m in re.finditer(r'[A-Za-z_][A-Za-z_0-9]*', text):
            word, called = m.group(0), text[m.end():].lstrip().startswith('(')
            if called:
                if word not in dir(sympy):           # OpenMM's own functions: step, select, deriv ...
                    names.setdefault(word, sympy.Function(word))
            elif word in ('Q', 'S', 'N', 'E', 'I', 'O', 'beta', 'gamma', 'zeta', 'lambda') or word not in dir(sympy):
                names.setdefault(word, sympy.Symbol(word))
        return original(text, local_dict=names, **kwargs)
    sympy_parser.parse_expr = parse_expr


def capture(reference):
    mm, unit = install_stand_in()
    patch_sympy()
    sys.path.insert(0, reference)
    import atomsmm                       # the REFERENCE (sys.path[0]); this script never imports atomsmm_amd
    assert os.path.realpath(os.path.dirname(atomsmm.__file__)).startswith(os.path.realpath(reference))
    ns = {'atomsmm': atomsmm, 'unit': unit, 'openmm': mm}
    programs, failed = {}, {}
    for name, ctor in PROGRAMS.items():
        try:
            integ = eval(ctor, ns)
        except Exception as exc:          # e.g. a scheme the reference itself cannot build: recorded, not hidden
            failed[name] = {'ctor': ctor, 'error': '%s: %s' % (type(exc).__name__, exc)}
            continue
        programs[name] = {'ctor': ctor, 'per_dof': list(integ.perdof), 'globals': [g[0] for g in integ.globals_],
                          'global_values': {g[0]: g[1] for g in integ.globals_}, 'steps': pretty_steps(integ)}
    forces = {}
    for name, ctor in FORCES.items():
        nb = mm.NonbondedForce()
        nb.setNonbondedMethod(nb.CutoffPeriodic)
        nb.addParticle(0.5, 0.3, 0.7)
        nb.addParticle(-0.5, 0.25, 0.2)
        nb.addParticle(0.1, 0.2, 0.1)
        nb.addException(0, 1, -0.1, 0.27, 0.3)
        try:
            force = eval(ctor, ns)
            force.importFrom(nb)
        except Exception as exc:
            failed[name] = {'ctor': ctor, 'error': '%s: %s' % (type(exc).__name__, exc)}
            continue
        forces[name] = {'ctor': ctor, 'energy': force.getEnergyFunction(),
                        'globals': {n: v for n, v in force.globals_} if hasattr(force, 'globals_') else {}}
    return programs, forces, failed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reference', default='/root/reference/src')
    ap.add_argument('--out', default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'programs.json'))
    args = ap.parse_args()
    programs, forces, failed = capture(args.reference)
    doc = {'_comment': 'Captured by scripts/capture_reference_text.py from the reference\'s Python layer (AtomsMM v0.1.0, /root/reference/src/atomsmm: '
                       'propagators.py, integrators.py, forces.py) under a recording stand-in for simtk; data only -- constructor expressions and the '
                       'text they emit.  "failed": constructor expressions the reference itself raises on (its defects, SURVEY Appendix A).',
           'programs': programs, 'forces': forces, 'failed': failed}
    with open(args.out, 'w') as fh:
        json.dump(doc, fh, indent=1, sort_keys=True)
    print('%d programs, %d force strings, %d failed -> %s' % (len(programs), len(forces), len(failed), args.out))
    for name, info in failed.items():
        print('  failed %s: %s' % (name, info['error']))


if __name__ == '__main__':
    main()
