"""Integrator hosts with the names of `atomsmm.integrators` (reference: src/atomsmm/integrators.py),
hot-path subset: `_AtomsMM_Integrator`, `GlobalThermostatIntegrator`, `MultipleTimeScaleIntegrator`.

`_AtomsMM_Integrator` keeps the reference's bookkeeping while a program is being emitted
(integrators.py:26-170): it adds the globals `mvv`, `NDOF` and the per-DOF `ndof`; inserts
`mvv <- sum(m*v*v)` before the first use of `mvv` after velocities changed; inserts one
"allow forces to update the context state" before the first use of a force; and splits a per-DOF
expression that references several force groups through `_f{k}_` buffers (a CustomIntegrator
computation may read one force group only).  The reference finds free symbols with sympy; an
identifier scan is equivalent for these expressions and keeps sympy off the import path.
"""
import math
import re

import numpy as np

from . import openmm
from . import propagators
from .utils import InputError, kB

_IDENT = re.compile(r'[A-Za-z_][A-Za-z_0-9]*')
_FUNCS = {'sqrt', 'exp', 'log', 'sin', 'cos', 'sec', 'csc', 'tan', 'cot', 'asin', 'acos', 'atan', 'atan2', 'sinh',
          'cosh', 'tanh', 'erf', 'erfc', 'min', 'max', 'abs', 'floor', 'ceil', 'step', 'delta', 'select', 'deriv'}


class _AtomsMM_Integrator(openmm.CustomIntegrator):
    def __init__(self, stepSize):
        super().__init__(stepSize)
        self.addGlobalVariable('mvv', 0.0)
        self.addGlobalVariable('NDOF', 0.0)
        self.addPerDofVariable('ndof', 0.0)
        self._obsoleteKinetic = True
        self._forceFinder = re.compile('^f[0-9]+$|^f$')
        self._obsoleteContextState = True
        self._random = np.random.RandomState()
        self._uninitialized = True

    def __repr__(self):
        """Human-readable program, same layout as the reference's (integrators.py:38-86)."""
        lines = ['Per-dof variables:',
                 '  ' + ', '.join(self.getPerDofVariableName(i) for i in range(self.getNumPerDofVariables())),
                 'Global variables:']
        for i in range(self.getNumGlobalVariables()):
            lines.append('  {} = {}'.format(self.getGlobalVariableName(i), self.getGlobalVariable(i)))
        lines.append('Computation steps:')
        lines += ['{:4d}: {}'.format(k, text) for k, text in enumerate(self.pretty_steps())]
        return '\n'.join(lines)

    def pretty_steps(self):
        """One line per computation, indented by block depth (the text captured in SURVEY.md 3.2)."""
        fmt = ['{target} <- {expr}', '{target} <- {expr}', '{target} <- sum({expr})', 'constrain positions',
               'constrain velocities', 'allow forces to update the context state', 'if ({expr}):', 'while ({expr}):', 'end']
        out, depth = [], 0
        for index in range(self.getNumComputations()):
            kind, target, expr = self.getComputationStep(index)
            if kind == self.EndBlock:
                depth -= 1
            out.append('   ' * depth + fmt[kind].format(target=target, expr=expr))
            if kind in (self.IfBlock, self.WhileBlock):
                depth += 1
        return out

    def _normalVec(self):
        return openmm.Vec3(self._random.normal(), self._random.normal(), self._random.normal())

    def _required_variables(self, variable, expression):
        """Names an assignment `variable <- expression` reads (excluding names it defines itself)."""
        defined, used = set(), set()
        for definition in '{}={}'.format(variable, expression).split(';'):
            name, expr = definition.split('=', 1)
            defined.add(name.strip())
            for m in _IDENT.finditer(expr):
                word = m.group(0)
                after = expr[m.end():].lstrip()
                if word in _FUNCS and after.startswith('('):
                    continue
                if re.fullmatch(r'[eE][0-9]*', word) and m.start() > 0 and (expr[m.start() - 1].isdigit() or expr[m.start() - 1] == '.'):
                    continue    # exponent of a float literal such as 1e5
                used.add(word)
        return sorted(used - defined)

    def _checkUpdate(self, requirements):
        if self._obsoleteKinetic and 'mvv' in requirements:
            openmm.CustomIntegrator.addComputeSum(self, 'mvv', 'm*v*v')
            self._obsoleteKinetic = False
        if self._obsoleteContextState and any(self._forceFinder.match(s) for s in requirements):
            openmm.CustomIntegrator.addUpdateContextState(self)
            self._obsoleteContextState = False

    def addUpdateContextState(self):
        if self._obsoleteContextState:
            openmm.CustomIntegrator.addUpdateContextState(self)
            self._obsoleteContextState = False

    def addComputeGlobal(self, variable, expression):
        if variable == 'mvv':
            raise InputError('Cannot assign value to global variable mvv')
        self._checkUpdate(self._required_variables(variable, expression))
        return openmm.CustomIntegrator.addComputeGlobal(self, variable, expression)

    def addComputePerDof(self, variable, expression):
        requirements = self._required_variables(variable, expression)
        self._checkUpdate(requirements)
        forces = sorted(s for s in requirements if self._forceFinder.match(s))
        if len(forces) > 1:
            # one force group per computation: stash all but the first in per-DOF buffers _f{k}_
            expression = re.sub(r'\bf([0-9]*)\b', '_f\\1_', expression)
            buffers = ['_{}_'.format(f) for f in forces]
            existing = [self.getPerDofVariableName(i) for i in range(self.getNumPerDofVariables())]
            for force, buffer in zip(forces[1:], buffers[1:]):
                if buffer not in existing:
                    self.addPerDofVariable(buffer, 0.0)
                self.addComputePerDof(buffer, force)
            expression = re.sub(r'\b{}\b'.format(buffers[0]), forces[0], expression)
        index = openmm.CustomIntegrator.addComputePerDof(self, variable, expression)
        if variable == 'v':
            self._obsoleteKinetic = True
        return index

    def setRandomNumberSeed(self, seed):
        self._random.seed(seed)
        openmm.CustomIntegrator.setRandomNumberSeed(self, self._random.tomaxint() % 2 ** 31)

    def step(self, steps):
        if self._uninitialized:
            if self._context is None:
                raise openmm.OpenMMException('This Integrator is not bound to a context!')
            self._ndof = NDOF = 3 * self._context.getSystem().getNumParticles()
            self.setGlobalVariableByName('NDOF', NDOF)
            self._context._engine.fill_per_dof('ndof', float(NDOF))
            self.initialize()
            self._uninitialized = False
        return openmm.CustomIntegrator.step(self, steps)

    def initialize(self):
        """Hook for subclasses: initialise velocities / random per-DOF variables."""
        pass


class GlobalThermostatIntegrator(_AtomsMM_Integrator):
    """NVE propagator wrapped as T^(1/2) NVE T^(1/2) by a global thermostat (integrators.py:173-211)."""

    def __init__(self, stepSize, nveIntegrator, thermostat=None):
        super().__init__(stepSize)
        propagator = nveIntegrator if thermostat is None else propagators.TrotterSuzukiPropagator(nveIntegrator, thermostat)
        propagator.addVariables(self)
        propagator.addSteps(self)


class MultipleTimeScaleIntegrator(_AtomsMM_Integrator):
    """RESPA integrator: MultipleTimeScalePropagator(loops, move, boost, bath, **kwargs) over stepSize
    (integrators.py:214-269)."""

    def __init__(self, stepSize, loops, move=None, boost=None, bath=None, **kwargs):
        super().__init__(stepSize)
        propagator = propagators.MultipleTimeScalePropagator(loops, move, boost, bath, **kwargs)
        propagator.addVariables(self)
        propagator.addSteps(self)


class Langevin_R_Integrator(MultipleTimeScaleIntegrator):
    """Multiple time scale Langevin integrator (integrators.py:296-323): RESPA with an Ornstein-Uhlenbeck bath
    dv = -gamma v dt + sqrt(2 gamma kT/m) dW placed by `scheme` (default 'middle')."""

    def __init__(self, stepSize, loops, temperature, frictionConstant, **kwargs):
        bath = propagators.OrnsteinUhlenbeckPropagator(temperature, frictionConstant)
        super().__init__(stepSize, loops, None, None, bath, **kwargs)


class NHL_R_Integrator(MultipleTimeScaleIntegrator):
    """Massive Nose-Hoover-Langevin RESPA integrator (integrators.py:272-318): every DOF carries a thermostat
    velocity v2 (inertia Q2 = kT tau^2) driven by m v^2 - kT and by its own Ornstein-Uhlenbeck bath; v <- v exp(-v2 dt)."""

    def __init__(self, stepSize, loops, temperature, timeScale, frictionConstant, **kwargs):
        scaling = propagators.GenericScalingPropagator('v', 'v2')
        DOU = propagators.OrnsteinUhlenbeckPropagator(temperature, frictionConstant, 'v2', 'Q2', 'm*v^2 - kT',
                                                      Q2=kB * temperature * timeScale ** 2, kT=kB * temperature)
        bath = propagators.TrotterSuzukiPropagator(DOU, scaling)
        super().__init__(stepSize, loops, None, None, bath, **kwargs)

    def initialize(self):
        kT = self.getGlobalVariableByName('kT')
        Q2 = self.getGlobalVariableByName('Q2')
        v2 = self.getPerDofVariableByName('v2')
        S = math.sqrt(kT / Q2)
        for i in range(len(v2)):
            v2[i] = S * self._normalVec()
        self.setPerDofVariableByName('v2', v2)
