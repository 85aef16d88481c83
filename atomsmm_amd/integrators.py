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


def _is_force_symbol(name):
    """`f` or `f<group>`: the per-DOF force symbols of a CustomIntegrator expression."""
    return name == 'f' or (name[:1] == 'f' and name[1:].isdigit())


class _AtomsMM_Integrator(openmm.CustomIntegrator):
    def __init__(self, stepSize):
        super().__init__(stepSize)
        # declaration order is contract (the captured programs list `mvv, NDOF` first and `ndof` as the first per-DOF variable)
        for name in ('mvv', 'NDOF'):
            self.addGlobalVariable(name, 0.0)
        self.addPerDofVariable('ndof', 0.0)
        # bookkeeping of the automatic insertions (integrators.py:106-147): is the stored sum(m v^2) older than v?  has this program
        # already let the forces update the Context?  has step() run its one-time initialisation?
        self._state = dict(mvv_stale=True, context_stale=True, first_step=True)
        self._rng = np.random.RandomState()

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
        return openmm.Vec3(self._rng.normal(), self._rng.normal(), self._rng.normal())

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
        if self._state['mvv_stale'] and 'mvv' in requirements:
            openmm.CustomIntegrator.addComputeSum(self, 'mvv', 'm*v*v')
            self._state['mvv_stale'] = False
        if self._state['context_stale'] and any(_is_force_symbol(s) for s in requirements):
            openmm.CustomIntegrator.addUpdateContextState(self)
            self._state['context_stale'] = False

    def addUpdateContextState(self):
        if self._state['context_stale']:
            openmm.CustomIntegrator.addUpdateContextState(self)
            self._state['context_stale'] = False

    def addComputeGlobal(self, variable, expression):
        if variable == 'mvv':
            raise InputError('Cannot assign value to global variable mvv')
        self._checkUpdate(self._required_variables(variable, expression))
        return openmm.CustomIntegrator.addComputeGlobal(self, variable, expression)

    def addComputePerDof(self, variable, expression):
        requirements = self._required_variables(variable, expression)
        self._checkUpdate(requirements)
        forces = sorted(s for s in requirements if _is_force_symbol(s))
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
            self._state['mvv_stale'] = True
        return index

    def setRandomNumberSeed(self, seed):
        self._rng.seed(seed)
        openmm.CustomIntegrator.setRandomNumberSeed(self, self._rng.tomaxint() % 2 ** 31)

    def step(self, steps):
        if self._state['first_step']:
            if self._context is None:
                raise openmm.OpenMMException('This Integrator is not bound to a context!')
            self._ndof = NDOF = 3 * self._context.getSystem().getNumParticles()
            self.setGlobalVariableByName('NDOF', NDOF)
            self._context._engine.fill_per_dof('ndof', float(NDOF))
            self.initialize()
            self._state['first_step'] = False
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


class SIN_R_Integrator(MultipleTimeScaleIntegrator):
    """Stochastic-Iso-NH-RESPA integrator (integrators.py:358-416): see propagators.SIN_R_Propagator.  `initialize` draws
    the thermostat velocities, v1 ~ N(0, (L+1)/L kT/Q1) and v2 ~ N(0, kT/Q2) per degree of freedom."""

    def __init__(self, stepSize, loops, temperature, timeScale, frictionConstant, **kwargs):
        _AtomsMM_Integrator.__init__(self, stepSize)
        propagator = propagators.SIN_R_Propagator(loops, temperature, timeScale, frictionConstant, **kwargs)
        propagator.addVariables(self)
        propagator.addSteps(self)

    def initialize(self):
        kT = self.getGlobalVariableByName('kT')
        Q1 = self.getGlobalVariableByName('Q1')
        Q2 = self.getGlobalVariableByName('Q2')
        L = round(self.getGlobalVariableByName('L'))
        S1, S2 = math.sqrt((L + 1) / L * kT / Q1), math.sqrt(kT / Q2)
        for i in range(L):
            for name, S in (('v1_{}'.format(i), S1), ('v2_{}'.format(i), S2)):
                values = self.getPerDofVariableByName(name)
                for j in range(len(values)):
                    values[j] = S * self._normalVec()
                self.setPerDofVariableByName(name, values)


class ExtendedSystemVariable(object):
    """An extended-space variable of Adiabatic Free Energy Dynamics (integrators.py:642-744): a global Context
    parameter `name` (e.g. `lambda_vdw`) with mass, its own temperature kT and a Nose-Hoover or Langevin thermostat, moving
    between `lower_limit` and `upper_limit` (elastic walls, or periodic)."""

    def __init__(self, name, mass, kT, time_scale, lower_limit=0, upper_limit=1, periodic=False,
                 thermostat='Nose-Hoover', friction_constant=None):
        from . import unit
        self._m_value = mass
        self._kT_value = kT
        self._Q_eta_value = kT * time_scale ** 2
        self._lower_limit, self._upper_limit, self._periodic = lower_limit, upper_limit, periodic
        self._gamma_value = 0.1 / unit.femtoseconds if friction_constant is None else friction_constant
        self._thermostat = thermostat
        self._x = name
        self._v, self._m, self._kT = '_v_' + name, '_m_' + name, '_kT_' + name
        self._kTbym, self._v_eta, self._Q_eta, self._gamma = '_kTbym_' + name, '_v_eta_' + name, '_Q_eta_' + name, '_gamma_' + name

    def add_global_variables(self, integrator):
        """Velocity and mass of the variable, then the bath's own globals -- in the reference's declaration order
        (integrators.py:690-699: programs.json pins names and order)."""
        bath = {'Nose-Hoover': [(self._v_eta, 0.0), (self._kT, self._kT_value), (self._Q_eta, self._Q_eta_value)],
                'Langevin': [(self._kTbym, self._kT_value / self._m_value), (self._gamma, self._gamma_value)]}
        for name, value in [(self._v, 0.0), (self._m, self._m_value)] + bath.get(self._thermostat, []):
            integrator.addGlobalVariable(name, value)

    def _apply_boundary_conditions(self, integrator):
        above_lower = 'step({}-({}))'.format(self._x, self._lower_limit)
        below_upper = 'step({}-{})'.format(self._upper_limit, self._x)
        integrator.beginIfBlock('{}*{} = 0'.format(above_lower, below_upper))
        if self._periodic:
            L = self._upper_limit - self._lower_limit
            integrator.addComputeGlobal(self._x, '{} + select({},{},{})'.format(self._x, above_lower, -L, L))
        else:
            integrator.addComputeGlobal(self._x, 'select({},{},{})-{}'.format(above_lower, 2 * self._upper_limit,
                                                                              2 * self._lower_limit, self._x))
            integrator.addComputeGlobal(self._v, '-{}'.format(self._v))
        integrator.endBlock()

    def add_integration_steps(self, integrator):
        move = '{} + 0.5*dt*{}'.format(self._x, self._v)
        integrator.addComputeGlobal(self._x, move)
        self._apply_boundary_conditions(integrator)
        if self._thermostat == 'Nose-Hoover':
            kick = '{0} + 0.5*dt*({1}*{2}^2-{3})/{4}'.format(self._v_eta, self._m, self._v, self._kT, self._Q_eta)
            integrator.addComputeGlobal(self._v_eta, kick)
            integrator.addComputeGlobal(self._v, '{}*exp(-dt*{})'.format(self._v, self._v_eta))
            integrator.addComputeGlobal(self._v_eta, kick)
        elif self._thermostat == 'Langevin':
            integrator.addComputeGlobal(self._v, 'z*{}+sqrt((1-z*z)*{})*gaussian; z=exp(-dt*{})'.format(self._v, self._kTbym,
                                                                                                          self._gamma))
        integrator.addComputeGlobal(self._x, move)
        self._apply_boundary_conditions(integrator)

    def update_velocity(self, integrator, divisor):
        integrator.addComputeGlobal(self._v, '{} - 0.5*(dt/{})*deriv(energy,{})/{}'.format(self._v, divisor, self._x, self._m))

    def initialize(self, integrator):
        from .unit import md_value
        sigma_v = math.sqrt(md_value(self._kT_value) / md_value(self._m_value))
        integrator.setGlobalVariableByName(self._v, sigma_v * integrator._rng.normal())
        if self._thermostat == 'Nose-Hoover':
            sigma_v_eta = math.sqrt(md_value(self._kT_value) / md_value(self._Q_eta_value))
            integrator.setGlobalVariableByName(self._v_eta, sigma_v_eta * integrator._rng.normal())


class AdiabaticDynamicsIntegrator(_AtomsMM_Integrator):
    """Adiabatic Free Energy Dynamics (interface of integrators.py:747-860).  One step of size 2 n dt is

        [ kick lambda ; atoms over dt ; kick lambda ]^n     lambda move + bath     [ ... ]^n

    where "atoms over dt" replays the program of `custom_integrator` with every `dt` rewritten to (dt/(2n)) and the kicks
    of the extended variables use deriv(energy, lambda).  The emitted program for RespaPropagator([2,1]), n = 2, is the
    44-step capture of SURVEY.md Appendix C.4 (tests/test_host_api.py)."""

    _SKIPPED_GLOBALS = frozenset(('mvv', 'NDOF'))       # owned by _AtomsMM_Integrator itself
    _SKIPPED_PER_DOF = frozenset(('ndof',))

    def __init__(self, custom_integrator, nsteps, variables):
        super().__init__(2 * nsteps * custom_integrator.getStepSize())
        self._variables = list(variables)
        self._nsteps = int(nsteps)
        self._counter = '_nsteps_counter' if self._nsteps > 1 else None
        if self._counter:
            self.addGlobalVariable(self._counter, 0)
        for variable in self._variables:
            variable.add_global_variables(self)
        self._adopt_state_of(custom_integrator)
        # the inner program, captured once: (step kind, variable, expression with dt -> dt/(2n))
        substep = '(dt/{})'.format(2 * self._nsteps)
        self._inner_program = [(kind, name, re.sub(r'\bdt\b', substep, text))
                               for kind, name, text in (custom_integrator.getComputationStep(k)
                                                        for k in range(custom_integrator.getNumComputations()))]
        self.addUpdateContextState()
        self._emit_half()
        for variable in self._variables:
            variable.add_integration_steps(self)
        self._emit_half()

    def _emit_half(self):
        """[kick ; inner program ; kick] repeated n times (a while-loop over the counter when n > 1)."""
        if self._counter:
            self.addComputeGlobal(self._counter, '0')
            self.beginWhileBlock('{} < {}'.format(self._counter, self._nsteps))
        self._kick_variables()
        self._replay_inner_program()
        self._kick_variables()
        if self._counter:
            self.addComputeGlobal(self._counter, '{} + 1'.format(self._counter))
            self.endBlock()

    def _kick_variables(self):
        for variable in self._variables:
            variable.update_velocity(self, 2 * self._nsteps)

    def _replay_inner_program(self):
        C = openmm.CustomIntegrator
        emit = {C.ComputeGlobal: lambda name, text: self.addComputeGlobal(name, text),
                C.ComputePerDof: lambda name, text: self.addComputePerDof(name, text),
                C.ComputeSum: lambda name, text: self.addComputeSum(name, text),
                C.ConstrainPositions: lambda name, text: self.addConstrainPositions(),
                C.ConstrainVelocities: lambda name, text: self.addConstrainVelocities(),
                C.UpdateContextState: lambda name, text: self.addUpdateContextState(),
                C.IfBlock: lambda name, text: self.beginIfBlock(text),
                C.WhileBlock: lambda name, text: self.beginWhileBlock(text),
                C.EndBlock: lambda name, text: self.endBlock()}
        for kind, name, text in self._inner_program:
            emit[kind](name, text)

    def _adopt_state_of(self, integrator):
        """Globals, per-DOF variables and the initialisation hook of the wrapped integrator become this one's."""
        for k in range(integrator.getNumGlobalVariables()):
            name = integrator.getGlobalVariableName(k)
            if name not in self._SKIPPED_GLOBALS:
                self.addGlobalVariable(name, integrator.getGlobalVariable(k))
        for k in range(integrator.getNumPerDofVariables()):
            name = integrator.getPerDofVariableName(k)
            if name not in self._SKIPPED_PER_DOF:
                self.addPerDofVariable(name, 0)
        self._inner_initialize = getattr(type(integrator), 'initialize', None)

    def initialize(self):
        if self._inner_initialize is not None:
            self._inner_initialize(self)
        for variable in self._variables:
            variable.initialize(self)
