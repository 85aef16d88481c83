"""Step-program builders with the class names and signatures of `atomsmm.propagators`
(reference: src/atomsmm/propagators.py), restricted to the hot path: the move / boost primitives,
the composition schemes and the RESPA multiple-timescale propagators.

A propagator does no arithmetic.  `addSteps(integrator, fraction, force)` appends CustomIntegrator
computations ("v <- v + (0.0625*dt)*(f0)/m" ...) to an integrator object; `atomsmm_amd.engine`
later unrolls that program into kick / move / copy / force-group-evaluation ops for the HIP library
(include/atomsmm_hip.h, amm_run_ops).  The emitted text is identical to the reference's
(tests/golden/goldens.json -> programs), which is the cheapest parity check there is.

The basic thermostat propagators are here too -- unconstrained velocity Verlet, stochastic
velocity rescaling, global and massive Nose-Hoover, Ornstein-Uhlenbeck / Langevin, generic boost and scaling; their
programs contain ComputeSum steps, random numbers and general per-DOF expressions, which the engine runs through
`amm_expr_eval` (atomsmm_amd/expr.py).  The 'regulated', 'limited-speed' and isokinetic families (reference :276-682,
:1452-2172) are not restated.
"""
import math

from . import unit
from .utils import InputError, kB


class Propagator:
    """Base class: global / per-DOF variable tables and `integrator(stepSize)` (propagators.py:24-75)."""

    def __init__(self):
        self.globalVariables = dict()
        self.perDofVariables = dict()

    def addVariables(self, integrator):
        for name, value in self.globalVariables.items():
            integrator.addGlobalVariable(name, value)
        for name, value in self.perDofVariables.items():
            integrator.addPerDofVariable(name, value)

    def absorbVariables(self, propagator):
        for table, label in ((propagator.globalVariables, 'Global'), (propagator.perDofVariables, 'Per-dof')):
            mine = self.globalVariables if label == 'Global' else self.perDofVariables
            for key, value in table.items():
                if key in mine and value != mine[key]:
                    raise InputError('{} variable inconsistency in merged propagators'.format(label))
        self.globalVariables.update(propagator.globalVariables)
        self.perDofVariables.update(propagator.perDofVariables)

    def addSteps(self, integrator, fraction=1.0, force='f'):
        pass

    def integrator(self, stepSize):
        """An `_AtomsMM_Integrator` that carries out this propagator over `stepSize`."""
        from .integrators import _AtomsMM_Integrator
        integrator = _AtomsMM_Integrator(stepSize)
        self.addVariables(integrator)
        self.addSteps(integrator)
        return integrator


class ChainedPropagator(Propagator):
    """C = A B ...: the listed propagators one after another (propagators.py:78-114)."""

    def __init__(self, propagators):
        super().__init__()
        self.propagators = propagators
        for propagator in propagators:
            self.absorbVariables(propagator)

    def addSteps(self, integrator, fraction=1.0, force='f'):
        for propagator in self.propagators:
            propagator.addSteps(integrator, fraction, force)


class SplitPropagator(Propagator):
    """A over dt as n applications of A over dt/n, emitted as a while-block on the global `nSplit`
    (propagators.py:117-149; like the reference, the `force` argument is not forwarded when n > 1)."""

    def __init__(self, A, n):
        super().__init__()
        self.A = A
        self.absorbVariables(A)
        self.n = n
        self.globalVariables['nSplit'] = 0

    def addSteps(self, integrator, fraction=1.0, force='f'):
        if self.n == 1:
            self.A.addSteps(integrator, fraction, force)
            return
        integrator.addComputeGlobal('nSplit', '0')
        integrator.beginWhileBlock('nSplit < {}'.format(self.n))
        self.A.addSteps(integrator, fraction / self.n)
        integrator.addComputeGlobal('nSplit', 'nSplit + 1')
        integrator.endBlock()


class TrotterSuzukiPropagator(Propagator):
    """C = B^(1/2) A B^(1/2)  (propagators.py:152-187)."""

    def __init__(self, A, B):
        super().__init__()
        self.A = A
        self.B = B
        self.absorbVariables(A)
        self.absorbVariables(B)

    def addSteps(self, integrator, fraction=1.0, force='f'):
        self.B.addSteps(integrator, 0.5 * fraction, force)
        self.A.addSteps(integrator, fraction, force)
        self.B.addSteps(integrator, 0.5 * fraction, force)


class SuzukiYoshidaPropagator(Propagator):
    """High-order symmetric factorisation with nsy in {1, 3, 7, 15} weights (propagators.py:190-226)."""

    _HALF_WEIGHTS = {
        15: [0.9148442462, 0.2536933366, -1.4448522369, -0.1582406354, 1.9381391376, -1.960610233, 0.1027998494],
        7: [0.784513610477560, 0.235573213359357, -1.17767998417887],
        3: [1.3512071919596578],
        1: [],
    }

    def __init__(self, A, nsy=3):
        super().__init__()
        if nsy not in self._HALF_WEIGHTS:
            raise InputError('SuzukiYoshidaPropagator accepts nsy = 1, 3, 7, or 15 only')
        self.A = A
        self.nsy = nsy
        self.absorbVariables(A)

    def addSteps(self, integrator, fraction=1.0, force='f'):
        half = self._HALF_WEIGHTS[self.nsy]
        for w in half + [1 - 2 * sum(half)] + half[::-1]:
            self.A.addSteps(integrator, fraction * w)


class TranslationPropagator(Propagator):
    """x <- x + (fraction*dt)*v ; the constrained variant saves x0, constrains positions and rebuilds
    v from the displacement (propagators.py:229-252)."""

    def __init__(self, constrained=True):
        super().__init__()
        self.constrained = constrained
        if constrained:
            self.perDofVariables['x0'] = 0

    def addSteps(self, integrator, fraction=1.0, force='f'):
        if self.constrained:
            integrator.addComputePerDof('x0', 'x')
        integrator.addComputePerDof('x', 'x + ({}*dt)*v'.format(fraction))
        if self.constrained:
            integrator.addConstrainPositions()
            integrator.addComputePerDof('v', '(x - x0)/({}*dt)'.format(fraction))


class VelocityBoostPropagator(Propagator):
    """v <- v + (fraction*dt)*force/m (+ velocity constraints)  (propagators.py:255-273)."""

    def __init__(self, constrained=True):
        super().__init__()
        self.constrained = constrained

    def addSteps(self, integrator, fraction=1.0, force='f'):
        integrator.addComputePerDof('v', 'v + ({}*dt)*{}/m'.format(fraction, force))
        if self.constrained:
            integrator.addConstrainVelocities()


class RespaPropagator(Propagator):
    """rRESPA with N force groups; group 0 in the innermost loop (propagators.py:830-973).

    loops[k] = iterations of level k per iteration of level k+1.  Level k kicks with force expression
    f0, f1, f2-f1, f3-f2, ... (the near force is *subtracted inside the integrator*).  Optional
    `core` (between two half moves) and `shell` {level: propagator} baths; keyword flags
    has_memory (default False, as in the reference's code), use_respa_switch, blitz.
    """

    def __init__(self, loops, move=None, boost=None, core=None, shell=None, **kwargs):
        super().__init__()
        self.loops = loops
        self.N = len(loops)
        self.move = move if move is not None else TranslationPropagator(constrained=False)
        self.boost = boost if boost is not None else VelocityBoostPropagator(constrained=False)
        self.core = core
        if shell is None:
            self.shell = dict()
        elif set(shell.keys()).issubset(range(self.N)):
            self.shell = shell
        else:
            raise InputError('invalid key(s) in RespaPropagator \'shell\' argument')
        for member in [self.move, self.boost, self.core] + list(self.shell.values()):
            if member is not None:
                self.absorbVariables(member)
        for level, n in enumerate(loops):
            if n > 1:
                self.globalVariables['n{}RESPA'.format(level)] = 0
        self.expr = ['f{}'.format(level) for level in range(self.N)]
        for level in range(2, self.N):
            self.expr[level] += '-f{}'.format(level - 1)
        self.force = list(self.expr)
        self._has_memory = kwargs.pop('has_memory', False)
        if self._has_memory:
            for level in range(1, self.N):
                self.perDofVariables['fm{}'.format(level)] = 0.0
                self.force[0] += '+fm{}'.format(level)
                self.force[level] += '-fm{}'.format(level)
        self.force = ['({})'.format(f) for f in self.force]
        self._use_respa_switch = kwargs.pop('use_respa_switch', False)
        self._blitz = kwargs.pop('blitz', False)

    def addSteps(self, integrator, fraction=1.0, force='f'):
        if self._use_respa_switch:
            integrator.addComputeGlobal('respa_switch', '1')
        self._addSubsteps(integrator, self.N - 1, fraction)
        if self._use_respa_switch:
            integrator.addComputeGlobal('respa_switch', '0')

    def _internalSplitting(self, integrator, timescale, fraction, shell):
        remembered = self._has_memory and timescale > 0
        if self._blitz:
            if remembered:
                integrator.addComputePerDof('F{}'.format(timescale), 'f{}'.format(timescale))
            else:
                self.boost.addSteps(integrator, fraction, self.force[timescale])
            self._addSubsteps(integrator, timescale - 1, fraction)
            return
        if shell:
            shell.addSteps(integrator, 0.5 * fraction, self.force[timescale])
        if remembered:
            integrator.addComputePerDof('fm{}'.format(timescale), self.expr[timescale])
        else:
            self.boost.addSteps(integrator, 0.5 * fraction, self.force[timescale])
        self._addSubsteps(integrator, timescale - 1, fraction)
        self.boost.addSteps(integrator, 0.5 * fraction, self.force[timescale])
        if shell:
            shell.addSteps(integrator, 0.5 * fraction, self.force[timescale])

    def _addSubsteps(self, integrator, timescale, fraction):
        if timescale < 0:
            if self.core is None:
                self.move.addSteps(integrator, fraction)
            else:
                self.move.addSteps(integrator, 0.5 * fraction)
                self.core.addSteps(integrator, fraction)
                self.move.addSteps(integrator, 0.5 * fraction)
            return
        n = self.loops[timescale]
        counter = 'n{}RESPA'.format(timescale)
        if n > 1:
            integrator.addComputeGlobal(counter, '0')
            integrator.beginWhileBlock('{} < {}'.format(counter, n))
        self._internalSplitting(integrator, timescale, fraction / n, self.shell.get(timescale, None))
        if n > 1:
            integrator.addComputeGlobal(counter, '{} + 1'.format(counter))
            integrator.endBlock()


class MultipleTimeScalePropagator(RespaPropagator):
    """RESPA with a bath placed by `scheme` in {middle, blitz, xi-respa, xo-respa, side}; the bath may be
    factorised by `nres` (SplitPropagator) and `nsy` (Suzuki-Yoshida)  (propagators.py:976-1042)."""

    def __init__(self, loops, move=None, boost=None, bath=None, **kwargs):
        scheme = kwargs.pop('scheme', 'middle')
        location = kwargs.pop('location', 0)
        nres = kwargs.pop('nres', 1)
        nsy = kwargs.pop('nsy', 1)
        if nres > 1:
            bath = SplitPropagator(bath, nres)
        if nsy > 1:
            bath = SuzukiYoshidaPropagator(bath, nsy)
        if scheme == 'middle':
            super().__init__(loops, move=move, boost=boost, core=bath, **kwargs)
        elif scheme == 'blitz':
            super().__init__(loops, move=move, boost=boost, core=bath, blitz=True, **kwargs)
        elif scheme in ('xi-respa', 'xo-respa', 'side'):
            level = {'side': location, 'xi-respa': 0, 'xo-respa': len(loops) - 1}[scheme]
            super().__init__(loops, move=move, boost=boost, shell={level: bath}, **kwargs)
        else:
            raise InputError('wrong value of scheme parameter')


class VelocityVerletPropagator(Propagator):
    """Velocity Verlet with constraints (propagators.py:1108-1133)."""

    def __init__(self):
        super().__init__()
        self.perDofVariables['x0'] = 0

    def addSteps(self, integrator, fraction=1.0, force='f'):
        Dt = '; Dt=%s*dt' % fraction
        integrator.addComputePerDof('v', 'v+0.5*Dt*f/m' + Dt)
        integrator.addComputePerDof('x0', 'x')
        integrator.addComputePerDof('x', 'x+Dt*v' + Dt)
        integrator.addConstrainPositions()
        integrator.addComputePerDof('v', '(x-x0)/Dt+0.5*Dt*f/m' + Dt)
        integrator.addConstrainVelocities()


class UnconstrainedVelocityVerletPropagator(Propagator):
    """Velocity Verlet without constraints (propagators.py:1136-1153)."""

    def addSteps(self, integrator, fraction=1.0, force='f'):
        integrator.addComputePerDof('v', 'v+0.5*{}*dt*f/m'.format(fraction))
        integrator.addComputePerDof('x', 'x+{}*dt*v'.format(fraction))
        integrator.addComputePerDof('v', 'v+0.5*{}*dt*f/m'.format(fraction))


class VelocityRescalingPropagator(Propagator):
    """Stochastic velocity rescaling of Bussi, Donadio and Parrinello (propagators.py:1156-1227): a gamma-distributed
    sum of squared Gaussians by Marsaglia-Tsang rejection in the global variables, then `v <- vscaling*v`."""

    def __init__(self, temperature, degreesOfFreedom, timeScale):
        super().__init__()
        self.tau = unit.md_value(timeScale)
        self.dof = degreesOfFreedom
        self.kT = unit.md_value(kB * temperature)
        for name in ('V', 'X', 'U', 'ready'):
            self.globalVariables[name] = 0

    def addSteps(self, integrator, fraction=1.0, force='f'):
        a = (self.dof - 2 + self.dof % 2) / 2
        d = a - 1 / 3
        c = 1 / math.sqrt(9 * d)
        integrator.addComputeGlobal('ready', '0')
        integrator.beginWhileBlock('ready < 0.5')
        integrator.addComputeGlobal('X', 'gaussian')
        integrator.addComputeGlobal('V', '1+%s*X' % c)
        integrator.beginWhileBlock('V <= 0.0')
        integrator.addComputeGlobal('X', 'gaussian')
        integrator.addComputeGlobal('V', '1+%s*X' % c)
        integrator.endBlock()
        integrator.addComputeGlobal('V', 'V^3')
        integrator.addComputeGlobal('U', 'random')
        integrator.addComputeGlobal('ready', 'step(1-0.0331*X^4-U)')
        integrator.beginIfBlock('ready < 0.5')
        integrator.addComputeGlobal('ready', 'step(0.5*X^2+%s*(1-V+log(V))-log(U))' % d)
        integrator.endBlock()
        integrator.endBlock()
        odd = self.dof % 2 == 1
        if odd:
            integrator.addComputeGlobal('X', 'gaussian')
        pieces = ['vscaling*v',
                  'vscaling = sqrt(A+C*B*(gaussian^2+sumRs)+2*sqrt(C*B*A)*gaussian)',
                  'C = %s/mvv' % self.kT,
                  'B = 1-A',
                  'A = exp(-dt*%s)' % (fraction / self.tau),
                  'sumRs = %s*V' % (2 * d) + ('+X^2' if odd else '')]
        integrator.addComputePerDof('v', '; '.join(pieces))


class NoseHooverPropagator(Propagator):
    """Global Nose-Hoover thermostat with `nloops` RESPA-like subdivisions (propagators.py:1230-1273)."""

    def __init__(self, temperature, degreesOfFreedom, timeScale, nloops=1):
        super().__init__()
        self.nloops = nloops
        self.globalVariables['LkT'] = degreesOfFreedom * kB * temperature
        self.globalVariables['Q'] = degreesOfFreedom * kB * temperature * timeScale ** 2
        for name in ('vscaling', 'p_eta', 'n_NH'):
            self.globalVariables[name] = 0

    def addSteps(self, integrator, fraction=1.0, force='f'):
        n = self.nloops
        subfrac = fraction / n
        integrator.addComputeGlobal('p_eta', 'p_eta + ({}*dt)*(mvv - LkT)'.format(0.5 * subfrac))
        integrator.addComputeGlobal('vscaling', 'exp(-({}*dt)*p_eta/Q)'.format(subfrac))
        if n > 2:
            integrator.addComputeGlobal('n_NH', '1')
            integrator.beginWhileBlock('n_NH < {}'.format(n))
            integrator.addComputeGlobal('p_eta', 'p_eta + ({}*dt)*(vscaling^2*mvv - LkT)'.format(subfrac))
            integrator.addComputeGlobal('vscaling', 'vscaling*exp(-({}*dt)*p_eta/Q)'.format(subfrac))
            integrator.addComputeGlobal('n_NH', 'n_NH + 1')
            integrator.endBlock()
        integrator.addComputeGlobal('p_eta', 'p_eta + ({}*dt)*(vscaling^2*mvv - LkT)'.format(0.5 * subfrac))
        integrator.addComputePerDof('v', 'vscaling*v')


class MassiveNoseHooverPropagator(Propagator):
    """One Nose-Hoover thermostat per degree of freedom (propagators.py:1276-1311)."""

    def __init__(self, temperature, timeScale, nloops=1):
        super().__init__()
        self.nloops = nloops
        self.globalVariables['kT'] = kB * temperature
        self.globalVariables['Q'] = kB * temperature * timeScale ** 2
        self.globalVariables['nMNH'] = 0
        self.perDofVariables['p_eta'] = 0

    def addSteps(self, integrator, fraction=1.0, force='f'):
        subfrac = fraction / self.nloops
        if self.nloops > 1:
            integrator.addComputeGlobal('nMNH', '0')
            integrator.beginWhileBlock('nMNH < {}'.format(self.nloops))
        integrator.addComputePerDof('p_eta', 'p_eta + ({}*dt)*(m*v^2 - kT)'.format(0.5 * subfrac))
        integrator.addComputePerDof('v', 'v*exp(-({}*dt)*p_eta/Q)'.format(subfrac))
        integrator.addComputePerDof('p_eta', 'p_eta + ({}*dt)*(m*v^2 - kT)'.format(0.5 * subfrac))
        if self.nloops > 1:
            integrator.addComputeGlobal('nMNH', 'nMNH + 1')
            integrator.endBlock()


class MassiveGeneralizedGaussianMomentPropagator(Propagator):
    """One generalized-Gaussian-moment thermostat per degree of freedom (propagators.py:1314-1359): two thermostat momenta
    p1 (driven by m v^2 - kT, inertia Q1 = kT tau^2) and p2 (driven by m^2 v^4/3 - kT^2, inertia Q2 = 2 kT^3 tau^2);
    the velocity step is a scaling, the exact solution of dv/dt = -alpha' v^3, and the scaling again."""

    def __init__(self, temperature, timeScale, nloops=1):
        super().__init__()
        self.nloops = nloops
        self.globalVariables['kT'] = kB * temperature
        self.globalVariables['Q1'] = kB * temperature * timeScale ** 2
        self.globalVariables['Q2'] = 2 * (kB * temperature) ** 3 * timeScale ** 2
        self.globalVariables['nGGM'] = 0
        self.perDofVariables['p1'] = 0
        self.perDofVariables['p2'] = 0

    def addSteps(self, integrator, fraction=1.0, force='f'):
        subfrac = fraction / self.nloops
        half = subfrac / 2
        boost1 = 'p1 + ({}*dt)*(m*v^2 - kT)'.format(half)
        boost2 = 'p2 + ({}*dt)*(m^2*v^4/3 - kT^2)'.format(half)
        scaling = 'exp(-{}*dt*(p1/Q1 + kT*p2/Q2))'.format(half)
        velocity = ['v2*{}'.format(scaling), 'v2 = v1/sqrt(1 + 2*v1^2*alpha*{}*dt)'.format(subfrac), 'alpha = p2/(3*m*Q2)',
                    'v1 = v*{}'.format(scaling)]
        if self.nloops > 1:
            integrator.addComputeGlobal('nGGM', '0')
            integrator.beginWhileBlock('nGGM < {}'.format(self.nloops))
        integrator.addComputePerDof('p1', boost1)
        integrator.addComputePerDof('p2', boost2)
        integrator.addComputePerDof('v', ';'.join(velocity))
        integrator.addComputePerDof('p2', boost2)
        integrator.addComputePerDof('p1', boost1)
        if self.nloops > 1:
            integrator.addComputeGlobal('nGGM', 'nGGM + 1')
            integrator.endBlock()


class _TwoStageGlobalThermostat(Propagator):
    """Shared by the Nose-Hoover chain and Nose-Hoover-Langevin propagators: the constants of the global thermostat
    written into the program text as numbers (kT, N kT, Q = N kT tau^2), default friction 1/tau."""

    def __init__(self, temperature, degreesOfFreedom, timeScale, frictionConstant=None):
        super().__init__()
        self.temperature, self.degreesOfFreedom, self.timeScale = temperature, degreesOfFreedom, timeScale
        self.frictionConstant = 1 / timeScale if frictionConstant is None else frictionConstant
        self.globalVariables['vscaling'] = 0

    def _constants(self):
        kT = unit.md_value(kB * self.temperature)
        NkT = self.degreesOfFreedom * kT
        tau = unit.md_value(self.timeScale)
        return kT, NkT, tau


class NoseHooverChainPropagator(_TwoStageGlobalThermostat):
    """Nose-Hoover chain of two global thermostats (propagators.py:1362-1449), Q1 = N kT tau^2, Q2 = kT tau^2, split as
    B2 S1 B1 S B1 S1 B2: boost of thermostat 2, scaling of thermostat 1, boost of thermostat 1 by mvv - N kT, scaling of
    the particle velocities."""

    def __init__(self, temperature, degreesOfFreedom, timeScale, frictionConstant=None):
        super().__init__(temperature, degreesOfFreedom, timeScale, frictionConstant)
        self.globalVariables['p_NHC_1'] = 0
        self.globalVariables['p_NHC_2'] = 0

    def addSteps(self, integrator, fraction=1.0, force='f'):
        kT, NkT, tau = self._constants()
        Q1, Q2 = NkT * tau ** 2, kT * tau ** 2
        half = 0.5 * fraction
        boost2 = 'p_NHC_2 + (p_NHC_1^2/{}-{})*{}*dt'.format(Q1, kT, half)
        scale1 = 'p_NHC_1*exp(-{}*p_NHC_2*dt)'.format(half / Q2)
        integrator.addComputeGlobal('p_NHC_2', boost2)
        integrator.addComputeGlobal('p_NHC_1', scale1)
        integrator.addComputeGlobal('p_NHC_1', 'p_NHC_1 + (mvv-{})*{}*dt'.format(NkT, half))
        integrator.addComputeGlobal('vscaling', 'exp(-{}*p_NHC_1*dt)'.format(fraction / Q1))
        integrator.addComputeGlobal('p_NHC_1', 'p_NHC_1 + (vscaling^2*mvv-{})*{}*dt'.format(NkT, half))
        integrator.addComputeGlobal('p_NHC_1', scale1)
        integrator.addComputeGlobal('p_NHC_2', boost2)
        integrator.addComputePerDof('v', 'vscaling*v')


class NoseHooverLangevinPropagator(_TwoStageGlobalThermostat):
    """Nose-Hoover-Langevin (propagators.py:1452-1536): one global thermostat momentum p_NHL (Q = N kT tau^2) boosted by
    mvv - N kT, with an Ornstein-Uhlenbeck step of its own in the middle; the velocities are scaled by the product of
    the two half scalings."""

    def __init__(self, temperature, degreesOfFreedom, timeScale, frictionConstant=None):
        super().__init__(temperature, degreesOfFreedom, timeScale, frictionConstant)
        self.globalVariables['p_NHL'] = 0

    def addSteps(self, integrator, fraction=1.0, force='f'):
        kT, NkT, tau = self._constants()
        Q = NkT * tau ** 2
        gamma = unit.md_value(self.frictionConstant)
        half = 0.5 * fraction
        integrator.addComputeGlobal('p_NHL', 'p_NHL + (mvv-{})*{}*dt'.format(NkT, half))
        integrator.addComputeGlobal('vscaling', 'exp(-{}*p_NHL*dt)'.format(half / Q))
        integrator.addComputeGlobal('p_NHL', 'p_NHL*x + sqrt({}*(1-x^2))*gaussian; x = exp(-{}*dt)'.format(kT / Q, gamma * fraction))
        integrator.addComputeGlobal('vscaling', 'vscaling*exp(-{}*p_NHL*dt)'.format(half / Q))
        integrator.addComputeGlobal('p_NHL', 'p_NHL + (vscaling^2*mvv-{})*{}*dt'.format(NkT, half))
        integrator.addComputePerDof('v', 'vscaling*v')


class OrnsteinUhlenbeckPropagator(Propagator):
    """Exact solution of dV = (F/M) dt - gamma V dt + sqrt(2 gamma kT/M) dW per degree of freedom
    (propagators.py:685-741): the Langevin bath of the 'middle' schemes."""

    def __init__(self, temperature, frictionConstant, velocity='v', mass='m', force=None, overall=False, **globals):
        super().__init__()
        self.globalVariables['kT'] = kB * temperature
        self.globalVariables['friction'] = frictionConstant
        self.velocity, self.mass, self.force, self.overall = velocity, mass, force, overall
        for key, value in globals.items():
            self.globalVariables[key] = value
        if velocity != 'v':
            (self.globalVariables if overall else self.perDofVariables)[velocity] = 0

    def addSteps(self, integrator, fraction=1.0, force='f'):
        expression = 'z*{} + sqrt(kT*(1 - z*z)/mass)*gaussian'.format(self.velocity, self.mass)
        if self.force is not None:
            expression += ' + force*(1 - z)/(mass*friction)'
            expression += '; force = {}'.format(self.force)
        expression += '; mass = {}'.format(self.mass)
        expression += '; z = exp(-({}*dt)*friction)'.format(fraction)
        if self.overall:
            integrator.addComputeGlobal(self.velocity, expression)
        else:
            integrator.addComputePerDof(self.velocity, expression)


class MassiveIsokineticPropagator(Propagator):
    """Unconstrained massive isokinetic propagator (propagators.py:276-355).  Every degree of freedom carries L thermostat
    velocities v1_i (inertia Q1 = kT tau^2) and obeys m v^2 + L/(L+1) Q1 sum_i v1_i^2 = L kT.  `forceDependent`: the exact
    solution of dv/dt = F/m - lambda_F v (v <- v cosh z + sqrt(LkT/m) sinh z, z = F t / sqrt(m LkT)); otherwise of the
    thermostat coupling v1_i <- v1_i exp(-v2_i t).  Both are followed by the rescaling H that restores the constraint."""

    def __init__(self, temperature, timeScale, L, forceDependent):
        super().__init__()
        self.L, self.forceDependent = L, forceDependent
        self.globalVariables['Q1'] = kB * temperature * timeScale ** 2
        self.globalVariables['L'] = L
        self.globalVariables['LkT'] = L * kB * temperature
        for i in range(L):
            self.perDofVariables['v1_{}'.format(i)] = 1 / timeScale
            self.perDofVariables['v2_{}'.format(i)] = 0
        self.perDofVariables['H'] = 0

    def addSteps(self, integrator, fraction=1.0, force='f'):
        v1 = ['v1_{}'.format(i) for i in range(self.L)]
        v2 = ['v2_{}'.format(i) for i in range(self.L)]
        if self.forceDependent:
            integrator.addComputePerDof('v', 'v*cosh(z) + sqrt(LkT/m)*sinh(z); z = ({}*dt)*{}/sqrt(m*LkT)'.format(fraction, force))
        else:
            for a, b in zip(v1, v2):
                integrator.addComputePerDof(a, '{}*exp(-({}*dt)*{})'.format(a, fraction, b))
        squares = '+'.join('{}^2'.format(a) for a in v1)
        integrator.addComputePerDof('H', 'sqrt(LkT/(m*v^2 + {}*Q1*({})))'.format(self.L / (self.L + 1), squares))
        integrator.addComputePerDof('v', 'H*v')
        for a in v1:
            integrator.addComputePerDof(a, 'H*{}'.format(a))


class SIN_R_Propagator(MultipleTimeScalePropagator):
    """Stochastic-Iso-NH-RESPA, SIN(R), of Leimkuhler, Margul and Tuckerman (propagators.py:1045-1105): the RESPA kicks
    are the force-dependent isokinetic propagator; the bath -- an Ornstein-Uhlenbeck process on every v2_i driven by
    Q1 v1_i^2 - kT (as part of the OU step, or as a separate boost with `split=True`), Trotter-split around the
    force-independent isokinetic propagator -- sits where `scheme` puts it.  Keywords L (thermostats per DOF, default 1)
    and split, then those of MultipleTimeScalePropagator."""

    def __init__(self, loops, temperature, timeScale, frictionConstant, **kwargs):
        L = kwargs.pop('L', 1)
        split = kwargs.pop('split', False)
        Q2 = kB * temperature * timeScale ** 2
        with_force = MassiveIsokineticPropagator(temperature, timeScale, L, forceDependent=True)
        without_force = MassiveIsokineticPropagator(temperature, timeScale, L, forceDependent=False)
        drive = ['Q1*v1_{}^2 - kT'.format(i) for i in range(L)]
        baths = [OrnsteinUhlenbeckPropagator(temperature, frictionConstant, 'v2_{}'.format(i), 'Q2',
                                             None if split else drive[i], Q2=Q2) for i in range(L)]
        DOU = ChainedPropagator(baths)
        if split:
            boosts = ChainedPropagator([GenericBoostPropagator('v2_{}'.format(i), 'Q2', drive[i], Q2=Q2) for i in range(L)])
            DOU = TrotterSuzukiPropagator(DOU, boosts)
        super().__init__(loops, None, with_force, TrotterSuzukiPropagator(DOU, without_force), **kwargs)


class GenericBoostPropagator(Propagator):
    """dV/dt = F/M for a named velocity / mass / force triple, per-DOF or global (propagators.py:744-790)."""

    def __init__(self, velocity='v', mass='m', force='f', perDof=True, **globals):
        super().__init__()
        self.velocity, self.mass, self.force, self.perDof = velocity, mass, force, perDof
        for key, value in globals.items():
            self.globalVariables[key] = value
        if velocity != 'v':
            (self.perDofVariables if perDof else self.globalVariables)[velocity] = 0

    def addSteps(self, integrator, fraction=1.0, force='f'):
        expression = '{} + ({}*dt)*F/M'.format(self.velocity, fraction)
        expression += '; F = {}'.format(self.force)
        expression += '; M = {}'.format(self.mass)
        (integrator.addComputePerDof if self.perDof else integrator.addComputeGlobal)(self.velocity, expression)


class GenericScalingPropagator(Propagator):
    """dV/dt = -damping*V for a named velocity and damping variable (propagators.py:793-827)."""

    def __init__(self, velocity, damping, perDof=True, **globals):
        super().__init__()
        self.velocity, self.damping, self.perDof = velocity, damping, perDof
        for key, value in globals.items():
            self.globalVariables[key] = value
        if perDof and velocity != 'v':
            self.perDofVariables[velocity] = 0
        elif not perDof:
            self.globalVariables[velocity] = 0

    def addSteps(self, integrator, fraction=1.0, force='f'):
        expression = '{}*exp(-({}*dt)*{})'.format(self.velocity, fraction, self.damping)
        (integrator.addComputePerDof if self.perDof else integrator.addComputeGlobal)(self.velocity, expression)
