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


def play(integrator, template, **values):
    """Append the steps a program template describes.  One step per line, `{name}` fields filled from `values`:

        dof v <- v + ({h}*dt)*{F}/m        per-DOF computation          global X <- gaussian        global computation
        sum mvv <- m*v*v                    sum over the DOFs            while V <= 0.0 / if ... / end    blocks
        constrain positions / constrain velocities

    The contract with the reference is the emitted program TEXT (tests/golden/goldens.json:
    programs); the propagators below state that text as templates instead of as call sequences."""
    for raw in template.strip().splitlines():
        line = raw.strip().format(**values)
        if not line:
            continue
        word, _, rest = line.partition(' ')
        if word in ('dof', 'global', 'sum'):
            name, _, expression = rest.partition(' <- ')
            {'dof': integrator.addComputePerDof, 'global': integrator.addComputeGlobal,
             'sum': integrator.addComputeSum}[word](name.strip(), expression.strip())
        elif word == 'while':
            integrator.beginWhileBlock(rest)
        elif word == 'if':
            integrator.beginIfBlock(rest)
        elif line == 'end':
            integrator.endBlock()
        elif line == 'constrain positions':
            integrator.addConstrainPositions()
        elif line == 'constrain velocities':
            integrator.addConstrainVelocities()
        else:
            raise ValueError('unknown step in program template: ' + line)


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
    """The move x <- x + (fraction*dt)*v.  With constraints the start is kept in x0, the positions are constrained and the
    velocity is rebuilt from the displacement actually made (interface of propagators.py:229-252)."""

    FREE = 'dof x <- x + ({h}*dt)*v'
    CONSTRAINED = """
        dof x0 <- x
        dof x <- x + ({h}*dt)*v
        constrain positions
        dof v <- (x - x0)/({h}*dt)
    """

    def __init__(self, constrained=True):
        super().__init__()
        self.constrained = constrained
        if constrained:
            self.perDofVariables['x0'] = 0

    def addSteps(self, integrator, fraction=1.0, force='f'):
        play(integrator, self.CONSTRAINED if self.constrained else self.FREE, h=fraction)


class VelocityBoostPropagator(Propagator):
    """The kick v <- v + (fraction*dt)*force/m, followed by velocity constraints when asked (propagators.py:255-273)."""

    def __init__(self, constrained=True):
        super().__init__()
        self.constrained = constrained

    def addSteps(self, integrator, fraction=1.0, force='f'):
        play(integrator, 'dof v <- v + ({h}*dt)*{F}/m' + ('\nconstrain velocities' if self.constrained else ''),
             h=fraction, F=force)


class RespaPropagator(Propagator):
    """rRESPA over N force groups, group 0 innermost (interface of propagators.py:830-973).

    loops[k] = iterations of level k per iteration of level k + 1.  Level k kicks with the force expression f0, f1, f2-f1,
    f3-f2, ...: the shorter-ranged force is subtracted inside the integrator.  `core`: a propagator placed between two half
    moves at the bottom of the recursion; `shell`: {level: propagator} wrapped around the kicks of a level.  Keyword flags:
    has_memory (default False, as in the reference's code), use_respa_switch, blitz.

    The whole program is one recursion (`_level`): level k is  [shell] kick/2 . (level k-1)^n_k . kick/2 [shell], and below
    level 0 sits the move (split around `core`).  The emitted text is pinned by the captures of SURVEY.md 3.2 / App. C."""

    def __init__(self, loops, move=None, boost=None, core=None, shell=None, **kwargs):
        super().__init__()
        self.loops = loops
        self.N = len(loops)
        self.move = move or TranslationPropagator(constrained=False)
        self.boost = boost or VelocityBoostPropagator(constrained=False)
        self.core = core
        self.shell = dict(shell or {})
        if not set(self.shell) <= set(range(self.N)):
            raise InputError("invalid key(s) in RespaPropagator 'shell' argument")
        for part in (self.move, self.boost, self.core, *self.shell.values()):
            if part is not None:
                self.absorbVariables(part)
        self.globalVariables.update({'n%dRESPA' % k: 0 for k, n in enumerate(loops) if n > 1})
        # force expression per level: f0, f1, f2-f1, ...; with memory, fm_k holds level k's force through the inner loops
        # (added to level 0's kicks, subtracted from level k's closing kick)
        self.expr = ['f%d' % k if k < 2 else 'f%d-f%d' % (k, k - 1) for k in range(self.N)]
        self._has_memory = kwargs.pop('has_memory', False)
        kick = list(self.expr)
        if self._has_memory:
            for k in range(1, self.N):
                self.perDofVariables['fm%d' % k] = 0.0
                kick[0] += '+fm%d' % k
                kick[k] += '-fm%d' % k
        self.force = ['(%s)' % text for text in kick]
        self._use_respa_switch = kwargs.pop('use_respa_switch', False)
        self._blitz = kwargs.pop('blitz', False)

    def addSteps(self, integrator, fraction=1.0, force='f'):
        if self._use_respa_switch:
            play(integrator, 'global respa_switch <- 1')
        self._level(integrator, self.N - 1, fraction)
        if self._use_respa_switch:
            play(integrator, 'global respa_switch <- 0')

    def _level(self, integrator, k, fraction):
        if k < 0:                                   # below the innermost force: the move, split around the core bath
            if self.core is None:
                self.move.addSteps(integrator, fraction)
            else:
                self.move.addSteps(integrator, 0.5 * fraction)
                self.core.addSteps(integrator, fraction)
                self.move.addSteps(integrator, 0.5 * fraction)
            return
        n, counter = self.loops[k], 'n%dRESPA' % k
        if n > 1:
            play(integrator, 'global {c} <- 0\nwhile {c} < {n}', c=counter, n=n)
        self._iteration(integrator, k, fraction / n)
        if n > 1:
            play(integrator, 'global {c} <- {c} + 1\nend', c=counter)

    def _iteration(self, integrator, k, h):
        """One iteration of level k over the fraction h of the step."""
        remembered = self._has_memory and k > 0
        if self._blitz:                             # one full kick up front, no closing kick
            if remembered:
                integrator.addComputePerDof('F%d' % k, 'f%d' % k)
            else:
                self.boost.addSteps(integrator, h, self.force[k])
            return self._level(integrator, k - 1, h)
        bath = self.shell.get(k)
        if bath:
            bath.addSteps(integrator, 0.5 * h, self.force[k])
        if remembered:                              # the opening half kick is carried by the inner kicks through fm_k
            integrator.addComputePerDof('fm%d' % k, self.expr[k])
        else:
            self.boost.addSteps(integrator, 0.5 * h, self.force[k])
        self._level(integrator, k - 1, h)
        self.boost.addSteps(integrator, 0.5 * h, self.force[k])
        if bath:
            bath.addSteps(integrator, 0.5 * h, self.force[k])


class MultipleTimeScalePropagator(RespaPropagator):
    """RESPA with a bath whose place is named by `scheme` (interface of propagators.py:976-1042):
    'middle' -- between the two half moves of the innermost loop; 'blitz' -- the same, with RespaPropagator's blitz flag;
    'xi-respa' / 'xo-respa' -- around the kicks of the innermost / outermost level; 'side' -- around those of level `location`.
    `nres` splits the bath into that many sub-steps, `nsy` applies a Suzuki-Yoshida factorisation on top."""

    def __init__(self, loops, move=None, boost=None, bath=None, **kwargs):
        scheme = kwargs.pop('scheme', 'middle')
        location = kwargs.pop('location', 0)
        nres, nsy = kwargs.pop('nres', 1), kwargs.pop('nsy', 1)
        bath = SplitPropagator(bath, nres) if nres > 1 else bath
        bath = SuzukiYoshidaPropagator(bath, nsy) if nsy > 1 else bath
        shell_level = {'xi-respa': 0, 'xo-respa': len(loops) - 1, 'side': location}
        if scheme in ('middle', 'blitz'):
            placement = dict(core=bath, blitz=True) if scheme == 'blitz' else dict(core=bath)
        elif scheme in shell_level:
            placement = dict(shell={shell_level[scheme]: bath})
        else:
            raise InputError('wrong value of scheme parameter')
        placement.update(kwargs)
        super().__init__(loops, move=move, boost=boost, **placement)


class VelocityVerletPropagator(Propagator):
    """Velocity Verlet with constraints: half kick, constrained move, velocity from the displacement plus the second half
    kick, velocity constraints (interface of propagators.py:1108-1133)."""

    PROGRAM = """
        dof v <- v+0.5*Dt*f/m; Dt={h}*dt
        dof x0 <- x
        dof x <- x+Dt*v; Dt={h}*dt
        constrain positions
        dof v <- (x-x0)/Dt+0.5*Dt*f/m; Dt={h}*dt
        constrain velocities
    """

    def __init__(self):
        super().__init__()
        self.perDofVariables['x0'] = 0

    def addSteps(self, integrator, fraction=1.0, force='f'):
        play(integrator, self.PROGRAM, h=fraction)


class UnconstrainedVelocityVerletPropagator(Propagator):
    """Velocity Verlet without constraints (interface of propagators.py:1136-1153)."""

    PROGRAM = """
        dof v <- v+0.5*{h}*dt*f/m
        dof x <- x+{h}*dt*v
        dof v <- v+0.5*{h}*dt*f/m
    """

    def addSteps(self, integrator, fraction=1.0, force='f'):
        play(integrator, self.PROGRAM, h=fraction)


class VelocityRescalingPropagator(Propagator):
    """Stochastic velocity rescaling of Bussi, Donadio and Parrinello (interface of propagators.py:1156-1227).  The sum of
    dof - 1 squared Gaussians is drawn as a gamma variate by Marsaglia-Tsang rejection (shape `a`, constants d = a - 1/3 and
    c = 1/sqrt(9 d)) held in global variables -- plus one more squared Gaussian when dof - 1 is odd -- and the velocities
    are scaled by the factor of the paper's Eq. (A7)."""

    GAMMA_VARIATE = """
        global ready <- 0
        while ready < 0.5
            global X <- gaussian
            global V <- 1+{c}*X
            while V <= 0.0
                global X <- gaussian
                global V <- 1+{c}*X
            end
            global V <- V^3
            global U <- random
            global ready <- step(1-0.0331*X^4-U)
            if ready < 0.5
                global ready <- step(0.5*X^2+{d}*(1-V+log(V))-log(U))
            end
        end
    """
    RESCALE = ('dof v <- vscaling*v; vscaling = sqrt(A+C*B*(gaussian^2+sumRs)+2*sqrt(C*B*A)*gaussian); C = {kT}/mvv; B = 1-A; '
               'A = exp(-dt*{rate}); sumRs = {twice_d}*V{extra}')

    def __init__(self, temperature, degreesOfFreedom, timeScale):
        super().__init__()
        self.tau = unit.md_value(timeScale)
        self.dof = degreesOfFreedom
        self.kT = unit.md_value(kB * temperature)
        self.globalVariables.update(V=0, X=0, U=0, ready=0)

    def addSteps(self, integrator, fraction=1.0, force='f'):
        one_more = self.dof % 2 == 1
        shape = (self.dof - 2 + self.dof % 2) / 2
        d = shape - 1 / 3
        play(integrator, self.GAMMA_VARIATE, c=1 / math.sqrt(9 * d), d=d)
        if one_more:
            play(integrator, 'global X <- gaussian')
        play(integrator, self.RESCALE, kT=self.kT, rate=fraction / self.tau, twice_d=2 * d, extra='+X^2' if one_more else '')


class NoseHooverPropagator(Propagator):
    """Global Nose-Hoover thermostat with `nloops` RESPA-like subdivisions (propagators.py:1230-1273)."""

    def __init__(self, temperature, degreesOfFreedom, timeScale, nloops=1):
        super().__init__()
        self.nloops = nloops
        self.globalVariables['LkT'] = degreesOfFreedom * kB * temperature
        self.globalVariables['Q'] = degreesOfFreedom * kB * temperature * timeScale ** 2
        for name in ('vscaling', 'p_eta', 'n_NH'):
            self.globalVariables[name] = 0

    HEAD = """
        global p_eta <- p_eta + ({half}*dt)*(mvv - LkT)
        global vscaling <- exp(-({h}*dt)*p_eta/Q)
    """
    LOOP = """
        global n_NH <- 1
        while n_NH < {n}
        global p_eta <- p_eta + ({h}*dt)*(vscaling^2*mvv - LkT)
        global vscaling <- vscaling*exp(-({h}*dt)*p_eta/Q)
        global n_NH <- n_NH + 1
        end
    """
    TAIL = """
        global p_eta <- p_eta + ({half}*dt)*(vscaling^2*mvv - LkT)
        dof v <- vscaling*v
    """

    def addSteps(self, integrator, fraction=1.0, force='f'):
        h = fraction / self.nloops
        play(integrator, self.HEAD, h=h, half=0.5 * h)
        if self.nloops > 2:            # as the reference emits it (propagators.py:1258: no inner pass for two loops)
            play(integrator, self.LOOP, h=h, n=self.nloops)
        play(integrator, self.TAIL, half=0.5 * h)


class MassiveNoseHooverPropagator(Propagator):
    """One Nose-Hoover thermostat per degree of freedom (propagators.py:1276-1311)."""

    def __init__(self, temperature, timeScale, nloops=1):
        super().__init__()
        self.nloops = nloops
        self.globalVariables['kT'] = kB * temperature
        self.globalVariables['Q'] = kB * temperature * timeScale ** 2
        self.globalVariables['nMNH'] = 0
        self.perDofVariables['p_eta'] = 0

    BODY = """
        dof p_eta <- p_eta + ({half}*dt)*(m*v^2 - kT)
        dof v <- v*exp(-({h}*dt)*p_eta/Q)
        dof p_eta <- p_eta + ({half}*dt)*(m*v^2 - kT)
    """

    def addSteps(self, integrator, fraction=1.0, force='f'):
        h = fraction / self.nloops
        repeated = self.nloops > 1
        if repeated:
            play(integrator, 'global nMNH <- 0\nwhile nMNH < {n}', n=self.nloops)
        play(integrator, self.BODY, h=h, half=0.5 * h)
        if repeated:
            play(integrator, 'global nMNH <- nMNH + 1\nend')


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

    BODY = """
        dof p1 <- p1 + ({half}*dt)*(m*v^2 - kT)
        dof p2 <- p2 + ({half}*dt)*(m^2*v^4/3 - kT^2)
        dof v <- v2*{scale};v2 = v1/sqrt(1 + 2*v1^2*alpha*{h}*dt);alpha = p2/(3*m*Q2);v1 = v*{scale}
        dof p2 <- p2 + ({half}*dt)*(m^2*v^4/3 - kT^2)
        dof p1 <- p1 + ({half}*dt)*(m*v^2 - kT)
    """

    def addSteps(self, integrator, fraction=1.0, force='f'):
        h = fraction / self.nloops
        repeated = self.nloops > 1
        if repeated:
            play(integrator, 'global nGGM <- 0\nwhile nGGM < {n}', n=self.nloops)
        play(integrator, self.BODY, h=h, half=h / 2, scale='exp(-{}*dt*(p1/Q1 + kT*p2/Q2))'.format(h / 2))
        if repeated:
            play(integrator, 'global nGGM <- nGGM + 1\nend')


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
        # one assignment: the update, then its definitions (`force` only when a constant force acts during the step)
        update = 'z*{} + sqrt(kT*(1 - z*z)/mass)*gaussian'.format(self.velocity)
        definitions = ['mass = {}'.format(self.mass), 'z = exp(-({}*dt)*friction)'.format(fraction)]
        if self.force is not None:
            update += ' + force*(1 - z)/(mass*friction)'
            definitions.insert(0, 'force = {}'.format(self.force))
        add = integrator.addComputeGlobal if self.overall else integrator.addComputePerDof
        add(self.velocity, '; '.join([update] + definitions))


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
