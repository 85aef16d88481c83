"""Pair-force builders with the class names and call signatures of `atomsmm.forces`
(reference: /root/reference/src/atomsmm/forces.py), re-designed for the HIP path.

In the reference every class *composes an energy string* and hands it to OpenMM, whose Lepton
library differentiates and evaluates it per pair.  Here every class still exposes that string
(`getEnergyFunction()`, `repr`) -- it is the observable API -- but next to it carries a structured
**descriptor** (`force._amm`): the family enum + numeric parameters that
`atomsmm_amd.engine` turns into an `amm_pair_desc` for the hand-written kernels
(include/atomsmm_hip.h).  No expression is ever interpreted on the device.

Energy strings whose family cannot be identified (`describe_energy` returns None) are rejected
with `InputError` when a Context is built: the HIP path implements the families the reference's
RESPA classes can emit, not a general expression compiler.
"""
import math
import re

from . import openmm, unit
from .unit import md_value
from .utils import InputError, exceptionOffsetParameters, particleOffsetParameters

KC = 138.935456   # kJ.nm/mol/e^2, hard-coded by the reference (forces.py:407,462,499,535)

_S_DEF = 'S = 1 + step(r - rs0)*u^3*(15*u - 6*u^2 - 10)'
_LJC = '4*epsilon*((sigma/r)^12-(sigma/r)^6) + Kc*chargeprod/r'
_F_POLY = (
    ('f12', '(6*b^2-21*b+28)*(b^3*(R^12-1)-12*b^2*u-66*b*u^2-220*u^3)/462+45*(7-2*b)*u^4/14-72*u^5/7'),
    ('f6', '(6*b^2-3*b+1)*(b^3*(R^6-1)-6*b^2*u-15*b*u^2-20*u^3)+45*(1-2*b)*u^4-36*u^5'),
    ('f1', '5*(b+1)^2*(6*b^3*R*log(R)-6*b^2*u-3*b*u^2+u^3)+u^4*(3*u-5*b-10)/2'),
)
ADJUSTMENTS = {None: 0, 'shift': 1, 'force-switch': 2}


def force_switch_constants(rs, rc):
    """b, f12c, f6c, f1c of the force-switched potential (closed forms of forces.py:559-563)."""
    b = rs / (rc - rs)
    f12c = (1 + b) ** 3 * (b ** 6 + 3 * b ** 5 + (30 / 7) * b ** 4 + (25 / 7) * b ** 3 + (25 / 14) * b ** 2 + (1 / 2) * b + 2 / 33) / b ** 9
    f6c = (1 + b) ** 3 / b ** 3
    f1c = (30 * (1 + b)) * (b ** 2 * (1 + b) ** 2 * math.log(1 / b + 1) - b ** 3 - (3 / 2) * b ** 2 - (1 / 3) * b + 1 / 12)
    return b, f12c, f6c, f1c


def _near_terms(cutoff_distance, switch_distance, adjustment, with_lj_only=False):
    """Expression list of the three near-potential families (same text as forces.py:539-567)."""
    if adjustment not in ADJUSTMENTS:
        raise InputError('unknown adjustment option')
    terms = []
    if adjustment is None:
        terms += ['S*({})'.format(_LJC), _S_DEF]
    elif adjustment == 'shift':
        lj = '4*epsilon*((sigma/r)^12-(sigma/r)^6-((sigma/rc0)^12-(sigma/rc0)^6))'
        terms += ['S*({}+{})'.format(lj, 'Kc*chargeprod*(1/r-1/rc0)'), _S_DEF]
    else:
        pot = '4*epsilon*(f12*(sigma/r)^12-f6*(sigma/r)^6) + Kc*chargeprod*f1/r'
        ref = '4*epsilon*(f12c*(sigma/rc0)^12-f6c*(sigma/rc0)^6) + Kc*chargeprod*f1c/rc0'
        terms.append('{}-({})'.format(pot, ref))
        terms += ['{}=1+step(r-rs0)*({})'.format(name, poly) for name, poly in _F_POLY]
        terms.append('R=u/b+1')
        b, f12c, f6c, f1c = force_switch_constants(switch_distance, cutoff_distance)
        terms += ['b={}'.format(b), 'f12c={}'.format(f12c), 'f6c={}'.format(f6c), 'f1c={}'.format(f1c)]
    terms.append('u=(r-rs0)/(rc0-rs0)')
    return terms


def nearForceExpressions(cutoff_distance, switch_distance, adjustment):
    """Near-potential expression list with rs0/rc0/Kc baked in as literals -- what RESPASystem feeds
    to its group-1 and group-31 CustomNonbondedForces (forces.py:469-500, systems.py:71-77)."""
    rc, rs = md_value(cutoff_distance), md_value(switch_distance)
    terms = _near_terms(cutoff_distance, switch_distance, adjustment)
    return terms + ['rs0={}'.format(rs), 'rc0={}'.format(rc), 'Kc={}'.format(KC)]


def _first_number(defs, name, fallback=None):
    for d in defs:
        m = re.match(r'^\s*%s\s*=\s*([-+0-9.eE]+)\s*$' % re.escape(name), d)
        if m:
            return float(m.group(1))
    return fallback


def describe_energy(energy, global_parameters=None):
    """Identify the pair family of an AtomsMM energy string.  Returns a descriptor dict
    (family, sign, guard, rc0, rs0, Kc, alpha, degree, rswitch) or None."""
    g = dict(global_parameters or {})
    parts = [p.strip() for p in energy.split(';') if p.strip()]
    if not parts:
        return None
    head = parts[0].replace(' ', '')
    # `U_name; U_name = <expression>; ...` (AlchemicalSystem, systems.py:345-365): the head is an alias of a definition
    if re.fullmatch(r'[A-Za-z_]\w*', head):
        for k, part in enumerate(parts[1:], start=1):
            lhs, _, rhs = part.partition('=')
            if lhs.strip() == head and rhs:
                parts = [rhs.strip()] + parts[1:k] + parts[k + 1:]
                head = parts[0].replace(' ', '')
                break
    sign, guard = 1.0, False
    for _ in range(3):
        if head.startswith('-step(rc0-r)*(') and head.endswith(')'):
            head, sign, guard = head[len('-step(rc0-r)*('):-1], -sign, True
        elif head.startswith('step(rc0-r)*(') and head.endswith(')'):
            head, guard = head[len('step(rc0-r)*('):-1], True
        elif head.startswith('-(') and head.endswith(')'):
            head, sign = head[2:-1], -sign
        elif head.startswith('-step(rc0-r)*'):
            head, sign, guard = head[len('-step(rc0-r)*'):], -sign, True
    ljc = _LJC.replace(' ', '')

    def num(name):
        v = _first_number(parts[1:], name)
        return g.get(name) if v is None else v

    desc = dict(sign=sign, guard=guard, Kc=num('Kc') or KC, rc0=num('rc0'), rs0=num('rs0'))
    if head in ('S*(%s)' % ljc, 'energy=S*(%s)' % ljc):
        desc['family'] = 'near-none'
    elif head.startswith('S*(4*epsilon*((sigma/r)^12-(sigma/r)^6-((sigma/rc0)^12-(sigma/rc0)^6))+Kc*chargeprod*(1/r-1/rc0))'):
        desc['family'] = 'near-shift'
    elif head.startswith('4*epsilon*(f12*(sigma/r)^12-f6*(sigma/r)^6)+Kc*chargeprod*f1/r-('):
        desc['family'] = 'near-force-switch'
    elif 'erfc(alpha*r)' in head and head.startswith('4*epsilon*((sigma/r)^12-(sigma/r)^6)'):
        desc.update(family='damped', degree=1, alpha=num('alpha'), rswitch=num('rswitch'))
    elif 'erfc(alpha*r)' in head and head.startswith('S*(4*epsilon*((sigma/r)^12-(sigma/r)^6)'):
        desc.update(family='damped', degree=int(num('d') or 2), alpha=num('alpha'), rswitch=num('rswitch'))
    elif head == '4*epsilon*x*(x-1)+Kc*chargeprod/r':
        desc['family'] = 'ljc'
    elif head == '24*epsilon*(2*(sigma/r)^12-(sigma/r)^6)':
        desc['family'] = 'lj-virial'          # ComputingSystem's dispersion virial (systems.py:894)
    else:
        aux = [p.replace(' ', '') for p in parts[1:]]
        num = r'([0-9.eE+-]+)'
        # a global parameter multiplying the whole energy (AlchemicalRespaSystem: respa_switch, systems.py:636-674)
        scale_name = None
        m = re.fullmatch(r'(\w+)\*(\(.*\)|4\*.*)', head)
        if m and m.group(1) in g and m.group(1) not in ('epsilon',) and not m.group(2).startswith('4*') or \
                (m and m.group(1) == 'respa_switch'):
            scale_name, head = m.group(1), m.group(2)
            if head.startswith('(') and head.endswith(')'):
                head = head[1:-1]
        # force-switched electrostatics of the solute-solvent pairs (systems.py:848-856): no Lennard-Jones part
        m = re.fullmatch(r'(\w+)\*\(1\+step\(r-' + num + r'\)\*f1\)\*' + num + r'\*chargeprod/r', head)
        if m:
            desc.update(family='near-force-switch', noshift=True, coulomb_only=True, rs0=float(m.group(2)), Kc=float(m.group(3)),
                        scale_name=m.group(1))
            return desc
        # `step(rc-r)*U; U = <expression>` bond wrappers (systems.py:643, 674)
        m = re.fullmatch(r'step\(' + num + r'-r\)\*U', head)
        if m and any(a.startswith('U=') for a in aux):
            inner = [a for a in aux if a.startswith('U=')][0][2:]
            sub = describe_energy(';'.join([inner] + [a for a in aux if not a.startswith('U=')]), g)
            if sub is None:
                return None
            sub.update(guard=True, rc0=float(m.group(1)))
            return sub
        # SolvationSystem's solute-solvent softcore Lennard-Jones (systems.py:268) and AlchemicalRespaSystem's (:712-713)
        m = re.fullmatch(r'4\*(\w+)\*epsilon\*\(1-x\)/x\^2', head)
        m2 = re.fullmatch(r'4\*(\w+)\*epsilon\*x\*\(x-1\)', head)
        if m and 'x=(r/sigma)^6+0.5*(1-%s)' % m.group(1) in aux:
            desc.update(family='softcore', lambda_name=m.group(1))
        elif m2 and 'x=1/((r/sigma)^6+0.5*(1-%s))' % m2.group(1) in aux:
            desc.update(family='softcore', lambda_name=m2.group(1))
            fixed = _first_number(parts[1:], m2.group(1))
            if fixed is not None:
                desc['lambda_value'] = fixed          # AlchemicalSoftcoreCVForce: `lambda = 0.4` among the definitions
        elif head == '4*epsilon*x*(x-1)' and 'x=(sigma/r)^6' in aux:
            desc.update(family='lj')                                   # systems.py:749: the collective variable
        elif re.fullmatch(r'4\*\((.+)\)\*epsilon\*x\*\(x-1\)', head) and 'x=(sigma/r)^6' in aux:
            # Lennard-Jones times a coupling function of the global parameters (AlchemicalSystem, systems.py:353-365):
            # the factor and the definitions it needs are handed to the engine as text
            factor = re.fullmatch(r'4\*\((.+)\)\*epsilon\*x\*\(x-1\)', head).group(1)
            keep = [a for a in aux if a.split('=')[0] not in ('x', 'sigma', 'epsilon')]
            desc.update(family='lj', scale_text=';'.join([factor] + keep))
        elif re.fullmatch(r'4\*epsilon\*x\*\(x-1\)\+' + num + r'\*chargeprod/r', head) and 'x=(sigma/r)^6' in aux:
            desc.update(family='ljc', Kc=float(re.fullmatch(r'4\*epsilon\*x\*\(x-1\)\+' + num + r'\*chargeprod/r', head).group(1)))
        else:
            # force-switched potentials without the constant shift (systems.py:823-846); the coefficients in the text
            # are functions of (rc, rs) only: rs is read from step(r-rs), rc is the cutoff of the force object
            ma = re.fullmatch(r'4\*epsilon\*x\*\(x-1\)\+' + num + r'\*chargeprod/r\+step\(r-' + num + r'\)\*perturbation', head)
            mb = re.fullmatch(r'4\*epsilon\*x\*\(x-1\)\+step\(r-' + num + r'\)\*perturbation', head)
            if ma:
                desc.update(family='near-force-switch', noshift=True, Kc=float(ma.group(1)), rs0=float(ma.group(2)))
            elif mb:
                desc.update(family='near-force-switch', noshift=True, Kc=0.0, lj_only=True, rs0=float(mb.group(1)))
            else:
                return None
        if scale_name:
            desc['scale_name'] = scale_name
    return desc


class _AtomsMM_Force:
    """Base of single-object AtomsMM forces: `addTo(system)` (forces.py:24-36)."""

    def addTo(self, system):
        system.addForce(self)
        return self


class _AtomsMM_CompoundForce:
    """Several force objects handled as one (forces.py:39-131): iteration, indexing, and any other
    method call is broadcast to the members that have it; the compound is returned for chaining."""

    def __init__(self, forces):
        self.forces = forces if isinstance(forces, list) else [forces]
        self.setForceGroup(0)

    def __iter__(self):
        return iter(self.forces)

    def __getitem__(self, i):
        return self.forces[i]

    def __len__(self):
        return len(self.forces)

    def __getattr__(self, method):
        if method.startswith('__') or method == 'forces':
            raise AttributeError(method)

        def broadcast(*args, **kwargs):
            for member in self.forces:
                if hasattr(member, method):
                    getattr(member, method)(*args, **kwargs)
            return self
        return broadcast

    def getForceGroup(self):
        return self.forces[0].getForceGroup()

    def addTo(self, system):
        for member in self.forces:
            system.addForce(member)
        return self

    def enableExceptions(self):
        exceptions = NonbondedExceptionsForce()
        exceptions.setForceGroup(self.getForceGroup())
        self.forces.append(exceptions)
        return self


class _AtomsMM_NonbondedForce(openmm.NonbondedForce, _AtomsMM_Force):
    """NonbondedForce without non-exclusion exceptions (forces.py:134-190): on import every exception
    of the source becomes an exclusion."""

    def __init__(self, cutoff_distance, switch_distance=None):
        super().__init__()
        self.setCutoffDistance(cutoff_distance)
        self.setUseSwitchingFunction(switch_distance is not None)
        if switch_distance is not None:
            self.setSwitchingDistance(switch_distance)

    def importFrom(self, force):
        for index in range(force.getNumParticles()):
            self.addParticle(*force.getParticleParameters(index))
        for index in range(force.getNumExceptions()):
            i, j, _, sigma, _ = force.getExceptionParameters(index)
            self.addException(i, j, 0.0, sigma, 0.0)
        self.setNonbondedMethod(force.getNonbondedMethod())
        self.setEwaldErrorTolerance(force.getEwaldErrorTolerance())
        self.setPMEParameters(*force.getPMEParameters())
        self.setUseDispersionCorrection(force.getUseDispersionCorrection())
        return self


class _AtomsMM_CustomNonbondedForce(openmm.CustomNonbondedForce, _AtomsMM_Force):
    """CustomNonbondedForce with per-particle (charge, sigma, epsilon) and Lorentz-Berthelot mixing
    rules appended on import (forces.py:193-323).  `None` for cutoff / switching / dispersion options
    means 'take it from the NonbondedForce passed to importFrom'."""

    def __init__(self, energy, cutoff_distance=None, use_switching_function=None, switch_distance=None,
                 use_dispersion_correction=None, **global_parameters):
        super().__init__(energy)
        self._defer = dict(cutoff=cutoff_distance is None, use_switch=use_switching_function is None,
                           switch=switch_distance is None, lrc=use_dispersion_correction is None)
        for name, value in global_parameters.items():
            self.addGlobalParameter(name, value)
        for parameter in ('charge', 'sigma', 'epsilon'):
            self.addPerParticleParameter(parameter)
        if cutoff_distance is not None:
            self.setCutoffDistance(cutoff_distance)
        if use_switching_function is not None:
            self.setUseSwitchingFunction(use_switching_function)
        if switch_distance is not None:
            self.setSwitchingDistance(switch_distance)
        if use_dispersion_correction is not None:
            self.setUseLongRangeCorrection(use_dispersion_correction)
        self._offset_parameters = []

    # kept for source compatibility with code that inspects these flags (forces.py:226-229)
    importCutoffDistance = property(lambda self: self._defer['cutoff'])
    importUseSwitchingFunction = property(lambda self: self._defer['use_switch'])
    importSwitchDistance = property(lambda self: self._defer['switch'])
    importUseDispersionCorrection = property(lambda self: self._defer['lrc'])

    def __repr__(self):
        return '\n'.join(term.strip(' \t') for term in self.getEnergyFunction().split(';'))

    def mixingRules(self, offset_parameters):
        """';chargeprod = ...;sigma = ...;epsilon = ...' with parameter offsets folded in (forces.py:247-258)."""
        sides = []
        for k in ('1', '2'):
            side = {}
            for prop in ('charge', 'sigma', 'epsilon'):
                text = prop + k
                for parameter in offset_parameters:
                    text += '+{}*{}Scale_{}{}'.format(parameter, prop, parameter, k)
                side[prop] = text
            sides.append(side)
        a, b = sides
        return (';chargeprod = ({})*({})'.format(a['charge'], b['charge']) +
                ';sigma = 0.5*({}+{})'.format(a['sigma'], b['sigma']) +
                ';epsilon = sqrt(({})*({}))'.format(a['epsilon'], b['epsilon']))

    def importFrom(self, nonbonded):
        nb, cn = openmm.NonbondedForce, openmm.CustomNonbondedForce
        method = {nb.NoCutoff: cn.NoCutoff, nb.CutoffNonPeriodic: cn.CutoffNonPeriodic,
                  nb.CutoffPeriodic: cn.CutoffPeriodic, nb.Ewald: cn.CutoffPeriodic, nb.PME: cn.CutoffPeriodic,
                  nb.LJPME: cn.CutoffPeriodic}[nonbonded.getNonbondedMethod()]
        self.setNonbondedMethod(method)
        if self._defer['cutoff']:
            self.setCutoffDistance(nonbonded.getCutoffDistance())
        if self._defer['use_switch']:
            self.setUseSwitchingFunction(nonbonded.getUseSwitchingFunction())
        if self._defer['switch']:
            self.setSwitchingDistance(nonbonded.getSwitchingDistance())
        if self._defer['lrc']:
            self.setUseLongRangeCorrection(nonbonded.getUseDispersionCorrection())
        offsets = particleOffsetParameters(nonbonded)
        self._offset_parameters = list(offsets)
        self.setEnergyFunction(self.getEnergyFunction() + self.mixingRules(offsets))
        for parameter, value in offsets.items():
            self.addGlobalParameter(parameter, value)
            for prop in ('charge', 'sigma', 'epsilon'):
                self.addPerParticleParameter('{}Scale_{}'.format(prop, parameter))
        blanks = [0.0] * (3 * len(offsets))
        for i in range(nonbonded.getNumParticles()):
            self.addParticle([md_value(p) for p in nonbonded.getParticleParameters(i)] + blanks)
        column = {name: 3 * (k + 1) for k, name in enumerate(offsets)}
        for index in range(nonbonded.getNumParticleParameterOffsets()):
            parameter, particle, qs, ss, es = nonbonded.getParticleParameterOffset(index)
            values = list(self.getParticleParameters(particle))
            values[column[parameter]:column[parameter] + 3] = [qs, ss, es]
            self.setParticleParameters(particle, values)
        for index in range(nonbonded.getNumExceptions()):
            i, j = nonbonded.getExceptionParameters(index)[:2]
            self.addExclusion(i, j)
        return self

    def getGlobalParameters(self):
        return {self.getGlobalParameterName(i): self.getGlobalParameterDefaultValue(i)
                for i in range(self.getNumGlobalParameters())}


class _AtomsMM_CustomBondForce(openmm.CustomBondForce, _AtomsMM_Force):
    """CustomBondForce holding the exceptions of a NonbondedForce as bonds with per-bond
    (chargeprod, sigma, epsilon) (forces.py:326-397)."""

    def __init__(self, energy, **globalParams):
        super().__init__(energy)
        for name, value in globalParams.items():
            self.addGlobalParameter(name, value)
        self._offset_parameters = []

    def offsetRules(self, offset_parameters):
        text = ''
        for prop in ('chargeprod', 'sigma', 'epsilon'):
            expr = prop + '0'
            for parameter in offset_parameters:
                expr += '+{}*{}Scale_{}'.format(parameter, prop, parameter)
            text += ';{} = {}'.format(prop, expr)
        return text

    def importFrom(self, nonbonded, extract=False):
        self.setUsesPeriodicBoundaryConditions(nonbonded.usesPeriodicBoundaryConditions())
        offsets = exceptionOffsetParameters(nonbonded)
        self._offset_parameters = list(offsets)
        props = ('chargeprod', 'sigma', 'epsilon')
        if offsets:
            self.setEnergyFunction(self.getEnergyFunction() + self.offsetRules(offsets))
            for prop in props:
                self.addPerBondParameter(prop + '0')
            for parameter, value in offsets.items():
                self.addGlobalParameter(parameter, value)
                for prop in props:
                    self.addPerBondParameter('{}Scale_{}'.format(prop, parameter))
        else:
            for prop in props:
                self.addPerBondParameter(prop)
        blanks = [0.0] * (3 * len(offsets))
        for index in range(nonbonded.getNumExceptions()):
            i, j, chargeprod, sigma, epsilon = nonbonded.getExceptionParameters(index)
            self.addBond(i, j, [md_value(chargeprod), md_value(sigma), md_value(epsilon)] + blanks)
            if extract:
                nonbonded.setExceptionParameters(index, i, j, 0.0, 1.0, 0.0)
        column = {name: 3 * (k + 1) for k, name in enumerate(offsets)}
        for index in range(nonbonded.getNumExceptionParameterOffsets()):
            parameter, bond, qqs, ss, es = nonbonded.getExceptionParameterOffset(index)
            i, j, stored = self.getBondParameters(bond)
            values = list(stored)
            values[column[parameter]:column[parameter] + 3] = [qqs, ss, es]
            self.setBondParameters(bond, i, j, values)
        return self


class NonbondedExceptionsForce(_AtomsMM_CustomBondForce):
    """Only the exceptions of a NonbondedForce: 4*epsilon*x*(x-1) + Kc*chargeprod/r over bonded pairs,
    no cutoff (forces.py:400-407)."""

    def __init__(self):
        super().__init__('4*epsilon*x*(x-1) + Kc*chargeprod/r; x=(sigma/r)^6',
                         Kc=KC * unit.kilojoules_per_mole / unit.nanometer)


class DampedSmoothedForce(_AtomsMM_CustomNonbondedForce):
    """Damped, smoothed Lennard-Jones/Coulomb potential (forces.py:410-466)

        V(r) = S(u) { 4 eps [(sigma/r)^12 - (sigma/r)^6] + Kc q1 q2 erfc(alpha r)/r },
        u = (r^n - rs^n)/(rc^n - rs^n),  S(u) = 1 + u^3 (15u - 6u^2 - 10) for r >= rs.

    degree n = 1 uses OpenMM's built-in switching function (same polynomial with u linear in r); no
    long-range dispersion correction.

    Parameters: alpha (1/length), cutoff_distance, switch_distance (0 <= rs < rc), degree=1.
    """

    def __init__(self, alpha, cutoff_distance, switch_distance, degree=1):
        rs, rc = md_value(switch_distance), md_value(cutoff_distance)
        if rs < 0.0 or rs >= rc:
            raise InputError('Switching distance must satisfy 0 <= r_switch < r_cutoff')
        lj, coul = '4*epsilon*((sigma/r)^12 - (sigma/r)^6)', 'Kc*chargeprod/r'
        if degree == 1:
            energy = '{} + erfc(alpha*r)*{}'.format(lj, coul)
        else:
            energy = ('S*({} + erfc(alpha*r)*{});'.format(lj, coul) +
                      'S = 1 + step(r - rswitch)*u^3*(15*u - 6*u^2 - 10);' +
                      'u = (r^d - rswitch^d)/(rcut^d - rswitch^d); d={}'.format(degree))
        super().__init__(energy=energy, cutoff_distance=cutoff_distance, use_switching_function=(degree == 1),
                         switch_distance=(switch_distance if degree == 1 else None), use_dispersion_correction=False,
                         Kc=KC * unit.kilojoules_per_mole / unit.nanometer, alpha=alpha, rswitch=switch_distance,
                         rcut=cutoff_distance)
        self._amm = dict(family='damped', sign=1.0, guard=False, Kc=KC, alpha=float(md_value(alpha)),
                         rswitch=float(rs), degree=int(degree))


class SoftcoreLennardJonesForce(_AtomsMM_CustomNonbondedForce):
    """Softened Lennard-Jones potential V = 4 lambda eps (1/s^2 - 1/s), s = (r/sigma)^6 + (1 - lambda)/2, with the global
    parameter `parameter` as lambda (forces.py:727-758).  On the HIP path it is evaluated over ONE interaction group
    (`addInteractionGroup(solute, solvent)`, the way every alchemical builder of the reference uses a softcore potential)."""

    def __init__(self, cutoff_distance=None, use_switching_function=None, switch_distance=None,
                 use_dispersion_correction=None, parameter='lambda'):
        text = '4*{0}*epsilon*x*(x-1);x = 1/((r/sigma)^6 + 0.5*(1-{0}))'.format(parameter)
        super().__init__(text, cutoff_distance, use_switching_function, switch_distance, use_dispersion_correction,
                         **{parameter: 1.0})


class SoftcoreForce(_AtomsMM_CustomNonbondedForce):
    """Softened Lennard-Jones plus scaled Coulomb in ONE expression (interface of forces.py:761-793): V = 4 lambda_vdw eps
    (1 - x)/x^2 + Kc lambda_coul q1 q2 / r with x = (r/sigma)^6 + (1 - lambda_vdw)/2, OpenMM's built-in switch from
    `switch_distance` to `cutoff_distance`.  The class, its energy text and its global parameters (Kc, lambda_vdw, lambda_coul)
    are the reference's (pinned by tests/golden/programs.json: 'softcore').  No system class on the RESPA hot path uses it
    (SolvationSystem / AlchemicalSystem use SoftcoreLennardJonesForce), and the HIP engine has no kernel for this text: creating a
    Context over a System that holds one raises the 'energy expression not recognised' error."""

    def __init__(self, cutoff_distance, switch_distance=None):
        terms = ['4*lambda_vdw*epsilon*(1-x)/x^2 + Kc*lambda_coul*chargeprod/r', 'x = (r/sigma)^6 + 0.5*(1-lambda_vdw)']
        parameters = dict(Kc=KC * unit.kilojoules_per_mole / unit.nanometer, lambda_vdw=1.0, lambda_coul=1.0)
        super().__init__(';'.join(terms), cutoff_distance, True, switch_distance, **parameters)


class NearForce(object):
    """Shared pieces of the near forces (forces.py:533-567)."""

    def _globalParams(self, cutoff_distance, switch_distance):
        return {'Kc': KC * unit.kilojoules_per_mole / unit.nanometer, 'rc0': cutoff_distance, 'rs0': switch_distance}

    def _expressions(self, cutoff_distance, switch_distance, adjustment):
        return _near_terms(cutoff_distance, switch_distance, adjustment)

    @staticmethod
    def _descriptor(cutoff_distance, switch_distance, adjustment, subtract, guard):
        family = {None: 'near-none', 'shift': 'near-shift', 'force-switch': 'near-force-switch'}[adjustment]
        return dict(family=family, sign=-1.0 if subtract else 1.0, guard=guard, Kc=KC,
                    rc0=float(md_value(cutoff_distance)), rs0=float(md_value(switch_distance)))


class NearNonbondedForce(_AtomsMM_CustomNonbondedForce, NearForce):
    """Short-range part of Lennard-Jones + Coulomb for RESPA-2 splitting (forces.py:570-670):
    V_LJC smoothed to zero between `switch_distance` and `cutoff_distance` by the quintic S(u).

    adjustment: None -> S*V ; 'shift' -> S*(V - V(rc)) ; 'force-switch' -> potential whose *force* is
    S*F_LJC.  subtract=True negates it; actual_cutoff sets the cutoff OpenMM would really use (the
    energy is then guarded by step(rc0-r)).
    """

    def __init__(self, cutoff_distance, switch_distance, adjustment=None, subtract=False, actual_cutoff=None):
        expressions = self._expressions(cutoff_distance, switch_distance, adjustment)
        if actual_cutoff is not None:
            expressions[0] = 'step(rc0-r)*({})'.format(expressions[0])
        if subtract:
            expressions[0] = '-({})'.format(expressions[0])
        super().__init__(energy='; '.join(expressions),
                         cutoff_distance=cutoff_distance if actual_cutoff is None else actual_cutoff,
                         use_switching_function=False, use_dispersion_correction=False,
                         **self._globalParams(cutoff_distance, switch_distance))
        self._amm = self._descriptor(cutoff_distance, switch_distance, adjustment, subtract, actual_cutoff is not None)


class NearExceptionForce(_AtomsMM_CustomBondForce, NearForce):
    """Near potential applied to exception pairs as bonds, guarded by step(rc0-r) (forces.py:673-680)."""

    def __init__(self, cutoff_distance, switch_distance, adjustment=None, subtract=False):
        expressions = self._expressions(cutoff_distance, switch_distance, adjustment)
        expressions[0] = 'step(rc0-r)*({})'.format(expressions[0])
        if subtract:
            expressions[0] = '-{}'.format(expressions[0])
        super().__init__('; '.join(expressions), **self._globalParams(cutoff_distance, switch_distance))
        self._amm = self._descriptor(cutoff_distance, switch_distance, adjustment, subtract, True)


class FarNonbondedForce(_AtomsMM_CompoundForce):
    """Complement of a NearNonbondedForce (forces.py:683-724): a compound of
    `total`  = plain NonbondedForce (cutoff, optional switch; method/Ewald settings imported) and
    `discount` = -step(rc0-r)*(near expression) evaluated out to the outer cutoff.
    The engine evaluates both members; near + far == the original NonbondedForce."""

    def __init__(self, preceding, cutoff_distance, switch_distance=None):
        if not isinstance(preceding, NearNonbondedForce):
            raise InputError('argument \'preceding\' must be of class NearNonbondedForce')
        potential = preceding.getEnergyFunction().split(';')
        potential[0] = '-step(rc0-r)*({})'.format(potential[0])
        discount = _AtomsMM_CustomNonbondedForce(energy=';'.join(potential), cutoff_distance=cutoff_distance,
                                                 use_switching_function=False, use_dispersion_correction=False,
                                                 **preceding.getGlobalParameters())
        near = dict(preceding._amm)
        near.update(sign=-near['sign'], guard=True)
        discount._amm = near
        total = _AtomsMM_NonbondedForce(cutoff_distance, switch_distance)
        super().__init__([total, discount])
