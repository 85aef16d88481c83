"""Host engine behind `openmm.Context`: turns a System + integrator *description* into calls of the
HIP library (include/atomsmm_hip.h) and takes over what OpenMM's C++ does for the reference
(SURVEY.md section 3.3): per-group force/energy evaluation (`getState(groups=...)`) and execution of
the CustomIntegrator step program (`integrator.step(n)`).

Design (MI355X-first, not OpenMM's):
  * every Force object becomes backend objects once, at Context creation -- a pair force (cell list ->
    Verlet list -> lanes-per-atom traversal) or bond-list terms (owner-computes CSR);
  * the step program is *unrolled on the host* into a flat op list (EVAL group / KICK / MOVE / COPY)
    with one force cache per group: a group is re-evaluated only if positions changed since its last
    evaluation.  For RespaPropagator([4,2,1]) that is 1/2/8 evaluations of groups 2/1/0 per outer step
    instead of the 2/4/10 OpenMM's single-buffer VM performs, with identical arithmetic;
  * the op list of a steady-state step is cached and replayed by `amm_run_ops` with no Python in the loop;
  * atom decomposition: with torch.distributed initialised (one process per GPU) every rank integrates all
    atoms redundantly, computes pair forces for its slice of the cell-sorted order and all-reduces the
    per-group force buffer over RCCL.

There is no CPU path: creating a Context without the built HIP library or without a GPU raises.
"""
import os
import threading
import math
import re

import numpy as np

from . import backend as B
from . import expr as X
from . import openmm as mm
from .forces import describe_energy
from .utils import InputError

_FAMILY = {'near-none': B.NEAR_NONE, 'near-shift': B.NEAR_SHIFT, 'near-force-switch': B.NEAR_FSWITCH,
           'damped': B.DAMPED}
_SAFE_FUNCS = {name: getattr(math, name) for name in ('sqrt', 'exp', 'log', 'sin', 'cos', 'tan', 'erf', 'erfc', 'floor', 'ceil')}
_SAFE_FUNCS.update(step=lambda x: 1.0 if x >= 0 else 0.0, delta=lambda x: 1.0 if x == 0 else 0.0,
                   select=lambda c, a, b: a if c != 0 else b, min=min, max=max, abs=abs)
ALLREDUCE = 'allreduce'
_AUX_FORCE = re.compile(r'_f[0-9]*_')      # auxiliary buffers of multi-force expressions (integrators.py:134-145)
_NO_ALIAS = os.environ.get('AMM_NO_ALIAS') is not None      # tuning knob (A/B)
# host-walked step programs evaluate the same few dozen texts every step: compiled code, split conditions, symbol lists
_CODE_CACHE, _CONDITION_CACHE, _PER_DOF_SYMBOLS = {}, {}, {}
_SCALARS = 2048        # device scalars of a host-walked program: deriv(energy, p) sums in flight and the globals computed from them
_COMPARE = {'<': lambda a, b: a < b, '>': lambda a, b: a > b, '<=': lambda a, b: a <= b, '>=': lambda a, b: a >= b,
            '=': lambda a, b: a == b, '!=': lambda a, b: a != b}
_context_factory = B.HipContext   # the only backend; tests of the host logic substitute a call recorder


def custom_long_range_correction(u, sigma, eps, box, rc, rswitch, codes=None, pairs=None):
    """Long-range correction of a CustomNonbondedForce as OpenMM defines it [recalled; pinned by tests/test_systems.py:39
    (interaction group, met to 2e-8) and tests/test_computers.py:31 (no groups)]: with I(s, e) = int_rc^inf u r^2 dr +
    int_rs^rc (1 - S) u r^2 dr for a pair of mixed parameters (s, e),
        E = 2 pi N^2 / V * [sum over class pairs of count * I] / (N (N + 1)/2),
    classes = atoms of equal (sigma, epsilon); count = n_i n_j for two classes, n_i (n_i + 1)/2 within a class; with an
    interaction group (`codes` 1 / 2) the count is the number of (set 1, set 2) pairs.  N = number of particles."""
    n = len(sigma)

    def refine(f):
        """OpenMM's quadrature (CustomNonbondedForceImpl::integrateInteraction [recalled]): midpoint rule on (0, 1),
        the number of points tripled (the old midpoints stay) until two sums agree to 1e-5 -- reproduced as is, because
        the reference literals carry ITS truncation error (tests/test_systems.py:39 is met to 1e-9 this way, to 2e-8
        with an exact quadrature; tests/test_computers.py:33 needs the former)."""
        total, points = 0.0, 1
        for iteration in range(9):
            idx = np.arange(points)
            xs = (idx[idx % 3 != 1] + 0.5) / points
            old = total
            total = float(np.sum(f(xs))) / points + old / 3.0
            if iteration > 2 and (total == 0.0 or abs((total - old) / total) < 1e-5):
                return total
            points *= 3
        raise RuntimeError('long-range correction did not converge')

    def integral(s, e):
        def tail(x):                                   # int_rc^inf u r^2 dr = (1/rc) int_0^1 u(rc/x) r^4 dx, r = rc/x
            r = rc / x
            return u(r, s, e) * r ** 4
        out = refine(tail) / rc
        if rswitch is not None:
            def shell(x):                              # int_rs^rc (1 - S) u r^2 dr
                r = rswitch + x * (rc - rswitch)
                return x ** 3 * (10.0 + x * (-15.0 + x * 6.0)) * u(r, s, e) * r * r
            out += refine(shell) * (rc - rswitch)
        return out

    if pairs is None:
        pairs = lrc_class_pairs(sigma, eps, codes)
    total = sum(weight * integral(s, e) for s, e, weight in pairs)
    return 2.0 * math.pi * n * n / float(np.prod(box)) * total / (n * (n + 1) / 2.0)


def lrc_class_pairs(sigma, eps, codes=None):
    """[(sigma_ij, eps_ij, number of pairs)] over the classes of equal (sigma, epsilon) -- the part of a long-range
    correction that depends on the per-particle parameters only (a 250 000-atom system has a handful of classes): the
    softcore force's correction is re-evaluated at every change of lambda, and twice per deriv(energy, lambda)."""
    table = np.stack([np.asarray(sigma, dtype=np.float64), np.asarray(eps, dtype=np.float64)], axis=1)

    def classes(members):
        values, counts = np.unique(table[members], axis=0, return_counts=True)
        return [(float(s_), float(e_), int(c_)) for (s_, e_), c_ in zip(values, counts)]
    out = []
    if codes is not None:
        for s1, e1, n1 in classes(np.where(np.asarray(codes) == 1.0)[0]):
            for s2, e2, n2 in classes(np.where(np.asarray(codes) == 2.0)[0]):
                s, e = 0.5 * (s1 + s2), math.sqrt(e1 * e2)
                if e != 0.0 and s > 0.0:
                    out.append((s, e, n1 * n2))
    else:
        every = classes(np.arange(len(table)))
        for a, (s1, e1, n1) in enumerate(every):
            for s2, e2, n2 in every[a:]:
                s, e = 0.5 * (s1 + s2), math.sqrt(e1 * e2)
                if e != 0.0 and s > 0.0:
                    out.append((s, e, n1 * (n1 + 1) / 2 if (s1, e1) == (s2, e2) else n1 * n2))
    return out


def softcore_long_range_correction(sigma, eps, codes, box, rc, rswitch, lam, pairs=None):
    """SolvationSystem's softcore force (systems.py:266-272): u = 4 lambda eps (1-x)/x^2, x = (r/sigma)^6 + (1-lambda)/2."""
    def u(r, s, e):
        x = (r / s) ** 6 + 0.5 * (1.0 - lam)
        return 4.0 * lam * e * (1.0 - x) / (x * x)
    return custom_long_range_correction(u, sigma, eps, box, rc, rswitch, codes, pairs)


def dispersion_correction(sigma, eps, box, rc, rswitch=None):
    """Long-range LJ correction of OpenMM's NonbondedForce, with the switching-function term
    (SURVEY.md Appendix B.6): (2 pi N^2/V) <int_rc^inf V r^2 dr + int_rs^rc (1-S) V r^2 dr> over type pairs."""
    classes = {}
    for s, e in zip(sigma, eps):
        classes[(s, e)] = classes.get((s, e), 0) + 1
    keys = list(classes)
    n = float(len(sigma))
    total = 0.0
    for a in range(len(keys)):
        for b in range(a, len(keys)):
            e = math.sqrt(keys[a][1] * keys[b][1])
            if e == 0.0:
                continue
            s = 0.5 * (keys[a][0] + keys[b][0])
            s6 = s ** 6
            s12 = s6 * s6
            w = 0.5 * classes[keys[a]] * (classes[keys[a]] + 1) if a == b else float(classes[keys[a]]) * classes[keys[b]]
            term = 4 * e * (s12 / (9 * rc ** 9) - s6 / (3 * rc ** 3))
            if rswitch is not None:
                r = np.linspace(rswitch, rc, 2001)
                t = (r - rswitch) / (rc - rswitch)
                S = 1 + t ** 3 * (15 * t - 6 * t * t - 10)
                g = (1 - S) * 4 * e * (s12 / r ** 12 - s6 / r ** 6) * r * r
                h = r[1] - r[0]
                term += h / 3 * (g[0] + g[-1] + 4 * g[1:-1:2].sum() + 2 * g[2:-1:2].sum())
            total += w * term
    total /= 0.5 * n * (n + 1)
    return 2 * math.pi * n * n * total / (box[0] * box[1] * box[2])


class LocalWorld:
    """W ranks inside ONE process, one thread each, all on the same GPU and the same stream: the multi-rank code path -- slices, exchange
    chunks, amm_exchange_finish, the launches that integrate a rank's own molecules -- with the collectives made by device-to-device
    copies between the ranks' buffers.  For tests at world sizes a one-GPU box cannot host as processes (8 ranks) and for the per-rank
    budgets of DESIGN.md section 5 (scripts/per_rank_step.py): a rank's kernels are the real ones, the exchange costs nothing.

        world = LocalWorld(8)
        results = world.run(lambda rank: job(rank))        # job builds its own Context; Engine finds the world it runs in
    """
    _tls = threading.local()

    def __init__(self, world):
        self.world = int(world)
        self._barrier = threading.Barrier(self.world)
        self._slots = [None] * self.world

    @classmethod
    def current(cls):
        return getattr(cls._tls, 'membership', None)

    def run(self, job):
        results, errors = [None] * self.world, [None] * self.world

        def body(rank):
            LocalWorld._tls.membership = (self, rank)
            try:
                results[rank] = job(rank)
            except BaseException as exc:      # noqa: BLE001 -- reported below; the others must not wait for this rank for ever
                errors[rank] = exc
                self._barrier.abort()
            finally:
                LocalWorld._tls.membership = None
        threads = [threading.Thread(target=body, args=(r,)) for r in range(self.world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        first = [e for e in errors if e is not None and not isinstance(e, threading.BrokenBarrierError)] or [e for e in errors if e is not None]
        if first:
            raise first[0]
        return results

    def all_gather(self, rank, buf, count):
        """chunk r of every rank's `buf` (count elements each) <- rank r's chunk r."""
        self._slots[rank] = buf
        self._barrier.wait()
        for r in range(self.world):
            if r != rank:
                buf[r * count:(r + 1) * count].copy_(self._slots[r][r * count:(r + 1) * count])
        self._barrier.wait()

    def all_reduce(self, rank, tensor, op='sum'):
        self._slots[rank] = tensor.clone()
        self._barrier.wait()
        total = self._slots[0].clone()
        for r in range(1, self.world):
            total = total + self._slots[r] if op == 'sum' else total.maximum(self._slots[r])
        self._barrier.wait()
        tensor.copy_(total)

    def broadcast(self, rank, value):
        if rank == 0:
            self._slots[0] = value
        self._barrier.wait()
        out = self._slots[0]
        self._barrier.wait()
        return out


class _Entry:
    """One System force translated to backend objects."""

    def __init__(self, force, group):
        self.force = force
        self.group = group
        self.pair_ids = []          # backend pair forces (sliced across ranks)
        self.bonded_id = None       # backend bonded set holding this force's bond-list terms
        self.terms = []             # (kind, idx, params, periodic, desc) for group-level merging
        self.constant = 0.0         # energy-only constant (dispersion correction)
        self.recip_group = None
        self.recip = None
        self.update = None          # callable(parameters) refreshing lambda-dependent parameters
        self.softcore = None        # softcore pair force: dict(pid, lambda_name, constant, depends)
        self.depends = set()        # global parameters this entry's backend data depend on


def slice_bounds(n_items, rank, world):
    """[begin, end) of rank's contiguous slice of n_items cell-sorted slots: ceil(n / world) each, the same formula as the
    HIP library (csrc/pair.hip first_build).  Atom decomposition: every rank holds all positions and
    integrates all atoms redundantly; the pair work is split by these slices of the cell-sorted order with full neighbour
    rows, so every force row has one producer and the exchange (all-gather of slices, or all-reduce of zero-filled
    buffers) is exact and identical on all ranks."""
    per = (n_items + world - 1) // world      # (bond-list sets: blocks of atom indices; the pair rows use backend.slice_per)
    begin = min(n_items, rank * per)
    return begin, min(n_items, begin + per)


class Engine:
    def __init__(self, system, integrator, properties):
        import torch
        self.torch = torch
        self.system = system
        self.integrator = integrator
        n = system.getNumParticles()
        if n == 0:
            raise mm.OpenMMException('System has no particles')
        try:
            vecs = system._box
        except AttributeError:
            vecs = None
        if vecs is None:
            raise InputError('the HIP path needs a periodic orthorhombic box: call System.setDefaultPeriodicBoxVectors')
        for i in range(3):
            for j in range(3):
                if i != j and abs(vecs[i][j]) > 1e-12:
                    raise InputError('only orthorhombic periodic boxes are supported by the HIP path')
        self.box = np.array([vecs[0][0], vecs[1][1], vecs[2][2]], dtype=np.float64)
        self.rank, self.world = 0, 1
        dist = torch.distributed
        # (ranks as threads of this process: LocalWorld above)
        self._local = None
        member = LocalWorld.current()
        if member is not None and member[0].world > 1:
            self._local, self.rank = member
            self.world = self._local.world
        elif dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            self.rank, self.world = dist.get_rank(), dist.get_world_size()
        # AMM_FORCE_COLLECTIVES=1: a 1-rank job takes the multi-rank code path (collectives over a 1-rank group), to
        # measure its host-side cost on a single GPU
        self._coll = self.world > 1 or (os.environ.get('AMM_FORCE_COLLECTIVES') == '1' and dist.is_available()
                                        and dist.is_initialized())
        device = int(properties.get('DeviceIndex', torch.cuda.current_device() if torch.cuda.is_available() else 0))
        self.ctx = _context_factory(n, self.box, device=device, rank=self.rank, world=self.world)
        self.n = n
        self._native_comm = False
        self._native_ops = {}
        if self._coll and self._local is None and hasattr(self.ctx, 'comm_init') and dist.get_backend() == 'nccl' \
                and os.environ.get('AMM_NATIVE_COMM', '1') != '0':
            self._init_native_comm(dist)
        dev = self.ctx.torch_device
        f64 = torch.float64
        # exchange of the owner-computed force slices: all-gather of the slices in cell-sorted order (1/world of the bytes
        # of the all-reduce of zero-filled buffers; AMM_EXCHANGE=reduce keeps the latter) for groups of ONE pair force
        self._gather = self._coll and hasattr(self.ctx, 'bind_exchange') and os.environ.get('AMM_EXCHANGE', 'gather') == 'gather'
        self._gather_groups = set()
        if self._gather:
            self._per = B.slice_per(n, self.world)
            self._xchg = torch.zeros(self.world * 2 * self._per * 3, dtype=f64, device=dev)
            self.ctx.bind_exchange(self._xchg)
        self.x = torch.zeros((n, 3), dtype=f64, device=dev)
        self.v = torch.zeros((n, 3), dtype=f64, device=dev)
        self.mass = torch.as_tensor(np.array(system._masses, dtype=np.float64), device=dev)
        self.time = 0.0
        self.parameters = {}
        self.entries = []
        self.skin = float(properties.get('Skin', -1.0))
        # The engine is the only writer of the bound position buffer outside the library (set_positions): amm_run_ops may trust
        # the displacement checks its own launches made at the end of the previous call
        if hasattr(self.ctx, 'positions_changed'):
            self.ctx.set_option('positions_private', 1)
        # 'Option.<name>': tuning / test options of the library context (include/atomsmm_hip.h: amm_set_option)
        for key, value in properties.items():
            if key.startswith('Option.') and hasattr(self.ctx, 'set_option'):
                self.ctx.set_option(key[len('Option.'):], float(value))
        if float(properties.get('OuterSkin', -1.0)) > 0:       # dual Verlet list: cell-built outer list pruned to the inner one
            self.ctx.set_outer_skin(float(properties['OuterSkin']))
        self._pair_info = {}
        for force in system.getForces():
            self._translate(force)
        self._share_lists()
        self._slots = {}
        self._buffers = {}
        self._arena_free = []
        self._arena_contiguous = True
        self._group_defs = {}
        self._group_bonded = []     # merged bond-list sets made by _define_group
        self._deriv_cache = {}      # deriv(energy, name) at the current positions and parameters (host-walked programs)
        self._emit_memo = {}        # what the per-DOF steps of a host-walked program emit (_emit_per_dof_memo)
        self._segment_memo, self._segment_ends, self._segment_symbols, self._segment_open = {}, {}, {}, None      # ... and whole runs of them
        self._pending = None        # device scalars that deferred globals wait for: (buffer, number in use)
        self._device_params = {}    # Context parameters whose number is a device scalar for now (name -> [pair force ids]): AFED's lambda
        self._scalar_code, self._scalar_consts = [], []      # assignments to device scalars not yet launched (one launch per batch)
        self._lrc_on_device = {}    # (pair force, lambda's scalar) -> the correction's lambda-derivative there
        self.n_scalar_evals = self.n_scalar_launches = self.n_settles = 0     # (tests / bench: global expressions evaluated on the device, blocking reads of the scalars)
        self.device_globals = True  # nonlinear uses of deferred globals are evaluated on the device (amm_expr_eval_scalar) instead of waited for
        self._valid = {}
        self._interpreted = None    # None: undecided; True: general (host-walked) step programs
        self._static_exprs = False
        self._expr_ids = {}
        self._seed_set = None
        self._host_rng = None
        self._expr_counter = 0
        self._mirror = {}           # per-DOF buffer -> force symbol it currently mirrors (`_f2_ <- f2`)
        self._programs = {}
        self._energy = torch.zeros(1, dtype=f64, device=dev)
        self._fwork = torch.zeros((n, 3), dtype=f64, device=dev)
        self._fwork2 = torch.zeros((n, 3), dtype=f64, device=dev)
        self.ctx.bind_state(self.x, self.v, self.mass)
        self._has_constraints = system.getNumConstraints() > 0
        if self._has_constraints:
            cons = system._constraints
            self.ctx.constraints_create(np.array([[c[0], c[1]] for c in cons], dtype=np.int32), np.array([c[2] for c in cons]),
                                        integrator.getConstraintTolerance() if hasattr(integrator, 'getConstraintTolerance') else 1e-5)
        if isinstance(integrator, mm.CustomIntegrator):
            integrator._n_hint = n

    # ------------------------------------------------------------------------------- translation
    def _register_globals(self, force):
        if hasattr(force, 'getNumGlobalParameters'):
            for i in range(force.getNumGlobalParameters()):
                self.parameters.setdefault(force.getGlobalParameterName(i), force.getGlobalParameterDefaultValue(i))

    def _translate(self, force):
        self._register_globals(force)
        if isinstance(force, mm.CMMotionRemover):
            return
        entry = _Entry(force, force.getForceGroup())
        if isinstance(force, mm.NonbondedForce):
            self._translate_nonbonded(force, entry)
        elif isinstance(force, mm.CustomNonbondedForce):
            self._translate_custom_nonbonded(force, entry)
        elif isinstance(force, mm.CustomBondForce):
            self._translate_custom_bond(force, entry)
        elif isinstance(force, mm.CustomCVForce):
            self._translate_cv(force, entry)
        elif isinstance(force, mm.HarmonicBondForce):
            b = np.array([[r[0], r[1]] for r in force._bonds], dtype=np.int32).reshape(-1, 2)
            p = np.array([[r[2], r[3]] for r in force._bonds], dtype=np.float64).reshape(-1, 2)
            entry.terms.append((B.BOND_HARMONIC, b, p, force.usesPeriodicBoundaryConditions(), None))
        elif isinstance(force, mm.HarmonicAngleForce):
            a = np.array([r[:3] for r in force._angles], dtype=np.int32).reshape(-1, 3)
            p = np.array([r[3:] for r in force._angles], dtype=np.float64).reshape(-1, 2)
            entry.terms.append((B.ANGLE_HARMONIC, a, p, force.usesPeriodicBoundaryConditions(), None))
        elif isinstance(force, mm.CustomAngleForce):
            if force.getEnergyFunction().replace(' ', '') != '0.5*(K0*(theta-t0)^2-Kn*(theta-tn)^2)':
                raise InputError('CustomAngleForce: only the redefine_angle difference potential is supported')
            # difference of two harmonic angles (RESPASystem.redefine_angle, systems.py:226)
            a = np.array([r[:3] for r in force._angles], dtype=np.int32).reshape(-1, 3)
            p = np.array([r[3] for r in force._angles], dtype=np.float64).reshape(-1, 4)
            periodic = force.usesPeriodicBoundaryConditions()
            entry.terms.append((B.ANGLE_HARMONIC, a, p[:, [0, 1]], periodic, None))
            entry.terms.append((B.ANGLE_HARMONIC, a, np.stack([p[:, 2], -p[:, 3]], axis=1), periodic, None))
        elif isinstance(force, mm.PeriodicTorsionForce):
            t = np.array([r[:4] for r in force._torsions], dtype=np.int32).reshape(-1, 4)
            p = np.array([[r[4], r[5], r[6]] for r in force._torsions], dtype=np.float64).reshape(-1, 3)
            entry.terms.append((B.TORSION_PERIODIC, t, p, force.usesPeriodicBoundaryConditions(), None))
        else:
            raise InputError('force type {} is not supported by the HIP path'.format(force.__class__.__name__))
        self._finish_entry(entry)
        self.entries.append(entry)

    def _pair_create(self, desc, q, sigma, eps, excl):
        pid = self.ctx.pair_create(desc, q, sigma, eps, excl, skin=self.skin)
        key = np.sort(np.sort(np.asarray(excl, dtype=np.int64).reshape(-1, 2), axis=1), axis=0).tobytes()
        # interaction-group forces keep a list of their own: it holds the (set 1, set 2) pairs only
        if desc.family == B.SOFTCORE or desc.flags & (B.GROUP_LJ | B.GROUP_Q):
            key = ('group', pid)
        # a force guarded by step(rc0 - r) reaches rc0 only (the discount of FarNonbondedForce): it can then walk the front part
        # of the total force's rows, and be evaluated on the total's pass
        reach = min(float(desc.rc), float(desc.rc0)) if (desc.flags & B.GUARD_RC0 and desc.rc0 > 0) else float(desc.rc)
        self._pair_info[pid] = (reach, key)
        return pid

    def _share_lists(self):
        """RESPASystem leaves a short-ranged copy (group 1, and its negative in group 31) and the full force over the
        same particles and exclusions: the shorter-ranged ones traverse the front part of the longest-ranged force's
        neighbour rows instead of building lists of their own (amm_pair_share_list)."""
        by_key = {}
        for pid, (rc, key) in self._pair_info.items():
            by_key.setdefault(key, []).append((rc, pid))
        self.shared = {}
        for members in by_key.values():
            members.sort(reverse=True)
            host_rc, host = members[0]
            for rc, pid in members[1:]:
                if rc < host_rc:
                    try:
                        self.ctx.pair_share_list(pid, host)
                        self.shared[pid] = host
                    except B.HipError:
                        pass          # incompatible radius with an earlier guest: keeps its own list

    def _finish_entry(self, entry):
        if entry.terms:
            entry.bonded_id = self._make_bonded(entry.terms, sliced=False)

    def _replace_bonded(self, entry, terms):
        """New bond-list set of a force whose terms changed; the set it replaces is freed (amm_bonded_release)."""
        old = entry.bonded_id
        entry.bonded_id = self._make_bonded(terms, sliced=False) if terms else None
        if old is not None:
            self.ctx.bonded_release(old)

    def _forget_groups(self):
        """Group definitions are stale (bond-list terms were rebuilt): drop them and free the merged sets they owned."""
        for bid in self._group_bonded:
            self.ctx.bonded_release(bid)
        del self._group_bonded[:]
        self._group_defs.clear()

    def _make_bonded(self, terms, sliced):
        bid = self.ctx.bonded_create()
        for kind, idx, par, periodic, desc in terms:
            if len(idx):
                self.ctx.bonded_add_terms(bid, kind, idx, par, periodic=periodic, desc=desc)
        self.ctx.bonded_finalize(bid, sliced=sliced)
        return bid

    @staticmethod
    def _effective(base, scales, names, parameters):
        """base (n,3) + sum_p lambda_p * scales[p] (n,3): parameter offsets (forces.py:247-258)."""
        out = base.copy()
        for k, name in enumerate(names):
            out += parameters[name] * scales[k]
        return out

    def _translate_nonbonded(self, nb, entry):
        method = nb.getNonbondedMethod()
        if method not in (nb.CutoffPeriodic, nb.Ewald, nb.PME):
            raise InputError('the HIP path evaluates periodic NonbondedForces only (CutoffPeriodic, Ewald, PME)')
        rc = nb._cutoff
        n = self.n
        base = np.array(nb._particles, dtype=np.float64).reshape(n, 3)
        names = []
        for rec in nb._particle_offsets:
            if rec[0] not in names:
                names.append(rec[0])
        scales = np.zeros((len(names), n, 3))
        for name, idx, qs, ss, es in nb._particle_offsets:
            scales[names.index(name), idx] = [qs, ss, es]
        exc = np.array([[r[0], r[1]] for r in nb._exceptions], dtype=np.int32).reshape(-1, 2)
        exc_base = np.array([r[2:] for r in nb._exceptions], dtype=np.float64).reshape(-1, 3)
        enames = []
        for rec in nb._exception_offsets:
            if rec[0] not in enames:
                enames.append(rec[0])
        escales = np.zeros((len(enames), len(exc), 3))
        for name, idx, qs, ss, es in nb._exception_offsets:
            escales[enames.index(name), idx] = [qs, ss, es]
        flags = B.SWITCH if nb._use_switch else 0
        alpha = krf = crf = 0.0
        ewald = method in (nb.Ewald, nb.PME)
        if ewald:
            alpha = nb._pme[0] if nb._pme[0] > 0 else math.sqrt(-math.log(2 * nb._ewald_tol)) / rc
            flags |= B.COULOMB_EWALD
        else:
            eps_rf = nb._rf_dielectric
            krf = (eps_rf - 1) / ((2 * eps_rf + 1) * rc ** 3)
            crf = 3 * eps_rf / ((2 * eps_rf + 1) * rc)
            flags |= B.COULOMB_RF
        desc = B.pair_desc(B.NONBONDED, rc, rswitch=nb._switch if nb._use_switch else 0.0, alpha=alpha, flags=flags,
                           krf=krf, crf=crf)
        eff = self._effective(base, scales, names, self.parameters)
        pid = self._pair_create(desc, eff[:, 0], eff[:, 1], eff[:, 2], exc)
        entry.pair_ids.append(pid)
        entry.alpha = alpha
        entry.ewald = ewald
        if ewald:
            entry.recip_group = nb._recip_group if nb._recip_group >= 0 else nb.getForceGroup()
            # PME mesh: explicit setPMEParameters, else OpenMM's rule ceil(2 alpha L / (3 tol^(1/5))) [recalled];
            # nonbondedMethod Ewald is evaluated on the same mesh (smooth PME stands in for the explicit k-sum)
            if nb._pme[0] > 0 and min(nb._pme[1:]) > 0:
                grid = [int(k) for k in nb._pme[1:]]
            else:
                grid = [max(6, int(math.ceil(2 * alpha * L / (3 * nb._ewald_tol ** 0.2)))) for L in self.box]
            entry.recip = self.ctx.pme_create(alpha, grid, eff[:, 0])
            entry.recip_grid = grid

        def bonded_terms(parameters):
            q = self._effective(base, scales, names, parameters)[:, 0]
            terms = []
            ep = self._effective(exc_base, escales, enames, parameters) if len(exc) else exc_base
            keep = (ep[:, 0] != 0.0) | (ep[:, 2] != 0.0) if len(exc) else np.zeros(0, bool)
            if keep.any():
                terms.append((B.BOND_LJC, exc[keep], ep[keep], True, B.pair_desc(B.NONBONDED, rc)))
            if ewald and len(exc):
                terms.append((B.BOND_EWALD_EXCL, exc, (q[exc[:, 0]] * q[exc[:, 1]]).reshape(-1, 1), True,
                              B.pair_desc(B.NONBONDED, rc, alpha=alpha)))
            return terms

        defaults = {nb.getGlobalParameterName(i): nb.getGlobalParameterDefaultValue(i)
                    for i in range(nb.getNumGlobalParameters())}

        def constant(parameters):
            # OpenMM evaluates the dispersion coefficient with the global parameters at their DEFAULT values and keeps
            # it when Context.setParameter changes them -- pinned by tests/test_systems.py:54 and :121 (sigma/epsilon
            # offsets at lambda_vdw = 0.5: the literals are met to 2e-7 kJ/mol only this way)
            if not nb._dispersion:
                return 0.0
            p = self._effective(base, scales, names, defaults)
            return dispersion_correction(p[:, 1], p[:, 2], self.box, rc, nb._switch if nb._use_switch else None)

        entry.terms = bonded_terms(self.parameters)
        entry.constant = constant(self.parameters)
        lam = set(names) | set(enames)

        def update(parameters, changed, force=False):
            if not (lam & changed) and not force:
                return False
            p = self._effective(base, scales, names, parameters)
            self.ctx.pair_set_params(pid, p[:, 0], p[:, 1], p[:, 2])
            if entry.recip is not None:
                self.ctx.pme_set_charges(entry.recip, p[:, 0])
            entry.terms = bonded_terms(parameters)
            self._replace_bonded(entry, entry.terms)
            entry.constant = constant(parameters)
            return True

        def reload():
            # NonbondedForce.updateParametersInContext: particle and exception parameters are read again (their number
            # and the exception pairs must not have changed, as in OpenMM)
            fresh = np.array(nb._particles, dtype=np.float64).reshape(n, 3)
            fresh_exc = np.array([r[2:] for r in nb._exceptions], dtype=np.float64).reshape(-1, 3)
            if fresh_exc.shape != exc_base.shape:
                raise mm.OpenMMException('updateParametersInContext: the number of exceptions has changed')
            base[:] = fresh
            exc_base[:] = fresh_exc
            return update(self.parameters, set(), force=True)

        entry.reload = reload
        if lam:
            entry.update = update
            entry.depends = set(lam)
            # parameters whose offsets touch the charges only: the energy is a quadratic form of them
            entry.quadratic_in = {nm for nm in lam
                                  if not (nm in names and np.any(scales[names.index(nm)][:, 1:]))
                                  and not (nm in enames and np.any(escales[enames.index(nm)][:, 1:]))}

    def _descriptor_of(self, force):
        desc = getattr(force, '_amm', None)
        if desc is None:
            globs = {force.getGlobalParameterName(i): force.getGlobalParameterDefaultValue(i)
                     for i in range(force.getNumGlobalParameters())}
            desc = describe_energy(force.getEnergyFunction(), globs)
        if desc is None:
            raise InputError('energy expression not recognised by the HIP path (supported: the AtomsMM near / '
                             'damped-smoothed / exception families): ' + force.getEnergyFunction().split(';')[0])
        return desc

    def _pair_desc_from(self, d, rc, builtin_switch=None):
        family = d['family']
        flags = B.GUARD_RC0 if d.get('guard') else 0
        if family == 'damped':
            degree = int(d.get('degree', 1))
            rswitch = builtin_switch if (degree == 1 and builtin_switch is not None) else d['rswitch']
            return B.pair_desc(B.DAMPED, rc, rswitch=rswitch, alpha=d['alpha'], degree=degree, sign=d.get('sign', 1.0),
                               Kc=d.get('Kc', B.KC))
        rc0, rs0 = d.get('rc0'), d.get('rs0')
        if rc0 is None or rs0 is None:
            raise InputError('near force without rc0/rs0')
        if d.get('noshift'):
            flags |= B.NO_SHIFT
        return B.pair_desc(_FAMILY[family], rc, rc0=rc0, rs0=rs0, flags=flags, sign=d.get('sign', 1.0), Kc=d.get('Kc', B.KC))

    def _translate_custom_nonbonded(self, force, entry):
        d = dict(self._descriptor_of(force))
        if d['family'] == 'ljc':
            raise InputError('the LJC exception expression belongs in a CustomBondForce')
        if force.getNonbondedMethod() != force.CutoffPeriodic:
            raise InputError('the HIP path evaluates CutoffPeriodic CustomNonbondedForces only')
        if d['family'] == 'softcore':
            return self._translate_softcore(force, entry, d)
        if d['family'] == 'lj-virial':
            return self._translate_lj_virial(force, entry)
        if d['family'] == 'lj' or d.get('noshift') or d.get('scale_name') or d.get('scale_text') or force.getNumInteractionGroups() > 0:
            return self._translate_alchemical_pair(force, entry, d)
        if force.getNumInteractionGroups() > 0:
            raise NotImplementedError('interaction groups are supported for the softcore solute-solvent force only')
        if force.getUseLongRangeCorrection():
            raise NotImplementedError('long-range correction is supported for the softcore solute-solvent force only')
        rc = force._cutoff
        for key in ('rc0', 'rs0', 'alpha', 'rswitch', 'Kc'):
            if d.get(key) is None and key in self.parameters:
                d[key] = self.parameters[key]
        builtin = force._switch if force.getUseSwitchingFunction() else None
        if builtin is not None and d['family'] != 'damped':
            raise NotImplementedError('built-in switching function on a near force')
        desc = self._pair_desc_from(d, rc, builtin)
        n = self.n
        if force.getNumParticles() != n:
            raise mm.OpenMMException('CustomNonbondedForce must have exactly as many particles as the System')
        allp = np.array(force._particles, dtype=np.float64).reshape(n, -1)
        names = list(getattr(force, '_offset_parameters', []))
        base = allp[:, :3]
        scales = np.stack([allp[:, 3 * (k + 1):3 * (k + 2)] for k in range(len(names))]) if names else np.zeros((0, n, 3))
        eff = self._effective(base, scales, names, self.parameters)
        excl = np.array(force._exclusions, dtype=np.int32).reshape(-1, 2)
        pid = self._pair_create(desc, eff[:, 0], eff[:, 1], eff[:, 2], excl)
        entry.pair_ids.append(pid)
        if names:
            lam = set(names)

            def update(parameters, changed):
                if not (lam & changed):
                    return False
                p = self._effective(base, scales, names, parameters)
                self.ctx.pair_set_params(pid, p[:, 0], p[:, 1], p[:, 2])
                return True
            entry.update = update
            entry.depends = set(lam)

    def _translate_alchemical_pair(self, force, entry, d, outer=None, outer_depends=()):
        """Pair forces of AlchemicalRespaSystem (systems.py:628-772): force-switched potentials without the constant
        shift, plain Lennard-Jones, optionally restricted to the (solute, solvent) interaction group, optionally
        multiplied by a global parameter (`respa_switch`) and -- as the collective variable of a CustomCVForce -- by
        `outer(parameters)`, the coupling function."""
        n = self.n
        if getattr(force, '_offset_parameters', []):
            raise NotImplementedError('alchemical pair force with parameter offsets')
        if force.getNonbondedMethod() != force.CutoffPeriodic:
            raise InputError('the HIP path evaluates CutoffPeriodic CustomNonbondedForces only')
        p = np.array(force._particles, dtype=np.float64).reshape(n, -1)
        if p.shape[1] == 2:                            # per-particle (sigma, epsilon) only: AlchemicalSystem (systems.py:378-379)
            p = np.concatenate([np.zeros((n, 1)), p], axis=1)
        p = p[:, :3]
        q = p[:, 0].copy()
        ngroups = force.getNumInteractionGroups()
        flags, Kc = 0, d.get('Kc', B.KC)
        codes = None
        if ngroups > 1:
            raise NotImplementedError('more than one interaction group')
        sigma, eps = p[:, 1], p[:, 2]
        if ngroups == 1:
            if not (d['family'] == 'lj' or d.get('lj_only') or d.get('coulomb_only')):
                raise NotImplementedError('interaction groups on a force with both electrostatics and Lennard-Jones')
            set1, set2 = force._groups[0]
            if set1 & set2:
                raise NotImplementedError('overlapping interaction-group sets')
            codes = np.zeros(n)
            codes[sorted(set1)] = 1.0
            codes[sorted(set2)] = 2.0
            if d.get('coulomb_only'):      # the expression has no sigma / epsilon: the sigma slot selects the pairs
                sigma, eps = 2.0 * codes, np.zeros(n)
                flags |= B.GROUP_Q
            else:
                q, Kc = codes, 1.0
                flags |= B.GROUP_LJ
        elif d['family'] == 'lj' or d.get('lj_only'):
            q = np.zeros(n)
        rc = force._cutoff
        rswitch = force._switch if force.getUseSwitchingFunction() else None
        scale_name = d.get('scale_name')
        scale_text = d.get('scale_text')
        scale_symbols = X.symbols(scale_text) if scale_text else set()
        for name in scale_symbols:
            if name not in self.parameters:          # e.g. the reference's `linear` coupling leaves two_pi undefined
                raise mm.OpenMMException('Unknown variable in expression: ' + name)

        def scale(parameters):
            value = d.get('sign', 1.0)
            if scale_name:
                value *= parameters[scale_name]
            if scale_text:
                value *= X.eval_global(scale_text, {k: parameters[k] for k in scale_symbols})
            if outer is not None:
                value *= outer(parameters)
            return value

        if d['family'] == 'lj':
            if rswitch is not None:
                flags |= B.SWITCH
            desc = B.pair_desc(B.NONBONDED, rc, rswitch=rswitch or 0.0, flags=flags, sign=scale(self.parameters), Kc=Kc)
        elif d['family'] == 'near-force-switch':
            if rswitch is not None:
                raise NotImplementedError('built-in switching function on a force-switched potential')
            if d.get('noshift'):
                flags |= B.NO_SHIFT
            if d.get('guard'):
                flags |= B.GUARD_RC0
            desc = B.pair_desc(B.NEAR_FSWITCH, rc, rc0=d.get('rc0') or rc, rs0=d['rs0'], flags=flags,
                               sign=scale(self.parameters), Kc=Kc)
        else:
            raise NotImplementedError('alchemical pair family ' + d['family'])
        excl = np.array(force._exclusions, dtype=np.int32).reshape(-1, 2)
        pid = self._pair_create(desc, q, sigma, eps, excl)
        entry.pair_ids.append(pid)
        if d.get('coulomb_only'):
            def reload():          # updateParametersInContext: the charges may have changed (reset_coulomb_scaling_factor)
                fresh = np.array(force._particles, dtype=np.float64).reshape(n, -1)[:, 0]
                self.ctx.pair_set_params(pid, fresh, sigma, eps)
                return 'values'
            entry.reload = reload
        lrc = 0.0
        if force.getUseLongRangeCorrection():
            if d['family'] != 'lj':
                raise NotImplementedError('long-range correction of this CustomNonbondedForce')
            lrc = custom_long_range_correction(lambda r, s_, e_: 4.0 * e_ * ((s_ / r) ** 12 - (s_ / r) ** 6), p[:, 1], p[:, 2],
                                               self.box, rc, rswitch, codes)
        entry.constant = lrc * scale(self.parameters)
        depends = set(outer_depends) | ({scale_name} if scale_name else set()) | scale_symbols
        if depends:
            def update(parameters, changed):
                if not (depends & changed):
                    return False
                self.ctx.pair_set_scale(pid, scale(parameters))
                entry.constant = lrc * scale(parameters)
                return 'values'
            entry.update = update
            entry.depends = set(depends)

    def _translate_cv(self, force, entry):
        """CustomCVForce of AlchemicalRespaSystem (systems.py:738-772): ((gt0-gt1)*S(lambda) + gt1) times ONE collective
        variable, the energy of a CustomNonbondedForce -- i.e. that pair force with a lambda-dependent overall factor."""
        if force.getNumCollectiveVariables() != 1:
            raise NotImplementedError('CustomCVForce with %d collective variables' % force.getNumCollectiveVariables())
        name, inner = force.getCollectiveVariableName(0), force.getCollectiveVariable(0)
        if not isinstance(inner, mm.CustomNonbondedForce):
            raise NotImplementedError('CustomCVForce over a ' + inner.__class__.__name__)
        text = force.getEnergyFunction()
        used = X.symbols(text) - {name}
        self._register_globals(inner)

        def outer(parameters):
            env = {k: parameters[k] for k in used}
            e1 = X.eval_global(text, dict(env, **{name: 1.0}))
            e0 = X.eval_global(text, dict(env, **{name: 0.0}))
            e2 = X.eval_global(text, dict(env, **{name: 2.0}))
            if e0 != 0.0 or abs(e2 - 2.0 * e1) > 1e-12 * max(1.0, abs(e1)):
                raise NotImplementedError('CustomCVForce energy is not proportional to its collective variable')
            return e1
        d = dict(self._descriptor_of(inner))
        self._translate_alchemical_pair(inner, entry, d, outer=outer, outer_depends=used)

    def _translate_lj_virial(self, force, entry):
        """ComputingSystem's dispersion virial (systems.py:893-897): a CustomNonbondedForce with the cutoff, switch and
        long-range-correction settings imported from the NonbondedForce."""
        n = self.n
        if force.getNumInteractionGroups() > 0 or getattr(force, '_offset_parameters', []):
            raise NotImplementedError('virial force with interaction groups or parameter offsets')
        p = np.array(force._particles, dtype=np.float64).reshape(n, -1)[:, :3]
        rc = force._cutoff
        rswitch = force._switch if force.getUseSwitchingFunction() else None
        excl = np.array(force._exclusions, dtype=np.int32).reshape(-1, 2)
        desc = B.pair_desc(B.LJ_VIRIAL, rc, rswitch=rswitch or 0.0, flags=B.SWITCH if rswitch is not None else 0)
        entry.pair_ids.append(self._pair_create(desc, p[:, 0], p[:, 1], p[:, 2], excl))
        if force.getUseLongRangeCorrection():
            entry.constant = custom_long_range_correction(
                lambda r, s, e: 24.0 * e * (2.0 * (s / r) ** 12 - (s / r) ** 6), p[:, 1], p[:, 2], self.box, rc, rswitch)

    def _translate_softcore(self, force, entry, d):
        """SolvationSystem's softcore CustomNonbondedForce (systems.py:266-272): one interaction group solute x
        solvent, cutoff / built-in switch / long-range correction imported from the NonbondedForce
        (forces.py:284-291), lambda_vdw a global parameter."""
        n = self.n
        if force.getNumInteractionGroups() != 1:
            raise NotImplementedError('softcore force: exactly one interaction group is supported')
        set1, set2 = force._groups[0]
        if set1 & set2:
            raise NotImplementedError('softcore force: overlapping interaction-group sets')
        names = list(getattr(force, '_offset_parameters', []))
        allp = np.array(force._particles, dtype=np.float64).reshape(n, -1)
        if allp.shape[1] == 2:                         # per-particle (sigma, epsilon) only: AlchemicalSoftcoreCVForce
            allp = np.concatenate([np.zeros((n, 1)), allp], axis=1)
        base = allp[:, :3]
        scales = np.stack([allp[:, 3 * (k + 1):3 * (k + 2)] for k in range(len(names))]) if names else np.zeros((0, n, 3))
        codes = np.zeros(n)
        codes[sorted(set1)] = 1.0
        codes[sorted(set2)] = 2.0
        lam_name = d['lambda_name']
        if 'lambda_value' in d:                        # a constant written into the expression, not a Context parameter
            lam_name = '__softcore_lambda_%d' % len(self.entries)
            self.parameters[lam_name] = float(d['lambda_value'])
        rc = force._cutoff
        rswitch = force._switch if force.getUseSwitchingFunction() else None
        excl = np.array(force._exclusions, dtype=np.int32).reshape(-1, 2)
        eff = self._effective(base, scales, names, self.parameters)
        scale_name = d.get('scale_name')
        desc = B.pair_desc(B.SOFTCORE, rc, rswitch=rswitch or 0.0, alpha=self.parameters[lam_name],
                           flags=B.SWITCH if rswitch is not None else 0, Kc=1.0,
                           sign=self.parameters[scale_name] if scale_name else 1.0)
        pid = self._pair_create(desc, codes, eff[:, 1], eff[:, 2], excl)
        entry.pair_ids.append(pid)
        use_lrc = force.getUseLongRangeCorrection()

        class_pairs = {}        # the (sigma, eps) classes depend on the offset parameters only, not on lambda

        def classes_of(parameters):
            key = tuple(parameters[name] for name in names)
            if key not in class_pairs:
                p = self._effective(base, scales, names, parameters)
                class_pairs.clear()
                class_pairs[key] = [lrc_class_pairs(p[:, 1], p[:, 2], codes), p[:, 1], p[:, 2], None]
            return class_pairs[key]

        def quadrature(parameters, lam_value):
            record = classes_of(parameters)
            return record, softcore_long_range_correction(record[1], record[2], codes, self.box, rc, rswitch, lam_value, record[0])

        def on_unit_interval(parameters):
            """The correction as a function of lambda on [0, 1] (it is analytic there): a Chebyshev interpolant through 24 of
            the quadrature's values, made once per set of offset parameters -- an AFED step asks for the correction and
            its lambda-derivative some twenty times, at 0.4 ms of host quadrature each otherwise."""
            record = classes_of(parameters)
            if record[3] is None:
                series = np.polynomial.Chebyshev.interpolate(
                    lambda nodes: np.array([quadrature(parameters, float(x))[1] for x in nodes]), 23, domain=[0.0, 1.0])
                record[3] = (series, series.deriv())
            return record[3]

        def constant(parameters):
            if not use_lrc:
                return 0.0
            lam_value = parameters[lam_name]
            if 0.0 <= lam_value <= 1.0:
                value = float(on_unit_interval(parameters)[0](lam_value))
            else:
                value = quadrature(parameters, lam_value)[1]
            return value * (parameters[scale_name] if scale_name else 1.0)

        def constant_derivative(parameters):
            """d(correction)/d(lambda)"""
            if not use_lrc:
                return 0.0
            lam_value = parameters[lam_name]
            if 0.0 <= lam_value <= 1.0:
                value = float(on_unit_interval(parameters)[1](lam_value))
            else:
                h = 1e-6
                value = (quadrature(parameters, lam_value + h)[1] - quadrature(parameters, lam_value - h)[1]) / (2 * h)
            return value * (parameters[scale_name] if scale_name else 1.0)

        def derivative_polynomial(parameters):
            """d(correction)/d(lambda) on [0, 1] as monomial coefficients in u = 2 lambda - 1 (lowest first), for the evaluation on
            the device while lambda lives there (amm_expr_eval_scalar: Horner) -- the interpolant's derivative, as constant_derivative."""
            if not use_lrc:
                return []
            record = classes_of(parameters)
            if len(record) < 5:          # (made once per set of offset parameters, like the interpolant)
                cheb = np.array(on_unit_interval(parameters)[1].coef, dtype=np.float64)
                # (the trailing Chebyshev terms of an analytic function are below rounding: |T_k| <= 1 bounds what is dropped)
                keep = len(cheb)
                while keep > 1 and np.abs(cheb[keep - 1:]).sum() < 1e-15 * np.abs(cheb).max():
                    keep -= 1
                record.append([float(c) for c in np.polynomial.chebyshev.cheb2poly(cheb[:keep])])
            scale = parameters[scale_name] if scale_name else 1.0
            return record[4] if scale == 1.0 else [c * scale for c in record[4]]

        entry.constant = constant(self.parameters)
        lam = set(names) | {lam_name} | ({scale_name} if scale_name else set())
        entry.softcore = dict(pid=pid, lambda_name=lam_name, constant=constant, constant_derivative=constant_derivative, depends=lam,
                              use_lrc=bool(use_lrc), derivative_polynomial=derivative_polynomial)

        def update(parameters, changed):
            if not (lam & changed):
                return False
            if set(names) & changed:
                p = self._effective(base, scales, names, parameters)
                self.ctx.pair_set_params(pid, codes, p[:, 1], p[:, 2])
            self.ctx.pair_set_lambda(pid, parameters[lam_name])
            if scale_name:
                self.ctx.pair_set_scale(pid, parameters[scale_name])
            entry.constant = constant(parameters)
            return 'values'            # no bond-list terms changed: group definitions stay
        entry.update = update
        entry.depends = set(lam)

    def _translate_custom_bond(self, force, entry):
        if force.getEnergyFunction().replace(' ', '') == '0.5*(K0*(r-r0)^2-Kn*(r-rn)^2)':
            # difference of two harmonic bonds (RESPASystem.redefine_bond, systems.py:162): +K0 at r0, -Kn at rn
            idx = np.array([[b[0], b[1]] for b in force._bonds], dtype=np.int32).reshape(-1, 2)
            par = np.array([b[2] for b in force._bonds], dtype=np.float64).reshape(-1, 4)
            periodic = force.usesPeriodicBoundaryConditions()
            entry.terms.append((B.BOND_HARMONIC, idx, par[:, [0, 1]], periodic, None))
            entry.terms.append((B.BOND_HARMONIC, idx, np.stack([par[:, 2], -par[:, 3]], axis=1), periodic, None))
            return
        if force.getEnergyFunction().replace(' ', '') == '-K*r*(r-r0)':
            # bond-stretching virial of ComputingSystem (systems.py:914): per-bond (r0, K)
            idx = np.array([[b[0], b[1]] for b in force._bonds], dtype=np.int32).reshape(-1, 2)
            par = np.array([b[2] for b in force._bonds], dtype=np.float64).reshape(-1, 2)
            entry.terms.append((B.BOND_VIRIAL_HARMONIC, idx, par, force.usesPeriodicBoundaryConditions(), None))
            return
        d = dict(self._descriptor_of(force))
        nb_ = force.getNumBonds()
        idx = np.array([[b[0], b[1]] for b in force._bonds], dtype=np.int32).reshape(-1, 2)
        allp = np.array([b[2] for b in force._bonds], dtype=np.float64).reshape(nb_, -1)
        names = list(getattr(force, '_offset_parameters', []))
        base = allp[:, :3]
        scales = np.stack([allp[:, 3 * (k + 1):3 * (k + 2)] for k in range(len(names))]) if names else np.zeros((0, nb_, 3))
        periodic = force.usesPeriodicBoundaryConditions()
        if d['family'] == 'ljc':
            kind, desc = B.BOND_LJC, B.pair_desc(B.NONBONDED, 1.0, Kc=d.get('Kc', B.KC))
        elif d['family'] == 'lj-virial':
            kind, desc = B.BOND_VIRIAL_LJ, None
        else:
            for key in ('rc0', 'rs0', 'Kc'):
                if d.get(key) is None and key in self.parameters:
                    d[key] = self.parameters[key]
            kind, desc = B.BOND_NEAR, self._pair_desc_from(d, d['rc0'])

        scale_name = d.get('scale_name')
        if scale_name and kind != B.BOND_NEAR:
            raise NotImplementedError('global scale factor on a bond force of this kind')

        def terms(parameters):
            dsc = desc
            if scale_name:      # respa_switch (systems.py:643, 674): the whole energy times a global parameter
                dsc = self._pair_desc_from(dict(d, sign=d.get('sign', 1.0) * parameters[scale_name]), d['rc0'])
            return [(kind, idx, self._effective(base, scales, names, parameters), periodic, dsc)]

        entry.terms = terms(self.parameters)
        if names or scale_name:
            lam = set(names) | ({scale_name} if scale_name else set())

            def update(parameters, changed):
                if not (lam & changed):
                    return False
                entry.terms = terms(parameters)
                self._replace_bonded(entry, entry.terms)
                return True
            entry.update = update
            entry.depends = set(lam)

    # ------------------------------------------------------------------------------- state
    def _invalidate_forces(self):
        for g in self._valid:
            self._valid[g] = False
        self._deriv_cache.clear()

    def set_positions(self, arr):
        self.x.copy_(self.torch.as_tensor(arr, device=self.x.device))
        if hasattr(self.ctx, 'positions_changed'):
            self.ctx.positions_changed()
        self._invalidate_forces()

    def set_velocities(self, arr):
        self.v.copy_(self.torch.as_tensor(arr, device=self.v.device))

    def set_parameter(self, name, value):
        if name not in self.parameters:
            raise mm.OpenMMException('Called setParameter() with invalid parameter name: ' + name)
        if self.parameters[name] == value:
            return
        self.parameters[name] = value
        changed = {name}
        groups, rebuilt = set(), False
        for entry in self.entries:
            if entry.update is not None:
                result = entry.update(self.parameters, changed)
                if result:
                    groups.add(entry.group)
                    if getattr(entry, 'recip', None) is not None:
                        groups.add(entry.recip_group)
                    if result != 'values':
                        rebuilt = True
                        self._forget_groups()       # bond-list terms were rebuilt
        if rebuilt:
            self._programs.clear()
            self._emit_memo.clear()
            self._segment_memo.clear()
            self._invalidate_forces()
        elif groups:
            # only VALUES changed (an extended variable moved: AFED sets lambda_vdw twice per step): the forces of the groups that
            # hold a force that depends on it are stale -- the other groups' buffers, the compiled programs and what the step
            # programs emit are not (invalidating everything cost config C5 a fifth outer + near evaluation per AFED step)
            for g in self._valid:
                if g in groups or g == 'all':
                    self._valid[g] = False
            self._deriv_cache.clear()
            self._programs.clear()          # (compiled programs may hold coefficients that name the parameter; the memos of the
                                            # host-walked path are keyed by the values of the globals they name)

    def get_parameter(self, name):
        return self.parameters[name]

    def update_force_parameters(self, force):
        """`force.updateParametersInContext(context)`: per-particle (and exception) parameters of an existing force are
        uploaded again -- what AlchemicalRespaSystem.reset_coulomb_scaling_factor relies on (systems.py:806-815)."""
        hits = [e for e in self.entries if e.force is force]
        if not hits:
            raise mm.OpenMMException('updateParametersInContext: the force does not belong to this Context')
        for entry in hits:
            reload = getattr(entry, 'reload', None)
            if reload is None:
                raise NotImplementedError('updateParametersInContext for ' + type(force).__name__ + ' with this energy expression')
            result = reload()
            if result:
                if result != 'values':
                    self._forget_groups()
                self._programs.clear()
                self._emit_memo.clear()
                self._segment_memo.clear()
                self._invalidate_forces()

    def invalidate_program(self):
        self._programs.clear()
        self._emit_memo.clear()
        self._segment_memo.clear()
        self._interpreted = None

    def reinitialize(self, preserveState=False):
        raise NotImplementedError('Context.reinitialize: create a new Context instead')

    # per-DOF variables live in named device buffers
    def _buffer(self, name, together=()):
        """Per-DOF buffer `name`.  Buffers are carved from arenas; `together` names buffers to create in the same breath as
        neighbours in memory: the force buffers of the sliced groups, whose all-reduces then travel as one message when
        they follow one another (AMM_OP_ALLREDUCE)."""
        if name not in self._buffers:
            names = [name] + [other for other in together if other != name and other not in self._buffers]
            if len(self._arena_free) < len(names) or (len(names) > 1 and not self._arena_contiguous):
                count = max(8, len(names))
                arena = self.torch.zeros((count, self.n, 3), dtype=self.torch.float64, device=self.x.device)
                leftovers = self._arena_free
                self._arena_free = [arena[k] for k in range(count)]
                self._arena_contiguous = True
            else:
                leftovers = []
            integ = self.integrator
            for nm in sorted(names):
                t = self._arena_free.pop(0)
                if isinstance(integ, mm.CustomIntegrator) and nm in integ._pnames:
                    t.fill_(integ._pvalues[integ._pnames.index(nm)])
                self._buffers[nm] = t
            if leftovers:
                self._arena_free += leftovers
                self._arena_contiguous = False
        return self._buffers[name]

    def fill_per_dof(self, name, value):
        self._buffer(name).fill_(value)
        self._mirror.pop(name, None)

    def get_per_dof(self, name):
        src = self._mirror.get(name)
        if src is not None and _AUX_FORCE.fullmatch(name) and src in self._buffers and not _NO_ALIAS:
            self._buffer(name).copy_(self._buffers[src])        # its copy op may have been dropped (_drop_dead_copies)
        return [mm.Vec3(*row) for row in self._buffer(name).cpu().numpy().tolist()]

    def set_per_dof(self, name, values):
        arr = np.array([list(v) for v in values], dtype=np.float64).reshape(self.n, 3)
        self._buffer(name).copy_(self.torch.as_tensor(arr, device=self.x.device))
        self._mirror.pop(name, None)

    # ------------------------------------------------------------------------------- getState
    def _check(self):
        """amm_check on every rank TOGETHER: a neighbour-row overflow or a constraint failure is detected by the rank
        that owns the row, and a rank that raised alone would leave its peers inside the next collective."""
        if not self._coll:
            return self.ctx.check()
        err = None
        try:
            self.ctx.check()
        except Exception as exc:       # noqa: BLE001 -- re-raised below, on every rank
            err = exc
        if self._local is not None:
            flag = self.torch.tensor([1 if err is not None else 0], dtype=self.torch.int32)
            self._local.all_reduce(self.rank, flag, op='max')
            if err is not None:
                raise err
            if int(flag.item()):
                raise B.HipError('another rank reported a failed check (neighbour-row overflow or constraint failure)')
            return
        dist = self.torch.distributed
        flag = self.torch.tensor([1 if err is not None else 0], dtype=self.torch.int32,
                                 device=self.x.device if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if err is not None:
            raise err
        if int(flag.item()):
            raise B.HipError('another rank reported a failed check (neighbour-row overflow or constraint failure)')

    def broadcast_from_rank0(self, array):
        """A host array every rank must hold bit for bit (e.g. randomly drawn velocities): rank 0's copy wins."""
        if not self._coll:
            return array
        if self._local is not None:
            return np.array(self._local.broadcast(self.rank, np.ascontiguousarray(array, dtype=np.float64)))
        dist = self.torch.distributed
        t = self.torch.as_tensor(np.ascontiguousarray(array, dtype=np.float64),
                                 device=self.x.device if dist.get_backend() == 'nccl' else 'cpu')
        dist.broadcast(t, src=0)
        return t.cpu().numpy()

    def _allreduce(self, tensor):
        if self._coll:
            if self._local is not None:
                self._local.all_reduce(self.rank, tensor)
            elif self._native_comm:
                self.ctx.comm_allreduce(tensor)
            else:
                self.torch.distributed.all_reduce(tensor)

    def get_state(self, want_pos, want_vel, want_forces, want_energy, mask):
        torch = self.torch
        energy = forces = None
        if want_forces or want_energy:
            e_pair = self._energy.zero_() if want_energy else None
            fp = self._fwork.zero_()
            fb = self._fwork2.zero_()
            e_bond = torch.zeros(1, dtype=torch.float64, device=self.x.device) if want_energy else None
            const = 0.0
            for entry in self.entries:
                if mask & (1 << entry.group):
                    for pid in entry.pair_ids:
                        self.ctx.force_eval(pid, self.x, fp, accumulate=True, energy=e_pair)
                    if entry.bonded_id is not None:
                        self.ctx.force_eval(entry.bonded_id, self.x, fb, accumulate=True, energy=e_bond)
                    const += entry.constant
                if entry.recip is not None and mask & (1 << entry.recip_group):
                    # reciprocal space: every rank evaluates all of it (spread + FFTs are not sharded)
                    self.ctx.pme_set_sliced(entry.recip, False)
                    self.ctx.force_eval(entry.recip, self.x, fb, accumulate=True, energy=e_bond)
                    self.ctx.pme_set_sliced(entry.recip, getattr(entry, 'recip_sliced', False))
            self._check()
            if self._coll:
                self._allreduce(fp)
                if want_energy:
                    self._allreduce(e_pair)
            if want_forces:
                forces = (fp + fb).cpu().numpy()
            if want_energy:
                energy = e_pair.item() + e_bond.item() + const
        kinetic = None
        if want_energy:
            out = torch.zeros(1, dtype=torch.float64, device=self.x.device)
            self.ctx.mvv(self.v, self.mass, out)
            kinetic = 0.5 * out.item()
        pos = self.x.cpu().numpy() if want_pos else None
        vel = self.v.cpu().numpy() if want_vel else None
        box = [(self.box[0], 0, 0), (0, self.box[1], 0), (0, 0, self.box[2])]
        return mm.State(energy, kinetic, forces, pos, vel, box, self.time)

    # ------------------------------------------------------------------------------- step programs
    def _slot(self, name):
        """Slot index of a named per-DOF buffer ('f3' = force of group 3, other names = per-DOF variables)."""
        if name == 'x':
            return B.SLOT_X
        if name == 'v':
            return B.SLOT_V
        if name not in self._slots:
            slot = len(self._slots)
            if slot >= B.SLOT_X:
                raise NotImplementedError('too many per-DOF buffers')
            self._slots[name] = slot
            self.ctx.bind_buffer(slot, self._buffer(name))
        return self._slots[name]

    def _define_group(self, g):
        """Backend group g = pair forces + one merged bonded set of all bond-list terms in the group."""
        if g in self._group_defs:
            return self._group_defs[g]
        members = [e for e in self.entries if (e.group == g if g != 'all' else True)]
        pair_ids = [pid for e in members for pid in e.pair_ids]
        terms = [t for e in members for t in e.terms]
        # multi-rank: a group is evaluated slice-wise and all-reduced as soon as it holds a pair force OR a reciprocal-space
        # force -- the "sliced" switch lives on the PME object, so it must mean the same in every group that evaluates it
        has_recip = any(e.recip is not None and (g == 'all' or e.recip_group == g) for e in self.entries)
        reduced = self._coll and (bool(pair_ids) or has_recip)
        gather = reduced and self._gather and g != 'all' and len(pair_ids) == 1 and not terms and not any(
            e.recip is not None and e.recip_group == g for e in self.entries)
        # a hybrid list (molecule rows + per-atom rows for the atoms outside the three-site molecules) has two sorted orders:
        # its forces are exchanged by all-reduce
        if gather and self.ctx.pair_stats(pair_ids[0]).get('n_rest_atoms', 0) > 0:
            gather = False
        ids = list(pair_ids)
        if terms:
            merged = self._make_bonded(terms, sliced=reduced)
            self._group_bonded.append(merged)
            # the first member of a group writes the buffer, the others add to it: an interaction-group force touches a few rows
            # only (it would have to clear the rest first), so the bond lists go first in its group
            if any(e.softcore is not None for e in members):
                ids.insert(0, merged)
            else:
                ids.append(merged)
        for e in self.entries:
            if e.recip is not None and (g == 'all' or e.recip_group == g):
                e.recip_sliced = reduced
                self.ctx.pme_set_sliced(e.recip, reduced)
                ids.append(e.recip)
        index = B.GROUP_ALL if g == 'all' else int(g)
        if reduced and g != 'all':
            self._buffer('f{}'.format(g), together=['f{}'.format(e.group) for e in self.entries if e.pair_ids])
        slot = self._slot('f' if g == 'all' else 'f{}'.format(g))
        self.ctx.group_define(index, slot, ids)
        if gather:       # the EVAL op exchanges the slices itself: no all-reduce of the buffer afterwards
            self.ctx.group_set_exchange(index, B.EXCHANGE_GATHER)
            self._gather_groups.add(index)
            reduced = False
        self._group_defs[g] = (index, slot, reduced)
        return self._group_defs[g]

    def _emit_constraint(self, kind, ops, valid):
        """addConstrainPositions / addConstrainVelocities (propagators.py:250, 272): identity without constraints."""
        if not self._has_constraints:
            return
        if kind == mm.CustomIntegrator.ConstrainPositions:
            ops.append(B.Op(B.OP_CONSTRAIN_X, 0, 0, 0, 0.0))
            for g in valid:
                valid[g] = False
            self._deriv_cache.clear()
        else:
            ops.append(B.Op(B.OP_CONSTRAIN_V, 0, 0, 0, 0.0))

    def apply_constraints(self):
        if self._has_constraints:
            self.ctx.run_ops([B.Op(B.OP_SAVE_REF, 0, 0, 0, 0.0), B.Op(B.OP_CONSTRAIN_X, 0, 0, 0, 0.0)], 1)
            self._invalidate_forces()
            self._check()

    def apply_velocity_constraints(self):
        if self._has_constraints:
            self.ctx.run_ops([B.Op(B.OP_CONSTRAIN_V, 0, 0, 0, 0.0)], 1)
            self._check()

    def _eval(self, expr, env):
        code = _CODE_CACHE.get(expr)
        if code is None:
            code = _CODE_CACHE[expr] = compile(expr.replace('^', '**').strip(), '<step program>', 'eval')
        return float(eval(code, {'__builtins__': {}}, env))

    def _compile(self):
        """Unroll one outer step of the CustomIntegrator program into backend ops (host-side control flow)."""
        integ = self.integrator
        steps = integ._steps
        C = mm.CustomIntegrator
        # block structure
        match, stack = {}, []
        for pc, (kind, _, _) in enumerate(steps):
            if kind in (C.IfBlock, C.WhileBlock):
                stack.append(pc)
            elif kind == C.EndBlock:
                begin = stack.pop()
                match[begin], match[pc] = pc, begin
        env = dict(_SAFE_FUNCS)
        env.update(self.parameters)
        env.update(zip(integ._gnames, integ._gvalues))
        env['dt'] = integ._dt
        valid = dict(self._valid)
        self._mirror_work = dict(self._mirror)
        self._static_exprs = True
        self._iso_found, self._plain_kick = None, False      # isokinetic (SIN(R)) kicks recognised / ordinary kicks emitted
        ops = [B.Op(B.OP_SAVE_REF, 0, 0, 0, 0.0)] if self._has_constraints else []
        pc = 0
        guard = 0
        while pc < len(steps):
            guard += 1
            if guard > 2_000_000:
                raise RuntimeError('step program does not terminate')
            kind, target, expr = steps[pc]
            if kind == C.ComputeGlobal:
                value = self._eval(expr, env)
                if target in self.parameters and target not in integ._gnames:
                    if value != self.parameters[target]:
                        raise NotImplementedError('step programs that change Context parameters (%s) are not supported' % target)
                env[target] = value
            elif kind == C.ComputePerDof:
                skip = self._emit_native_bath_block(steps, pc, env, ops) or self._emit_native_iso_block(steps, pc, env, ops, valid)
                if skip:
                    pc += skip
                    continue
                self._emit_per_dof(target, expr, env, ops, valid)
            elif kind == C.ComputeSum:
                raise NotImplementedError('ComputeSum steps (thermostat propagators) are outside this round\'s scope')
            elif kind in (C.ConstrainPositions, C.ConstrainVelocities):
                self._emit_constraint(kind, ops, valid)
            elif kind == C.UpdateContextState:
                pass
            elif kind in (C.IfBlock, C.WhileBlock):
                if not self._condition(expr, env):
                    pc = match[pc]
            elif kind == C.EndBlock:
                if steps[match[pc]][0] == C.WhileBlock:
                    pc = match[pc] - 1
            pc += 1
        self._static_exprs = False
        if self._iso_found and self._plain_kick:
            # isokinetic AND ordinary kicks in one program: the context-wide isokinetic mode cannot serve both -- run the
            # isokinetic ones as general expressions
            self._no_native_iso = True
            return self._compile()
        if hasattr(self.ctx, 'iso_define'):
            if self._iso_found:
                LkT, Q1, v1 = self._iso_found
                self.ctx.iso_define(True, LkT, Q1, self._slot(v1))
            else:
                self.ctx.iso_define(False)
        finals = {name: env[name] for name in integ._gnames}
        return self._drop_dead_copies(self._pair_up_evals(ops)), valid, finals, dict(self._mirror_work)

    _NHL_SCALE = re.compile(r'v\*exp\(-\((.+)\*dt\)\*(\w+)\)')
    _NHL_UPDATE = re.compile(r'z\*(\w+)\+sqrt\(kT\*\(1-z\*z\)/mass\)\*gaussian\+force\*\(1-z\)/\(mass\*friction\);force=m\*v\^2-kT;'
                             r'mass=(\w+);z=exp\(-\((.+)\*dt\)\*friction\)')

    def _emit_native_bath_block(self, steps, pc, env, ops):
        """The three per-DOF steps of a Nose-Hoover-Langevin bath (NHL_R_Integrator, integrators.py:272-330:
        `v <- v*exp(-(h*dt)*v2)` ; `v2 <- z*v2 + sqrt(kT*(1 - z*z)/mass)*gaussian + force*(1 - z)/(mass*friction); ...` ;
        `v <- v*exp(-(h*dt)*v2)`) become ONE native bath op, which the inner-loop kernel carries between its two half moves
        like the Ornstein-Uhlenbeck step of Langevin_R.  Returns the number of program steps consumed (0: not such a block)."""
        C = mm.CustomIntegrator
        if pc + 2 >= len(steps) or any(steps[pc + k][0] != C.ComputePerDof for k in range(3)):
            return 0
        (_, t0, e0), (_, t1, e1), (_, t2, e2) = steps[pc:pc + 3]
        a, b, c = self._NHL_SCALE.fullmatch(e0.replace(' ', '')), self._NHL_UPDATE.fullmatch(e1.replace(' ', '')), \
            self._NHL_SCALE.fullmatch(e2.replace(' ', ''))
        if not (a and b and c and t0 == 'v' and t2 == 'v' and e0 == e2):
            return 0
        w = a.group(2)
        if not (t1 == w and b.group(1) == w and w in self.integrator._pnames and b.group(2) in env and 'kT' in env and 'friction' in env):
            return 0
        h = self._eval(a.group(1), env) * env['dt']
        z = math.exp(-(self._eval(b.group(3), env) * env['dt']) * env['friction'])
        key = ('nhl', h, z, float(env['kT']), float(env[b.group(2)]), float(env['friction']), w)
        if key not in self._expr_ids:
            self._expr_ids[key] = self.ctx.bath_define_nhl(h, z, float(env['kT']), float(env[b.group(2)]), float(env['friction']),
                                                           self._slot(w))
        ops.append(B.Op(B.OP_BATH, self._expr_ids[key], B.SLOT_V, 0, 0.0))
        self._mirror_work.pop(w, None)
        return 3

    _ISO_KICK = re.compile(r'v\*cosh\(z\)\+sqrt\(LkT/m\)\*sinh\(z\);z=(.+)/sqrt\(m\*LkT\)')
    _ISO_H = re.compile(r'sqrt\(LkT/\(m\*v\^2\+0\.5\*Q1\*\((\w+)\^2\)\)\)')
    _SIN_SCALE = re.compile(r'(\w+)\*exp\(-\((.+)\*dt\)\*(\w+)\)')
    _SIN_UPDATE = re.compile(r'z\*(\w+)\+sqrt\(kT\*\(1-z\*z\)/mass\)\*gaussian\+force\*\(1-z\)/\(mass\*friction\);force=Q1\*(\w+)\^2-kT;'
                             r'mass=(\w+);z=exp\(-\((.+)\*dt\)\*friction\)')

    def _iso_rescale_at(self, steps, pc, env):
        """`H <- sqrt(LkT/(m*v^2 + 0.5*Q1*(v1^2)))` ; `v <- H*v` ; `v1 <- H*v1` at steps[pc:pc+3] (SIN(R) with L = 1,
        propagators.py:300-320): the name of v1, or None."""
        C = mm.CustomIntegrator
        if pc + 2 >= len(steps) or any(steps[pc + k][0] != C.ComputePerDof for k in range(3)):
            return None
        (_, t0, e0), (_, t1, e1), (_, t2, e2) = steps[pc:pc + 3]
        m = self._ISO_H.fullmatch(e0.replace(' ', ''))
        if not (m and t0 == 'H' and t1 == 'v' and e1.replace(' ', '') == 'H*v' and t2 == m.group(1) and
                e2.replace(' ', '') == 'H*' + m.group(1) and m.group(1) in self.integrator._pnames and 'LkT' in env and 'Q1' in env):
            return None
        found = (float(env['LkT']), float(env['Q1']), m.group(1))
        if self._iso_found not in (None, found):
            return None
        return m.group(1)

    def _emit_native_iso_block(self, steps, pc, env, ops, valid):
        """SIN(R) with one thermostat per DOF (SIN_R_Integrator, integrators.py:358-416).  (a) The isokinetic kick --
        `v <- v*cosh(z) + sqrt(LkT/m)*sinh(z); z = (c*dt)*(F)/sqrt(m*LkT)` followed by the rescale triple -- becomes an ordinary
        KICK op of a context in isokinetic mode (amm_iso_define); (b) the bath block between the two half moves -- `v1 <-
        v1*exp(-(h*dt)*v2)`, rescale, the Ornstein-Uhlenbeck-like update of v2 driven by Q1*v1^2 - kT, the scaling and the rescale
        again -- becomes ONE native bath op.  Returns the number of program steps consumed (0: neither)."""
        if getattr(self, '_no_native_iso', False) or not hasattr(self.ctx, 'iso_define'):
            return 0
        C = mm.CustomIntegrator
        kind, target, expr = steps[pc]
        text = expr.replace(' ', '')
        kick = self._ISO_KICK.fullmatch(text) if target == 'v' else None
        if kick:
            v1 = self._iso_rescale_at(steps, pc + 1, env)
            parts = self._split_leading_group(kick.group(1))
            terms = self._signed_terms(parts[1]) if parts else None
            if not (v1 and terms and len(terms) <= 2 and terms[0][0] == 1):
                return 0
            coef = self._eval(parts[0], env)
            a = self._force_ref(terms[0][1], ops, valid)
            b, plus = -1, 0
            if len(terms) == 2:
                b = self._force_ref(terms[1][1], ops, valid)
                plus = 1 if terms[1][0] == 1 else 0
            ops.append(B.Op(B.OP_KICK, a, b, plus, coef))
            self._iso_found = (float(env['LkT']), float(env['Q1']), v1)
            return 4
        scale = self._SIN_SCALE.fullmatch(text)
        if scale and scale.group(1) == target and pc + 8 < len(steps):
            v1, v2 = target, scale.group(3)
            if self._iso_rescale_at(steps, pc + 1, env) != v1 or self._iso_rescale_at(steps, pc + 6, env) != v1:
                return 0
            (k4, t4, e4), (k5, t5, e5) = steps[pc + 4], steps[pc + 5]
            upd = self._SIN_UPDATE.fullmatch(e4.replace(' ', '')) if k4 == C.ComputePerDof else None
            if not (upd and t4 == v2 and upd.group(1) == v2 and upd.group(2) == v1 and k5 == C.ComputePerDof and t5 == v1 and e5 == expr and
                    v2 in self.integrator._pnames and upd.group(3) in env and 'kT' in env and 'friction' in env):
                return 0
            h = self._eval(scale.group(2), env) * env['dt']
            z = math.exp(-(self._eval(upd.group(4), env) * env['dt']) * env['friction'])
            key = ('sin', h, z, float(env['kT']), float(env[upd.group(3)]), float(env['friction']), v2)
            if key not in self._expr_ids:
                self._expr_ids[key] = self.ctx.bath_define_sin(h, z, float(env['kT']), float(env[upd.group(3)]), float(env['friction']),
                                                               self._slot(v2))
            ops.append(B.Op(B.OP_BATH, self._expr_ids[key], B.SLOT_V, 0, 0.0))
            self._iso_found = (float(env['LkT']), float(env['Q1']), v1)
            self._slot(v1)
            return 9
        return 0

    def _pair_up_evals(self, ops):
        """RESPA evaluates the near force (group 1) and, one kick later, the outer force (group 2) at the same positions
        (propagators.py:940-973); the two traverse one neighbour list (_share_lists).  Move the second EVAL (and its
        all-reduce marker) right behind the first, so that the backend evaluates both in one pass: legal when nothing in
        between moves the atoms or touches the second group's buffer."""
        fusable = set()
        for gid, host in self.shared.items():
            ga = [e.group for e in self.entries if gid in e.pair_ids]
            gb = [e.group for e in self.entries if host in e.pair_ids]
            if ga and gb:
                fusable.add(frozenset((ga[0], gb[0])))

        def is_eval(op):
            return not isinstance(op, tuple) and op.op == B.OP_EVAL

        def touches(op, slot):
            if isinstance(op, tuple):
                return op[1] == slot
            if op.op == B.OP_KICK:
                return slot in (op.a, op.b)
            if op.op == B.OP_COPY:
                return slot in (op.a, op.b)
            if op.op == B.OP_COMBINE:
                return slot in (op.a, op.b, op.c)
            return False

        slot_of = {index: slot for (index, slot, _) in self._group_defs.values()}
        out = list(ops)
        k = 0
        while k < len(out):
            if is_eval(out[k]):
                first = k
                end = first + 1
                while end < len(out) and isinstance(out[end], tuple):      # the first EVAL's all-reduce marker
                    end += 1
                j = end
                while j < len(out):
                    op = out[j]
                    if is_eval(op):
                        if frozenset((out[first].a, op.a)) in fusable:
                            slot2 = slot_of.get(op.a)
                            if not any(touches(mid, slot2) for mid in out[end:j]):
                                block = [op]
                                nxt = j + 1
                                while nxt < len(out) and isinstance(out[nxt], tuple):
                                    block.append(out[nxt])
                                    nxt += 1
                                del out[j:nxt]
                                evals = [out[first], block[0]]
                                markers = out[first + 1:end] + block[1:]
                                out[first:end] = evals + markers
                        break
                    if isinstance(op, tuple) or op.op in (B.OP_KICK,) or (op.op in (B.OP_COPY, B.OP_COMBINE) and op.a != B.SLOT_X):
                        j += 1
                        continue
                    break                       # MOVE / writes to x: positions change
            k += 1
        return out

    def _condition(self, expr, env):
        parts = _CONDITION_CACHE.get(expr)
        if parts is None:
            m = re.match(r'^(.*?)(<=|>=|!=|=|<|>)(.*)$', expr)
            if not m:
                raise NotImplementedError('unsupported block condition: ' + expr)
            parts = _CONDITION_CACHE[expr] = (m.group(1), _COMPARE[m.group(2)], m.group(3))
        return parts[1](self._eval(parts[0], env), self._eval(parts[2], env))

    @staticmethod
    def _split_leading_group(text):
        """'(A)*rest' -> ('A', 'rest') with balanced parentheses."""
        if not text.startswith('('):
            return None
        depth = 0
        for k, ch in enumerate(text):
            depth += ch == '('
            depth -= ch == ')'
            if depth == 0:
                if text[k + 1:k + 2] != '*':
                    return None
                return text[1:k], text[k + 2:]
        return None

    def _force_ref(self, name, ops, valid):
        """Slot of a force symbol / per-DOF buffer; emits the EVAL (and all-reduce marker) when stale."""
        m = re.fullmatch(r'f([0-9]*)', name)
        if m:
            g = 'all' if m.group(1) == '' else int(m.group(1))
            index, slot, reduced = self._define_group(g)
            if not valid.get(g, False):
                ops.append(B.Op(B.OP_EVAL, index, 0, 0, 0.0))
                if reduced:
                    ops.append((ALLREDUCE, slot))
                valid[g] = True
                for dst in [d for d, src in self._mirror_work.items() if src == name]:
                    del self._mirror_work[dst]          # copies of the old contents are no longer copies
            return slot
        integ = self.integrator
        if name not in integ._pnames and name not in ('x', 'v'):
            raise NotImplementedError('unknown per-DOF symbol in step program: ' + name)
        # `_f2_` while it mirrors f2 (`_f2_ <- f2`, integrators.py:139-144, and no evaluation of group 2 since): read the
        # group's buffer itself, so that the copy has no reader left and can be dropped (_drop_dead_copies)
        src = self._mirror_work.get(name)
        if src is not None and _AUX_FORCE.fullmatch(name) and not _NO_ALIAS:
            return self._define_group('all' if src == 'f' else int(src[1:]))[1]
        return self._slot(name)

    def _drop_dead_copies(self, ops):
        """Remove COPY ops into the integrator's auxiliary force buffers (`_f2_ <- f2`) that no later op reads before the
        buffer is written again -- the program is cyclic, so the search wraps around.  get_per_dof materialises such a
        buffer from the force it mirrors when the user asks for it."""
        if _NO_ALIAS:
            return ops
        aux = {slot for name, slot in self._slots.items() if _AUX_FORCE.fullmatch(name)}
        if not aux:
            return ops

        def reads(op, slot):
            if isinstance(op, tuple):
                return op[1] == slot
            if op.op == B.OP_KICK:
                return slot in (op.a, op.b)
            if op.op == B.OP_COPY:
                return op.b == slot
            if op.op == B.OP_COMBINE:
                return slot in (op.b, op.c)
            return op.op in (B.OP_EXPR,)            # expression programs may read any buffer

        def writes(op, slot):
            return not isinstance(op, tuple) and op.op in (B.OP_COPY, B.OP_COMBINE) and op.a == slot

        keep = [True] * len(ops)
        for k, op in enumerate(ops):
            if isinstance(op, tuple) or op.op != B.OP_COPY or op.a not in aux:
                continue
            dead = True
            for j in list(range(k + 1, len(ops))) + list(range(0, k)):
                if reads(ops[j], op.a):
                    dead = False
                    break
                if writes(ops[j], op.a):
                    break
            keep[k] = not dead
        return [op for op, kept in zip(ops, keep) if kept]

    def _run_end(self, steps, pc):
        """First step at or after pc that is not a ComputePerDof (cached per program position)."""
        ends = self._segment_ends
        if pc not in ends:
            C = mm.CustomIntegrator
            q = pc
            while q < len(steps) and steps[q][0] == C.ComputePerDof:
                q += 1
            for k in range(pc, q):
                ends[k] = q
        return ends[pc]

    def _segment_key(self, pc, pc_end, steps, env, valid):
        names = self._segment_symbols.get((pc, pc_end))
        if names is None:
            found = set()
            for k in range(pc, pc_end):
                found |= set(X.symbols(steps[k][2]))
            names = self._segment_symbols[(pc, pc_end)] = tuple(sorted(found))
        return (pc, pc_end, tuple(env.get(name) for name in names), tuple(sorted(valid.items(), key=str)),
                tuple(sorted(self._mirror_work.items())), getattr(self, '_static_exprs', False))

    def _replay_segment(self, pc, pc_end, steps, env, ops, valid):
        """Emit the remembered ops of the run of per-DOF steps [pc, pc_end); False when this combination has not been seen (the
        walker then goes through it step by step and _record_segment remembers it) or cannot be keyed (a deferred global)."""
        self._segment_open = None
        try:
            key = self._segment_key(pc, pc_end, steps, env, valid)
            hit = self._segment_memo.get(key)
        except TypeError:
            return False
        if hit is None:
            if len(self._segment_memo) < 512:
                self._segment_open = [key, pc_end, len(ops), True]      # key, end, ops emitted so far, every step emitted natively
            return False
        new_ops, valid_after, mirror_after, moved, kicked = hit
        ops.extend(new_ops)
        valid.clear()
        valid.update(valid_after)
        self._mirror_work.clear()
        self._mirror_work.update(mirror_after)
        if moved:
            self._deriv_cache.clear()
        if kicked:
            self._plain_kick = True
        return True

    def _record_segment(self, pc, emitted, ops, valid):
        rec = getattr(self, '_segment_open', None)
        if rec is None:
            return
        if not emitted or len(ops) < rec[2]:        # a general expression (run outside the op list) or a flush in between: not a unit
            self._segment_open = None
            return
        if pc + 1 == rec[1]:
            key, _, before, _ = rec
            new_ops = tuple(ops[before:])
            moved = any(op.op == B.OP_MOVE or (op.op in (B.OP_COPY, B.OP_COMBINE) and op.a == B.SLOT_X) or
                        (op.op == B.OP_EXPR and op.b == B.SLOT_X) for op in new_ops if not isinstance(op, tuple))
            self._segment_memo[key] = (new_ops, dict(valid), dict(self._mirror_work), moved, self._plain_kick)
            self._segment_open = None

    def _emit_per_dof_memo(self, pc, target, expr, env, ops, valid):
        """_emit_per_dof for the host-walked path, remembered: what a per-DOF step of the program emits is a function of the
        values of the globals its text names, of which force groups are valid and of which auxiliary buffers mirror a force --
        a RESPA block inside an AFED step meets the same few dozen combinations every step, and re-deriving them (regular
        expressions, eval) was what made the host slower than the GPU at config C5."""
        names = _PER_DOF_SYMBOLS.get(expr)
        if names is None:
            names = _PER_DOF_SYMBOLS[expr] = tuple(sorted(X.symbols(expr)))
        try:
            key = (pc, tuple(env.get(name) for name in names), tuple(sorted(valid.items(), key=str)),
                   tuple(sorted(self._mirror_work.items())), getattr(self, '_static_exprs', False))
            hit = self._emit_memo.get(key)
        except TypeError:                # a deferred global (unhashable) among the values: no memo for this one
            return self._emit_per_dof(target, expr, env, ops, valid)
        if hit is None:
            before = len(ops)
            self._emit_per_dof(target, expr, env, ops, valid)
            if len(self._emit_memo) < 4096:
                self._emit_memo[key] = (tuple(ops[before:]), dict(valid), dict(self._mirror_work), target == 'x', self._plain_kick)
            return
        new_ops, valid_after, mirror_after, moved, kicked = hit
        ops.extend(new_ops)
        valid.clear()
        valid.update(valid_after)
        self._mirror_work.clear()
        self._mirror_work.update(mirror_after)
        if moved:
            self._deriv_cache.clear()
        if kicked:
            self._plain_kick = True

    def _emit_per_dof(self, target, expr, env, ops, valid):
        text = expr.replace(' ', '')
        if ';' in text:
            text = ''         # auxiliary definitions: never a plain kick / move / copy -> general expression below
        # kick: v <- v + (coef)*FORCE/m
        if target == 'v' and text.startswith('v+') and text.endswith('/m'):
            parts = self._split_leading_group(text[2:-2])
            if parts:
                coef = self._eval(parts[0], env)
                terms = self._signed_terms(parts[1])
                if terms and len(terms) <= 2 and terms[0][0] == 1:
                    a = self._force_ref(terms[0][1], ops, valid)
                    b, plus = -1, 0
                    if len(terms) == 2:
                        b = self._force_ref(terms[1][1], ops, valid)
                        plus = 1 if terms[1][0] == 1 else 0
                    ops.append(B.Op(B.OP_KICK, a, b, plus, coef))
                    self._plain_kick = True
                    return
        # the same kick written as a bare product, `v + 0.5*1.0*dt*f/m` (UnconstrainedVelocityVerletPropagator, propagators.py:1136-1153):
        # OpenMM's parser associates a*b*c*f/m as (((a*b)*c)*f)/m, i.e. (coef * f) / m with coef = the product of the factors in
        # front of the force symbol -- what AMM_OP_KICK computes -- provided the force is the LAST factor
        if target == 'v' and text.startswith('v+') and text.endswith('/m'):
            factors = self._split_product(text[2:-2])
            if factors and len(factors) >= 2 and self._is_global_product(factors[:-1], env):
                terms = self._signed_terms(factors[-1])
                if terms and len(terms) <= 2 and terms[0][0] == 1 and all(self._is_force_symbol(t[1]) for t in terms):
                    coef = self._eval('*'.join(factors[:-1]), env)
                    a = self._force_ref(terms[0][1], ops, valid)
                    b, plus = -1, 0
                    if len(terms) == 2:
                        b = self._force_ref(terms[1][1], ops, valid)
                        plus = 1 if terms[1][0] == 1 else 0
                    ops.append(B.Op(B.OP_KICK, a, b, plus, coef))
                    self._plain_kick = True
                    return
        # move: x <- x + (coef)*v
        if target == 'x' and text.startswith('x+') and text.endswith('*v'):
            parts = self._split_leading_group(text[2:])
            factors = None if parts else self._split_product(text[2:])
            coef_text = parts[0] if parts and parts[1] == 'v' else None
            if coef_text is None and factors and len(factors) >= 2 and factors[-1] == 'v' and self._is_global_product(factors[:-1], env):
                coef_text = '*'.join(factors[:-1])          # `x + 1.0*dt*v`: ((1.0*dt)*v), the move's arithmetic
            if coef_text is not None:
                ops.append(B.Op(B.OP_MOVE, 0, 0, 0, self._eval(coef_text, env)))
                for g in valid:
                    valid[g] = False
                self._deriv_cache.clear()
                return
        # copies / differences of buffers: `_f2_ <- f2`, `fm1 <- f1`, `fm2 <- f2-f1`, `x0 <- x`
        if target not in ('x', 'v'):
            terms = self._signed_terms(text)
            if terms and terms[0][0] == 1 and len(terms) <= 2:
                src = self._force_ref(terms[0][1], ops, valid)
                dst = self._slot(target)
                for d in [d for d, sname in self._mirror_work.items() if sname == target]:
                    del self._mirror_work[d]
                if len(terms) == 1:
                    # `_f2_ <- f2` opens AND closes the RESPA program (propagators.py:940-973): the copy at the top
                    # of the next step finds _f2_ still equal to the unchanged f2 and is dropped
                    if self._mirror_work.get(target) == terms[0][1]:
                        return
                    ops.append(B.Op(B.OP_COPY, dst, src, 0, 0.0))
                    if re.fullmatch(r'f[0-9]*', terms[0][1]):
                        self._mirror_work[target] = terms[0][1]
                    else:
                        self._mirror_work.pop(target, None)
                else:
                    second = self._force_ref(terms[1][1], ops, valid)
                    ops.append(B.Op(B.OP_COMBINE, dst, src, second, float(terms[1][0])))
                    self._mirror_work.pop(target, None)
                return
        # any other per-DOF expression whose globals are known now (a bath step inside a RESPA loop, ...): registered
        # once with the backend and replayed by amm_run_ops as an EXPR op
        if getattr(self, '_static_exprs', False):
            integ = self.integrator

            def resolve(name):
                if name == 'm':
                    return ('mass',)
                if name in ('x', 'v') or re.fullmatch(r'f[0-9]*', name) or name in integ._pnames:
                    return ('buf', self._force_ref(name, ops, valid))
                if name in env and not callable(env[name]):
                    return ('global',)
                return None
            if target not in ('x', 'v') and target not in integ._pnames:
                raise mm.OpenMMException('unknown per-DOF variable: ' + target)
            # the Ornstein-Uhlenbeck bath on (v, m) without force, as OrnsteinUhlenbeckPropagator writes it
            # (propagators.py:727-733): a native op, which the inner-loop kernel can carry (Langevin_R)
            ou = re.fullmatch(r'z\*v\+sqrt\(kT\*\(1-z\*z\)/mass\)\*gaussian;mass=m;z=exp\(-\((.+)\*dt\)\*friction\)',
                              expr.replace(' ', ''))
            if ou and target == 'v' and 'kT' in env and 'friction' in env:
                z = math.exp(-(self._eval(ou.group(1), env) * env['dt']) * env['friction'])
                key = ('ou', z, float(env['kT']))
                if key not in self._expr_ids:
                    self._expr_ids[key] = self.ctx.bath_define(z, float(env['kT']))
                ops.append(B.Op(B.OP_BATH, self._expr_ids[key], B.SLOT_V, 0, 0.0))
                return
            prog = X.compile_per_dof(expr, resolve)
            gvals = tuple(float(env[name]) for name in prog.globals_)
            key = (expr, gvals)
            if key not in self._expr_ids:
                self._expr_ids[key] = self.ctx.expr_define(prog.code, prog.consts, list(gvals))
            ops.append(B.Op(B.OP_EXPR, self._expr_ids[key], self._slot(target), 0, 0.0))
            self._mirror_work.pop(target, None)
            for d in [d for d, sname in self._mirror_work.items() if sname == target]:
                del self._mirror_work[d]
            if target == 'x':
                for g in valid:
                    valid[g] = False
                self._deriv_cache.clear()
            return
        raise NotImplementedError('per-DOF computation outside the RESPA hot path: {} <- {}'.format(target, expr))

    @staticmethod
    def _split_product(text):
        """'0.5*1.0*dt*(f1-f2)' -> ['0.5', '1.0', 'dt', '(f1-f2)']; None when the text is not a plain product at its top level."""
        parts, depth, start = [], 0, 0
        for k, ch in enumerate(text):
            if ch == '(':
                depth += 1
            elif ch == ')':
                depth -= 1
                if depth < 0:
                    return None
            elif depth == 0:
                if ch == '*':
                    parts.append(text[start:k])
                    start = k + 1
                elif ch in '+-/^;' and k > 0 and text[k - 1] not in 'eE':      # (a sign inside 1e-3 is part of the number)
                    return None
        parts.append(text[start:])
        return parts if depth == 0 and all(parts) else None

    def _is_force_symbol(self, name):
        return bool(re.fullmatch(r'f[0-9]*', name)) or name in self.integrator._pnames

    def _is_global_product(self, factors, env):
        """Every factor a number or a global / Context parameter known now (no per-DOF symbol, no function call)."""
        per_dof = set(self.integrator._pnames) | {'x', 'v', 'm', 'f'}
        for f in factors:
            if re.fullmatch(r'[0-9.]+([eE][+-]?[0-9]+)?', f):
                continue
            if re.fullmatch(r'[A-Za-z_][A-Za-z_0-9]*', f) and f not in per_dof and not re.fullmatch(r'f[0-9]+', f) \
                    and f in env and not callable(env[f]):
                continue
            return False
        return True

    @staticmethod
    def _signed_terms(text):
        """'(f0+fm1)' -> [(1,'f0'), (1,'fm1')]; None if not a signed sum of identifiers."""
        while text.startswith('(') and text.endswith(')'):
            text = text[1:-1]
        if not re.fullmatch(r'[A-Za-z_][A-Za-z_0-9]*([+-][A-Za-z_][A-Za-z_0-9]*)*', text):
            return None
        return [(-1 if s == '-' else 1, name) for s, name in re.findall(r'([+-]?)([A-Za-z_][A-Za-z_0-9]*)', text)]

    def _init_native_comm(self, dist):
        """Give the library its own RCCL communicator (csrc/comm.hip): the all-reduce after the EVAL of a sliced group
        becomes an op of the step program and whole runs of steps stay inside ONE amm_run_ops call, instead of ~10 host
        round trips per outer step (measured 480 us/step of host time).  torch.distributed only carries the 128-byte id."""
        box = [None]
        if self.rank == 0:
            try:
                box[0] = self.ctx.comm_unique_id()
            except Exception as exc:        # e.g. librccl cannot be bound: every rank then keeps torch.distributed
                box[0] = 'failed: %s' % exc
        dist.broadcast_object_list(box, src=0)
        ok = isinstance(box[0], (bytes, bytearray))
        if ok:
            try:
                self.ctx.comm_init(box[0])
            except Exception as exc:
                ok = False
                box[0] = 'failed: %s' % exc
        # all ranks or none: a rank without the communicator would leave the others waiting in the first collective
        flag = self.torch.tensor([1 if ok else 0], device=self.ctx.torch_device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            self._native_comm = True
        else:
            import warnings
            warnings.warn('library-owned RCCL communicator not available (%s): collectives stay in torch.distributed'
                          % (box[0] if not ok else 'another rank failed'))
            if ok:
                self.ctx.comm_destroy()

    def _run(self, ops, repeat, cache=True):
        """cache=False: `ops` is a throw-away list (the interpreted path flushes a fresh one several times per step); its
        plan is not kept -- the cache is keyed by the list's identity and would otherwise grow without bound."""
        if not self._coll:
            self.ctx.run_ops([op for op in ops if not isinstance(op, tuple)], repeat)
            return
        if self._native_comm:
            native = self._native_ops.get(id(ops)) if cache else None
            if native is None or native[0] is not ops:
                native = (ops, [B.Op(B.OP_ALLREDUCE, op[1], 0, 0, 0.0) if isinstance(op, tuple) else op for op in ops])
                if cache:
                    self._native_ops[id(ops)] = native
            self.ctx.run_ops(native[1], repeat)
            return
        # collectives driven from here (torch.distributed): the op list is cut after every all-reduce marker and after
        # every EVAL of an all-gather group (the library leaves the rank's chunk in the exchange buffer and waits)
        plan = self._native_ops.get(id(ops)) if cache else None
        if plan is None or plan[0] is not ops:
            segments, current = [], []
            for op in ops:
                if isinstance(op, tuple):
                    segments.append((current, ('reduce', op[1])))
                    current = []
                else:
                    current.append(op)
            plan = (ops, segments, current)
            if cache:
                self._native_ops[id(ops)] = plan
        _, segments, tail = plan
        inv = {slot: name for name, slot in self._slots.items()}
        # the all-gather exchanges INSIDE a run of ops are the library's to ask for: it runs up to an exchanged evaluation, hands back,
        # the chunks are gathered here, and it goes on where it stopped (amm_run_ops_from) -- so that the launches which integrate
        # their rows' molecules and exchange positions and velocities (state exchange) see the ops that follow their EVAL
        if not segments:            # no all-reduce markers (groups of one pair force each: the bench): whole repetitions in one go
            if tail:
                self.ctx.run_ops_host_exchanges(tail, repeat, self._host_gather)
            return
        for _ in range(repeat):
            for seg, (kind, slot) in segments:
                if seg:
                    self.ctx.run_ops_host_exchanges(seg, 1, self._host_gather)
                self._allreduce(self._buffers[inv[slot]])
            if tail:
                self.ctx.run_ops_host_exchanges(tail, 1, self._host_gather)

    def _host_gather(self, nf):
        """All-gather of the exchange chunks by torch.distributed (no library-owned communicator), then the unsort."""
        dist = self.torch.distributed
        count = nf * self._per * 3
        if self._local is not None:
            self._local.all_gather(self.rank, self._xchg, count)
            self.ctx.exchange_finish()
            return
        chunks = [self._xchg[r * count:(r + 1) * count] for r in range(self.world)]
        if dist.get_backend() == 'nccl':
            dist.all_gather_into_tensor(self._xchg[:self.world * count], chunks[self.rank].clone())
        else:                     # gloo (CPU tests / several ranks sharing one card): staged through the host
            mine = chunks[self.rank].cpu()
            parts = [self.torch.empty_like(mine) for _ in range(self.world)]
            dist.all_gather(parts, mine)
            for r in range(self.world):
                if r != self.rank:
                    chunks[r].copy_(parts[r])
        self.ctx.exchange_finish()

    @staticmethod
    def _program_key(valid, mirror):
        return (tuple(sorted((str(g), ok) for g, ok in valid.items())), tuple(sorted(mirror.items())))

    def _deriv(self, what, name):
        if what != 'energy':
            raise NotImplementedError('deriv(%s, ...): only deriv(energy, parameter) is supported' % what)
        return self.energy_derivative(name)

    def _energy_of(self, entries):
        """Potential energy of a subset of the translated forces (pair + bond-list + reciprocal-space terms + constants)."""
        torch = self.torch
        e_pair = torch.zeros(1, dtype=torch.float64, device=self.x.device)
        e_bond = torch.zeros(1, dtype=torch.float64, device=self.x.device)
        fp, fb = self._fwork.zero_(), self._fwork2.zero_()
        const = 0.0
        for entry in entries:
            for pid in entry.pair_ids:
                self.ctx.force_eval(pid, self.x, fp, accumulate=True, energy=e_pair)
            if entry.bonded_id is not None:
                self.ctx.force_eval(entry.bonded_id, self.x, fb, accumulate=True, energy=e_bond)
            if entry.recip is not None:
                self.ctx.pme_set_sliced(entry.recip, False)
                self.ctx.force_eval(entry.recip, self.x, fb, accumulate=True, energy=e_bond)
                self.ctx.pme_set_sliced(entry.recip, getattr(entry, 'recip_sliced', False))
            const += entry.constant
        self._check()
        if self._coll:
            self._allreduce(e_pair)
        return e_pair.item() + e_bond.item() + const

    def energy_derivative(self, name):
        """deriv(energy, name): d(total potential energy)/d(global parameter) at the current positions
        (ExtendedSystemVariable.update_velocity, integrators.py:735-737).

        * lambda of a softcore pair force: the pair kernel in derivative mode + the long-range correction's derivative;
        * an overall coupling factor (AlchemicalSystem's `linear` / `spline` / `art` / custom couplings, systems.py:349-365):
          the factor's derivative times the energy at unit coupling;
        * parameter offsets of a NonbondedForce (SolvationSystem's `lambda_coul`, systems.py:289-308): the energy is a
          quadratic form of the charges, so the central difference over a whole unit of lambda is exact; offsets that act
          on sigma / epsilon (`use_softcore=False`, systems.py:309-312) take a small central step instead (O(h^2))."""
        if name not in self.parameters:
            raise mm.OpenMMException('deriv(energy, %s): no such Context parameter' % name)
        torch = self.torch
        total = 0.0
        by_difference = []
        for entry in self.entries:
            sc = entry.softcore
            if sc is not None and sc['lambda_name'] == name:
                out = torch.zeros(1, dtype=torch.float64, device=self.x.device)
                self.ctx.pair_energy_derivative(sc['pid'], self.x, out)
                if self._coll:
                    self._allreduce(out)
                total += out.item() + sc['constant_derivative'](self.parameters)
            elif name in getattr(entry, 'depends', ()):
                if entry.update is None:
                    raise NotImplementedError('deriv(energy, %s): the force that depends on it cannot be re-parameterised' % name)
                by_difference.append(entry)
        if by_difference:
            exact = all(getattr(e, 'quadratic_in', None) and name in e.quadratic_in for e in by_difference)
            here = self.parameters[name]
            h = 1.0 if exact else 1e-4
            lo = here - h
            if not exact and lo < 0.0 <= here:
                lo = here                      # sqrt(epsilon) offsets are not analytic below zero: one-sided step
            values, rebuilt = [], False
            for point in (here + h, lo, here):
                trial = dict(self.parameters)
                trial[name] = point
                for entry in by_difference:
                    result = entry.update(trial, {name})
                    rebuilt = rebuilt or (result and result != 'values')
                if point != here:
                    values.append(self._energy_of(by_difference))
            if rebuilt:
                self._forget_groups()
                self._programs.clear()
                self._emit_memo.clear()
                self._segment_memo.clear()
            self._invalidate_forces()
            total += (values[0] - values[1]) / (here + h - lo)
        return total

    def _deriv_deferred(self, what, name, settle):
        """deriv(energy, name) of a host-walked program without waiting for the GPU: the softcore pair kernel in derivative
        mode leaves its sum in the next free slot of the pending buffer and the value is handed on as a deferred global
        (expr.Deferred); the same positions and parameters give the same object again (a RESPA block ends and the next one
        begins with the same derivative).  Forces that need difference quotients take the waiting path."""
        if what != 'energy':
            raise NotImplementedError('deriv(%s, ...): only deriv(energy, parameter) is supported' % what)
        if name in self._deriv_cache:
            return self._deriv_cache[name]
        if name not in self.parameters:
            raise mm.OpenMMException('deriv(energy, %s): no such Context parameter' % name)
        if any(e.softcore is None and name in getattr(e, 'depends', ()) for e in self.entries):
            settle()
            value = self.energy_derivative(name)
        else:
            if self._pending is None:
                self._pending = [self.torch.zeros(_SCALARS, dtype=self.torch.float64, device=self.x.device), 0]
            value = 0.0
            for entry in self.entries:
                sc = entry.softcore
                if sc is not None and sc['lambda_name'] == name:
                    if self._pending[1] >= _SCALARS:
                        settle()
                    slot = self._pending[1]
                    self._pending[1] += 1
                    lam = self.parameters[sc['lambda_name']]
                    # (lambda is a device scalar: so is the correction's derivative at it -- queued first: one launch with the block that moved lambda)
                    correction = self._lrc_derivative_on_device(sc, lam, settle) if isinstance(lam, X.Deferred) else None
                    self._flush_scalars()                 # (lambda itself may be an assignment still queued)
                    self.ctx.pair_energy_derivative(sc['pid'], self.x, self._pending[0][slot:slot + 1])
                    if isinstance(lam, X.Deferred):
                        if correction is not None:
                            value = value + correction
                        value = value + X.Deferred(0.0, {slot: 1.0})
                    else:
                        value = value + X.Deferred(sc['constant_derivative'](self.parameters), {slot: 1.0})
        self._deriv_cache[name] = value
        return value

    def _device_scalars_ok(self):
        """Deferred globals may be worked on where they are: single rank (the ranks' partial sums are added up at a settle), a
        backend with the scalar kernel."""
        return self.device_globals and not self._coll and hasattr(self.ctx, 'expr_eval_scalar')

    def _scalar_slot(self, settle):
        if self._pending is None:
            self._pending = [self.torch.zeros(_SCALARS, dtype=self.torch.float64, device=self.x.device), 0]
        if self._pending[1] >= _SCALARS:
            settle()
        slot = self._pending[1]
        self._pending[1] += 1
        return slot

    def _eval_global_on_device(self, expr, env, rng, settle, predicate=None, keep=None):
        """A global expression whose operands wait on device results, evaluated by one thread on the stream (amm_expr_eval_scalar):
        its value is a new device scalar, the host goes on without it (AFED: integrators.py:701-737, the extended variable's move,
        walls and thermostat).  Returns the Deferred that names the scalar."""
        return self._queue_scalar(X.compile_scalar(expr, env, rng, predicate, keep), settle)

    def _queue_scalar(self, prog, settle):
        self.n_scalar_evals += 1
        slot = self._scalar_slot(settle)
        # queued: consecutive assignments go out as ONE launch (_flush_scalars: before anything that reads a scalar is enqueued)
        if len(self._scalar_code) + len(prog.code) + 1 > 640 or len(self._scalar_consts) + len(prog.consts) > 96:
            self._flush_scalars()
        base = len(self._scalar_consts)
        const_ops = (X.OPCODES['CONST'], X.OPCODES['HORNER'])      # (words whose argument is a constant's index)
        self._scalar_code += [w + (base << 8) if (w & 0xff) in const_ops else w for w in prog.code]
        self._scalar_code.append(X.OPCODES['OUT'] | (slot << 8))
        self._scalar_consts += prog.consts
        return X.Deferred(0.0, {slot: 1.0})

    def _flush_scalars(self):
        if self._scalar_code:
            self.ctx.expr_eval_scalar(self._scalar_code, self._scalar_consts, self._pending[0])
            self.n_scalar_launches += 1
            self._scalar_code, self._scalar_consts = [], []

    def _lrc_derivative_on_device(self, sc, lam, settle):
        """d(long-range correction)/d(lambda) of a softcore force at a lambda that is a device scalar: the interpolant's monomial form
        in u = 2 lambda - 1, Horner, as one more assignment (queued with the block that moved lambda: no launch of its own)."""
        coef = sc['derivative_polynomial'](self.parameters)
        if not coef:
            return None
        key = (sc['pid'], tuple(sorted(lam.terms.items())), lam.const)
        if key not in self._lrc_on_device:
            self._lrc_on_device[key] = self._queue_scalar(X.compile_polynomial(coef, 2.0, -1.0, lam), settle)
        return self._lrc_on_device[key]

    def _parameter_to_device(self, name, value, settle):
        """context.setParameter(name, <a device scalar>): possible when every force that depends on the parameter is a softcore
        pair force with this lambda and the list-free kernel (its long-range correction's derivative follows as a polynomial) -- the
        solute-solvent force of SolvationSystem under AFED.  Returns False when the number is needed after all."""
        pids = []
        for entry in self.entries:
            sc = entry.softcore
            if sc is not None and name in sc['depends']:
                if sc['lambda_name'] != name:
                    return False
                pids.append(sc['pid'])
            elif name in getattr(entry, 'depends', ()):
                return False
        if not pids:
            return False
        if len(value.terms) != 1 or value.const != 0.0 or list(value.terms.values()) != [1.0]:
            value = self._eval_global_on_device('__v', {'__v': value}, None, settle)         # (a linear form: one scalar of its own)
        slot = next(iter(value.terms))
        try:
            for pid in pids:
                self.ctx.pair_set_lambda_dev(pid, self._pending[0], slot)
        except B.HipError:
            for pid in pids:
                self.ctx.pair_set_lambda_dev(pid, None, 0)
            return False
        self.parameters[name] = value
        self._device_params[name] = pids
        groups = {e.group for e in self.entries if e.softcore is not None and e.softcore['pid'] in pids}
        for g in self._valid:
            if g in groups or g == 'all':
                self._valid[g] = False
        self._deriv_cache.clear()
        self._programs.clear()
        return True

    def _settle(self, containers):
        """Read the pending device scalars (one synchronising copy; summed over the ranks first) and turn every deferred
        global in `containers` (dicts / lists) into its number."""
        if self._pending is None or self._pending[1] == 0:
            return
        self._flush_scalars()
        self._lrc_on_device = {}
        buf, used = self._pending
        if self._coll:
            self._allreduce(buf)
        values = buf[:used].cpu().numpy()
        self.n_settles += 1
        for box in containers + [self._deriv_cache]:
            keys = range(len(box)) if isinstance(box, list) else list(box)
            for k in keys:
                if isinstance(box[k], X.Deferred):
                    box[k] = box[k].resolve(values)
        # Context parameters that lived on the device (an extended variable between two reads): the host has their numbers again, and
        # the library its launch argument -- the same value its kernels read from the scalar until now: no force becomes stale
        for name, pids in self._device_params.items():
            value = self.parameters[name].resolve(values)
            self.parameters[name] = value
            for pid in pids:
                self.ctx.pair_set_lambda(pid, value)
            for entry in self.entries:
                if entry.softcore is not None and entry.softcore['pid'] in pids:
                    entry.constant = entry.softcore['constant'](self.parameters)
        self._device_params = {}
        buf.zero_()
        self._pending[1] = 0

    # ------------------------------------------------------------------------------- general step programs
    def _step_interpreted(self, n):
        """Programs with data-dependent globals (ComputeSum results, random numbers) or per-DOF expressions beyond kick /
        move / copy -- the thermostat propagators (propagators.py:276-827, 1108-2172): the host walks the program step by
        step; runs of kick / move / copy / EVAL ops still go to amm_run_ops, every other per-DOF or sum expression is
        compiled (atomsmm_amd/expr.py) and interpreted on the GPU (amm_expr_eval); global expressions are evaluated on
        the host.  A ComputeSum costs one device -> host read."""
        torch = self.torch
        integ = self.integrator
        steps = integ._steps
        C = mm.CustomIntegrator
        match, stack = {}, []
        for pc, (kind, _, _) in enumerate(steps):
            if kind in (C.IfBlock, C.WhileBlock):
                stack.append(pc)
            elif kind == C.EndBlock:
                begin = stack.pop()
                match[begin], match[pc] = pc, begin
        if self._host_rng is None:
            self._host_rng = np.random.default_rng(integ.getRandomNumberSeed())
        seed = int(integ.getRandomNumberSeed()) & (2 ** 64 - 1)
        total = torch.zeros(1, dtype=torch.float64, device=self.x.device)
        try:
            self._walk_program(int(n), steps, match, seed, total)
        finally:
            self._settle([integ._gvalues])         # no deferred global outlives the call
        self._check()

    def _walk_program(self, n, steps, match, seed, total):
        torch = self.torch
        integ = self.integrator
        C = mm.CustomIntegrator
        for _ in range(n):
            env = dict(_SAFE_FUNCS)
            env.update(self.parameters)
            env.update(zip(integ._gnames, integ._gvalues))
            env['dt'] = integ._dt

            # After a blocking read the GPU has nothing queued: the next ops go out in small batches (at the first moves, i.e. at
            # the end of a kick ... move pattern that amm_run_ops fuses) instead of waiting for the host to walk the whole block
            # -- at config C5 the GPU sat idle for 0.26 ms per AFED step behind the read of deriv(energy, lambda)
            eager = [0]

            preds = []          # if-blocks whose condition waits on the device: [end pc, predicate] (their steps are evaluated there, predicated)

            def settle():
                if self._pending is not None and self._pending[1] > 0 and not self._has_constraints:
                    eager[0] = 3          # (constraint ops stay in one batch with the move they follow)
                if self._device_params:
                    flush()               # (launches recorded so far read the parameter where it is now)
                held = [p[1] for p in preds]
                self._settle([env, integ._gvalues, held])
                for p, value in zip(preds, held):
                    p[1] = value

            def deferred_in(text):
                """Does the expression name a global whose number is still on the device?"""
                if self._pending is None or self._pending[1] == 0:
                    return False
                m = re.match(r'^(.*?)(<=|>=|!=|=|<|>)(.*)$', text) if re.search(r'[<>=]', text) else None      # (a block's condition)
                names = X.symbols(m.group(1)) | X.symbols(m.group(3)) if m else X.symbols(text)
                return any(isinstance(env.get(name), X.Deferred) for name in names)

            env['__deriv__'] = lambda what, name: self._deriv_deferred(what, name, settle)
            valid = self._valid
            self._mirror_work = self._mirror
            ops = [B.Op(B.OP_SAVE_REF, 0, 0, 0, 0.0)] if self._has_constraints else []

            def flush():
                # (a run of per-DOF steps that is being recorded as one unit is no unit any more: the ops before the flush are gone)
                self._segment_open = None
                if ops:
                    self._flush_scalars()     # (assignments to device scalars queued so far: the launches below may read them)
                    # a RESPA block between two host-evaluated steps is a static run of ops: pair the near and the outer
                    # evaluation of one list into a single traversal, as the compiled path does
                    self._run(self._pair_up_evals(list(ops)), 1, cache=False)
                    del ops[:]

            def resolve(name):
                if name == 'm':
                    return ('mass',)
                if name in ('x', 'v') or re.fullmatch(r'f[0-9]*', name) or name in integ._pnames:
                    return ('buf', self._force_ref(name, ops, valid))
                if name in env and not callable(env[name]):
                    return ('global',)
                return None

            pc = 0
            guard = 0
            while pc < len(steps):
                guard += 1
                if guard > 2_000_000:
                    raise RuntimeError('step program does not terminate')
                kind, target, expr = steps[pc]
                if self._pending is not None and self._pending[1] > _SCALARS - 64:
                    settle()            # (room for the scalars a step may ask for; what was deferred is a number from here on)
                if kind == C.ComputePerDof:
                    # a straight run of per-DOF steps (a RESPA block of the atoms inside an AFED step: ~60 kicks, moves and
                    # copies) emits the same ops whenever the globals it names, the valid force groups and the mirrored buffers
                    # are the same: remembered as ONE unit (the host was slower than the GPU at config C5 walking it step by step)
                    pc_end = self._run_end(steps, pc)
                    # (while THIS run is being recorded its later steps must not ask again: _replay_segment would drop the record
                    # and open one for the remainder -- only the last four steps of a run were ever remembered as a unit)
                    rec = self._segment_open
                    recording = rec is not None and rec[1] == pc_end and not eager[0]
                    if not recording and pc_end - pc >= 4 and not eager[0] and self._replay_segment(pc, pc_end, steps, env, ops, valid):
                        pc = pc_end
                        continue
                if kind == C.ComputeGlobal:
                    if 'deriv(' in expr:
                        flush()                                   # the derivative is taken at the current positions
                    rng_state = self._host_rng.bit_generator.state
                    pred = preds[-1][1] if preds else None
                    if isinstance(pred, X.Deferred):
                        # inside an if-block whose condition is still on the device: target <- select(condition, expression, target)
                        value = self._eval_global_on_device(expr, env, self._host_rng, settle, pred, env[target])
                    elif pred is not None and not pred:
                        pc += 1                                   # (the condition was read after all, and is false)
                        continue
                    else:
                        try:
                            value = X.eval_global(expr, env, self._host_rng)
                        except X.NeedsValue:                      # more than sums and multiples of a deferred global
                            self._host_rng.bit_generator.state = rng_state
                            value = None
                            if self._device_scalars_ok():
                                try:
                                    value = self._eval_global_on_device(expr, env, self._host_rng, settle)
                                except (X.NeedsValue, X.ExpressionError):
                                    self._host_rng.bit_generator.state = rng_state
                                    value = None
                            if value is None:
                                settle()
                                value = X.eval_global(expr, env, self._host_rng)
                    if target in self.parameters and target not in integ._gnames:
                        done = False
                        if isinstance(value, X.Deferred):         # a Context parameter: on the device if its forces can read it there
                            if self._device_scalars_ok():
                                flush()
                                done = self._parameter_to_device(target, value, settle)
                                if done:
                                    value = self.parameters[target]
                                    valid = self._valid
                            if not done:
                                env[target] = value
                                settle()
                                value = env[target]
                        if not done and (isinstance(self.parameters[target], X.Deferred) or value != self.parameters[target]):
                            flush()
                            if isinstance(self.parameters[target], X.Deferred):
                                settle()
                            self.set_parameter(target, value)     # an extended-system variable (AFED): forces change
                            valid = self._valid
                    env[target] = value
                elif kind in (C.ComputePerDof, C.ComputeSum):
                    if deferred_in(expr):
                        settle()
                    done = False
                    if kind == C.ComputePerDof:
                        try:
                            self._emit_per_dof_memo(pc, target, expr, env, ops, valid)
                            done = True
                        except (NotImplementedError, NameError, SyntaxError, TypeError):
                            done = False
                        self._record_segment(pc, done, ops, valid)
                        if done and eager[0] and target == 'x':
                            eager[0] -= 1
                            flush()
                    if not done:
                        prog = X.compile_per_dof(expr, resolve)
                        flush()                                   # EVALs emitted by resolve() run first
                        gvals = [float(env[name]) for name in prog.globals_]
                        self._expr_counter += 1
                        if kind == C.ComputePerDof:
                            if target not in ('x', 'v') and target not in integ._pnames:
                                raise mm.OpenMMException('unknown per-DOF variable: ' + target)
                            dst = self.x if target == 'x' else (self.v if target == 'v' else self._buffer(target))
                            self._slot(target)
                            self.ctx.expr_eval(prog.code, prog.consts, gvals, seed, self._expr_counter, dst=dst)
                            self._mirror_work.pop(target, None)
                            for d in [d for d, sname in self._mirror_work.items() if sname == target]:
                                del self._mirror_work[d]
                            if target == 'x':
                                for g in valid:
                                    valid[g] = False
                                self._deriv_cache.clear()
                        else:
                            self.ctx.expr_eval(prog.code, prog.consts, gvals, seed, self._expr_counter, total=total)
                            env[target] = total.item()          # device -> host (synchronises)
                elif kind in (C.ConstrainPositions, C.ConstrainVelocities):
                    self._emit_constraint(kind, ops, valid)
                elif kind == C.UpdateContextState:
                    pass
                elif kind in (C.IfBlock, C.WhileBlock):
                    if deferred_in(expr):
                        body = steps[pc + 1:match[pc]]
                        if (kind == C.IfBlock and self._device_scalars_ok() and body and all(b[0] == C.ComputeGlobal for b in body)
                                and not any('deriv(' in b[2] for b in body)):
                            # the condition stays on the device (1 or 0) and the block's steps are evaluated there, predicated --
                            # the reflecting walls of an extended variable (integrators.py:701-714)
                            lhs, op, rhs = re.match(r'^(.*?)(<=|>=|!=|=|<|>)(.*)$', expr).groups()
                            diff = '((%s)-(%s))' % (lhs, rhs)
                            text = {'=': 'delta(%s)', '!=': '1-delta(%s)', '>=': 'step(%s)', '<': '1-step(%s)',
                                    '<=': 'step(-%s)', '>': '1-step(-%s)'}[op] % diff
                            preds.append([match[pc], self._eval_global_on_device(text, env, None, settle)])
                            pc += 1
                            continue
                        settle()
                    if not self._condition(expr, env):
                        pc = match[pc]
                elif kind == C.EndBlock:
                    if preds and preds[-1][0] == pc:
                        preds.pop()
                    if steps[match[pc]][0] == C.WhileBlock:
                        pc = match[pc] - 1
                pc += 1
            flush()
            for k, name in enumerate(integ._gnames):
                integ._gvalues[k] = env[name]
            self.time += integ._dt

    def step(self, n):
        integ = self.integrator
        if not isinstance(integ, mm.CustomIntegrator):
            raise NotImplementedError('only CustomIntegrator step programs run on the HIP path')
        if self._seed_set != integ.getRandomNumberSeed():
            self._seed_set = integ.getRandomNumberSeed()
            self.ctx.expr_seed(self._seed_set)
        if self._interpreted is None:
            # static programs (RESPA, also with bath steps inside the loops) are unrolled once and replayed; programs
            # with data-dependent globals (ComputeSum results, random globals) go through the general path
            try:
                self._compile()
                self._interpreted = False
            except (NotImplementedError, NameError, SyntaxError, TypeError, KeyError, X.ExpressionError):
                self._static_exprs = False
                self._interpreted = True
        if self._interpreted:
            return self._step_interpreted(n)
        remaining = int(n)
        while remaining > 0:
            key = self._program_key(self._valid, self._mirror)
            if key not in self._programs:
                self._programs[key] = self._compile()
            ops, valid_after, finals, mirror_after = self._programs[key]
            after_key = self._program_key(valid_after, mirror_after)
            count = remaining if after_key == key else 1
            self._run(ops, count)
            self._valid = dict(valid_after)
            self._mirror = dict(mirror_after)
            for name, value in finals.items():
                integ._gvalues[integ._gnames.index(name)] = value
            remaining -= count
            self.time += count * integ._dt
        self._check()

    # measurement helpers (bench / tests)
    def pair_force_ids(self, group):
        return [pid for e in self.entries if e.group == group for pid in e.pair_ids]

    def recip_force_ids(self, group):
        """Library ids of the PME reciprocal-space forces of a force group (amm_pme_create)."""
        return [e.recip for e in self.entries if getattr(e, 'recip', None) is not None and e.recip_group == group]
