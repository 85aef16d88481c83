"""ctypes binding of include/atomsmm_hip.h (libatomsmm_hip.so, gfx950).

There is NO CPU fallback: if the shared library is missing or no HIP device is visible, every entry
point raises.  Device memory, streams and collectives come from torch (plumbing only).
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('AMM_LIB') or os.path.join(HERE, 'libatomsmm_hip.so')      # AMM_LIB: an experimental build (kernel tuning)

NEAR_NONE, NEAR_SHIFT, NEAR_FSWITCH, DAMPED, NONBONDED, SOFTCORE, LJ_VIRIAL = range(7)
GUARD_RC0, COULOMB_EWALD, COULOMB_RF, SWITCH, NO_SHIFT, GROUP_LJ, GROUP_Q = 1, 2, 4, 8, 16, 32, 64
BOND_HARMONIC, ANGLE_HARMONIC, BOND_LJC, BOND_NEAR, TORSION_PERIODIC, BOND_EWALD_EXCL = range(6)
BOND_VIRIAL_HARMONIC, BOND_VIRIAL_LJ = 6, 7
OP_EVAL, OP_KICK, OP_MOVE, OP_COPY, OP_COMBINE, OP_EXPR, OP_BATH = 1, 2, 3, 4, 5, 6, 7
OP_SAVE_REF, OP_CONSTRAIN_X, OP_CONSTRAIN_V = 8, 9, 10
OP_ALLREDUCE = 11
EXCHANGE_REDUCE, EXCHANGE_GATHER = 0, 1
COMM_ID_BYTES = 128
MAX_SLOTS, SLOT_X, SLOT_V = 64, 62, 63
GROUP_ALL = 32   # pseudo-group of the force symbol `f` (all groups)
KC = 138.935456   # forces.py:407
ARITY = {BOND_HARMONIC: 2, ANGLE_HARMONIC: 3, BOND_LJC: 2, BOND_NEAR: 2, TORSION_PERIODIC: 4, BOND_EWALD_EXCL: 2,
         BOND_VIRIAL_HARMONIC: 2, BOND_VIRIAL_LJ: 2}
NPAR = {BOND_HARMONIC: 2, ANGLE_HARMONIC: 2, BOND_LJC: 3, BOND_NEAR: 3, TORSION_PERIODIC: 3, BOND_EWALD_EXCL: 1,
        BOND_VIRIAL_HARMONIC: 2, BOND_VIRIAL_LJ: 3}

EXPORTS = [
    'amm_abi_version', 'amm_last_error', 'amm_create', 'amm_destroy', 'amm_set_stream', 'amm_set_slice',
    'amm_synchronize', 'amm_check', 'amm_pair_create', 'amm_pair_set_params', 'amm_pair_share_list', 'amm_bonded_create',
    'amm_bonded_add_terms', 'amm_bonded_finalize', 'amm_bonded_set_sliced', 'amm_bonded_release', 'amm_force_eval', 'amm_kick',
    'amm_move', 'amm_copy', 'amm_mvv', 'amm_bind_state', 'amm_bind_buffer', 'amm_group_define', 'amm_run_ops',
    'amm_set_fuse_inner', 'amm_set_outer_skin',
    'amm_pair_get_stats', 'amm_profile_enable', 'amm_profile_read', 'amm_pair_count_within', 'amm_pair_row_padding', 'amm_kernel_revision',
    'amm_pme_create', 'amm_pme_set_charges', 'amm_pme_set_sliced', 'amm_pair_set_lambda', 'amm_pair_set_lambda_dev', 'amm_expr_eval', 'amm_expr_eval_scalar', 'amm_expr_define', 'amm_expr_seed', 'amm_bath_define', 'amm_bath_define_nhl', 'amm_bath_define_sin', 'amm_iso_define', 'amm_pair_energy_derivative', 'amm_constraints_create', 'amm_pair_set_scale',
    'amm_comm_unique_id', 'amm_comm_init', 'amm_comm_destroy', 'amm_comm_allreduce', 'amm_comm_stats', 'amm_group_set_exchange', 'amm_bind_exchange', 'amm_exchange_finish',
    'amm_set_option', 'amm_positions_changed', 'amm_exchange_per', 'amm_run_stats', 'amm_run_ops_from', 'amm_exchange_pending',
]


class PairDesc(C.Structure):
    _fields_ = [('family', C.c_int32), ('flags', C.c_int32), ('degree', C.c_int32), ('pad_', C.c_int32),
                ('sign', C.c_double), ('rc', C.c_double), ('rswitch', C.c_double), ('rc0', C.c_double),
                ('rs0', C.c_double), ('alpha', C.c_double), ('Kc', C.c_double), ('krf', C.c_double),
                ('crf', C.c_double)]


class Op(C.Structure):
    _fields_ = [('op', C.c_int32), ('a', C.c_int32), ('b', C.c_int32), ('c', C.c_int32), ('coef', C.c_double)]


class PairStats(C.Structure):
    _fields_ = [('n_builds', C.c_int64), ('n_evals', C.c_int64), ('n_list_pairs', C.c_int64),
                ('n_slice_atoms', C.c_int64), ('capacity', C.c_int32), ('max_neighbors', C.c_int32),
                ('lanes_per_atom', C.c_int32), ('n_cells', C.c_int32), ('rlist', C.c_double),
                ('shares_list', C.c_int32), ('list_kind', C.c_int32), ('n_outer_builds', C.c_int64),
                ('n_outer_pairs', C.c_int64), ('rlist_outer', C.c_double), ('tab_error', C.c_double),
                ('has_table', C.c_int32), ('rode_along', C.c_int32),
                ('has_site_table', C.c_int32), ('n_rest_atoms', C.c_int32), ('site_tab_error', C.c_double),
                ('n_candidates', C.c_int32), ('n_candidate_walks', C.c_int32), ('chargeless', C.c_int32), ('build_split', C.c_int32)]


def slice_per(n, world):
    """Slots of the cell-sorted order per rank: whole molecules of three (csrc/amm_ctx.h: amm_slice_per)."""
    return 3 * (((n + 2) // 3 + world - 1) // world)


def pair_desc(family, rc, rc0=0.0, rs0=0.0, rswitch=0.0, alpha=0.0, degree=1, flags=0, sign=1.0, Kc=KC,
              krf=0.0, crf=0.0):
    return PairDesc(int(family), int(flags), int(degree), 0, float(sign), float(rc), float(rswitch), float(rc0),
                    float(rs0), float(alpha), float(Kc), float(krf), float(crf))


class HipError(RuntimeError):
    pass


class _LiveSet:
    """Identity set of the open contexts (weak references: a context that is garbage-collected closes itself)."""

    def __init__(self):
        import weakref
        self._refs = weakref.WeakValueDictionary()

    def add(self, obj):
        self._refs[id(obj)] = obj

    def discard(self, obj):
        self._refs.pop(id(obj), None)

    def close_all(self):
        for obj in list(self._refs.values()):
            try:
                obj.close()
            except Exception:
                pass


_LIVE = _LiveSet()
import atexit  # noqa: E402
atexit.register(_LIVE.close_all)


_LIB = None


def kernel_revision():
    return lib().amm_kernel_revision().decode()


def lib():
    """Load libatomsmm_hip.so; raises HipError when it has not been built (no fallback)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise HipError('libatomsmm_hip.so not found at %s: build it with `python -m atomsmm_amd.build` '
                           '(hipcc --offload-arch=gfx950). The HIP path has no CPU fallback.' % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        vp, dp, ip = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32)
        L.amm_kernel_revision.restype = C.c_char_p
        revision = L.amm_kernel_revision().decode()
        if revision.endswith('-tune') and os.environ.get('AMM_ALLOW_TUNE') != '1':
            # a kernel-tuning / measurement build (scripts/build_variant.sh): only part of the kernels, or kernels with arithmetic
            # removed -- never what a bench line or a test may run on
            raise HipError('%s is a tuning build (kernel revision %s): rebuild with `python -m atomsmm_amd.build --force`, or set '
                           'AMM_ALLOW_TUNE=1 for a probe script' % (LIB_PATH, revision))
        L.amm_abi_version.restype = C.c_int
        L.amm_last_error.restype = C.c_char_p
        L.amm_create.argtypes = [C.c_int32, dp, C.c_int32, vp, C.POINTER(vp)]
        L.amm_destroy.argtypes = [vp]
        L.amm_set_stream.argtypes = [vp, vp]
        L.amm_set_slice.argtypes = [vp, C.c_int32, C.c_int32]
        L.amm_synchronize.argtypes = [vp]
        L.amm_comm_unique_id.argtypes = [C.c_char_p, C.c_char_p]
        L.amm_comm_init.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_int32, C.c_int32]
        L.amm_comm_allreduce.argtypes = [vp, vp, C.c_int64]
        L.amm_comm_destroy.argtypes = [vp]
        L.amm_comm_stats.argtypes = [vp, C.POINTER(C.c_int64)]
        L.amm_run_stats.argtypes = [vp, C.POINTER(C.c_int64)]
        L.amm_group_set_exchange.argtypes = [vp, C.c_int32, C.c_int32]
        L.amm_bind_exchange.argtypes = [vp, vp, C.c_int64]
        L.amm_exchange_finish.argtypes = [vp]
        L.amm_exchange_pending.argtypes = [vp, C.POINTER(C.c_int32)]
        L.amm_check.argtypes = [vp]
        L.amm_pair_create.argtypes = [vp, C.POINTER(PairDesc), dp, dp, dp, ip, C.c_int32, C.c_double, ip]
        L.amm_pair_set_params.argtypes = [vp, C.c_int32, dp, dp, dp]
        L.amm_pair_share_list.argtypes = [vp, C.c_int32, C.c_int32]
        L.amm_bonded_create.argtypes = [vp, ip]
        L.amm_bonded_add_terms.argtypes = [vp, C.c_int32, C.c_int32, ip, dp, C.c_int32, C.c_int32, C.POINTER(PairDesc)]
        L.amm_bonded_finalize.argtypes = [vp, C.c_int32]
        L.amm_bonded_set_sliced.argtypes = [vp, C.c_int32, C.c_int32]
        L.amm_bonded_release.argtypes = [vp, C.c_int32]
        L.amm_force_eval.argtypes = [vp, C.c_int32, vp, vp, C.c_int32, vp]
        L.amm_kick.argtypes = [vp, vp, vp, vp, C.c_int32, vp, C.c_double]
        L.amm_move.argtypes = [vp, vp, vp, C.c_double]
        L.amm_copy.argtypes = [vp, vp, vp]
        L.amm_mvv.argtypes = [vp, vp, vp, vp]
        L.amm_bind_state.argtypes = [vp, vp, vp, vp]
        L.amm_bind_buffer.argtypes = [vp, C.c_int32, vp]
        L.amm_group_define.argtypes = [vp, C.c_int32, C.c_int32, ip, C.c_int32]
        L.amm_run_ops.argtypes = [vp, C.POINTER(Op), C.c_int32, C.c_int32]
        L.amm_run_ops_from.argtypes = [vp, C.POINTER(Op), C.c_int32, C.c_int32, C.POINTER(C.c_int64)]
        L.amm_set_fuse_inner.argtypes = [vp, C.c_int32]
        L.amm_set_outer_skin.argtypes = [vp, C.c_double]
        L.amm_pair_get_stats.argtypes = [vp, C.c_int32, C.POINTER(PairStats)]
        L.amm_profile_enable.argtypes = [vp, C.c_int32]
        L.amm_pair_count_within.argtypes = [vp, C.c_int32, vp, C.c_double, C.POINTER(C.c_int64)]
        L.amm_pair_row_padding.argtypes = [vp, C.c_int32, C.POINTER(C.c_int64)]
        L.amm_kernel_revision.restype = C.c_char_p
        L.amm_kernel_revision.argtypes = []
        L.amm_profile_read.argtypes = [vp, C.c_int32, C.POINTER(C.c_int64), dp]
        L.amm_pme_create.argtypes = [vp, C.c_double, C.c_int32, C.c_int32, C.c_int32, C.c_double, dp, ip]
        L.amm_pme_set_charges.argtypes = [vp, C.c_int32, dp]
        L.amm_pme_set_sliced.argtypes = [vp, C.c_int32, C.c_int32]
        L.amm_pair_set_lambda.argtypes = [vp, C.c_int32, C.c_double]
        L.amm_pair_set_lambda_dev.argtypes = [vp, C.c_int32, vp]
        L.amm_expr_eval_scalar.argtypes = [vp, ip, C.c_int32, dp, C.c_int32, vp, C.c_int32]
        L.amm_pair_set_scale.argtypes = [vp, C.c_int32, C.c_double]
        L.amm_expr_define.argtypes = [vp, ip, C.c_int32, dp, C.c_int32, dp, C.c_int32, ip]
        L.amm_expr_seed.argtypes = [vp, C.c_uint64]
        L.amm_constraints_create.argtypes = [vp, ip, dp, C.c_int32, C.c_double]
        L.amm_pair_energy_derivative.argtypes = [vp, C.c_int32, vp, vp]
        L.amm_bath_define.argtypes = [vp, C.c_double, C.c_double, ip]
        L.amm_bath_define_nhl.argtypes = [vp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32, ip]
        L.amm_bath_define_sin.argtypes = [vp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32, ip]
        L.amm_iso_define.argtypes = [vp, C.c_int32, C.c_double, C.c_double, C.c_int32]
        L.amm_set_option.argtypes = [vp, C.c_char_p, C.c_double]
        L.amm_positions_changed.argtypes = [vp]
        L.amm_exchange_per.argtypes = [vp, ip]
        L.amm_expr_eval.argtypes = [vp, ip, C.c_int32, dp, C.c_int32, dp, C.c_int32, C.c_uint64, C.c_uint64, vp, vp]
        for name in EXPORTS:
            if name not in ('amm_last_error', 'amm_kernel_revision'):
                getattr(L, name).restype = C.c_int
        L.amm_last_error.restype = C.c_char_p
        L.amm_kernel_revision.restype = C.c_char_p
        _LIB = _Serialised(L)
    return _LIB


class _Serialised:
    """The loaded library with ONE caller at a time: ctypes releases the GIL during a call, and the library's host code (launch
    caches, upload flags) is not written for two threads inside it at once.  Matters only for engine.LocalWorld -- ranks as threads of
    one process -- and costs a quarter of a microsecond per call otherwise."""

    def __init__(self, library):
        import threading
        self._library = library
        self._lock = threading.RLock()

    def __getattr__(self, name):
        fn = getattr(self._library, name)          # (AttributeError for a missing symbol, as ctypes raises it)
        lock = self._lock

        def call(*args):
            with lock:
                return fn(*args)
        self.__dict__[name] = call
        return call


def _chk(rc):
    if rc != 0:
        raise HipError(lib().amm_last_error().decode())


def _hd(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


def _hi(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(C.POINTER(C.c_int32))


def _ptr(t):
    """Device pointer of a torch tensor (fp64, contiguous, on the GPU) or None."""
    if t is None:
        return None
    import torch
    assert isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float64 and t.is_contiguous(), \
        'expected a contiguous float64 CUDA/HIP tensor'
    return C.c_void_p(t.data_ptr())


class HipContext:
    """Thin object wrapper over the C-ABI; all tensors are torch float64 tensors on the context's device."""

    def __init__(self, n_atoms, box, device=0, stream=None, rank=0, world=1):
        import torch
        if not torch.cuda.is_available():
            raise HipError('no HIP device visible to torch: the HIP path has no CPU fallback')
        L = lib()
        self.n = int(n_atoms)
        self.device = int(device)
        self.torch_device = torch.device('cuda', self.device)
        torch.cuda.set_device(self.device)
        if stream is None:
            stream = torch.cuda.current_stream(self.torch_device).cuda_stream
        b, bp = _hd(np.asarray(box, dtype=np.float64).reshape(3))
        h = C.c_void_p()
        _chk(L.amm_create(self.n, bp, self.device, C.c_void_p(stream), C.byref(h)))
        self.h = h
        self.rank, self.world = rank, world
        self.has_comm = False
        if world > 1:
            _chk(L.amm_set_slice(self.h, rank, world))
        self._keep = []

    def close(self):
        """Release the context (and its RCCL communicator) NOW.  Idempotent.  Contexts still open when the interpreter exits are
        closed by an atexit hook, while the HIP runtime and RCCL are still whole: left to __del__ during interpreter
        finalisation, ncclCommDestroy can run after the runtime it needs has begun to shut down (the stuck RCCL worker of
        round 3)."""
        if getattr(self, 'h', None):
            _LIVE.discard(self)
            lib().amm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- forces
    def pair_create(self, desc, q, sigma, eps, excl_pairs=None, skin=-1.0):
        q_, qp = _hd(q); s_, sp = _hd(sigma); e_, ep = _hd(eps)
        assert len(q_) == len(s_) == len(e_) == self.n
        ex = np.zeros((0, 2), np.int32) if excl_pairs is None else np.asarray(excl_pairs, dtype=np.int32).reshape(-1, 2)
        ex_, exp_ = _hi(ex)
        fid = C.c_int32(-1)
        _chk(lib().amm_pair_create(self.h, C.byref(desc), qp, sp, ep, exp_, len(ex_), float(skin), C.byref(fid)))
        return fid.value

    def pair_share_list(self, fid, host_fid):
        _chk(lib().amm_pair_share_list(self.h, fid, host_fid))

    def pair_set_params(self, fid, q, sigma, eps):
        q_, qp = _hd(q); s_, sp = _hd(sigma); e_, ep = _hd(eps)
        _chk(lib().amm_pair_set_params(self.h, fid, qp, sp, ep))

    def pair_energy_derivative(self, fid, pos, out):
        _chk(lib().amm_pair_energy_derivative(self.h, fid, _ptr(pos), _ptr(out)))

    def pair_set_scale(self, fid, value):
        _chk(lib().amm_pair_set_scale(self.h, fid, float(value)))

    def pair_set_lambda(self, fid, value):
        _chk(lib().amm_pair_set_lambda(self.h, fid, float(value)))

    def pair_set_lambda_dev(self, fid, scalars, index):
        """lambda of a softcore force = scalars[index] (device tensor of doubles) at every later launch; scalars = None: back to the
        number of the last pair_set_lambda."""
        ptr = None if scalars is None else C.c_void_p(scalars.data_ptr() + 8 * int(index))
        _chk(lib().amm_pair_set_lambda_dev(self.h, fid, ptr))

    def expr_eval_scalar(self, code, consts, scalars):
        """Run a scalar program (atomsmm_amd.expr.compile_scalar pieces, each closed by OUT dst): scalars[dst] <- value, in order; DEVG
        operands are entries of `scalars`."""
        c_, cp = _hi(code)
        k_, kp = _hd(consts if len(consts) else [0.0])
        _chk(lib().amm_expr_eval_scalar(self.h, cp, len(c_), kp, len(consts), _ptr(scalars), int(scalars.numel())))

    def bonded_create(self):
        fid = C.c_int32(-1)
        _chk(lib().amm_bonded_create(self.h, C.byref(fid)))
        return fid.value

    def bonded_add_terms(self, fid, kind, idx, params, periodic=False, desc=None):
        idx_, ip = _hi(np.asarray(idx).reshape(-1, ARITY[kind]))
        par_, pp = _hd(np.asarray(params, dtype=np.float64).reshape(-1, NPAR[kind]))
        assert len(idx_) == len(par_)
        _chk(lib().amm_bonded_add_terms(self.h, fid, kind, ip, pp, len(idx_), int(bool(periodic)),
                                        C.byref(desc) if desc is not None else None))

    def bonded_finalize(self, fid, sliced=False):
        _chk(lib().amm_bonded_finalize(self.h, fid))
        if sliced:
            _chk(lib().amm_bonded_set_sliced(self.h, fid, 1))

    def bonded_release(self, fid):
        """Free a bond-list set that has been replaced (its id is retired)."""
        _chk(lib().amm_bonded_release(self.h, fid))

    def pme_create(self, alpha, grid, q, Kc=KC):
        """Reciprocal space of a PME / Ewald NonbondedForce (smooth PME, order 5) on a grid[0] x grid[1] x grid[2] mesh."""
        q_, qp = _hd(q)
        fid = C.c_int32(-1)
        _chk(lib().amm_pme_create(self.h, float(alpha), int(grid[0]), int(grid[1]), int(grid[2]), float(Kc), qp, C.byref(fid)))
        return fid.value

    def pme_set_charges(self, fid, q):
        q_, qp = _hd(q)
        _chk(lib().amm_pme_set_charges(self.h, fid, qp))

    def pme_set_sliced(self, fid, on=True):
        _chk(lib().amm_pme_set_sliced(self.h, fid, int(bool(on))))

    def expr_define(self, code, consts, globals_):
        c_, cp = _hi(code)
        k_, kp = _hd(consts if len(consts) else [0.0])
        g_, gp = _hd(globals_ if len(globals_) else [0.0])
        eid = C.c_int32(-1)
        _chk(lib().amm_expr_define(self.h, cp, len(c_), kp, len(consts), gp, len(globals_), C.byref(eid)))
        return eid.value

    def constraints_create(self, pairs, distances, tolerance=1e-5):
        p_, pp = _hi(np.asarray(pairs).reshape(-1, 2))
        d_, dp_ = _hd(distances)
        assert len(p_) == len(d_)
        _chk(lib().amm_constraints_create(self.h, pp, dp_, len(d_), float(tolerance)))

    # --- the library's own RCCL communicator (csrc/comm.hip)
    @staticmethod
    def rccl_path():
        """The librccl that torch already carries in this process (one copy must serve both)."""
        import torch
        path = os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so')
        return path.encode() if os.path.exists(path) else None

    @classmethod
    def comm_unique_id(cls, library=None):
        """library: path of the RCCL to bind (default: the one torch carries; the error-path tests pass a stand-in)."""
        buf = C.create_string_buffer(COMM_ID_BYTES)
        _chk(lib().amm_comm_unique_id(library.encode() if library else cls.rccl_path(), buf))
        return buf.raw

    def comm_init(self, id_bytes, library=None):
        if len(id_bytes) != COMM_ID_BYTES:
            raise ValueError('communicator id must be %d bytes' % COMM_ID_BYTES)
        _chk(lib().amm_comm_init(self.h, library.encode() if library else self.rccl_path(), bytes(id_bytes), self.rank, self.world))
        self.has_comm = True

    def comm_destroy(self):
        _chk(lib().amm_comm_destroy(self.h))
        self.has_comm = False

    def comm_allreduce(self, tensor):
        _chk(lib().amm_comm_allreduce(self.h, _ptr(tensor), tensor.numel()))

    def comm_stats(self):
        out = (C.c_int64 * 2)()
        _chk(lib().amm_comm_stats(self.h, out))
        return dict(calls=out[0], doubles=out[1])

    def run_stats(self):
        """What amm_run_ops fused so far: launches that carried the inner RESPA loop as an epilogue, evaluations without a gather launch."""
        out = (C.c_int64 * 4)()
        _chk(lib().amm_run_stats(self.h, out))
        return dict(epilogues=out[0], copies_current=out[1], state_exchanges=out[2])

    def bath_define(self, z, kT):
        bid = C.c_int32(-1)
        _chk(lib().amm_bath_define(self.h, float(z), float(kT), C.byref(bid)))
        return bid.value

    def bath_define_nhl(self, h, z, kT, Q, friction, slot):
        bid = C.c_int32(-1)
        _chk(lib().amm_bath_define_nhl(self.h, float(h), float(z), float(kT), float(Q), float(friction), int(slot), C.byref(bid)))
        return bid.value

    def bath_define_sin(self, h, z, kT, Q2, friction, slot_v2):
        bid = C.c_int32(-1)
        _chk(lib().amm_bath_define_sin(self.h, float(h), float(z), float(kT), float(Q2), float(friction), int(slot_v2), C.byref(bid)))
        return bid.value

    def iso_define(self, on, LkT=0.0, Q1=0.0, slot_v1=-1):
        _chk(lib().amm_iso_define(self.h, int(bool(on)), float(LkT), float(Q1), int(slot_v1)))

    def expr_seed(self, seed):
        _chk(lib().amm_expr_seed(self.h, int(seed) & (2 ** 64 - 1)))

    def expr_eval(self, code, consts, globals_, seed, counter, dst=None, total=None):
        """Per-DOF postfix program (atomsmm_amd.expr): dst <- values, total <- their sum (device tensors or None)."""
        c_, cp = _hi(code)
        k_, kp = _hd(consts if len(consts) else [0.0])
        g_, gp = _hd(globals_ if len(globals_) else [0.0])
        _chk(lib().amm_expr_eval(self.h, cp, len(c_), kp, len(consts), gp, len(globals_), int(seed) & (2 ** 64 - 1),
                                 int(counter) & (2 ** 64 - 1), _ptr(dst), _ptr(total)))

    def force_eval(self, fid, pos, force, accumulate=False, energy=None):
        _chk(lib().amm_force_eval(self.h, fid, _ptr(pos), _ptr(force), int(bool(accumulate)), _ptr(energy)))

    # ---- step primitives
    def kick(self, v, f, mass, coef, fsub=None, fadd=None):
        second = fsub if fsub is not None else fadd
        _chk(lib().amm_kick(self.h, _ptr(v), _ptr(f), _ptr(second), int(fadd is not None), _ptr(mass), float(coef)))

    def move(self, x, v, coef):
        _chk(lib().amm_move(self.h, _ptr(x), _ptr(v), float(coef)))

    def copy(self, dst, src):
        _chk(lib().amm_copy(self.h, _ptr(dst), _ptr(src)))

    def mvv(self, v, mass, out):
        _chk(lib().amm_mvv(self.h, _ptr(v), _ptr(mass), _ptr(out)))

    def bind_state(self, x, v, mass):
        self._keep += [x, v, mass]
        _chk(lib().amm_bind_state(self.h, _ptr(x), _ptr(v), _ptr(mass)))

    def bind_buffer(self, slot, buf):
        self._keep.append(buf)
        _chk(lib().amm_bind_buffer(self.h, slot, _ptr(buf)))

    def group_define(self, group, slot, force_ids):
        ids, p = _hi(np.asarray(force_ids, dtype=np.int32).reshape(-1))
        _chk(lib().amm_group_define(self.h, group, slot, p, len(ids)))

    def group_set_exchange(self, group, mode):
        _chk(lib().amm_group_set_exchange(self.h, int(group), int(mode)))

    def bind_exchange(self, tensor):
        """Exchange buffer of the all-gather mode: world * 2 * exchange_per() * 3 doubles, owned by the caller."""
        self._keep.append(tensor)
        _chk(lib().amm_bind_exchange(self.h, _ptr(tensor), tensor.numel()))

    def exchange_finish(self):
        _chk(lib().amm_exchange_finish(self.h))

    def run_ops(self, ops, repeat=1):
        arr = (Op * len(ops))(*ops)
        _chk(lib().amm_run_ops(self.h, arr, len(ops), int(repeat)))

    def run_ops_host_exchanges(self, ops, repeat, exchange):
        """amm_run_ops for several ranks whose collectives the HOST makes: runs the program to its end, calling `exchange()` -- which
        must all-gather the chunks of the exchange buffer and call exchange_finish -- whenever an exchanged evaluation waits for it
        (include/atomsmm_hip.h: amm_run_ops_from)."""
        arr = (Op * len(ops))(*ops)
        cursor = C.c_int64(0)
        total = len(ops) * int(repeat)
        while True:
            _chk(lib().amm_run_ops_from(self.h, arr, len(ops), int(repeat), C.byref(cursor)))
            nf = C.c_int32(0)
            _chk(lib().amm_exchange_pending(self.h, C.byref(nf)))
            if nf.value:
                exchange(nf.value)          # (also the exchange of the program's last op: one more call winds the program up)
            elif cursor.value >= total:
                return

    def set_outer_skin(self, skin_out):
        _chk(lib().amm_set_outer_skin(self.h, float(skin_out)))

    def set_option(self, name, value):
        """Tuning / test option of the context (include/atomsmm_hip.h: amm_set_option); set before the first evaluation."""
        _chk(lib().amm_set_option(self.h, name.encode(), float(value)))

    def positions_changed(self):
        """The bound position buffer was written outside the library (option 'positions_private')."""
        _chk(lib().amm_positions_changed(self.h))

    def exchange_per(self):
        """Slots of the cell-sorted order per rank (whole molecules of three): chunk geometry of the exchange buffer."""
        per = C.c_int32(0)
        _chk(lib().amm_exchange_per(self.h, C.byref(per)))
        return per.value

    def set_fuse_inner(self, on=True):
        _chk(lib().amm_set_fuse_inner(self.h, int(bool(on))))

    def synchronize(self):
        _chk(lib().amm_synchronize(self.h))

    def check(self):
        _chk(lib().amm_check(self.h))

    def pair_stats(self, fid):
        st = PairStats()
        _chk(lib().amm_pair_get_stats(self.h, fid, C.byref(st)))
        return {name: getattr(st, name) for name, _ in PairStats._fields_}

    def pair_row_padding(self, fid):
        """(lane-trips executed, row entries) of the molecule-row traversal of this force; (0, 0) for per-atom rows."""
        out = (C.c_int64 * 2)()
        _chk(lib().amm_pair_row_padding(self.h, fid, out))
        return int(out[0]), int(out[1])

    def pair_count_within(self, fid, pos, r_within):
        """Directed list entries of force `fid` with r < r_within at `pos` (fp64 count on the device; synchronises)."""
        n = C.c_int64()
        _chk(lib().amm_pair_count_within(self.h, fid, _ptr(pos), float(r_within), C.byref(n)))
        return n.value

    def profile_enable(self, on=True, only=None):
        _chk(lib().amm_profile_enable(self.h, -(int(only) + 1) if (on and only is not None) else int(bool(on))))

    def profile_read(self, fid):
        n = C.c_int64(); ms = C.c_double()
        _chk(lib().amm_profile_read(self.h, fid, C.byref(n), C.byref(ms)))
        return n.value, ms.value
