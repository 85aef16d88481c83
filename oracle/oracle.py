"""ctypes wrapper of oracle/libamm_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package `atomsmm_amd` never does.  See oracle/amm_oracle.h for what is restated and how the
oracle is pinned (reference golden energies, tests/golden/goldens.json).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

NEAR_NONE, NEAR_SHIFT, NEAR_FSWITCH, DAMPED, NONBONDED, SOFTCORE, LJ_VIRIAL = range(7)
GUARD_RC0, COULOMB_EWALD, COULOMB_RF, SWITCH, NO_SHIFT, GROUP_LJ, GROUP_Q = 1, 2, 4, 8, 16, 32, 64
KC = 138.935456   # forces.py:407
ADJ = {None: NEAR_NONE, 'shift': NEAR_SHIFT, 'force-switch': NEAR_FSWITCH}


class PairDesc(C.Structure):
    _fields_ = [('family', C.c_int), ('flags', C.c_int), ('degree', C.c_int), ('pad_', C.c_int),
                ('sign', C.c_double), ('rc', C.c_double), ('rswitch', C.c_double),
                ('rc0', C.c_double), ('rs0', C.c_double), ('alpha', C.c_double), ('Kc', C.c_double),
                ('krf', C.c_double), ('crf', C.c_double)]


def desc(family, rc, rc0=0.0, rs0=0.0, rswitch=0.0, alpha=0.0, degree=1, flags=0, sign=1.0, Kc=KC,
         krf=0.0, crf=0.0):
    return PairDesc(family, flags, degree, 0, sign, rc, rswitch, rc0, rs0, alpha, Kc, krf, crf)


def build():
    subprocess.check_call(['make', '-s', '-C', HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(HERE, 'libamm_oracle.so')
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int)
        L.ammo_pair_eval.restype = C.c_long
        L.ammo_pair_eval.argtypes = [C.POINTER(PairDesc), C.c_int, dp, dp, dp, dp, dp, ip, ip, dp, dp, C.c_int]
        L.ammo_pair_kernel.restype = None
        L.ammo_pair_kernel.argtypes = [C.POINTER(PairDesc), C.c_double, C.c_double, C.c_double, C.c_double, dp, dp]
        L.ammo_ewald_exclusion.restype = None
        L.ammo_ewald_exclusion.argtypes = [C.c_int, ip, dp, dp, dp, C.c_double, C.c_double, dp, dp]
        L.ammo_ewald_reciprocal.restype = None
        L.ammo_ewald_reciprocal.argtypes = [C.c_int, dp, dp, dp, C.c_double, C.c_double, C.c_int, dp, dp]
        L.ammo_dispersion_correction.restype = C.c_double
        L.ammo_dispersion_correction.argtypes = [C.c_int, dp, dp, dp, C.c_double, C.c_double, C.c_int]
        L.ammo_ljc_bonds.restype = None
        L.ammo_ljc_bonds.argtypes = [C.c_int, ip, dp, dp, dp, C.c_double, dp, dp, C.c_int, dp, dp]
        L.ammo_near_bonds.restype = None
        L.ammo_near_bonds.argtypes = [C.POINTER(PairDesc), C.c_int, ip, dp, dp, dp, dp, dp, C.c_int, dp, dp]
        L.ammo_harmonic_bonds.restype = None
        L.ammo_harmonic_bonds.argtypes = [C.c_int, ip, dp, dp, dp, dp, C.c_int, dp, dp]
        L.ammo_harmonic_angles.restype = None
        L.ammo_harmonic_angles.argtypes = [C.c_int, ip, dp, dp, dp, dp, C.c_int, dp, dp]
        L.ammo_periodic_torsions.restype = None
        L.ammo_periodic_torsions.argtypes = [C.c_int, ip, ip, dp, dp, dp, dp, C.c_int, dp, dp]
        L.ammo_kick.restype = None
        L.ammo_kick.argtypes = [C.c_int, dp, dp, dp, dp, C.c_double]
        L.ammo_move.restype = None
        L.ammo_move.argtypes = [C.c_int, dp, dp, C.c_double]
        L.ammo_mvv.restype = C.c_double
        L.ammo_mvv.argtypes = [C.c_int, dp, dp]
        L.ammo_num_threads.restype = C.c_int
        _LIB = L
    return _LIB


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(C.POINTER(C.c_int))


def exclusion_csr(n, pairs):
    """Symmetric CSR (ptr, idx) of excluded partners from an (E,2) pair list."""
    pairs = np.asarray(pairs, dtype=np.int64).reshape(-1, 2)
    both = np.concatenate([pairs, pairs[:, ::-1]]) if len(pairs) else np.zeros((0, 2), np.int64)
    order = np.lexsort((both[:, 1], both[:, 0])) if len(both) else np.zeros(0, np.int64)
    both = both[order]
    ptr = np.zeros(n + 1, dtype=np.int32)
    np.add.at(ptr, both[:, 0] + 1, 1)
    ptr = np.cumsum(ptr).astype(np.int32)
    return ptr, both[:, 1].astype(np.int32)


def pair_eval(d, pos, box, q, sigma, eps, excl_pairs=None, want_forces=True, use_cells=False, csr=None):
    """csr = exclusion_csr(n, pairs) may be passed instead of excl_pairs to avoid rebuilding it per call."""
    n = len(pos)
    pos_, pp = _d(pos)
    box_, bp = _d(box)
    q_, qp = _d(q)
    s_, sp = _d(sigma)
    e_, ep = _d(eps)
    if csr is not None or (excl_pairs is not None and len(excl_pairs)):
        ptr, idx = csr if csr is not None else exclusion_csr(n, excl_pairs)
        ptr_, ptrp = _i(ptr)
        idx_, idxp = _i(idx)
    else:
        ptrp = idxp = None
    en = C.c_double(0.0)
    f = np.zeros((n, 3)) if want_forces else None
    fp = f.ctypes.data_as(C.POINTER(C.c_double)) if want_forces else None
    npairs = lib().ammo_pair_eval(C.byref(d), n, pp, bp, qp, sp, ep, ptrp, idxp, C.byref(en), fp, int(use_cells))
    if npairs < 0:
        raise RuntimeError('oracle: cell traversal needs >= 3 cells per axis')
    return en.value, f, npairs


class VerletList:
    """CPU BASELINE helper (bench.py cpu_baseline leg): full neighbour list within rc + skin, rebuilt when an atom has moved
    more than skin / 2 since the last build (oracle/amm_oracle.c: ammo_nlist_build / ammo_pair_eval_nlist)."""

    def __init__(self, n, box, rlist, skin, csr):
        self.n, self.box, self.rlist, self.skin, self.csr = n, np.ascontiguousarray(box, dtype=np.float64), rlist, skin, csr
        self.ptr = np.zeros(n + 1, dtype=np.int64)
        self.idx = np.zeros(1, dtype=np.int32)
        self.xref = None
        self.builds = 0

    def _build(self, pos):
        L = lib()
        L.ammo_nlist_build.restype = C.c_long
        ptr, idx = self.csr if self.csr is not None else (None, None)
        ptrp = ptr.ctypes.data_as(C.POINTER(C.c_int)) if ptr is not None else None
        idxp = idx.ctypes.data_as(C.POINTER(C.c_int)) if idx is not None else None
        while True:
            r = L.ammo_nlist_build(self.n, pos.ctypes.data_as(C.POINTER(C.c_double)), self.box.ctypes.data_as(C.POINTER(C.c_double)),
                                   C.c_double(self.rlist), ptrp, idxp, C.c_long(len(self.idx)),
                                   self.ptr.ctypes.data_as(C.POINTER(C.c_long)), self.idx.ctypes.data_as(C.POINTER(C.c_int)))
            if r >= 0:
                break
            if r == -1:
                raise RuntimeError('oracle: the neighbour-list build needs >= 3 cells per axis')
            self.idx = np.zeros(int(1.2 * (-(r + 2))) + 1024, dtype=np.int32)
        self.xref = pos.copy()
        self.builds += 1

    def update(self, pos):
        pos = np.ascontiguousarray(pos, dtype=np.float64)
        if self.xref is None or np.max(np.sum((pos - self.xref) ** 2, axis=1)) > (0.5 * self.skin) ** 2:
            self._build(pos)

    def eval(self, d, pos, q, sigma, eps, want_forces=True):
        pos_, pp = _d(pos)
        q_, qp = _d(q); s_, sp = _d(sigma); e_, ep = _d(eps)
        en = C.c_double(0.0)
        f = np.zeros((self.n, 3)) if want_forces else None
        L = lib()
        L.ammo_pair_eval_nlist.restype = C.c_long
        npairs = L.ammo_pair_eval_nlist(C.byref(d), self.n, pp, self.box.ctypes.data_as(C.POINTER(C.c_double)), qp, sp, ep,
                                        self.ptr.ctypes.data_as(C.POINTER(C.c_long)), self.idx.ctypes.data_as(C.POINTER(C.c_int)),
                                        C.byref(en), f.ctypes.data_as(C.POINTER(C.c_double)) if want_forces else None)
        return en.value, f, npairs


def pair_kernel(d, r2, qq, sig, eps):
    e, fr = C.c_double(), C.c_double()
    lib().ammo_pair_kernel(C.byref(d), r2, qq, sig, eps, C.byref(e), C.byref(fr))
    return e.value, fr.value


def ewald_exclusion(pairs, pos, box, q, alpha, Kc=KC, want_forces=True):
    pr_, prp = _i(np.asarray(pairs).reshape(-1, 2))
    pos_, pp = _d(pos); box_, bp = _d(box); q_, qp = _d(q)
    en = C.c_double()
    f = np.zeros((len(pos), 3)) if want_forces else None
    fp = f.ctypes.data_as(C.POINTER(C.c_double)) if want_forces else None
    lib().ammo_ewald_exclusion(len(pr_), prp, pp, bp, qp, alpha, Kc, C.byref(en), fp)
    return en.value, f


def ewald_reciprocal(pos, box, q, alpha, kmax, Kc=KC, want_forces=False):
    pos_, pp = _d(pos); box_, bp = _d(box); q_, qp = _d(q)
    en = C.c_double()
    f = np.zeros((len(pos), 3)) if want_forces else None
    fp = f.ctypes.data_as(C.POINTER(C.c_double)) if want_forces else None
    lib().ammo_ewald_reciprocal(len(pos_), pp, bp, qp, alpha, Kc, kmax, C.byref(en), fp)
    return en.value, f


def dispersion_correction(sigma, eps, box, rc, rswitch=None):
    s_, sp = _d(sigma); e_, ep = _d(eps); b_, bp = _d(box)
    return lib().ammo_dispersion_correction(len(s_), sp, ep, bp, rc, rswitch or 0.0, int(rswitch is not None))


def softcore_lrc(sigma, eps, codes, box, rc, rswitch, lam):
    """Long-range correction of a softcore CustomNonbondedForce with one interaction group (OpenMM
    CustomNonbondedForceImpl::calcLongRangeCorrection [recalled]): (4 pi / V) N/(N+1) sum over (set 1, set 2) pairs
    of int_rc^inf u r^2 dr + int_rs^rc (1 - S) u r^2 dr.  scipy adaptive quadrature (independent of the product's
    Gauss-Legendre rule)."""
    from scipy.integrate import quad

    def u(r, s, e):
        x = (r / s) ** 6 + 0.5 * (1.0 - lam)
        return 4.0 * lam * e * (1.0 - x) / (x * x)

    n = len(sigma)
    cls = {}
    for i in np.where(np.asarray(codes) == 1)[0]:
        for j in np.where(np.asarray(codes) == 2)[0]:
            e = np.sqrt(eps[i] * eps[j])
            if e > 0:
                key = (0.5 * (sigma[i] + sigma[j]), e)
                cls[key] = cls.get(key, 0) + 1
    total = 0.0
    for (s, e), count in cls.items():
        integral = quad(lambda r: u(r, s, e) * r * r, rc, np.inf, epsabs=0, epsrel=1e-12)[0]
        if rswitch is not None:
            def f(r):
                t = (r - rswitch) / (rc - rswitch)
                return t ** 3 * (10 - 15 * t + 6 * t * t) * u(r, s, e) * r * r
            integral += quad(f, rswitch, rc, epsabs=0, epsrel=1e-12)[0]
        total += count * integral
    return 4 * np.pi * n / (n + 1.0) * total / float(np.prod(box))


def custom_lrc(u, sigma, eps, box, rc, rswitch=None):
    """Long-range correction of a CustomNonbondedForce WITHOUT interaction groups (OpenMM
    CustomNonbondedForceImpl::calcLongRangeCorrection [recalled]; pinned by tests/test_computers.py:31): classes of
    equal (sigma, epsilon); class pairs i <= j count n_i n_j (n_i (n_i + 1)/2 for i == j); the sum is divided by
    N (N + 1)/2 and multiplied by 2 pi N^2 / V.  u(r, sigma_ij, eps_ij) is the pair energy."""
    from scipy.integrate import quad
    n = len(sigma)
    cls = {}
    for s_, e_ in zip(sigma, eps):
        cls[(s_, e_)] = cls.get((s_, e_), 0) + 1
    keys = list(cls)
    total = 0.0
    for a, k1 in enumerate(keys):
        for k2 in keys[a:]:
            s_, e_ = 0.5 * (k1[0] + k2[0]), np.sqrt(k1[1] * k2[1])
            if e_ == 0:
                continue
            count = cls[k1] * (cls[k1] + 1) / 2 if k1 == k2 else cls[k1] * cls[k2]
            integral = quad(lambda r: u(r, s_, e_) * r * r, rc, np.inf, epsabs=0, epsrel=1e-12)[0]
            if rswitch is not None:
                def f(r):
                    t = (r - rswitch) / (rc - rswitch)
                    return t ** 3 * (10 - 15 * t + 6 * t * t) * u(r, s_, e_) * r * r
                integral += quad(f, rswitch, rc, epsabs=0, epsrel=1e-12)[0]
            total += count * integral
    return 2 * np.pi * n * n / float(np.prod(box)) * total / (n * (n + 1) / 2)


def _bonded(fn, idx, params, pos, box, periodic, want_forces, pre=()):
    idx_, ip = _i(idx)
    keep = [_d(p) for p in params]
    pos_, pp = _d(pos); box_, bp = _d(box)
    en = C.c_double()
    f = np.zeros((len(pos_), 3)) if want_forces else None
    fp = f.ctypes.data_as(C.POINTER(C.c_double)) if want_forces else None
    fn(*pre, len(idx_), ip, *[k[1] for k in keep], pp, bp, int(periodic), C.byref(en), fp)
    return en.value, f


def ljc_bonds(ij, qq, sig, eps, pos, box, periodic=True, want_forces=True, Kc=KC):
    idx_, ip = _i(ij)
    a, ap = _d(qq); b, bp_ = _d(sig); c, cp = _d(eps)
    pos_, pp = _d(pos); box_, bp = _d(box)
    en = C.c_double()
    f = np.zeros((len(pos_), 3)) if want_forces else None
    fp = f.ctypes.data_as(C.POINTER(C.c_double)) if want_forces else None
    lib().ammo_ljc_bonds(len(idx_), ip, ap, bp_, cp, Kc, pp, bp, int(periodic), C.byref(en), fp)
    return en.value, f


def near_bonds(d, ij, qq, sig, eps, pos, box, periodic=True, want_forces=True):
    return _bonded(lib().ammo_near_bonds, ij, [qq, sig, eps], pos, box, periodic, want_forces, pre=(C.byref(d),))


def harmonic_bonds(ij, r0, k, pos, box, periodic=False, want_forces=True):
    return _bonded(lib().ammo_harmonic_bonds, ij, [r0, k], pos, box, periodic, want_forces)


def harmonic_angles(ijk, t0, k, pos, box, periodic=False, want_forces=True):
    return _bonded(lib().ammo_harmonic_angles, ijk, [t0, k], pos, box, periodic, want_forces)


def periodic_torsions(ijkl, per, phase, k, pos, box, periodic=False, want_forces=True):
    idx_, ip = _i(ijkl)
    per_, perp = _i(per)
    a, ap = _d(phase); b, bp_ = _d(k)
    pos_, pp = _d(pos); box_, bp = _d(box)
    en = C.c_double()
    f = np.zeros((len(pos_), 3)) if want_forces else None
    fp = f.ctypes.data_as(C.POINTER(C.c_double)) if want_forces else None
    lib().ammo_periodic_torsions(len(idx_), ip, perp, ap, bp_, pp, bp, int(periodic), C.byref(en), fp)
    return en.value, f


def kick(v, f, m, coef, fsub=None):
    """v <- v + coef*(f - fsub)/m  in place (v: (n,3) float64 C-contiguous)."""
    assert v.flags.c_contiguous and v.dtype == np.float64
    f_, fp = _d(f); m_, mp = _d(m)
    sp = None
    if fsub is not None:
        s_, sp = _d(fsub)
    lib().ammo_kick(len(m_), v.ctypes.data_as(C.POINTER(C.c_double)), fp, sp, mp, coef)


def move(x, v, coef):
    assert x.flags.c_contiguous and x.dtype == np.float64
    v_, vp = _d(v)
    lib().ammo_move(len(x), x.ctypes.data_as(C.POINTER(C.c_double)), vp, coef)


def mvv(v, m):
    v_, vp = _d(v); m_, mp = _d(m)
    return lib().ammo_mvv(len(m_), vp, mp)


def num_threads():
    return lib().ammo_num_threads()
