"""ctypes wrapper of oracle/cpu_port.c -- the CPU port behind bench.py's `cpu_baseline` (kind "port", not OpenMM).
TEST / MEASUREMENT INFRASTRUCTURE: loaded by bench.py's cpu_baseline leg and by tests/ only."""
import ctypes as C
import os
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(HERE, 'libamm_cpuport.so')
        if not os.path.exists(path):
            import subprocess
            subprocess.check_call(['make', '-s', '-C', HERE])
        L = C.CDLL(path)
        dp = C.POINTER(C.c_double)
        L.port_create.restype = C.c_void_p
        L.port_create.argtypes = [C.c_int, dp, dp, dp, dp, dp, dp, dp] + [C.c_double] * 10 + [C.c_int, C.c_int, C.c_double, C.c_double]
        L.port_step.argtypes = [C.c_void_p, C.c_int]
        L.port_get.argtypes = [C.c_void_p, dp, dp, dp, dp, dp, C.POINTER(C.c_long)]
        L.port_destroy.argtypes = [C.c_void_p]
        L.port_threads.restype = C.c_int
        L.port_set_threads.argtypes = [C.c_int]
        _LIB = L
    return _LIB


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


class RespaPort:
    """RespaPropagator([n0, n1, 1]) on a flexible three-site water case (atomsmm_amd.testing.tip3p_box layout: atoms O, H, H per
    molecule, bonds O-H, angle H-O-H), near force-switch (rc_in / rs_in) + DampedSmoothedForce(alpha, rc, rs) outer force."""

    def __init__(self, case, rc_in=0.7, rs_in=0.5, rc=1.0, rs=0.9, alpha=2.9, loops=(4, 2, 1), dt=0.004, skin=0.1, Kc=138.935456):
        n = len(case['positions'])
        bonds, angles = np.asarray(case['bonds']), np.asarray(case['angles'])
        assert n % 3 == 0 and len(bonds) == 2 * (n // 3) and len(angles) == n // 3, 'three-site molecules expected'
        assert (bonds[:, 0] % 3 == 0).all() and (bonds[:, 0] // 3 == bonds[:, 1] // 3).all(), 'bonds O-H within a molecule expected'
        assert (angles[:, 1] % 3 == 0).all(), 'angles H-O-H expected'
        for key in ('bond_r0', 'bond_k', 'angle_theta0', 'angle_k'):
            assert np.ptp(case[key]) == 0.0, 'one bond / angle type expected'
        self.n = n
        keep = [_d(case[k]) for k in ('box', 'positions', 'velocities', 'mass', 'charge', 'sigma', 'epsilon')]
        self.h = lib().port_create(n, *[p for _, p in keep], float(case['bond_r0'][0]), float(case['bond_k'][0]),
                                   float(case['angle_theta0'][0]), float(case['angle_k'][0]), rc_in, rs_in, rc, rs, alpha, Kc,
                                   int(loops[0]), int(loops[1]), float(dt), float(skin))

    def step(self, nsteps=1):
        lib().port_step(self.h, int(nsteps))

    def state(self):
        out = [np.zeros((self.n, 3)) for _ in range(5)]
        stats = (C.c_long * 5)()
        lib().port_get(self.h, *[a.ctypes.data_as(C.POINTER(C.c_double)) for a in out], stats)
        return dict(x=out[0], v=out[1], f0=out[2], f1=out[3], f2=out[4], builds=stats[0], evals=(stats[1], stats[2], stats[3]),
                    list_entries=stats[4])

    def close(self):
        if self.h:
            lib().port_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def threads():
    return int(lib().port_threads())


def set_threads(n):
    lib().port_set_threads(int(n))


def cpu_quota():
    """CPUs this process may use according to its cgroup (cpu.max of cgroup v2, cfs_quota of v1); None: no limit stated."""
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            quota, period = f.read().split()
        return None if quota == 'max' else float(quota) / float(period)
    except (OSError, ValueError):
        pass
    try:
        with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as f:
            quota = float(f.read())
        with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as f:
            period = float(f.read())
        return None if quota <= 0 else quota / period
    except (OSError, ValueError):
        return None


def best_thread_count(case, candidates=None, steps=12, **kw):
    """The thread count that runs the step fastest on this host: a container's CPU quota can be far below the number of logical
    CPUs it sees, and over-subscribed OpenMP threads run slower, not faster.  `steps` timed steps per candidate -- at least two
    list-rebuild intervals of the CPU's Verlet buffer, so that every candidate pays its share of list builds (five steps, less
    than one interval, made the scan disagree with the sample it chose for by up to 1.5 x)."""
    ncpu = os.cpu_count() or 1
    quota = cpu_quota()
    if candidates is None:
        candidates = set(c for c in (8, 16, 32, 64) if c <= ncpu)
        if quota:
            candidates |= {max(1, int(round(quota))), max(1, int(round(2 * quota)))}
        candidates = sorted(c for c in candidates if c <= ncpu) or [ncpu]
    sim = RespaPort(case, **kw)
    sim.step(1)
    timing = {}
    for t in candidates:
        set_threads(t)
        sim.step(1)
        t0 = time.perf_counter()
        sim.step(steps)
        timing[t] = (time.perf_counter() - t0) / steps
    sim.close()
    best = min(timing, key=timing.get)
    set_threads(best)
    return best, timing


def time_port(case, warmup=2, steps=20, **kw):
    """Seconds per outer step on this host's cores (with the thread count set by set_threads / best_thread_count, or OpenMP's default)."""
    sim = RespaPort(case, **kw)
    sim.step(warmup)
    t0 = time.perf_counter()
    sim.step(steps)
    sec = (time.perf_counter() - t0) / steps
    st = sim.state()
    sim.close()
    return sec, st
