"""CPU driver of the RESPA step program on the oracle -- TEST INFRASTRUCTURE / bench cpu_baseline leg only.

Runs the exact op sequence RespaPropagator([n0,n1,1]) emits on a flexible-water
case (groups: 0 = harmonic bonds+angles, 1 = near force-switch, 2 = DampedSmoothed total; slow force f2-f1)
with one force cache per group, i.e. the same 1/2/8 evaluations per outer step as the HIP engine.
Pair forces use the OpenMP cell-list traversal of oracle/amm_oracle.c."""
import time

import numpy as np

from . import oracle as O


class RespaCPU:
    def __init__(self, case, rc_in=0.7, rs_in=0.5, rc=1.0, rs=0.9, alpha=2.9, loops=(4, 2, 1), dt=0.004, verlet_skin=None):
        self.c = case
        self.loops, self.dt = loops, dt
        n = len(case['positions'])
        self.csr = O.exclusion_csr(n, case['exc_pairs'])
        self.dn = O.desc(O.NEAR_FSWITCH, rc=rc_in, rc0=rc_in, rs0=rs_in)
        self.dd = O.desc(O.DAMPED, rc=rc, rswitch=rs, alpha=alpha, degree=1)
        self.x = np.ascontiguousarray(case['positions'], dtype=np.float64).copy()
        self.v = np.ascontiguousarray(case['velocities'], dtype=np.float64).copy()
        self.m = np.ascontiguousarray(case['mass'], dtype=np.float64)
        self.F = {}
        self.evals = {0: 0, 1: 0, 2: 0}
        # verlet_skin: the CPU-baseline mode -- Verlet lists (one per cutoff) instead of a 27-cell walk per evaluation
        self.lists = None
        if verlet_skin:
            self.lists = {1: O.VerletList(n, case['box'], rc_in + verlet_skin, verlet_skin, self.csr),
                          2: O.VerletList(n, case['box'], rc + verlet_skin, verlet_skin, self.csr)}

    def f(self, g):
        c = self.c
        if g not in self.F:
            self.evals[g] += 1
            if g == 0:
                self.F[g] = (O.harmonic_bonds(c['bonds'], c['bond_r0'], c['bond_k'], self.x, c['box'])[1] +
                             O.harmonic_angles(c['angles'], c['angle_theta0'], c['angle_k'], self.x, c['box'])[1])
            else:
                d = self.dn if g == 1 else self.dd
                if self.lists is not None:
                    self.lists[g].update(self.x)
                    self.F[g] = self.lists[g].eval(d, self.x, c['charge'], c['sigma'], c['epsilon'])[1]
                    return self.F[g]
                cells = min(c['box']) / d.rc >= 3.0       # the 27-cell stencil needs >= 3 cells per axis
                self.F[g] = O.pair_eval(d, self.x, c['box'], c['charge'], c['sigma'], c['epsilon'], use_cells=cells,
                                        csr=self.csr)[1]
        return self.F[g]

    def step(self, nsteps=1):
        n0, n1, _ = self.loops
        dt = self.dt
        for _ in range(nsteps):
            O.kick(self.v, self.f(2), self.m, 0.5 * dt, fsub=self.f(1))
            for _a in range(n1):
                O.kick(self.v, self.f(1), self.m, 0.5 * dt / n1)
                for _b in range(n0):
                    O.kick(self.v, self.f(0), self.m, 0.5 * dt / (n0 * n1))
                    O.move(self.x, self.v, dt / (n0 * n1))
                    self.F.clear()
                    O.kick(self.v, self.f(0), self.m, 0.5 * dt / (n0 * n1))
                O.kick(self.v, self.f(1), self.m, 0.5 * dt / n1)
            O.kick(self.v, self.f(2), self.m, 0.5 * dt, fsub=self.f(1))


def time_respa(case, warmup=1, steps=3, **kw):
    """Seconds per outer step of the oracle on this host's cores (bounded sample)."""
    sim = RespaCPU(case, **kw)
    sim.step(warmup)
    t0 = time.perf_counter()
    sim.step(steps)
    return (time.perf_counter() - t0) / steps, sim
