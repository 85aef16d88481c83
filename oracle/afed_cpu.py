"""CPU driver of the AFED step program on the oracle -- TEST INFRASTRUCTURE (checker only).

Evaluates, with numpy and the C oracle, the program AdiabaticDynamicsIntegrator emits around a RespaPropagator([n0, n1, 1])
integrator for ONE extended variable lambda_vdw with a Nose-Hoover bath and reflecting walls (the
capture of /root/reference/src/atomsmm/integrators.py:642-860):

    one step of size DT = 2 n dt:   H  .  lambda-block  .  H        with   H = [ kick_lambda ; RESPA(dt) ; kick_lambda ]^n
    kick_lambda : v_l <- v_l - 0.5*(DT/(2n))*deriv(energy, lambda)/m_l
    lambda-block: l += DT/2 v_l (walls) ; v_eta += DT/2 (m_l v_l^2 - kT)/Q ; v_l *= exp(-DT v_eta) ; v_eta += ... ; l += DT/2 v_l (walls)

on the system RESPASystem(SolvationSystem(case, solute)) + DampedSmoothedForce outer force builds (systems.py:240-315, 62-95):
group 0 = harmonic bonds / angles / torsions + the exception bonds (NonbondedExceptionsForce) + the softcore
solute-solvent force; group 1 = near force; group 2 = damped outer force; the solute has no charge and no LJ site in
groups 1 / 2.  deriv(energy, lambda) = d/dlambda of the softcore energy and of its long-range correction, by central
differences of the oracle's energies."""
import numpy as np

from . import oracle as O


class AfedCPU:
    def __init__(self, case, loops=(2, 2, 1), dt=0.001, nsteps=2, mass=50.0, kT=2.5, tau=0.02, lam=0.8, v_lam=0.0,
                 rc_in=0.7, rs_in=0.5, rc=1.0, rs=0.9, alpha=2.9):
        c = self.c = case
        n = self.n = len(c['positions'])
        self.loops, self.dt, self.nsteps = loops, dt, nsteps
        self.m_l, self.kT_l, self.Q = mass, kT, kT * tau * tau
        self.lam, self.v_lam, self.v_eta = lam, v_lam, 0.0
        self.x = np.ascontiguousarray(c['positions'], dtype=np.float64).copy()
        self.v = np.ascontiguousarray(c['velocities'], dtype=np.float64).copy()
        self.m = np.ascontiguousarray(c['mass'], dtype=np.float64)
        solute = np.zeros(n, dtype=bool)
        solute[c['solute']] = True
        self.codes = np.where(solute, 1.0, 2.0)
        # NonbondedForce after SolvationSystem: solute parameters (0, 0, 0); every solute-solute pair is an exception
        self.q = np.where(solute, 0.0, c['charge'])
        self.sig = np.where(solute, 0.0, c['sigma'])
        self.eps = np.where(solute, 0.0, c['epsilon'])
        have = {tuple(sorted(p)) for p in c['exc_pairs'].tolist()}
        extra = [(int(i), int(j)) for k, i in enumerate(c['solute']) for j in c['solute'][k + 1:] if (int(i), int(j)) not in have]
        ex_pairs = np.concatenate([c['exc_pairs'], np.array(extra, dtype=np.int32).reshape(-1, 2)])
        ei, ej = np.array(extra, dtype=np.int64).reshape(-1, 2).T if extra else (np.zeros(0, np.int64), np.zeros(0, np.int64))
        self.ex_pairs = ex_pairs
        self.ex_qq = np.concatenate([c['exc_chargeprod'], c['charge'][ei] * c['charge'][ej]])
        self.ex_sig = np.concatenate([c['exc_sigma'], 0.5 * (c['sigma'][ei] + c['sigma'][ej])])
        self.ex_eps = np.concatenate([c['exc_epsilon'], np.sqrt(c['epsilon'][ei] * c['epsilon'][ej])])
        self.csr = O.exclusion_csr(n, ex_pairs)
        self.dn = O.desc(O.NEAR_FSWITCH, rc=rc_in, rc0=rc_in, rs0=rs_in)
        self.dd = O.desc(O.DAMPED, rc=rc, rswitch=rs, alpha=alpha, degree=1)
        self.rc, self.rs = rc, rs
        self.cells = min(c['box']) / rc >= 3.0
        self.F = {}
        self._lrc_memo = {}

    def softcore(self, lam, want_forces=True):
        c = self.c
        d = O.desc(O.SOFTCORE, rc=self.rc, rswitch=self.rs, alpha=lam, flags=O.SWITCH, Kc=1.0)
        return O.pair_eval(d, self.x, c['box'], self.codes, c['sigma'], c['epsilon'], want_forces=want_forces, use_cells=self.cells,
                           csr=self.csr)

    def _lrc(self, lam):
        # (a function of lambda alone -- not of the positions -- and a slow quadrature: remembered per value; a step asks 2 n times
        # for the same two values)
        c = self.c
        key = float(lam)
        if key not in self._lrc_memo:
            self._lrc_memo[key] = O.softcore_lrc(c['sigma'], c['epsilon'], self.codes, c['box'], self.rc, self.rs, lam)
        return self._lrc_memo[key]

    def dE_dlambda(self, h=1e-5):
        e = [self.softcore(self.lam + s * h, want_forces=False)[0] + self._lrc(self.lam + s * h) for s in (1, -1)]
        return (e[0] - e[1]) / (2 * h)

    def group_energy_forces(self, g):
        c = self.c
        if g == 0:
            parts = [O.harmonic_bonds(c['bonds'], c['bond_r0'], c['bond_k'], self.x, c['box']),
                     O.harmonic_angles(c['angles'], c['angle_theta0'], c['angle_k'], self.x, c['box']),
                     O.periodic_torsions(c['torsions'], c['torsion_n'], c['torsion_phase'], c['torsion_k'], self.x, c['box']),
                     O.ljc_bonds(self.ex_pairs, self.ex_qq, self.ex_sig, self.ex_eps, self.x, c['box']),
                     self.softcore(self.lam)[:2]]
            return sum(p[0] for p in parts), sum(p[1] for p in parts)
        d = self.dn if g == 1 else self.dd
        e, f, _ = O.pair_eval(d, self.x, c['box'], self.q, self.sig, self.eps, use_cells=self.cells, csr=self.csr)
        return e, f

    def f(self, g):
        if g not in self.F:
            self.F[g] = self.group_energy_forces(g)[1]
        return self.F[g]

    def respa(self, dt):
        n0, n1, _ = self.loops
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

    def _walls(self):
        if not (0.0 <= self.lam <= 1.0):
            self.lam = (2.0 if self.lam >= 0.0 else 0.0) - self.lam
            self.v_lam = -self.v_lam

    def _half(self, DT):
        n = self.nsteps
        for _ in range(n):
            self.v_lam -= 0.5 * (DT / (2 * n)) * self.dE_dlambda() / self.m_l
            self.respa(DT / (2 * n))
            self.v_lam -= 0.5 * (DT / (2 * n)) * self.dE_dlambda() / self.m_l

    def step(self, nsteps=1):
        DT = 2 * self.nsteps * self.dt
        for _ in range(nsteps):
            self._half(DT)
            self.lam += 0.5 * DT * self.v_lam
            self._walls()
            self.F.pop(0, None)             # group 0 holds the softcore force: it depends on lambda
            self.v_eta += 0.5 * DT * (self.m_l * self.v_lam ** 2 - self.kT_l) / self.Q
            self.v_lam *= np.exp(-DT * self.v_eta)
            self.v_eta += 0.5 * DT * (self.m_l * self.v_lam ** 2 - self.kT_l) / self.Q
            self.lam += 0.5 * DT * self.v_lam
            self._walls()
            self.F.pop(0, None)
            self._half(DT)
