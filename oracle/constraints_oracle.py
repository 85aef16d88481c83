"""oracle/constraints_oracle.py -- CPU restatement of what `addConstrainPositions` / `addConstrainVelocities` denote
(reference call sites: propagators.py:246-252, 272-273, 1126-1133).  TEST INFRASTRUCTURE ONLY: imported by tests/, never by the
product path.

What the reference's engine does [OpenMM, recalled; the library is absent from this image]: the Reference platform's
`ReferenceConstraints` solves rigid three-site waters (three atoms, three constraints, two equal legs) with the analytic SETTLE
algorithm (`ReferenceSETTLEAlgorithm`) and every other constraint with CCMA (`ReferenceCCMAAlgorithm`), both to the integrator's
constraint tolerance.  Both are solvers of ONE problem, which is therefore the specification restated here from its definition
(Ryckaert, Ciccotti & Berendsen, J. Comput. Phys. 23, 327 (1977); Andersen, J. Comput. Phys. 52, 24 (1983); Miyamoto &
Kollman, J. Comput. Chem. 13, 952 (1992)):

  positions:  x' = x + M^-1 sum_k lambda_k grad sigma_k(x_ref),   sigma_k(x') = |x'_i - x'_j|^2 - d_k^2 = 0  for every k
              (the displacements act along the bond vectors of the REFERENCE positions, mass-weighted);
  velocities: v' = v + M^-1 sum_k mu_k (x_i - x_j),               (v'_i - v'_j) . (x_i - x_j) = 0           for every k.

Three independent routes to those solutions, none of them the GPU kernel's Gauss-Seidel sweep (csrc/constraints.hip):
  * `shake_exact`   Newton's method on all multipliers of a cluster at once (dense Jacobian), to machine precision;
  * `settle`        the closed form of Miyamoto & Kollman for a three-site rigid molecule -- no iteration at all;
  * `rattle_exact`  one dense linear solve per cluster (the velocity conditions are linear in mu).
`tests/test_oracle_golden.py` pins `settle` == `shake_exact` (two derivations of one definition); the GPU tests compare the HIP
kernels with them.  Parity against OpenMM's own round-off (tolerance-dependent last digits of SETTLE / CCMA) is UNPINNED: no
reference test holds a literal for a constrained quantity that does not also need OpenMM's random velocities.
"""
import numpy as np


def shake_exact(x, xref, mass, pairs, dist, iters=50):
    """Exact solution of the position-constraint equations for ONE cluster.
    x, xref: (na, 3) unconstrained / reference positions; pairs: [(i, j)] local indices; dist: target distances."""
    x = np.asarray(x, dtype=np.float64)
    xref = np.asarray(xref, dtype=np.float64)
    im = 1.0 / np.asarray(mass, dtype=np.float64)
    pairs = [tuple(p) for p in pairs]
    nc = len(pairs)
    s = np.array([xref[i] - xref[j] for i, j in pairs])            # reference bond vectors

    def positions(lam):
        y = x.copy()
        for k, (i, j) in enumerate(pairs):
            y[i] += lam[k] * im[i] * s[k]
            y[j] -= lam[k] * im[j] * s[k]
        return y

    lam = np.zeros(nc)
    for _ in range(iters):
        y = positions(lam)
        d = np.array([y[i] - y[j] for i, j in pairs])
        sigma = (d * d).sum(1) - np.asarray(dist, dtype=np.float64) ** 2
        if np.abs(sigma).max() < 1e-28:
            break
        # d sigma_k / d lambda_l = 2 d_k . d(y_i - y_j)/d lambda_l
        J = np.zeros((nc, nc))
        for k, (i, j) in enumerate(pairs):
            for l, (p, q) in enumerate(pairs):
                coef = (im[i] if i == p else 0.0) - (im[i] if i == q else 0.0) - (im[j] if j == p else 0.0) + (im[j] if j == q else 0.0)
                J[k, l] = 2.0 * coef * (d[k] @ s[l])
        step = np.linalg.solve(J, -sigma)
        lam += step
        if np.abs(step).max() <= 1e-17 * max(1.0, np.abs(lam).max()):
            break
    return positions(lam)


def rattle_exact(x, v, mass, pairs):
    """Exact solution of the velocity-constraint equations for ONE cluster (linear)."""
    x = np.asarray(x, dtype=np.float64)
    v = np.asarray(v, dtype=np.float64)
    im = 1.0 / np.asarray(mass, dtype=np.float64)
    pairs = [tuple(p) for p in pairs]
    nc = len(pairs)
    d = np.array([x[i] - x[j] for i, j in pairs])
    A = np.zeros((nc, nc))
    b = np.zeros(nc)
    for k, (i, j) in enumerate(pairs):
        b[k] = -(v[i] - v[j]) @ d[k]
        for l, (p, q) in enumerate(pairs):
            coef = (im[i] if i == p else 0.0) - (im[i] if i == q else 0.0) - (im[j] if j == p else 0.0) + (im[j] if j == q else 0.0)
            A[k, l] = coef * (d[l] @ d[k])
    mu = np.linalg.solve(A, b)
    w = v.copy()
    for k, (i, j) in enumerate(pairs):
        w[i] += mu[k] * im[i] * d[k]
        w[j] -= mu[k] * im[j] * d[k]
    return w


def settle(x0, x1, m_o, m_h, d_oh, d_hh):
    """Analytic SETTLE (Miyamoto & Kollman 1992, section 2) for one three-site molecule, atoms ordered (O, H, H).
    x0: (3, 3) positions that satisfy the constraints (the reference); x1: (3, 3) unconstrained positions.
    Returns the constrained positions: the rigid triangle placed so that the displacement x' - x1 is a mass-weighted
    combination of the bond vectors of x0 -- the same conditions `shake_exact` solves numerically."""
    x0 = np.asarray(x0, dtype=np.float64)
    x1 = np.asarray(x1, dtype=np.float64)
    mt = m_o + 2.0 * m_h
    rc = 0.5 * d_hh
    height = np.sqrt(d_oh * d_oh - rc * rc)          # distance of O from the H-H line
    ra = 2.0 * m_h * height / mt                     # O above the centre of mass of the canonical triangle
    rb = height - ra                                 # the H-H line below it
    b0, c0 = x0[1] - x0[0], x0[2] - x0[0]
    com = (m_o * x1[0] + m_h * (x1[1] + x1[2])) / mt
    a1, b1, c1 = x1[0] - com, x1[1] - com, x1[2] - com
    # orthonormal frame: Z normal to the reference plane, X = a1 x Z, Y = Z x X
    ez = np.cross(b0, c0)
    ex = np.cross(a1, ez)
    ey = np.cross(ez, ex)
    ex, ey, ez = ex / np.linalg.norm(ex), ey / np.linalg.norm(ey), ez / np.linalg.norm(ez)
    T = np.stack([ex, ey, ez])                       # rows: lab -> frame
    b0d, c0d = T @ b0, T @ c0
    a1d, b1d, c1d = T @ a1, T @ b1, T @ c1
    # tilt of the canonical triangle out of the reference plane (phi about X, psi about Y) from the z coordinates
    sinphi = a1d[2] / ra
    cosphi = np.sqrt(1.0 - sinphi * sinphi)
    sinpsi = (b1d[2] - c1d[2]) / (2.0 * rc * cosphi)
    cospsi = np.sqrt(1.0 - sinpsi * sinpsi)
    ya2 = ra * cosphi
    xb2 = -rc * cospsi
    yb2 = -rb * cosphi - rc * sinpsi * sinphi
    yc2 = -rb * cosphi + rc * sinpsi * sinphi
    # rotation theta about Z: no net torque of the constraint forces about Z (they act along the reference bonds)
    alpha = xb2 * (b0d[0] - c0d[0]) + b0d[1] * yb2 + c0d[1] * yc2
    beta = xb2 * (c0d[1] - b0d[1]) + b0d[0] * yb2 + c0d[0] * yc2
    gamma = b0d[0] * b1d[1] - b1d[0] * b0d[1] + c0d[0] * c1d[1] - c1d[0] * c0d[1]
    a2b2 = alpha * alpha + beta * beta
    sintheta = (alpha * gamma - beta * np.sqrt(a2b2 - gamma * gamma)) / a2b2
    costheta = np.sqrt(1.0 - sintheta * sintheta)
    a3 = np.array([-ya2 * sintheta, ya2 * costheta, a1d[2]])
    b3 = np.array([xb2 * costheta - yb2 * sintheta, xb2 * sintheta + yb2 * costheta, b1d[2]])
    c3 = np.array([-xb2 * costheta - yc2 * sintheta, -xb2 * sintheta + yc2 * costheta, c1d[2]])
    return np.stack([T.T @ a3 + com, T.T @ b3 + com, T.T @ c3 + com])
