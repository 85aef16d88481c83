/*
 * oracle/amm_oracle.c -- TEST INFRASTRUCTURE ONLY (see amm_oracle.h).
 *
 * CPU fp64 restatement of the pair/bond energy expressions AtomsMM emits and of the OpenMM
 * Reference-platform pair-loop semantics they are evaluated under:
 *   - pair set: all i<j, not an exception of the source NonbondedForce, minimum-image r < cutoff
 *   - mixing  : chargeprod = q1*q2 ; sigma = 0.5*(s1+s2) ; epsilon = sqrt(e1*e2)   (forces.py:255-257)
 *   - step(x) = 1 for x >= 0
 * Forces are the analytic -dE/dr of exactly these expressions (the reference never tests forces;
 * tests/test_oracle_golden.py cross-checks them against central differences of the energy).
 */
#include "amm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#define TWO_OVER_SQRT_PI 1.1283791670955125739

int ammo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* S(u) = 1 + u^3 (15u - 6u^2 - 10)   forces.py:543 ; dS/du = -30 u^2 (1-u)^2 */
static inline double sw_S(double u) { return 1.0 + u * u * u * (15.0 * u - 6.0 * u * u - 10.0); }
static inline double sw_dS(double u) { double w = u * (1.0 - u); return -30.0 * w * w; }

static inline double min_image(double d, double L) { return d - L * rint(d / L); }

/* force-switch constants, forces.py:559-563 */
static void fswitch_consts(double rs0, double rc0, double *b, double *f12c, double *f6c, double *f1c) {
    double bb = rs0 / (rc0 - rs0);
    *b = bb;
    *f12c = pow(1 + bb, 3) * (pow(bb, 6) + 3 * pow(bb, 5) + (30.0 / 7) * pow(bb, 4) + (25.0 / 7) * pow(bb, 3) +
                              (25.0 / 14) * bb * bb + 0.5 * bb + 2.0 / 33) / pow(bb, 9);
    *f6c = pow(1 + bb, 3) / pow(bb, 3);
    *f1c = (30 * (1 + bb)) * (bb * bb * (1 + bb) * (1 + bb) * log(1 / bb + 1) - bb * bb * bb - 1.5 * bb * bb - bb / 3 + 1.0 / 12);
}

void ammo_pair_kernel(const ammo_pair_desc *d, double r2, double qq, double sig, double eps,
                      double *e_out, double *fr_out) {
    double r = sqrt(r2);
    double e = 0.0, dedr = 0.0;
    if ((d->flags & AMMO_GUARD_RC0) && !(d->rc0 - r >= 0.0)) { /* step(rc0-r) forces.py:661,714 */
        *e_out = 0.0; *fr_out = 0.0; return;
    }
    if (d->flags & AMMO_GROUP_LJ) {
        if (qq != 2.0) { *e_out = 0.0; *fr_out = 0.0; return; }
        qq = 0.0;
    }
    if (d->flags & AMMO_GROUP_Q) {
        if (sig != 3.0) { *e_out = 0.0; *fr_out = 0.0; return; }
        sig = 1.0;
        eps = 0.0;
    }
    double inv = 1.0 / r;
    double s = sig * inv, s2 = s * s, s6 = s2 * s2 * s2, s12 = s6 * s6;
    double lj = 4.0 * eps * (s12 - s6);
    double dlj = 4.0 * eps * (-12.0 * s12 + 6.0 * s6) * inv;
    double coul = d->Kc * qq * inv;
    double dcoul = -coul * inv;
    switch (d->family) {
    case AMMO_NEAR_NONE: {       /* forces.py:542-543 */
        double u = (r - d->rs0 >= 0.0) ? (r - d->rs0) / (d->rc0 - d->rs0) : 0.0;
        double S = sw_S(u), dS = sw_dS(u) / (d->rc0 - d->rs0);
        e = S * (lj + coul);
        dedr = dS * (lj + coul) + S * (dlj + dcoul);
    } break;
    case AMMO_NEAR_SHIFT: {      /* forces.py:545-548 */
        double u = (r - d->rs0 >= 0.0) ? (r - d->rs0) / (d->rc0 - d->rs0) : 0.0;
        double S = sw_S(u), dS = sw_dS(u) / (d->rc0 - d->rs0);
        double sc = sig / d->rc0, sc2 = sc * sc, sc6 = sc2 * sc2 * sc2, sc12 = sc6 * sc6;
        double V = 4.0 * eps * (s12 - s6 - (sc12 - sc6)) + d->Kc * qq * (inv - 1.0 / d->rc0);
        e = S * V;
        dedr = dS * V + S * (dlj + dcoul);
    } break;
    case AMMO_NEAR_FSWITCH: {    /* forces.py:550-563 ; V'(r) = S(u) V'_LJC(r), forces.py:628 */
        double b, f12c, f6c, f1c;
        fswitch_consts(d->rs0, d->rc0, &b, &f12c, &f6c, &f1c);
        double f12 = 1.0, f6 = 1.0, f1 = 1.0, S = 1.0;
        if (r - d->rs0 >= 0.0) {
            double u = (r - d->rs0) / (d->rc0 - d->rs0);
            double R = u / b + 1.0;
            double u2 = u * u, u3 = u2 * u, u4 = u3 * u, u5 = u4 * u;
            double b2 = b * b, b3 = b2 * b;
            f12 = 1 + ((6 * b2 - 21 * b + 28) * (b3 * (pow(R, 12) - 1) - 12 * b2 * u - 66 * b * u2 - 220 * u3) / 462 +
                       45 * (7 - 2 * b) * u4 / 14 - 72 * u5 / 7);
            f6 = 1 + ((6 * b2 - 3 * b + 1) * (b3 * (pow(R, 6) - 1) - 6 * b2 * u - 15 * b * u2 - 20 * u3) +
                      45 * (1 - 2 * b) * u4 - 36 * u5);
            f1 = 1 + (5 * (b + 1) * (b + 1) * (6 * b3 * R * log(R) - 6 * b2 * u - 3 * b * u2 + u3) +
                      u4 * (3 * u - 5 * b - 10) / 2);
            S = sw_S(u);
        }
        double sc = sig / d->rc0, sc2 = sc * sc, sc6 = sc2 * sc2 * sc2, sc12 = sc6 * sc6;
        e = 4.0 * eps * (f12 * s12 - f6 * s6) + d->Kc * qq * f1 * inv;
        if (!(d->flags & AMMO_NO_SHIFT)) e -= 4.0 * eps * (f12c * sc12 - f6c * sc6) + d->Kc * qq * f1c / d->rc0;
        dedr = S * (dlj + dcoul);
    } break;
    case AMMO_DAMPED: {          /* forces.py:448-455 ; degree 1 == OpenMM built-in switch (:459-460) */
        double ar = d->alpha * r;
        double ec = erfc(ar);
        double V = lj + ec * coul;
        double dV = dlj + ec * dcoul - coul * d->alpha * TWO_OVER_SQRT_PI * exp(-ar * ar);
        double S = 1.0, dS = 0.0;
        if (r - d->rswitch >= 0.0) {
            int dg = d->degree;
            double den = pow(d->rc, dg) - pow(d->rswitch, dg);
            double u = (pow(r, dg) - pow(d->rswitch, dg)) / den;
            S = sw_S(u);
            dS = sw_dS(u) * dg * pow(r, dg - 1) / den;
        }
        e = S * V;
        dedr = dS * V + S * dV;
    } break;
    case AMMO_NONBONDED: {       /* OpenMM NonbondedForce direct space, SURVEY.md Appendix B.4-5 */
        double S = 1.0, dS = 0.0;
        if ((d->flags & AMMO_SWITCH) && r > d->rswitch) {
            double t = (r - d->rswitch) / (d->rc - d->rswitch);
            S = sw_S(t);
            dS = sw_dS(t) / (d->rc - d->rswitch);
        }
        e = S * lj;
        dedr = dS * lj + S * dlj;
        if (d->flags & AMMO_COULOMB_EWALD) {
            double ar = d->alpha * r;
            double ec = erfc(ar);
            e += ec * coul;
            dedr += ec * dcoul - coul * d->alpha * TWO_OVER_SQRT_PI * exp(-ar * ar);
        } else if (d->flags & AMMO_COULOMB_RF) {
            e += d->Kc * qq * (inv + d->krf * r2 - d->crf);
            dedr += d->Kc * qq * (-inv * inv + 2.0 * d->krf * r);
        } else {
            e += coul;
            dedr += dcoul;
        }
    } break;
    case AMMO_SOFTCORE: {        /* systems.py:268 with OpenMM's built-in switch (imported flags, forces.py:286-289) */
        if (qq != 2.0 || !(sig > 0.0)) break;                     /* interaction group: set 1 x set 2 only */
        double lam = d->alpha;
        double rs_ = r / sig, rs2 = rs_ * rs_, rs6 = rs2 * rs2 * rs2;
        double x = rs6 + 0.5 * (1.0 - lam);
        double V = 4.0 * lam * eps * (1.0 - x) / (x * x);
        double dV = 4.0 * lam * eps * (x - 2.0) / (x * x * x) * 6.0 * rs6 * inv;
        double S = 1.0, dS = 0.0;
        if ((d->flags & AMMO_SWITCH) && r > d->rswitch) {
            double t = (r - d->rswitch) / (d->rc - d->rswitch);
            S = sw_S(t);
            dS = sw_dS(t) / (d->rc - d->rswitch);
        }
        e = S * V;
        dedr = dS * V + S * dV;
    } break;
    case AMMO_LJ_VIRIAL: {       /* systems.py:894 */
        double W = 24.0 * eps * (2.0 * s12 - s6);
        double dW = 24.0 * eps * (-24.0 * s12 + 6.0 * s6) * inv;
        double S = 1.0, dS = 0.0;
        if ((d->flags & AMMO_SWITCH) && r > d->rswitch) {
            double t = (r - d->rswitch) / (d->rc - d->rswitch);
            S = sw_S(t);
            dS = sw_dS(t) / (d->rc - d->rswitch);
        }
        e = S * W;
        dedr = dS * W + S * dW;
    } break;
    default: break;
    }
    *e_out = d->sign * e;
    *fr_out = -d->sign * dedr * inv;
}

static int is_excluded(const int *excl_ptr, const int *excl_idx, int i, int j) {
    if (!excl_ptr) return 0;
    for (int k = excl_ptr[i]; k < excl_ptr[i + 1]; k++)
        if (excl_idx[k] == j) return 1;
    return 0;
}

static long pair_eval_n2(const ammo_pair_desc *d, int n, const double *pos, const double *box,
                         const double *q, const double *sigma, const double *eps,
                         const int *excl_ptr, const int *excl_idx, double *energy, double *f) {
    double rc2 = d->rc * d->rc;
    double etot = 0.0;
    long npairs = 0;
    for (int i = 0; i < n; i++) {
        for (int j = i + 1; j < n; j++) {
            double dx = min_image(pos[3 * i] - pos[3 * j], box[0]);
            double dy = min_image(pos[3 * i + 1] - pos[3 * j + 1], box[1]);
            double dz = min_image(pos[3 * i + 2] - pos[3 * j + 2], box[2]);
            double r2 = dx * dx + dy * dy + dz * dz;
            if (r2 >= rc2) continue;
            if (is_excluded(excl_ptr, excl_idx, i, j)) continue;
            double e, fr;
            ammo_pair_kernel(d, r2, q[i] * q[j], 0.5 * (sigma[i] + sigma[j]), sqrt(eps[i] * eps[j]), &e, &fr);
            etot += e;
            npairs++;
            if (f) {
                f[3 * i] += fr * dx; f[3 * i + 1] += fr * dy; f[3 * i + 2] += fr * dz;
                f[3 * j] -= fr * dx; f[3 * j + 1] -= fr * dy; f[3 * j + 2] -= fr * dz;
            }
        }
    }
    if (energy) *energy = etot;
    return npairs;
}

/* cell-list traversal, owner-computes per i atom (no write conflicts), OpenMP over cells */
static long pair_eval_cells(const ammo_pair_desc *d, int n, const double *pos, const double *box,
                            const double *q, const double *sigma, const double *eps,
                            const int *excl_ptr, const int *excl_idx, double *energy, double *f) {
    int nc[3];
    double cw[3];
    for (int k = 0; k < 3; k++) {
        nc[k] = (int)floor(box[k] / d->rc);
        if (nc[k] < 3) return -1; /* need >= 3 cells per axis for the 27-stencil to be duplicate-free */
        cw[k] = box[k] / nc[k];
    }
    int ncell = nc[0] * nc[1] * nc[2];
    int *cell_of = (int *)malloc(sizeof(int) * n);
    int *start = (int *)calloc(ncell + 1, sizeof(int));
    int *order = (int *)malloc(sizeof(int) * n);
    double *wp = (double *)malloc(sizeof(double) * 3 * n);
    for (int i = 0; i < n; i++) {
        int c[3];
        for (int k = 0; k < 3; k++) {
            double x = pos[3 * i + k] - box[k] * floor(pos[3 * i + k] / box[k]);
            if (x >= box[k]) x -= box[k];
            wp[3 * i + k] = x;
            c[k] = (int)(x / cw[k]);
            if (c[k] >= nc[k]) c[k] = nc[k] - 1;
        }
        cell_of[i] = (c[2] * nc[1] + c[1]) * nc[0] + c[0];
        start[cell_of[i] + 1]++;
    }
    for (int c = 0; c < ncell; c++) start[c + 1] += start[c];
    int *fill = (int *)malloc(sizeof(int) * ncell);
    memcpy(fill, start, sizeof(int) * ncell);
    for (int i = 0; i < n; i++) order[fill[cell_of[i]]++] = i;
    free(fill);
    double rc2 = d->rc * d->rc;
    double etot = 0.0;
    long npairs = 0;
#pragma omp parallel for schedule(dynamic, 8) reduction(+ : etot, npairs)
    for (int c = 0; c < ncell; c++) {
        int cx = c % nc[0], cy = (c / nc[0]) % nc[1], cz = c / (nc[0] * nc[1]);
        for (int a = start[c]; a < start[c + 1]; a++) {
            int i = order[a];
            double xi = wp[3 * i], yi = wp[3 * i + 1], zi = wp[3 * i + 2];
            double fx = 0, fy = 0, fz = 0, ei = 0;
            long np_i = 0;
            for (int dz = -1; dz <= 1; dz++)
                for (int dy = -1; dy <= 1; dy++)
                    for (int dx = -1; dx <= 1; dx++) {
                        int c2 = (((cz + dz + nc[2]) % nc[2]) * nc[1] + (cy + dy + nc[1]) % nc[1]) * nc[0] +
                                 (cx + dx + nc[0]) % nc[0];
                        for (int b2 = start[c2]; b2 < start[c2 + 1]; b2++) {
                            int j = order[b2];
                            if (j == i) continue;
                            double ddx = min_image(xi - wp[3 * j], box[0]);
                            double ddy = min_image(yi - wp[3 * j + 1], box[1]);
                            double ddz = min_image(zi - wp[3 * j + 2], box[2]);
                            double r2 = ddx * ddx + ddy * ddy + ddz * ddz;
                            if (r2 >= rc2) continue;
                            if (is_excluded(excl_ptr, excl_idx, i, j)) continue;
                            double e, fr;
                            ammo_pair_kernel(d, r2, q[i] * q[j], 0.5 * (sigma[i] + sigma[j]),
                                             sqrt(eps[i] * eps[j]), &e, &fr);
                            ei += e;
                            np_i++;
                            fx += fr * ddx; fy += fr * ddy; fz += fr * ddz;
                        }
                    }
            etot += 0.5 * ei;
            npairs += np_i;
            if (f) { f[3 * i] += fx; f[3 * i + 1] += fy; f[3 * i + 2] += fz; }
        }
    }
    free(cell_of); free(start); free(order); free(wp);
    if (energy) *energy = etot;
    return npairs / 2;
}

long ammo_pair_eval(const ammo_pair_desc *d, int n, const double *pos, const double *box,
                    const double *q, const double *sigma, const double *eps,
                    const int *excl_ptr, const int *excl_idx, double *energy, double *f, int use_cells) {
    if (use_cells) return pair_eval_cells(d, n, pos, box, q, sigma, eps, excl_ptr, excl_idx, energy, f);
    return pair_eval_n2(d, n, pos, box, q, sigma, eps, excl_ptr, excl_idx, energy, f);
}

/* ---------------------------------------------------------------------------------------------------------------------
 * CPU BASELINE (bench.py cpu_baseline leg only): Verlet neighbour list, OpenMP.  The traversal above re-walks the
 * 27-cell stencil and searches the exclusion list at every evaluation -- fine for a checker, not what a CPU engine
 * does.  This port builds a full (both-direction) list of non-excluded neighbours within rlist once per skin/2 of
 * displacement and walks it with the same ammo_pair_kernel arithmetic; same results (checked by the tests), the cost
 * profile of a production CPU code.  list: nbr_ptr[n+1], nbr_idx[]; returns the number of entries or -1. */
long ammo_nlist_build(int n, const double *pos, const double *box, double rlist, const int *excl_ptr, const int *excl_idx,
                      long capacity, long *nbr_ptr, int *nbr_idx) {
    int nc[3];
    double cw[3];
    for (int k = 0; k < 3; k++) {       /* cell edge >= rlist / 2: neighbours within +-2 cells (1.7 x fewer candidates than +-1 of rlist) */
        nc[k] = (int)floor(box[k] / (0.5 * rlist));
        if (nc[k] < 5) return -1;
        cw[k] = box[k] / nc[k];
    }
    int ncell = nc[0] * nc[1] * nc[2];
    int *cell_of = (int *)malloc(sizeof(int) * n);
    int *start = (int *)calloc(ncell + 1, sizeof(int));
    int *order = (int *)malloc(sizeof(int) * n);
    double *wp = (double *)malloc(sizeof(double) * 3 * n);
    for (int i = 0; i < n; i++) {
        int c[3];
        for (int k = 0; k < 3; k++) {
            double x = pos[3 * i + k] - box[k] * floor(pos[3 * i + k] / box[k]);
            if (x >= box[k]) x -= box[k];
            wp[3 * i + k] = x;
            c[k] = (int)(x / cw[k]);
            if (c[k] >= nc[k]) c[k] = nc[k] - 1;
        }
        cell_of[i] = (c[2] * nc[1] + c[1]) * nc[0] + c[0];
        start[cell_of[i] + 1]++;
    }
    for (int c = 0; c < ncell; c++) start[c + 1] += start[c];
    int *fill = (int *)malloc(sizeof(int) * ncell);
    memcpy(fill, start, sizeof(int) * ncell);
    for (int i = 0; i < n; i++) order[fill[cell_of[i]]++] = i;
    free(fill);
    const double rl2 = rlist * rlist;
    /* pass 1: counts; pass 2: fill (rows in atom order, so that the traversal streams the list) */
    for (int pass = 0; pass < 2; pass++) {
#pragma omp parallel for schedule(dynamic, 64)
        for (int i = 0; i < n; i++) {
            int c = cell_of[i];
            int cx = c % nc[0], cy = (c / nc[0]) % nc[1], cz = c / (nc[0] * nc[1]);
            double xi = wp[3 * i], yi = wp[3 * i + 1], zi = wp[3 * i + 2];
            long count = 0, base = pass ? nbr_ptr[i] : 0;
            for (int dz = -2; dz <= 2; dz++)
                for (int dy = -2; dy <= 2; dy++)
                    for (int dx = -2; dx <= 2; dx++) {
                        /* the periodic image of the whole neighbour cell: one shift per cell instead of a rounding per candidate */
                        const int nx = cx + dx, ny = cy + dy, nz = cz + dz;
                        const double sx = nx < 0 ? -box[0] : (nx >= nc[0] ? box[0] : 0.0), sy = ny < 0 ? -box[1] : (ny >= nc[1] ? box[1] : 0.0),
                                     sz = nz < 0 ? -box[2] : (nz >= nc[2] ? box[2] : 0.0);
                        int c2 = (((nz + nc[2]) % nc[2]) * nc[1] + (ny + nc[1]) % nc[1]) * nc[0] + (nx + nc[0]) % nc[0];
                        for (int b2 = start[c2]; b2 < start[c2 + 1]; b2++) {
                            int j = order[b2];
                            if (j == i) continue;
                            double ddx = xi - (wp[3 * j] + sx);
                            double ddy = yi - (wp[3 * j + 1] + sy);
                            double ddz = zi - (wp[3 * j + 2] + sz);
                            if (ddx * ddx + ddy * ddy + ddz * ddz >= rl2) continue;
                            if (is_excluded(excl_ptr, excl_idx, i, j)) continue;
                            if (pass) nbr_idx[base + count] = j;
                            count++;
                        }
                    }
            if (!pass) nbr_ptr[i + 1] = count;
        }
        if (!pass) {
            nbr_ptr[0] = 0;
            for (int i = 0; i < n; i++) nbr_ptr[i + 1] += nbr_ptr[i];
            if (nbr_ptr[n] > capacity) {
                free(cell_of); free(start); free(order); free(wp);
                return -2 - nbr_ptr[n];      /* caller reallocates: needs -(ret + 2) entries */
            }
        }
    }
    free(cell_of); free(start); free(order); free(wp);
    return nbr_ptr[n];
}

long ammo_pair_eval_nlist(const ammo_pair_desc *d, int n, const double *pos, const double *box, const double *q,
                          const double *sigma, const double *eps, const long *nbr_ptr, const int *nbr_idx, double *energy,
                          double *f) {
    const double rc2 = d->rc * d->rc;
    double etot = 0.0;
    long npairs = 0;
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : etot, npairs)
    for (int i = 0; i < n; i++) {
        const double xi = pos[3 * i], yi = pos[3 * i + 1], zi = pos[3 * i + 2], qi = q[i], si = sigma[i], ei_ = eps[i];
        double fx = 0, fy = 0, fz = 0, ei = 0;
        for (long k = nbr_ptr[i]; k < nbr_ptr[i + 1]; k++) {
            const int j = nbr_idx[k];
            const double ddx = min_image(xi - pos[3 * j], box[0]);
            const double ddy = min_image(yi - pos[3 * j + 1], box[1]);
            const double ddz = min_image(zi - pos[3 * j + 2], box[2]);
            const double r2 = ddx * ddx + ddy * ddy + ddz * ddz;
            if (r2 >= rc2) continue;
            double e, fr;
            ammo_pair_kernel(d, r2, qi * q[j], 0.5 * (si + sigma[j]), sqrt(ei_ * eps[j]), &e, &fr);
            ei += e;
            npairs++;
            fx += fr * ddx; fy += fr * ddy; fz += fr * ddz;
        }
        etot += 0.5 * ei;
        if (f) { f[3 * i] += fx; f[3 * i + 1] += fy; f[3 * i + 2] += fz; }
    }
    if (energy) *energy = etot;
    return npairs / 2;
}

void ammo_ewald_exclusion(int npairs, const int *pairs, const double *pos, const double *box,
                          const double *q, double alpha, double Kc, double *energy, double *f) {
    double etot = 0.0;
    for (int p = 0; p < npairs; p++) {
        int i = pairs[2 * p], j = pairs[2 * p + 1];
        double dx = min_image(pos[3 * i] - pos[3 * j], box[0]);
        double dy = min_image(pos[3 * i + 1] - pos[3 * j + 1], box[1]);
        double dz = min_image(pos[3 * i + 2] - pos[3 * j + 2], box[2]);
        double r2 = dx * dx + dy * dy + dz * dz, r = sqrt(r2), ar = alpha * r;
        double qq = Kc * q[i] * q[j];
        double e = -qq * erf(ar) / r;
        /* dE/dr = -qq [ (2 alpha/sqrt(pi)) exp(-a^2 r^2)/r - erf(ar)/r^2 ] */
        double dedr = -qq * (alpha * TWO_OVER_SQRT_PI * exp(-ar * ar) / r - erf(ar) / r2);
        etot += e;
        if (f) {
            double fr = -dedr / r;
            f[3 * i] += fr * dx; f[3 * i + 1] += fr * dy; f[3 * i + 2] += fr * dz;
            f[3 * j] -= fr * dx; f[3 * j + 1] -= fr * dy; f[3 * j + 2] -= fr * dz;
        }
    }
    if (energy) *energy = etot;
}

void ammo_ewald_reciprocal(int n, const double *pos, const double *box, const double *q,
                           double alpha, double Kc, int kmax, double *energy, double *f) {
    double V = box[0] * box[1] * box[2];
    double etot = 0.0;
    double pref = 2.0 * M_PI * Kc / V;
    int nk = 2 * kmax + 1;
    long total = (long)nk * nk * nk;
#pragma omp parallel
    {
        double *cs = (double *)malloc(sizeof(double) * 2 * n);
        double *floc = f ? (double *)calloc(3 * n, sizeof(double)) : NULL;
        double eloc = 0.0;
#pragma omp for schedule(dynamic, 64)
        for (long idx = 0; idx < total; idx++) {
            int kx = (int)(idx % nk) - kmax, ky = (int)((idx / nk) % nk) - kmax, kz = (int)(idx / ((long)nk * nk)) - kmax;
            /* half space: count each +-k pair once */
            if (kz < 0 || (kz == 0 && (ky < 0 || (ky == 0 && kx <= 0)))) continue;
            double gx = 2 * M_PI * kx / box[0], gy = 2 * M_PI * ky / box[1], gz = 2 * M_PI * kz / box[2];
            double k2 = gx * gx + gy * gy + gz * gz;
            double ak = exp(-k2 / (4 * alpha * alpha)) / k2;
            double sr = 0, si = 0;
            for (int i = 0; i < n; i++) {
                double ph = gx * pos[3 * i] + gy * pos[3 * i + 1] + gz * pos[3 * i + 2];
                double c = cos(ph), s = sin(ph);
                cs[2 * i] = c; cs[2 * i + 1] = s;
                sr += q[i] * c; si += q[i] * s;
            }
            eloc += 2.0 * pref * ak * (sr * sr + si * si);
            if (floc) {
                for (int i = 0; i < n; i++) {
                    /* F_i = 2*pref*ak * 2 q_i (s_i*Sr - c_i*Si) k   (factor 2: half space) */
                    double g = 4.0 * pref * ak * q[i] * (cs[2 * i + 1] * sr - cs[2 * i] * si);
                    floc[3 * i] += g * gx; floc[3 * i + 1] += g * gy; floc[3 * i + 2] += g * gz;
                }
            }
        }
#pragma omp critical
        {
            etot += eloc;
            if (floc) for (int i = 0; i < 3 * n; i++) f[i] += floc[i];
        }
        free(cs);
        if (floc) free(floc);
    }
    double self = 0.0;
    for (int i = 0; i < n; i++) self += q[i] * q[i];
    etot -= Kc * alpha / sqrt(M_PI) * self;
    if (energy) *energy = etot;
}

/* (2 pi N^2 / V) sum_classes w_c [ int_rc^inf V r^2 dr + int_rs^rc (S-1)... ]
 * OpenMM: sum over class pairs of count * ( eps sig^12 / (9 rc^9) - eps sig^6 / (3 rc^3) ) * 8 pi N^2/V /(N(N+1)/2) ...
 * restated as: E = 2 pi N^2 / V * < int_{rc}^{inf} V_LJ r^2 dr - int_{rs}^{rc} (S_b(r) - 1)... >.
 * With switching the energy actually summed inside rs..rc is S*V, so the correction adds
 * int_{rs}^{rc} (1 - S) V r^2 dr + int_{rc}^{inf} V r^2 dr. */
double ammo_dispersion_correction(int n, const double *sigma, const double *eps, const double *box,
                                  double rc, double rswitch, int use_switch) {
    /* collect classes of distinct (sigma, eps) */
    int ncls = 0, cap = 16;
    double *cs = (double *)malloc(sizeof(double) * cap), *ce = (double *)malloc(sizeof(double) * cap);
    long *cnt = (long *)malloc(sizeof(long) * cap);
    for (int i = 0; i < n; i++) {
        int k;
        for (k = 0; k < ncls; k++)
            if (cs[k] == sigma[i] && ce[k] == eps[i]) break;
        if (k == ncls) {
            if (ncls == cap) {
                cap *= 2;
                cs = (double *)realloc(cs, sizeof(double) * cap);
                ce = (double *)realloc(ce, sizeof(double) * cap);
                cnt = (long *)realloc(cnt, sizeof(long) * cap);
            }
            cs[ncls] = sigma[i]; ce[ncls] = eps[i]; cnt[ncls] = 0; ncls++;
        }
        cnt[k]++;
    }
    double sum = 0.0;
    for (int a = 0; a < ncls; a++)
        for (int b = a; b < ncls; b++) {
            double e = sqrt(ce[a] * ce[b]);
            if (e == 0.0) continue;
            double s = 0.5 * (cs[a] + cs[b]);
            double s6 = pow(s, 6), s12 = s6 * s6;
            double w = (a == b) ? 0.5 * cnt[a] * (cnt[a] + 1) : (double)cnt[a] * cnt[b];
            /* tail: int_rc^inf 4e(s12 r^-12 - s6 r^-6) r^2 dr = 4e( s12/(9 rc^9) - s6/(3 rc^3) ) */
            double term = 4 * e * (s12 / (9 * pow(rc, 9)) - s6 / (3 * pow(rc, 3)));
            if (use_switch) {
                /* + int_rs^rc (1 - S(t)) V(r) r^2 dr by composite Simpson (smooth integrand) */
                int m = 2000;
                double h = (rc - rswitch) / m, acc = 0.0;
                for (int k = 0; k <= m; k++) {
                    double r = rswitch + k * h, t = (r - rswitch) / (rc - rswitch);
                    double S = sw_S(t);
                    double v = 4 * e * (s12 / pow(r, 12) - s6 / pow(r, 6));
                    double g = (1 - S) * v * r * r;
                    acc += g * ((k == 0 || k == m) ? 1 : ((k & 1) ? 4 : 2));
                }
                term += acc * h / 3;
            }
            sum += w * term;
        }
    free(cs); free(ce); free(cnt);
    double V = box[0] * box[1] * box[2];
    double npart = (double)n;
    sum /= 0.5 * npart * (npart + 1);
    return 2 * M_PI * npart * npart * sum / V;
}

static inline void delta(const double *pos, const double *box, int periodic, int i, int j, double *d) {
    for (int k = 0; k < 3; k++) {
        d[k] = pos[3 * i + k] - pos[3 * j + k];
        if (periodic) d[k] = min_image(d[k], box[k]);
    }
}

void ammo_ljc_bonds(int nb, const int *ij, const double *qq, const double *sig, const double *eps,
                    double Kc, const double *pos, const double *box, int periodic, double *energy, double *f) {
    double etot = 0;
    for (int b = 0; b < nb; b++) {
        int i = ij[2 * b], j = ij[2 * b + 1];
        double d[3];
        delta(pos, box, periodic, i, j, d);
        double r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2], r = sqrt(r2);
        double x = pow(sig[b] / r, 6);   /* forces.py:406: 4*epsilon*x*(x-1) + Kc*chargeprod/r; x=(sigma/r)^6 */
        etot += 4 * eps[b] * x * (x - 1) + Kc * qq[b] / r;
        if (f) {
            double dedr = 4 * eps[b] * (-12 * x * x + 6 * x) / r - Kc * qq[b] / r2;
            double fr = -dedr / r;
            for (int k = 0; k < 3; k++) { f[3 * i + k] += fr * d[k]; f[3 * j + k] -= fr * d[k]; }
        }
    }
    if (energy) *energy = etot;
}

void ammo_near_bonds(const ammo_pair_desc *dsc, int nb, const int *ij, const double *qq, const double *sig,
                     const double *eps, const double *pos, const double *box, int periodic,
                     double *energy, double *f) {
    double etot = 0;
    for (int b = 0; b < nb; b++) {
        int i = ij[2 * b], j = ij[2 * b + 1];
        double d[3];
        delta(pos, box, periodic, i, j, d);
        double r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        double e, fr;
        ammo_pair_kernel(dsc, r2, qq[b], sig[b], eps[b], &e, &fr);
        etot += e;
        if (f) for (int k = 0; k < 3; k++) { f[3 * i + k] += fr * d[k]; f[3 * j + k] -= fr * d[k]; }
    }
    if (energy) *energy = etot;
}

void ammo_harmonic_bonds(int nb, const int *ij, const double *r0, const double *kk,
                         const double *pos, const double *box, int periodic, double *energy, double *f) {
    double etot = 0;
    for (int b = 0; b < nb; b++) {
        int i = ij[2 * b], j = ij[2 * b + 1];
        double d[3];
        delta(pos, box, periodic, i, j, d);
        double r = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        double dr = r - r0[b];
        etot += 0.5 * kk[b] * dr * dr;
        if (f) {
            double fr = -kk[b] * dr / r;
            for (int k = 0; k < 3; k++) { f[3 * i + k] += fr * d[k]; f[3 * j + k] -= fr * d[k]; }
        }
    }
    if (energy) *energy = etot;
}

void ammo_harmonic_angles(int na, const int *ijk, const double *t0, const double *kk,
                          const double *pos, const double *box, int periodic, double *energy, double *f) {
    double etot = 0;
    for (int a = 0; a < na; a++) {
        int i = ijk[3 * a], j = ijk[3 * a + 1], k = ijk[3 * a + 2];
        double d1[3], d2[3];
        delta(pos, box, periodic, i, j, d1);   /* r_i - r_j */
        delta(pos, box, periodic, k, j, d2);   /* r_k - r_j */
        double r1 = sqrt(d1[0] * d1[0] + d1[1] * d1[1] + d1[2] * d1[2]);
        double r2 = sqrt(d2[0] * d2[0] + d2[1] * d2[1] + d2[2] * d2[2]);
        double c = (d1[0] * d2[0] + d1[1] * d2[1] + d1[2] * d2[2]) / (r1 * r2);
        if (c > 1) c = 1;
        if (c < -1) c = -1;
        double th = acos(c), dth = th - t0[a];
        etot += 0.5 * kk[a] * dth * dth;
        if (f) {
            double s = sqrt(1 - c * c);
            if (s < 1e-12) s = 1e-12;
            double dEdth = kk[a] * dth;
            /* dtheta/dr_i = -(1/sin) * d(cos)/dr_i ; d(cos)/dr_i = (d2/r2 - c d1/r1)/r1 */
            for (int x = 0; x < 3; x++) {
                double dci = (d2[x] / r2 - c * d1[x] / r1) / r1;
                double dck = (d1[x] / r1 - c * d2[x] / r2) / r2;
                double fi = dEdth * dci / s, fk = dEdth * dck / s;
                f[3 * i + x] += fi; f[3 * k + x] += fk; f[3 * j + x] -= fi + fk;
            }
        }
    }
    if (energy) *energy = etot;
}

static inline void cross(const double *a, const double *b, double *c) {
    c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}
static inline double dot(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

void ammo_periodic_torsions(int nt, const int *ijkl, const int *per, const double *phase, const double *kk,
                            const double *pos, const double *box, int periodic, double *energy, double *f) {
    /* E = k (1 + cos(n phi - phase)); phi and its gradient after Blondel & Karplus (1996):
     * F = r_i - r_j, G = r_j - r_k, H = r_l - r_k, A = F x G, B = H x G,
     * cos phi = A.B/(|A||B|), sin phi = (B x A).G/(|A||B||G|). */
    double etot = 0;
    for (int t = 0; t < nt; t++) {
        int i = ijkl[4 * t], j = ijkl[4 * t + 1], k = ijkl[4 * t + 2], l = ijkl[4 * t + 3];
        double F[3], G[3], H[3], A[3], B[3], BA[3];
        delta(pos, box, periodic, i, j, F);
        delta(pos, box, periodic, j, k, G);
        delta(pos, box, periodic, l, k, H);
        cross(F, G, A);
        cross(H, G, B);
        cross(B, A, BA);
        double Gn = sqrt(dot(G, G));
        double phi = atan2(dot(BA, G) / Gn, dot(A, B));
        etot += kk[t] * (1 + cos(per[t] * phi - phase[t]));
        if (f) {
            double dEdphi = -kk[t] * per[t] * sin(per[t] * phi - phase[t]);
            double A2 = dot(A, A), B2 = dot(B, B);
            double FG = dot(F, G), HG = dot(H, G);
            for (int a = 0; a < 3; a++) {
                double gi = -Gn / A2 * A[a];
                double gl = Gn / B2 * B[a];
                double gj = Gn / A2 * A[a] + FG / (A2 * Gn) * A[a] - HG / (B2 * Gn) * B[a];
                double gk = -Gn / B2 * B[a] - FG / (A2 * Gn) * A[a] + HG / (B2 * Gn) * B[a];
                f[3 * i + a] -= dEdphi * gi; f[3 * j + a] -= dEdphi * gj;
                f[3 * k + a] -= dEdphi * gk; f[3 * l + a] -= dEdphi * gl;
            }
        }
    }
    if (energy) *energy = etot;
}

/* v <- v + coef*(f - fsub)/m   propagators.py:271 ; fsub may be NULL */
void ammo_kick(int n, double *v, const double *f, const double *fsub, const double *m, double coef) {
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) {
            double ff = fsub ? (f[3 * i + k] - fsub[3 * i + k]) : f[3 * i + k];
            v[3 * i + k] = v[3 * i + k] + coef * ff / m[i];
        }
}

/* x <- x + coef*v   propagators.py:249 */
void ammo_move(int n, double *x, const double *v, double coef) {
    for (int i = 0; i < 3 * n; i++) x[i] = x[i] + coef * v[i];
}

double ammo_mvv(int n, const double *v, const double *m) {
    double s = 0;
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) s += m[i] * v[3 * i + k] * v[3 * i + k];
    return s;
}
