/* oracle/cpu_port.c -- CPU port of the RESPA hot path for bench.py's `cpu_baseline` leg (kind "port", NOT OpenMM).
 *
 * TEST / MEASUREMENT INFRASTRUCTURE: only bench.py's cpu_baseline leg and tests/ load this library; the product path
 * (atomsmm_amd/) never does.  It restates, for a flexible three-site water box, the step program RespaPropagator([n0, n1, 1])
 * emits (/root/reference/src/atomsmm/propagators.py:933-973) over the force groups RESPASystem makes
 * (systems.py:62-95): group 0 harmonic bonds + angles, group 1 near force (force-switched, forces.py:549-563, V' = S V'_LJC :628),
 * group 2 DampedSmoothedForce (forces.py:448-455, degree 1 = built-in switch), slow force f2 - f1 -- the same arithmetic as
 * oracle/amm_oracle.c (checked against it by tests/test_oracle_golden.py), organised the way a CPU MD code would:
 *   - the whole step loop in C (the Python driver of oracle/respa_cpu.py spent most of a step outside the pair loops);
 *   - one Verlet list (rc + skin) built from a cell grid in parallel, rows in cell-sorted order with the partners inside the
 *     near list radius first, shared by both pair forces; rebuilt when an atom has moved more than skin / 2;
 *   - every pair evaluated once (Newton's third law) into per-thread copies of the force arrays, summed in thread order (no
 *     atomics); OpenMP over rows / molecules / atoms everywhere, first-touch allocation by the threads that later use the pages;
 *   - one force cache per group: 1 outer + n1 near + n0 n1 inner evaluations per step, the outer and the last near one in a
 *     single traversal.
 * Exclusions are the pairs inside a molecule (atoms 3m, 3m+1, 3m+2): what the reference's exceptions -> exclusions give for water.
 */
#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int n, n0, n1;
    double box[3], dt, skin;
    double rc, rs, alpha, rc0, rs0, Kc;
    double *x, *v, *m, *q, *hsig, *seps2;       /* positions, velocities, masses; charge, sigma / 2, 2 sqrt(eps) */
    double bond_r0, bond_k, angle_t0, angle_k;
    double *f0, *f1, *f2, *xref;
    /* cell-sorted Verlet list */
    int *order, *nbr_idx;
    long *nbr_ptr, cap;
    int *nfront;                                /* entries of a row within the near list radius (they come first) */
    long builds, evals[3];
    double f12c, f6c, f1c, b;
    double *scratch;                            /* per-thread copies of f1 and f2 (pair_pass) */
    int scratch_threads;
} port_t;

static inline double min_image(double d, double L) { return d - L * nearbyint(d / L); }

/* near force over r per unit: S(u) (-dV_LJC/dr) / r  (forces.py:628) */
static inline double near_fr(const port_t *p, double r2, double qq, double sig, double eps4) {
    const double rinv = 1.0 / sqrt(r2), r = r2 * rinv, rinv2 = rinv * rinv;
    const double s2 = sig * sig * rinv2, s6 = s2 * s2 * s2, s12 = s6 * s6;
    const double dlj = eps4 * (12.0 * s12 - 6.0 * s6) * rinv2, dc = p->Kc * qq * rinv * rinv2;
    const double du = r - p->rs0;
    double S = 1.0;
    if (du > 0.0) {
        const double u = du / (p->rc0 - p->rs0);
        S = 1.0 + u * u * u * (15.0 * u - 6.0 * u * u - 10.0);
    }
    return S * (dlj + dc);
}

/* DampedSmoothedForce, degree 1: S_b(t) (LJ + erfc(alpha r) Kc qq / r), S_b the built-in switch  (forces.py:448-460) */
static inline double damped_fr(const port_t *p, double r2, double qq, double sig, double eps4) {
    const double rinv = 1.0 / sqrt(r2), r = r2 * rinv, rinv2 = rinv * rinv;
    const double s2 = sig * sig * rinv2, s6 = s2 * s2 * s2, s12 = s6 * s6;
    const double ar = p->alpha * r, ec = erfc(ar), ex = exp(-ar * ar);
    const double coul = p->Kc * qq * rinv;
    const double V = eps4 * (s12 - s6) + ec * coul;
    const double mdV_r = eps4 * (12.0 * s12 - 6.0 * s6) * rinv2 + ec * coul * rinv2 + coul * p->alpha * 1.1283791670955125739 * ex * rinv;
    double S = 1.0, dS = 0.0;
    if (r > p->rs) {
        const double t = (r - p->rs) / (p->rc - p->rs), w = t * (1.0 - t);
        S = 1.0 + t * t * t * (15.0 * t - 6.0 * t * t - 10.0);
        dS = -30.0 * w * w / (p->rc - p->rs);
    }
    return S * mdV_r - dS * V * rinv;
}

static void build_list(port_t *p) {
    const int n = p->n;
    const double rlist = p->rc + p->skin, rnear = p->rc0 + p->skin, rl2 = rlist * rlist, rn2 = rnear * rnear;
    int nc[3];
    double cw[3];
    int small = 0;
    for (int k = 0; k < 3; k++) {
        nc[k] = (int)floor(p->box[k] / (0.5 * rlist));
        if (nc[k] < 5) small = 1;           /* the +-2 stencil would wrap onto itself: small boxes take the all-pairs build */
        if (nc[k] < 1) nc[k] = 1;
        cw[k] = p->box[k] / nc[k];
    }
    if (small) {
        for (int i = 0; i < n; i++) p->order[i] = i;
        for (int pass = 0; pass < 2; pass++) {
#pragma omp parallel for schedule(dynamic, 32)
            for (int s = 0; s < n; s++) {
                long front = 0, backn = 0;
                const long base = pass ? p->nbr_ptr[s] : 0, end = pass ? p->nbr_ptr[s + 1] : 0;
                for (int j = 0; j < n; j++) {
                    if (j / 3 == s / 3) continue;
                    const double ddx = min_image(p->x[3 * s] - p->x[3 * j], p->box[0]), ddy = min_image(p->x[3 * s + 1] - p->x[3 * j + 1], p->box[1]),
                                 ddz = min_image(p->x[3 * s + 2] - p->x[3 * j + 2], p->box[2]);
                    const double r2 = ddx * ddx + ddy * ddy + ddz * ddz;
                    if (r2 >= rl2) continue;
                    if (pass) {
                        if (r2 < rn2) p->nbr_idx[base + front] = j;
                        else p->nbr_idx[end - 1 - backn] = j;
                    }
                    if (r2 < rn2) front++;
                    else backn++;
                }
                if (!pass) p->nbr_ptr[s + 1] = front + backn;
                else p->nfront[s] = (int)front;
            }
            if (!pass) {
                p->nbr_ptr[0] = 0;
                for (int s = 0; s < n; s++) p->nbr_ptr[s + 1] += p->nbr_ptr[s];
                if (p->nbr_ptr[n] > p->cap) {
                    free(p->nbr_idx);
                    p->cap = (long)(1.2 * p->nbr_ptr[n]) + 1024;
                    p->nbr_idx = (int *)malloc(sizeof(int) * p->cap);
                }
            }
        }
        memcpy(p->xref, p->x, sizeof(double) * 3 * n);
        p->builds++;
        return;
    }
    const int ncell = nc[0] * nc[1] * nc[2];
    int *cell_of = (int *)malloc(sizeof(int) * n), *start = (int *)calloc(ncell + 1, sizeof(int)), *fill = (int *)malloc(sizeof(int) * ncell);
    double *wp = (double *)malloc(sizeof(double) * 3 * n);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) {
        int c[3];
        for (int k = 0; k < 3; k++) {
            double x = p->x[3 * i + k] - p->box[k] * floor(p->x[3 * i + k] / p->box[k]);
            if (x >= p->box[k]) x -= p->box[k];
            wp[3 * i + k] = x;
            c[k] = (int)(x / cw[k]);
            if (c[k] >= nc[k]) c[k] = nc[k] - 1;
        }
        cell_of[i] = (c[2] * nc[1] + c[1]) * nc[0] + c[0];
    }
    for (int i = 0; i < n; i++) start[cell_of[i] + 1]++;
    for (int c = 0; c < ncell; c++) start[c + 1] += start[c];
    memcpy(fill, start, sizeof(int) * ncell);
    for (int i = 0; i < n; i++) p->order[fill[cell_of[i]]++] = i;
    for (int pass = 0; pass < 2; pass++) {
#pragma omp parallel for schedule(dynamic, 32)
        for (int s = 0; s < n; s++) {              /* rows in cell-sorted order: a thread's rows are neighbours in space */
            const int i = p->order[s], c = cell_of[i], mol = i / 3;
            const int cx = c % nc[0], cy = (c / nc[0]) % nc[1], cz = c / (nc[0] * nc[1]);
            const double xi = wp[3 * i], yi = wp[3 * i + 1], zi = wp[3 * i + 2];
            long front = 0, backn = 0;
            const long base = pass ? p->nbr_ptr[s] : 0, end = pass ? p->nbr_ptr[s + 1] : 0;
            for (int dz = -2; dz <= 2; dz++)
                for (int dy = -2; dy <= 2; dy++)
                    for (int dx = -2; dx <= 2; dx++) {
                        const int nx = cx + dx, ny = cy + dy, nz = cz + dz;
                        const double sx = nx < 0 ? -p->box[0] : (nx >= nc[0] ? p->box[0] : 0.0), sy = ny < 0 ? -p->box[1] : (ny >= nc[1] ? p->box[1] : 0.0),
                                     sz = nz < 0 ? -p->box[2] : (nz >= nc[2] ? p->box[2] : 0.0);
                        const int c2 = (((nz + nc[2]) % nc[2]) * nc[1] + (ny + nc[1]) % nc[1]) * nc[0] + (nx + nc[0]) % nc[0];
                        for (int b2 = start[c2]; b2 < start[c2 + 1]; b2++) {
                            const int j = p->order[b2];
                            if (j / 3 == mol) continue;                 /* same molecule: excluded (and self) */
                            const double ddx = xi - (wp[3 * j] + sx), ddy = yi - (wp[3 * j + 1] + sy), ddz = zi - (wp[3 * j + 2] + sz);
                            const double r2 = ddx * ddx + ddy * ddy + ddz * ddz;
                            if (r2 >= rl2) continue;
                            if (pass) {
                                if (r2 < rn2) p->nbr_idx[base + front] = j;
                                else p->nbr_idx[end - 1 - backn] = j;
                            }
                            if (r2 < rn2) front++;
                            else backn++;
                        }
                    }
            if (!pass) p->nbr_ptr[s + 1] = front + backn;
            else p->nfront[s] = (int)front;
        }
        if (!pass) {
            p->nbr_ptr[0] = 0;
            for (int s = 0; s < n; s++) p->nbr_ptr[s + 1] += p->nbr_ptr[s];
            if (p->nbr_ptr[n] > p->cap) {
                free(p->nbr_idx);
                p->cap = (long)(1.2 * p->nbr_ptr[n]) + 1024;
                p->nbr_idx = (int *)malloc(sizeof(int) * p->cap);
#pragma omp parallel for schedule(static)
                for (long k = 0; k < p->cap; k += 1024) p->nbr_idx[k] = 0;      /* first touch */
            }
        }
    }
    memcpy(p->xref, p->x, sizeof(double) * 3 * n);
    free(cell_of); free(start); free(fill); free(wp);
    p->builds++;
}

static void check_list(port_t *p) {
    const double thr2 = 0.25 * p->skin * p->skin;
    int stale = p->builds == 0;
    if (!stale) {
#pragma omp parallel for schedule(static) reduction(| : stale)
        for (int i = 0; i < p->n; i++) {
            const double dx = p->x[3 * i] - p->xref[3 * i], dy = p->x[3 * i + 1] - p->xref[3 * i + 1], dz = p->x[3 * i + 2] - p->xref[3 * i + 2];
            if (dx * dx + dy * dy + dz * dz > thr2) stale |= 1;
        }
    }
    if (stale) build_list(p);
}

/* which: 1 near force only (front parts) -> f1; 2 outer force over the whole rows AND the near force over the front parts -> f2, f1.
 * Every pair is evaluated ONCE (Newton's third law: the row of the atom with the smaller index does it -- the rows are full, so it
 * is there) into per-thread copies of the force arrays, which are then added up in thread order. */
static void pair_pass(port_t *p, int which) {
    check_list(p);
    const double rc2 = p->rc * p->rc, rc02 = p->rc0 * p->rc0;
    const int nt = omp_get_max_threads();
    const size_t n3 = 3 * (size_t)p->n;
    if (p->scratch_threads < nt) {
        free(p->scratch);
        p->scratch = (double *)malloc(sizeof(double) * 2 * n3 * (size_t)nt);
        p->scratch_threads = nt;
    }
#pragma omp parallel
    {
        const int t = omp_get_thread_num();
        double *g1 = p->scratch + (size_t)t * 2 * n3, *g2 = g1 + n3;
        memset(g1, 0, sizeof(double) * n3);
        if (which == 2) memset(g2, 0, sizeof(double) * n3);
#pragma omp for schedule(dynamic, 32)
        for (int s = 0; s < p->n; s++) {
            const int i = p->order[s];
            const double xi = p->x[3 * i], yi = p->x[3 * i + 1], zi = p->x[3 * i + 2], qi = p->q[i], hi = p->hsig[i], ei = p->seps2[i];
            double fx = 0, fy = 0, fz = 0, gx = 0, gy = 0, gz = 0;
            const long b = p->nbr_ptr[s], e = which == 2 ? p->nbr_ptr[s + 1] : b + p->nfront[s], fe = b + p->nfront[s];
            for (long k = b; k < e; k++) {
                const int j = p->nbr_idx[k];
                if (j < i) continue;                              /* the row of j has this pair */
                const double dx = min_image(xi - p->x[3 * j], p->box[0]), dy = min_image(yi - p->x[3 * j + 1], p->box[1]),
                             dz = min_image(zi - p->x[3 * j + 2], p->box[2]);
                const double r2 = dx * dx + dy * dy + dz * dz;
                const double qq = qi * p->q[j], sig = hi + p->hsig[j], eps4 = ei * p->seps2[j];
                if (which == 2 && r2 < rc2) {
                    const double fr = damped_fr(p, r2, qq, sig, eps4);
                    fx += fr * dx; fy += fr * dy; fz += fr * dz;
                    g2[3 * j] -= fr * dx; g2[3 * j + 1] -= fr * dy; g2[3 * j + 2] -= fr * dz;
                }
                if (k < fe && r2 < rc02) {
                    const double fr = near_fr(p, r2, qq, sig, eps4);
                    gx += fr * dx; gy += fr * dy; gz += fr * dz;
                    g1[3 * j] -= fr * dx; g1[3 * j + 1] -= fr * dy; g1[3 * j + 2] -= fr * dz;
                }
            }
            if (which == 2) { g2[3 * i] += fx; g2[3 * i + 1] += fy; g2[3 * i + 2] += fz; }
            g1[3 * i] += gx; g1[3 * i + 1] += gy; g1[3 * i + 2] += gz;
        }
        /* (implicit barrier) the threads' copies, added in thread order */
#pragma omp for schedule(static)
        for (long c = 0; c < (long)n3; c++) {
            double a1 = 0.0, a2 = 0.0;
            for (int u = 0; u < nt; u++) {
                a1 += p->scratch[(size_t)u * 2 * n3 + c];
                if (which == 2) a2 += p->scratch[(size_t)u * 2 * n3 + n3 + c];
            }
            p->f1[c] = a1;
            if (which == 2) p->f2[c] = a2;
        }
    }
    p->evals[1]++;
    if (which == 2) p->evals[2]++;
}

/* group 0: two O-H bonds and the H-O-H angle of every molecule (atoms O, H, H), owner = the molecule */
static void bonded(port_t *p) {
    const int nm = p->n / 3;
#pragma omp parallel for schedule(static)
    for (int m = 0; m < nm; m++) {
        const double *o = p->x + 9 * m, *h1 = o + 3, *h2 = o + 6;
        double d1[3], d2[3], f[9] = {0};
        for (int k = 0; k < 3; k++) { d1[k] = h1[k] - o[k]; d2[k] = h2[k] - o[k]; }
        const double r1 = sqrt(d1[0] * d1[0] + d1[1] * d1[1] + d1[2] * d1[2]), r2 = sqrt(d2[0] * d2[0] + d2[1] * d2[1] + d2[2] * d2[2]);
        const double fb1 = -p->bond_k * (r1 - p->bond_r0) / r1, fb2 = -p->bond_k * (r2 - p->bond_r0) / r2;
        double c = (d1[0] * d2[0] + d1[1] * d2[1] + d1[2] * d2[2]) / (r1 * r2);
        c = c > 1 ? 1 : (c < -1 ? -1 : c);
        double sn = sqrt(1 - c * c);
        if (sn < 1e-12) sn = 1e-12;
        const double dE = p->angle_k * (acos(c) - p->angle_t0);
        for (int k = 0; k < 3; k++) {
            const double fi = dE * ((d2[k] / r2 - c * d1[k] / r1) / r1) / sn, fk = dE * ((d1[k] / r1 - c * d2[k] / r2) / r2) / sn;
            f[3 + k] = fb1 * d1[k] + fi;
            f[6 + k] = fb2 * d2[k] + fk;
            f[k] = -(fb1 * d1[k] + fb2 * d2[k]) - (fi + fk);
        }
        memcpy(p->f0 + 9 * m, f, sizeof(f));
    }
    p->evals[0]++;
}

static void kick(port_t *p, const double *fa, const double *fsub, double coef) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < p->n; i++)
        for (int k = 0; k < 3; k++) {
            const double ff = fsub ? fa[3 * i + k] - fsub[3 * i + k] : fa[3 * i + k];
            p->v[3 * i + k] += coef * ff / p->m[i];
        }
}

static void move(port_t *p, double coef) {
#pragma omp parallel for schedule(static)
    for (long k = 0; k < 3L * p->n; k++) p->x[k] += coef * p->v[k];
}

/* ---- C-callable surface (ctypes) ---- */
port_t *port_create(int n, const double *box, const double *x, const double *v, const double *m, const double *q, const double *sigma,
                    const double *eps, double bond_r0, double bond_k, double angle_t0, double angle_k, double rc0, double rs0, double rc,
                    double rs, double alpha, double Kc, int n0, int n1, double dt, double skin) {
    port_t *p = (port_t *)calloc(1, sizeof(port_t));
    p->n = n; p->n0 = n0; p->n1 = n1; p->dt = dt; p->skin = skin;
    memcpy(p->box, box, sizeof(p->box));
    p->rc = rc; p->rs = rs; p->alpha = alpha; p->rc0 = rc0; p->rs0 = rs0; p->Kc = Kc;
    p->bond_r0 = bond_r0; p->bond_k = bond_k; p->angle_t0 = angle_t0; p->angle_k = angle_k;
    double **arrs[] = {&p->x, &p->v, &p->f0, &p->f1, &p->f2, &p->xref};
    for (unsigned a = 0; a < sizeof(arrs) / sizeof(arrs[0]); a++) *arrs[a] = (double *)malloc(sizeof(double) * 3 * n);
    p->m = (double *)malloc(sizeof(double) * n); p->q = (double *)malloc(sizeof(double) * n);
    p->hsig = (double *)malloc(sizeof(double) * n); p->seps2 = (double *)malloc(sizeof(double) * n);
    p->order = (int *)malloc(sizeof(int) * n); p->nfront = (int *)malloc(sizeof(int) * n);
    p->nbr_ptr = (long *)malloc(sizeof(long) * (n + 1));
#pragma omp parallel for schedule(static)           /* first touch by the threads that will own these atoms */
    for (int i = 0; i < n; i++) {
        for (int k = 0; k < 3; k++) {
            p->x[3 * i + k] = x[3 * i + k]; p->v[3 * i + k] = v[3 * i + k];
            p->f0[3 * i + k] = p->f1[3 * i + k] = p->f2[3 * i + k] = 0.0; p->xref[3 * i + k] = x[3 * i + k];
        }
        p->m[i] = m[i]; p->q[i] = q[i]; p->hsig[i] = 0.5 * sigma[i]; p->seps2[i] = 2.0 * sqrt(eps[i]);
        p->order[i] = i; p->nfront[i] = 0; p->nbr_ptr[i] = 0;
    }
    p->nbr_ptr[n] = 0;
    return p;
}

/* nsteps outer steps of RespaPropagator([n0, n1, 1]) with one force cache per group */
void port_step(port_t *p, int nsteps) {
    const double dt = p->dt;
    const int n0 = p->n0, n1 = p->n1;
    if (p->evals[2] == 0) { bonded(p); pair_pass(p, 2); }              /* forces at the starting positions */
    for (int st = 0; st < nsteps; st++) {
        kick(p, p->f2, p->f1, 0.5 * dt);
        for (int a = 0; a < n1; a++) {
            kick(p, p->f1, NULL, 0.5 * dt / n1);
            for (int b = 0; b < n0; b++) {
                kick(p, p->f0, NULL, 0.5 * dt / (n0 * n1));
                move(p, dt / (n0 * n1));
                bonded(p);
                kick(p, p->f0, NULL, 0.5 * dt / (n0 * n1));
            }
            pair_pass(p, a == n1 - 1 ? 2 : 1);       /* the last near evaluation of a step rides on the outer force's traversal */
            kick(p, p->f1, NULL, 0.5 * dt / n1);
        }
        kick(p, p->f2, p->f1, 0.5 * dt);
    }
}

void port_get(const port_t *p, double *x, double *v, double *f0, double *f1, double *f2, long *stats) {
    const size_t b = sizeof(double) * 3 * p->n;
    if (x) memcpy(x, p->x, b);
    if (v) memcpy(v, p->v, b);
    if (f0) memcpy(f0, p->f0, b);
    if (f1) memcpy(f1, p->f1, b);
    if (f2) memcpy(f2, p->f2, b);
    if (stats) { stats[0] = p->builds; stats[1] = p->evals[0]; stats[2] = p->evals[1]; stats[3] = p->evals[2]; stats[4] = p->nbr_ptr[p->n]; }
}

int port_threads(void) { return omp_get_max_threads(); }
void port_set_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }

void port_destroy(port_t *p) {
    double *d[] = {p->x, p->v, p->m, p->q, p->hsig, p->seps2, p->f0, p->f1, p->f2, p->xref};
    for (unsigned a = 0; a < sizeof(d) / sizeof(d[0]); a++) free(d[a]);
    free(p->order); free(p->nbr_idx); free(p->nbr_ptr); free(p->nfront); free(p->scratch);
    free(p);
}
