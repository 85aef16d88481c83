// atomsmm_amd/csrc/constraints.hip -- distance constraints for the constrained move / boost of the reference's
// propagators (propagators.py:246-252 `addConstrainPositions`, :272-273 `addConstrainVelocities`, :1126-1133 velocity
// Verlet with constraints): SHAKE on positions, RATTLE on velocities (Ryckaert, Ciccotti, Berendsen, J. Comput. Phys.
// 23, 327; Andersen, J. Comput. Phys. 52, 24).
//
// OpenMM semantics restated [recalled]: ConstrainPositions moves the current positions onto the constraint surface
// along the bond vectors of the REFERENCE positions -- those at the start of the step or right after the previous
// ConstrainPositions -- and then makes the constrained positions the new reference; ConstrainVelocities removes the
// velocity components along the constrained bonds.  Both iterate to a relative tolerance (default 1e-5).
//
// MI355X mapping: constraints only couple atoms of one small cluster (a rigid water: 3 atoms / 3 constraints; X-H
// groups: up to 4 atoms): ONE THREAD solves one cluster entirely in private memory -- no inter-thread traffic, no
// atomics, deterministic.  Positions are not wrapped (a molecule never straddles the box in the engine's arrays).
#include "amm_ctx.h"

#define AMM_CLUSTER_ATOMS 8
#define AMM_CLUSTER_CONS 16

struct ConstraintSet {
    int ncluster = 0;
    double tol = 1e-5;
    int *d_cptr = nullptr;       // [ncluster+1] constraints of each cluster
    int *d_aptr = nullptr;       // [ncluster+1] atoms of each cluster
    int *d_atoms = nullptr;      // atom indices, cluster by cluster
    int2 *d_pair = nullptr;      // constraint -> (local i, local j) within its cluster
    double *d_dist = nullptr;    // constraint -> distance
    double *d_xref = nullptr;    // [n][3] reference positions (see above)
    int *d_fail = nullptr;       // set when a cluster does not converge
};

struct ConsArgs {
    int ncluster;
    const int *cptr, *aptr, *atoms;
    const int2 *pair;
    const double *dist;
    const double *mass;
    double tol;
    int *fail;
};

__global__ void __launch_bounds__(128) k_shake(ConsArgs A, double *x, double *xref) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= A.ncluster) return;
    const int a0 = A.aptr[c], na = A.aptr[c + 1] - a0, c0 = A.cptr[c], nc = A.cptr[c + 1] - c0;
    double p[AMM_CLUSTER_ATOMS][3], r[AMM_CLUSTER_ATOMS][3], im[AMM_CLUSTER_ATOMS];
    for (int k = 0; k < na; ++k) {
        const int i = A.atoms[a0 + k];
        im[k] = 1.0 / A.mass[i];
        for (int j = 0; j < 3; ++j) {
            p[k][j] = x[3 * i + j];
            r[k][j] = xref[3 * i + j];
        }
    }
    const double lower = 1.0 - 2.0 * A.tol + A.tol * A.tol, upper = 1.0 + 2.0 * A.tol + A.tol * A.tol;
    bool done = false;
    for (int it = 0; it < 500 && !done; ++it) {
        done = true;
        for (int q = 0; q < nc; ++q) {
            const int2 ij = A.pair[c0 + q];
            const double d2 = A.dist[c0 + q] * A.dist[c0 + q];
            double dp[3], dr[3], pp = 0.0, rp = 0.0;
            for (int j = 0; j < 3; ++j) {
                dp[j] = p[ij.x][j] - p[ij.y][j];
                dr[j] = r[ij.x][j] - r[ij.y][j];
                pp += dp[j] * dp[j];
                rp += dr[j] * dp[j];
            }
            if (pp < lower * d2 || pp > upper * d2) {
                done = false;
                const double g = (d2 - pp) / (2.0 * (im[ij.x] + im[ij.y]) * rp);
                for (int j = 0; j < 3; ++j) {
                    p[ij.x][j] += g * im[ij.x] * dr[j];
                    p[ij.y][j] -= g * im[ij.y] * dr[j];
                }
            }
        }
    }
    if (!done) *A.fail = 1;
    for (int k = 0; k < na; ++k) {
        const int i = A.atoms[a0 + k];
        for (int j = 0; j < 3; ++j) {
            x[3 * i + j] = p[k][j];
            xref[3 * i + j] = p[k][j];       // the constrained positions are the next reference
        }
    }
}

__global__ void __launch_bounds__(128) k_rattle(ConsArgs A, const double *x, double *v) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= A.ncluster) return;
    const int a0 = A.aptr[c], na = A.aptr[c + 1] - a0, c0 = A.cptr[c], nc = A.cptr[c + 1] - c0;
    double p[AMM_CLUSTER_ATOMS][3], w[AMM_CLUSTER_ATOMS][3], im[AMM_CLUSTER_ATOMS];
    for (int k = 0; k < na; ++k) {
        const int i = A.atoms[a0 + k];
        im[k] = 1.0 / A.mass[i];
        for (int j = 0; j < 3; ++j) {
            p[k][j] = x[3 * i + j];
            w[k][j] = v[3 * i + j];
        }
    }
    bool done = false;
    for (int it = 0; it < 500 && !done; ++it) {
        done = true;
        for (int q = 0; q < nc; ++q) {
            const int2 ij = A.pair[c0 + q];
            double dp[3], dot = 0.0, pp = 0.0;
            for (int j = 0; j < 3; ++j) {
                dp[j] = p[ij.x][j] - p[ij.y][j];
                dot += dp[j] * (w[ij.x][j] - w[ij.y][j]);
                pp += dp[j] * dp[j];
            }
            // relative rate of change of the bond length, d ln|r| / dt, against the tolerance (1/ps)
            if (fabs(dot) > A.tol * pp) {
                done = false;
                const double g = -dot / ((im[ij.x] + im[ij.y]) * pp);
                for (int j = 0; j < 3; ++j) {
                    w[ij.x][j] += g * im[ij.x] * dp[j];
                    w[ij.y][j] -= g * im[ij.y] * dp[j];
                }
            }
        }
    }
    if (!done) *A.fail = 1;
    for (int k = 0; k < na; ++k) {
        const int i = A.atoms[a0 + k];
        for (int j = 0; j < 3; ++j) v[3 * i + j] = w[k][j];
    }
}

template <class T>
static int cons_upload(T **dptr, const std::vector<T> &h) {
    AMM_HIP(hipMalloc(dptr, sizeof(T) * std::max<size_t>(h.size(), 1)));
    if (!h.empty()) AMM_HIP(hipMemcpy(*dptr, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice));
    return 0;
}

int amm_constraints_create_impl(amm_ctx *ctx, const int32_t *h_pairs, const double *h_dist, int n_cons, double tol,
                                ConstraintSet **out) {
    const int n = ctx->n;
    std::vector<int> parent(n);
    for (int i = 0; i < n; ++i) parent[i] = i;
    auto find = [&](int a) {
        while (parent[a] != a) {
            parent[a] = parent[parent[a]];
            a = parent[a];
        }
        return a;
    };
    for (int q = 0; q < n_cons; ++q) {
        const int i = h_pairs[2 * q], j = h_pairs[2 * q + 1];
        if (i < 0 || j < 0 || i >= n || j >= n || i == j || !(h_dist[q] > 0.0)) {
            amm_set_error("amm_constraints_create: bad constraint (atom index or distance)");
            return 1;
        }
        const int a = find(i), b = find(j);
        if (a != b) parent[std::max(a, b)] = std::min(a, b);
    }
    // clusters = components with at least one constraint, numbered by their smallest atom
    std::vector<char> has_cons(n, 0);
    for (int q = 0; q < n_cons; ++q) has_cons[find(h_pairs[2 * q])] = 1;
    std::vector<int> cluster_of(n, -1), cluster_id(n, -1);
    std::vector<std::vector<int>> atoms;
    int ncl = 0;
    for (int i = 0; i < n; ++i) {
        const int r = find(i);
        if (!has_cons[r]) continue;
        if (cluster_id[r] < 0) {
            cluster_id[r] = ncl++;
            atoms.emplace_back();
        }
        cluster_of[i] = cluster_id[r];
        atoms[cluster_id[r]].push_back(i);
    }
    std::vector<std::vector<int>> cons(ncl);
    for (int q = 0; q < n_cons; ++q) cons[cluster_of[h_pairs[2 * q]]].push_back(q);
    std::vector<int> cptr(ncl + 1, 0), aptr(ncl + 1, 0), flat_atoms;
    std::vector<int2> pair;
    std::vector<double> dist;
    for (int c = 0; c < ncl; ++c) {
        if ((int)atoms[c].size() > AMM_CLUSTER_ATOMS || (int)cons[c].size() > AMM_CLUSTER_CONS) {
            amm_set_error("amm_constraints_create: a constraint cluster exceeds 8 atoms / 16 constraints");
            return 1;
        }
        for (int a : atoms[c]) flat_atoms.push_back(a);
        for (int q : cons[c]) {
            int li = -1, lj = -1;
            for (int k = 0; k < (int)atoms[c].size(); ++k) {
                if (atoms[c][k] == h_pairs[2 * q]) li = k;
                if (atoms[c][k] == h_pairs[2 * q + 1]) lj = k;
            }
            pair.push_back(make_int2(li, lj));
            dist.push_back(h_dist[q]);
        }
        aptr[c + 1] = (int)flat_atoms.size();
        cptr[c + 1] = (int)pair.size();
    }
    ConstraintSet *cs = new ConstraintSet();
    cs->ncluster = ncl;
    cs->tol = tol > 0 ? tol : 1e-5;
    if (cons_upload(&cs->d_cptr, cptr) || cons_upload(&cs->d_aptr, aptr) || cons_upload(&cs->d_atoms, flat_atoms) ||
        cons_upload(&cs->d_pair, pair) || cons_upload(&cs->d_dist, dist))
        return 1;
    AMM_HIP(hipMalloc(&cs->d_xref, sizeof(double) * 3 * (size_t)n));
    AMM_HIP(hipMalloc(&cs->d_fail, sizeof(int)));
    AMM_HIP(hipMemset(cs->d_fail, 0, sizeof(int)));
    *out = cs;
    return 0;
}

static ConsArgs cons_args(amm_ctx *ctx, ConstraintSet *cs) {
    ConsArgs A;
    A.ncluster = cs->ncluster;
    A.cptr = cs->d_cptr;
    A.aptr = cs->d_aptr;
    A.atoms = cs->d_atoms;
    A.pair = cs->d_pair;
    A.dist = cs->d_dist;
    A.mass = ctx->d_mass;
    A.tol = cs->tol;
    A.fail = cs->d_fail;
    return A;
}

int amm_constraints_save_reference(amm_ctx *ctx, ConstraintSet *cs, const double *d_x) {
    AMM_HIP(hipMemcpyAsync(cs->d_xref, d_x, sizeof(double) * 3 * (size_t)ctx->n, hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
}

int amm_constrain_positions(amm_ctx *ctx, ConstraintSet *cs, double *d_x) {
    if (cs->ncluster == 0) return 0;
    hipLaunchKernelGGL(k_shake, dim3((cs->ncluster + 127) / 128), dim3(128), 0, ctx->stream, cons_args(ctx, cs), d_x, cs->d_xref);
    AMM_HIP(hipGetLastError());
    return 0;
}

int amm_constrain_velocities(amm_ctx *ctx, ConstraintSet *cs, const double *d_x, double *d_v) {
    if (cs->ncluster == 0) return 0;
    hipLaunchKernelGGL(k_rattle, dim3((cs->ncluster + 127) / 128), dim3(128), 0, ctx->stream, cons_args(ctx, cs), d_x, d_v);
    AMM_HIP(hipGetLastError());
    return 0;
}

int amm_constraints_failed(amm_ctx *ctx, ConstraintSet *cs) {
    int fail = 0;
    if (hipMemcpy(&fail, cs->d_fail, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    if (fail) (void)hipMemset(cs->d_fail, 0, sizeof(int));
    (void)ctx;
    return fail;
}

int amm_constraints_free(ConstraintSet *cs) {
    void *ptrs[] = {cs->d_cptr, cs->d_aptr, cs->d_atoms, cs->d_pair, cs->d_dist, cs->d_xref, cs->d_fail};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    delete cs;
    return 0;
}
