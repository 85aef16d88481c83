// atomsmm_amd/csrc/pair.hip -- cell list, Verlet neighbour list and pair-force traversal (gfx950, fp64).
//
// Takes over what OpenMM does for the reference's CustomNonbondedForce / NonbondedForce objects
// (forces.py:225, 260-313, 655-670, 710-724; systems.py:97-111): neighbour search + per-pair
// evaluation of the energy expression and of its derivative.
//
// Data layout in HBM (per pair force):
//   posq_s[s] = (x,y,z wrapped into the box, q)   double4, cell-sorted order  (32 B / atom)
//   lj_s[s]   = (sigma/2, 2 sqrt(eps))            double2, cell-sorted order  (16 B / atom)
//   nl[a*cap + k]  int32 sorted-slot index of the k-th neighbour of slice atom a (full list: both
//                  directions -> owner-computes, no atomics, deterministic summation order)
//   perm[s]   original atom index of sorted slot s; forces are written to f[perm[s]].
// Work decomposition: `lpa` lanes of a 64-wide wavefront share one i-atom and stride through its
// neighbour row (coalesced 4*lpa-byte reads); partial forces are combined with wavefront shuffles.
#include <cmath>
#include <cstdio>

#include "amm_ctx.h"
#include "pair_math.h"

// ------------------------------------------------------------------------------------------------
// host: constants
int amm_pair_build_consts(const amm_pair_desc &d, PairConsts &pc) {
    pc.family = d.family;
    pc.flags = d.flags;
    pc.degree = d.degree < 1 ? 1 : d.degree;
    pc.cmode = (d.flags & AMM_COULOMB_EWALD) ? 1 : ((d.flags & AMM_COULOMB_RF) ? 2 : 0);
    pc.sign = d.sign;
    pc.rc = d.rc;
    pc.rc2 = d.rc * d.rc;
    pc.rc0 = d.rc0;
    pc.rs0 = d.rs0;
    pc.inv_dr0 = (d.rc0 > d.rs0) ? 1.0 / (d.rc0 - d.rs0) : 0.0;
    pc.rswitch = d.rswitch;
    pc.alpha = d.alpha;
    pc.two_alpha_over_sqrtpi = d.alpha * 1.1283791670955125739;
    pc.Kc = d.Kc;
    pc.krf = d.krf;
    pc.crf = d.crf;
    pc.inv_rc0 = d.rc0 > 0 ? 1.0 / d.rc0 : 0.0;
    pc.inv_rc0_2 = pc.inv_rc0 * pc.inv_rc0;
    pc.b = pc.f12c = pc.f6c = pc.f1c = 0.0;
    if (d.family == AMM_NEAR_FSWITCH) {
        if (!(d.rs0 > 0.0 && d.rc0 > d.rs0)) {
            amm_set_error("force-switch needs 0 < rs0 < rc0");
            return 1;
        }
        // forces.py:559-563
        double b = d.rs0 / (d.rc0 - d.rs0);
        pc.b = b;
        pc.f12c = pow(1 + b, 3) * (pow(b, 6) + 3 * pow(b, 5) + (30.0 / 7) * pow(b, 4) + (25.0 / 7) * pow(b, 3) +
                                   (25.0 / 14) * b * b + 0.5 * b + 2.0 / 33) / pow(b, 9);
        pc.f6c = pow(1 + b, 3) / pow(b, 3);
        pc.f1c = (30 * (1 + b)) * (b * b * (1 + b) * (1 + b) * log(1 / b + 1) - b * b * b - 1.5 * b * b - b / 3 + 1.0 / 12);
    }
    pc.sw_den = 1.0;
    pc.inv_sw_dr = 0.0;
    if (d.family == AMM_DAMPED) pc.sw_den = pow(d.rc, pc.degree) - pow(d.rswitch, pc.degree);
    if (d.family == AMM_NONBONDED && (d.flags & AMM_SWITCH)) pc.inv_sw_dr = 1.0 / (d.rc - d.rswitch);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// K1: cell list
__device__ __forceinline__ double wrap1(double x, double L, double invL) {
    double w = x - L * floor(x * invL);
    if (w >= L) w -= L;
    if (w < 0.0) w = 0.0;
    return w;
}

__global__ void k_check_displacement(int n, const double *__restrict__ pos, const double *__restrict__ xref,
                                     double thr2, int *flags) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double dx = pos[3 * i] - xref[3 * i], dy = pos[3 * i + 1] - xref[3 * i + 1], dz = pos[3 * i + 2] - xref[3 * i + 2];
    double d2 = dx * dx + dy * dy + dz * dz;
    if (!(d2 <= thr2)) flags[0] = 1;   // benign race: every writer stores 1 (NaN also triggers)
}

__global__ void k_cell_assign(int n, const double *__restrict__ pos, Box box, CellGrid g, int *cell_of, int *count,
                              double *xref, const int *flags, int force) {
    if (!force && !flags[0]) return;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double x = pos[3 * i + k];
        xref[3 * i + k] = x;
        double w = wrap1(x, box.L[k], box.invL[k]);
        int ck = (int)(w * g.inv_cw[k]);
        c[k] = ck >= g.nc[k] ? g.nc[k] - 1 : ck;
    }
    int cell = (c[2] * g.nc[1] + c[1]) * g.nc[0] + c[0];
    cell_of[i] = cell;
    atomicAdd(&count[cell], 1);
}

// single block: exclusive scan of count[0..ncell) -> start[0..ncell], fill <- start, count <- 0
__global__ void k_cell_scan(int ncell, int *count, int *start, int *fill, const int *flags, int force) {
    if (!force && !flags[0]) return;
    __shared__ int part[1024];
    __shared__ int carry;
    int t = threadIdx.x;
    if (t == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < ncell; base += 1024) {
        int idx = base + t;
        int v = idx < ncell ? count[idx] : 0;
        part[t] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            int add = t >= off ? part[t - off] : 0;
            __syncthreads();
            part[t] += add;
            __syncthreads();
        }
        int excl = part[t] - v + carry;
        if (idx < ncell) {
            start[idx] = excl;
            fill[idx] = excl;
            count[idx] = 0;
        }
        __syncthreads();
        if (t == 1023) carry += part[1023];
        __syncthreads();
    }
    if (t == 0) start[ncell] = carry;
}

__global__ void k_cell_fill(int n, const int *__restrict__ cell_of, int *fill, int *perm_tmp, const int *flags, int force) {
    if (!force && !flags[0]) return;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int slot = atomicAdd(&fill[cell_of[i]], 1);
    perm_tmp[slot] = i;
}

// one wavefront per cell: rank sort by atom index -> deterministic order whatever the atomics did
__global__ void k_cell_sort(int ncell, const int *__restrict__ start, const int *__restrict__ perm_tmp, int *perm,
                            const int *flags, int force) {
    if (!force && !flags[0]) return;
    int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    int lane = threadIdx.x & 63;
    if (wave >= ncell) return;
    int b = start[wave], e = start[wave + 1];
    for (int a = b + lane; a < e; a += 64) {
        int me = perm_tmp[a];
        int rank = 0;
        for (int k = b; k < e; ++k) rank += perm_tmp[k] < me;
        perm[b + rank] = me;
    }
}

// sorted copies of positions (wrapped) and parameters; run before every evaluation
__global__ void k_gather_sorted(int n, const int *__restrict__ perm, const double *__restrict__ pos,
                                const double *__restrict__ q, const double *__restrict__ hsig,
                                const double *__restrict__ seps2, Box box, double4 *posq_s, double2 *lj_s) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    int i = perm[s];
    double4 p;
    p.x = wrap1(pos[3 * i], box.L[0], box.invL[0]);
    p.y = wrap1(pos[3 * i + 1], box.L[1], box.invL[1]);
    p.z = wrap1(pos[3 * i + 2], box.L[2], box.invL[2]);
    p.w = q[i];
    posq_s[s] = p;
    lj_s[s] = make_double2(hsig[i], seps2[i]);
}

// ------------------------------------------------------------------------------------------------
// neighbour-list build: `lpb` lanes per i-atom sweep the candidate cells; ordered (ballot) compaction
template <bool COUNT_ONLY>
__global__ void k_build_nlist(int s_begin, int s_end, int lpb_shift, const int *__restrict__ perm,
                              const int *__restrict__ cell_of, const int *__restrict__ cell_start,
                              const double *__restrict__ pos, Box box, CellGrid g, double rlist2,
                              const int *__restrict__ excl_ptr, const int *__restrict__ excl_idx, int cap, int *nl,
                              int *nnb, int *flags, unsigned long long *counters, int force) {
    if (!force && !flags[0]) return;
    const int lpb = 1 << lpb_shift;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int a = tid >> lpb_shift;
    const int sub = tid & (lpb - 1);
    const int lane = threadIdx.x & 63;
    const int gbase = lane & ~(lpb - 1);              // first lane of my group inside the wavefront
    const unsigned long long gmask = (lpb == 64 ? ~0ull : ((1ull << lpb) - 1ull)) << gbase;
    const int s = s_begin + a;
    const bool valid = s < s_end;
    int i = 0, ci = 0;
    double xi = 0, yi = 0, zi = 0;
    if (valid) {
        i = perm[s];
        ci = cell_of[i];
        xi = wrap1(pos[3 * i], box.L[0], box.invL[0]);
        yi = wrap1(pos[3 * i + 1], box.L[1], box.invL[1]);
        zi = wrap1(pos[3 * i + 2], box.L[2], box.invL[2]);
    }
    const int cx = ci % g.nc[0], cy = (ci / g.nc[0]) % g.nc[1], cz = ci / (g.nc[0] * g.nc[1]);
    const int eb = valid ? excl_ptr[i] : 0, ee = valid ? excl_ptr[i + 1] : 0;
    int count = 0;
    // wave-uniform trip counts: every lane walks the same stencil; ranges differ per group
    for (int oz = 0; oz < g.nstencil[2]; ++oz)
        for (int oy = 0; oy < g.nstencil[1]; ++oy)
            for (int ox = 0; ox < g.nstencil[0]; ++ox) {
                // offsets: 3 -> {-1,0,1}; 2 -> {0,1}; 1 -> {0}
                int dx = g.nstencil[0] == 3 ? ox - 1 : ox, dy = g.nstencil[1] == 3 ? oy - 1 : oy,
                    dz = g.nstencil[2] == 3 ? oz - 1 : oz;
                int nx = cx + dx, ny = cy + dy, nz = cz + dz;
                nx = nx < 0 ? nx + g.nc[0] : (nx >= g.nc[0] ? nx - g.nc[0] : nx);
                ny = ny < 0 ? ny + g.nc[1] : (ny >= g.nc[1] ? ny - g.nc[1] : ny);
                nz = nz < 0 ? nz + g.nc[2] : (nz >= g.nc[2] ? nz - g.nc[2] : nz);
                int c2 = (nz * g.nc[1] + ny) * g.nc[0] + nx;
                int jb = valid ? cell_start[c2] : 0, je = valid ? cell_start[c2 + 1] : 0;
                // all lanes of the wavefront must take part in every ballot: iterate to the wave-wide maximum
                int len = je - jb;
                int maxlen = len;
                for (int off = 32; off > 0; off >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off));
                for (int base = 0; base < maxlen; base += lpb) {
                    int js = jb + base + sub;
                    bool pass = false;
                    if (valid && base + sub < len) {
                        int j = perm[js];
                        double ddx = amm_min_image(xi - wrap1(pos[3 * j], box.L[0], box.invL[0]), box.L[0], box.invL[0]);
                        double ddy = amm_min_image(yi - wrap1(pos[3 * j + 1], box.L[1], box.invL[1]), box.L[1], box.invL[1]);
                        double ddz = amm_min_image(zi - wrap1(pos[3 * j + 2], box.L[2], box.invL[2]), box.L[2], box.invL[2]);
                        double r2 = ddx * ddx + ddy * ddy + ddz * ddz;
                        pass = (r2 < rlist2) && (j != i);
                        if (pass) {
                            for (int k = eb; k < ee; ++k)
                                if (excl_idx[k] == j) pass = false;
                        }
                    }
                    unsigned long long bal = __ballot(pass) & gmask;
                    if (pass) {
                        int pos_in = count + __popcll(bal & ((1ull << lane) - 1ull));
                        if (!COUNT_ONLY) {
                            if (pos_in < cap) nl[(size_t)a * cap + pos_in] = js;
                        }
                    }
                    count += __popcll(bal);
                }
            }
    if (valid && sub == 0) {
        if (!COUNT_ONLY) {
            nnb[a] = count < cap ? count : cap;
            if (count > cap) flags[1] = 1;
            atomicAdd(&counters[1], (unsigned long long)count);
        }
        atomicMax(&flags[2], count);
    }
}

__global__ void k_finish_build(int *flags, unsigned long long *counters, int force) {
    if (!force && !flags[0]) return;
    flags[0] = 0;
    counters[0] += 1;
}
__global__ void k_begin_build(int *flags, unsigned long long *counters, int force) {
    if (!force && !flags[0]) return;
    flags[2] = 0;
    counters[1] = 0;
}

// ------------------------------------------------------------------------------------------------
// K2/K3: pair traversal.  lpa lanes per i-atom; wavefront-shuffle reduction of the partial forces.
struct PairArgs {
    int s_begin, s_end, lpa_shift, cap;
    const int *perm;
    const int *nl;
    const int *nnb;
    const double4 *posq_s;
    const double2 *lj_s;
    double *force;     // original order [n][3]
    double *epart;     // per-block energy partials
    int accumulate;
    Box box;
};

template <int FAM, int CMODE, bool GUARD, bool EN>
__global__ void __launch_bounds__(256) k_pair_nlist(PairArgs A, PairConsts c) {
    const int lpa = 1 << A.lpa_shift;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int a = tid >> A.lpa_shift;
    const int sub = tid & (lpa - 1);
    const int s = A.s_begin + a;
    const bool valid = s < A.s_end;
    double fx = 0.0, fy = 0.0, fz = 0.0, esum = 0.0;
    if (valid) {
        const double4 pi = A.posq_s[s];
        const double2 li = A.lj_s[s];
        const double qi = c.Kc * pi.w;
        const int nn = A.nnb[a];
        const int *row = A.nl + (size_t)a * A.cap;
        for (int k = sub; k < nn; k += lpa) {
            const int js = row[k];
            const double4 pj = A.posq_s[js];
            const double2 lj = A.lj_s[js];
            double dx = amm_min_image(pi.x - pj.x, A.box.L[0], A.box.invL[0]);
            double dy = amm_min_image(pi.y - pj.y, A.box.L[1], A.box.invL[1]);
            double dz = amm_min_image(pi.z - pj.z, A.box.L[2], A.box.invL[2]);
            double r2 = dx * dx + dy * dy + dz * dz;
            if (r2 < c.rc2) {
                double e, fr;
                amm_pair_math<FAM, CMODE, GUARD, EN>(c, r2, qi * pj.w, li.x + lj.x, li.y * lj.y, e, fr);
                fx += fr * dx;
                fy += fr * dy;
                fz += fr * dz;
                if (EN) esum += e;
            }
        }
    }
    // combine the lpa partial sums of each atom (fixed butterfly order -> deterministic)
    for (int off = lpa >> 1; off > 0; off >>= 1) {
        fx += __shfl_xor(fx, off);
        fy += __shfl_xor(fy, off);
        fz += __shfl_xor(fz, off);
    }
    if (valid && sub == 0) {
        const int i = A.perm[s];
        if (A.accumulate) {
            A.force[3 * i] += fx;
            A.force[3 * i + 1] += fy;
            A.force[3 * i + 2] += fz;
        } else {
            A.force[3 * i] = fx;
            A.force[3 * i + 1] = fy;
            A.force[3 * i + 2] = fz;
        }
    }
    if (EN) {
        __shared__ double red[4];
        for (int off = 32; off > 0; off >>= 1) esum += __shfl_xor(esum, off);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = esum;
        __syncthreads();
        if (threadIdx.x == 0) A.epart[blockIdx.x] = 0.5 * (((red[0] + red[1]) + red[2]) + red[3]);
    }
}

template <int FAM, int CMODE>
static void launch_pair(dim3 grid, dim3 block, hipStream_t st, bool guard, bool en, const PairArgs &A, const PairConsts &c) {
    if (guard) {
        if (en) hipLaunchKernelGGL((k_pair_nlist<FAM, CMODE, true, true>), grid, block, 0, st, A, c);
        else hipLaunchKernelGGL((k_pair_nlist<FAM, CMODE, true, false>), grid, block, 0, st, A, c);
    } else {
        if (en) hipLaunchKernelGGL((k_pair_nlist<FAM, CMODE, false, true>), grid, block, 0, st, A, c);
        else hipLaunchKernelGGL((k_pair_nlist<FAM, CMODE, false, false>), grid, block, 0, st, A, c);
    }
}

// deterministic single-block reduction: *out += scale * sum(part[0..n))
__global__ void k_reduce_add(const double *__restrict__ part, int n, double scale, double *out) {
    __shared__ double sh[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += part[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out += scale * sh[0];
}

int amm_reduce_add(amm_ctx *ctx, const double *d_part, int n, double scale, double *d_out) {
    hipLaunchKernelGGL(k_reduce_add, dim3(1), dim3(256), 0, ctx->stream, d_part, n, scale, d_out);
    AMM_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// host orchestration
static int ilog2(int v) {
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}

static int setup_grid(amm_ctx *ctx, PairForce *pf) {
    CellGrid &g = pf->grid;
    g.ncell = 1;
    for (int k = 0; k < 3; ++k) {
        double L = ctx->box.L[k];
        if (pf->desc.rc > 0.5 * L * (1 + 1e-12)) {
            amm_set_error("pair cutoff exceeds half the box edge (minimum image needs rc <= L/2)");
            return 1;
        }
        int nc = (int)floor(L / pf->rlist);
        if (nc < 1) nc = 1;
        g.nc[k] = nc;
        g.nstencil[k] = nc >= 3 ? 3 : nc;
        g.cw[k] = L / nc;
        g.inv_cw[k] = nc / L;
        g.ncell *= nc;
    }
    return 0;
}

static int build_chain(amm_ctx *ctx, PairForce *pf, const double *d_pos, int force, bool count_only) {
    hipStream_t st = ctx->stream;
    const int n = pf->n;
    const int nb = (n + 255) / 256;
    hipLaunchKernelGGL(k_begin_build, dim3(1), dim3(1), 0, st, pf->d_flags, pf->d_counters, force);
    hipLaunchKernelGGL(k_cell_assign, dim3(nb), dim3(256), 0, st, n, d_pos, ctx->box, pf->grid, pf->d_cell_of,
                       pf->d_cell_count, pf->d_xref, pf->d_flags, force);
    hipLaunchKernelGGL(k_cell_scan, dim3(1), dim3(1024), 0, st, pf->grid.ncell, pf->d_cell_count, pf->d_cell_start,
                       pf->d_cell_fill, pf->d_flags, force);
    hipLaunchKernelGGL(k_cell_fill, dim3(nb), dim3(256), 0, st, n, pf->d_cell_of, pf->d_cell_fill, pf->d_perm_tmp,
                       pf->d_flags, force);
    hipLaunchKernelGGL(k_cell_sort, dim3((pf->grid.ncell * 64 + 255) / 256), dim3(256), 0, st, pf->grid.ncell,
                       pf->d_cell_start, pf->d_perm_tmp, pf->d_perm, pf->d_flags, force);
    const int nslice = pf->s_end - pf->s_begin;
    const int lpb_shift = ilog2(pf->lpb);
    const long threads = (long)nslice << lpb_shift;
    dim3 grid((unsigned)((threads + 255) / 256));
    if (count_only)
        hipLaunchKernelGGL((k_build_nlist<true>), grid, dim3(256), 0, st, pf->s_begin, pf->s_end, lpb_shift, pf->d_perm,
                           pf->d_cell_of, pf->d_cell_start, d_pos, ctx->box, pf->grid, pf->rlist * pf->rlist,
                           pf->d_excl_ptr, pf->d_excl_idx, pf->cap, pf->d_nl, pf->d_nnb, pf->d_flags, pf->d_counters, force);
    else
        hipLaunchKernelGGL((k_build_nlist<false>), grid, dim3(256), 0, st, pf->s_begin, pf->s_end, lpb_shift, pf->d_perm,
                           pf->d_cell_of, pf->d_cell_start, d_pos, ctx->box, pf->grid, pf->rlist * pf->rlist,
                           pf->d_excl_ptr, pf->d_excl_idx, pf->cap, pf->d_nl, pf->d_nnb, pf->d_flags, pf->d_counters, force);
    if (!count_only) hipLaunchKernelGGL(k_finish_build, dim3(1), dim3(1), 0, st, pf->d_flags, pf->d_counters, force);
    AMM_HIP(hipGetLastError());
    return 0;
}

static int first_build(amm_ctx *ctx, PairForce *pf, const double *d_pos) {
    // slice of the sorted order owned by this rank
    const int n = pf->n;
    const int per = (n + ctx->world - 1) / ctx->world;
    pf->s_begin = std::min(n, ctx->rank * per);
    pf->s_end = std::min(n, pf->s_begin + per);
    const int nslice = pf->s_end - pf->s_begin;
    AMM_HIP(hipMalloc(&pf->d_nnb, sizeof(int) * std::max(nslice, 1)));
    // lanes per atom: aim at >= 8 wavefronts per SIMD (1024 SIMDs) for latency hiding
    int lpa = 1;
    while (lpa < 64 && (long)nslice * lpa < 64L * 1024 * 8) lpa <<= 1;
    pf->lpa = lpa;
    pf->lpb = 16;
    // pass 1: count only -> capacity
    pf->cap = 0;
    if (build_chain(ctx, pf, d_pos, 1, true)) return 1;
    int flags[4];
    AMM_HIP(hipMemcpyAsync(flags, pf->d_flags, sizeof(flags), hipMemcpyDeviceToHost, ctx->stream));
    AMM_HIP(hipStreamSynchronize(ctx->stream));
    int maxnb = flags[2];
    pf->cap = ((int)(maxnb * 1.5) + 32 + 15) / 16 * 16;   // head-room for density fluctuations between rebuilds
    AMM_HIP(hipMalloc(&pf->d_nl, sizeof(int) * (size_t)std::max(nslice, 1) * pf->cap));
    if (build_chain(ctx, pf, d_pos, 1, false)) return 1;
    pf->built = true;
    return 0;
}

int amm_pair_eval_impl(amm_ctx *ctx, PairForce *pf, const double *d_pos, double *d_force, int accumulate,
                       double *d_energy) {
    hipStream_t st = ctx->stream;
    const int n = pf->n;
    const int nb = (n + 255) / 256;
    if (!pf->built) {
        if (first_build(ctx, pf, d_pos)) return 1;
    } else {
        const double thr = 0.5 * pf->skin;
        hipLaunchKernelGGL(k_check_displacement, dim3(nb), dim3(256), 0, st, n, d_pos, pf->d_xref, thr * thr, pf->d_flags);
        if (build_chain(ctx, pf, d_pos, 0, false)) return 1;
    }
    hipLaunchKernelGGL(k_gather_sorted, dim3(nb), dim3(256), 0, st, n, pf->d_perm, d_pos, pf->d_q, pf->d_hsig,
                       pf->d_seps2, ctx->box, pf->d_posq_s, pf->d_lj_s);
    const int nslice = pf->s_end - pf->s_begin;
    if (!accumulate && ctx->world > 1) AMM_HIP(hipMemsetAsync(d_force, 0, sizeof(double) * 3 * (size_t)n, st));
    if (nslice > 0) {
        PairArgs A;
        A.s_begin = pf->s_begin;
        A.s_end = pf->s_end;
        A.lpa_shift = ilog2(pf->lpa);
        A.cap = pf->cap;
        A.perm = pf->d_perm;
        A.nl = pf->d_nl;
        A.nnb = pf->d_nnb;
        A.posq_s = pf->d_posq_s;
        A.lj_s = pf->d_lj_s;
        A.force = d_force;
        A.accumulate = accumulate;
        A.box = ctx->box;
        const long threads = (long)nslice << A.lpa_shift;
        const int nblk = (int)((threads + 255) / 256);
        const bool en = d_energy != nullptr;
        if (en && nblk > pf->n_epart) {
            if (pf->d_epart) AMM_HIP(hipFree(pf->d_epart));
            AMM_HIP(hipMalloc(&pf->d_epart, sizeof(double) * nblk));
            pf->n_epart = nblk;
        }
        A.epart = pf->d_epart;
        const bool guard = (pf->desc.flags & AMM_GUARD_RC0) != 0;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (ctx->profile) {
            if (pf->ev_used + 2 > pf->ev.size()) {
                for (int k = 0; k < 64; ++k) {
                    hipEvent_t ev;
                    AMM_HIP(hipEventCreate(&ev));
                    pf->ev.push_back(ev);
                }
            }
            e0 = pf->ev[pf->ev_used++];
            e1 = pf->ev[pf->ev_used++];
            AMM_HIP(hipEventRecord(e0, st));
        }
        dim3 grid(nblk), block(256);
        switch (pf->desc.family) {
        case AMM_NEAR_NONE: launch_pair<AMM_NEAR_NONE, 0>(grid, block, st, guard, en, A, pf->pc); break;
        case AMM_NEAR_SHIFT: launch_pair<AMM_NEAR_SHIFT, 0>(grid, block, st, guard, en, A, pf->pc); break;
        case AMM_NEAR_FSWITCH: launch_pair<AMM_NEAR_FSWITCH, 0>(grid, block, st, guard, en, A, pf->pc); break;
        case AMM_DAMPED: launch_pair<AMM_DAMPED, 0>(grid, block, st, false, en, A, pf->pc); break;
        case AMM_NONBONDED:
            if (pf->pc.cmode == 1) launch_pair<AMM_NONBONDED, 1>(grid, block, st, false, en, A, pf->pc);
            else if (pf->pc.cmode == 2) launch_pair<AMM_NONBONDED, 2>(grid, block, st, false, en, A, pf->pc);
            else launch_pair<AMM_NONBONDED, 0>(grid, block, st, false, en, A, pf->pc);
            break;
        default: amm_set_error("unknown pair family"); return 1;
        }
        if (ctx->profile) AMM_HIP(hipEventRecord(e1, st));
        AMM_HIP(hipGetLastError());
        if (en) {
            if (amm_reduce_add(ctx, pf->d_epart, nblk, 1.0, d_energy)) return 1;
        }
    }
    pf->n_evals++;
    return 0;
}

int amm_pair_free(PairForce *pf) {
    void *ptrs[] = {pf->d_q, pf->d_hsig, pf->d_seps2, pf->d_excl_ptr, pf->d_excl_idx, pf->d_cell_of, pf->d_cell_count,
                    pf->d_cell_start, pf->d_cell_fill, pf->d_perm_tmp, pf->d_perm, pf->d_posq_s, pf->d_lj_s, pf->d_xref,
                    pf->d_nl, pf->d_nnb, pf->d_flags, pf->d_counters, pf->d_epart};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (hipEvent_t e : pf->ev) (void)hipEventDestroy(e);
    return 0;
}

// exported for abi.hip
int amm_pair_setup_grid(amm_ctx *ctx, PairForce *pf) { return setup_grid(ctx, pf); }
