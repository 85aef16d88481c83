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
#include <cstdlib>
#include <cstring>

#include "amm_ctx.h"
#include "bonded_terms.h"
#include "cluster.h"
#include "device_utils.h"
#include "pair_math.h"
#include "pair_tab.h"

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
    pc.inv_sw_den = 1.0 / pc.sw_den;
    pc.rswitch_d = pow(d.rswitch, pc.degree);
    if ((d.family == AMM_NONBONDED || d.family == AMM_SOFTCORE || d.family == AMM_LJ_VIRIAL) && (d.flags & AMM_SWITCH)) pc.inv_sw_dr = 1.0 / (d.rc - d.rswitch);
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

// Two Verlet buffers (dual list): flags[0] asks for a re-PRUNE of the inner list (some atom moved more than
// skin_in/2 since the last prune), flags[4] for a rebuild of the outer list (moved more than (skin_out-skin_in)/2
// since the last cell-based build; the outer list then no longer covers rc + skin_in).
__global__ void k_check_displacement(int n, const double *__restrict__ pos, const double *__restrict__ xref_in,
                                     const double *__restrict__ xref_out, double thr_in2, double thr_out2, int *flags) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = pos[3 * i], y = pos[3 * i + 1], z = pos[3 * i + 2];
    double dx = x - xref_in[3 * i], dy = y - xref_in[3 * i + 1], dz = z - xref_in[3 * i + 2];
    double d2 = dx * dx + dy * dy + dz * dz;
    if (!(d2 <= thr_in2)) flags[0] = 1;   // benign race: every writer stores 1 (NaN also triggers)
    dx = x - xref_out[3 * i]; dy = y - xref_out[3 * i + 1]; dz = z - xref_out[3 * i + 2];
    d2 = dx * dx + dy * dy + dz * dz;
    if (!(d2 <= thr_out2)) flags[4] = 1;
}

// cell index of every atom + per-cell counts; the atom's arrival rank in its cell places it in the cell's fixed-size
// member table (capc entries per cell, sized at the first build: no scatter pass that would need the scan first);
// the last block turns the counts into the exclusive scan start[0..ncell] (count <- 0) and records the fullest cell
__global__ void __launch_bounds__(256) k_cell_assign(int n, const double *__restrict__ pos, Box box, CellGrid g, int *cell_of,
                              int *count, int *start, int *members, int capc, double *xref, int *flags, int *ticket,
                              int which, int force, const int *__restrict__ cls, int *count_lj, int *start_lj) {
    if (!force && !flags[which]) return;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        int c[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double x = pos[3 * i + k];
            xref[3 * i + k] = x;
            double w = wrap1(x, box.L[k], box.invL[k]);
            int ck = (w == w) ? (int)(w * g.inv_cw[k]) : 0;      // NaN-safe: a blown-up trajectory must not index out of range
            c[k] = ck >= g.nc[k] ? g.nc[k] - 1 : (ck < 0 ? 0 : ck);
        }
        int cell = (c[2] * g.nc[1] + c[1]) * g.nc[0] + c[0];
        cell_of[i] = cell;
        // one counter per cell: atoms (low 16 bits; the member tables hold far fewer) | those with a Lennard-Jones site << 16 (they
        // come first in their cell and in row_order)
        const int rank = atomicAdd(&count[cell], cls[i] ? 1 : 0x10001) & 0xffff;
        if (members) {
            if (rank < capc) members[(size_t)cell * capc + rank] = i;
            else flags[7] = 1;            // reported by amm_check: the density grew beyond the member tables
        }
    }
    if (!amm_last_block(ticket)) return;
    const int fullest = amm_block_scan_packed(g.ncell, count, start, start_lj);
    if (threadIdx.x == 0) flags[6] = fullest;
}

// rebuild: one wavefront per cell ranks the cell's members by atom index -> deterministic order whatever the atomics
// did; the lane that places atom i at sorted slot s also writes the fp32 copy of its wrapped position (the list build
// only has to find a SUPERSET of the pairs within rlist -- the traversal re-tests r^2 < rc^2 in fp64 -- so it runs on
// the fp32 pipe with a margin), inv_perm[i] = s and, when asked (posq_s), the fp64 sorted copies of the evaluation
// that follows.  No rebuild: only those sorted copies, with the permutation in place (the k_gather_sorted of this
// evaluation) -- so a step without rebuild pays two conditional launches (this and k_cell_assign), not four.
__global__ void __launch_bounds__(256) k_cell_sort_gather(int ncell, int n, const int *__restrict__ start,
                                                          const int *__restrict__ members, int capc, int *perm,
                                                          const double *__restrict__ pos, Box box, float4 *pos4f_s,
                                                          int *inv_perm, const int *flags, int which, int force,
                                                          const double *__restrict__ q, const double *__restrict__ hsig,
                                                          const double *__restrict__ seps2, double4 *posq_s, double2 *lj_s,
                                                          const int *__restrict__ cls, const int *__restrict__ start_lj,
                                                          int *row_order, int s_begin, int s_end, int *n_lj_out, const float *__restrict__ member,
                                                          int *cell_sets, int filter_mode, int copies_current) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (!force && !flags[which]) {
        // (copies_current: the launch that moved the atoms wrote the copies of these positions already -- k_pair_tab's epilogue)
        if (!copies_current && posq_s && gid < n) {
            const int i = perm[gid];
            double4 p;
            p.x = wrap1(pos[3 * i], box.L[0], box.invL[0]);
            p.y = wrap1(pos[3 * i + 1], box.L[1], box.invL[1]);
            p.z = wrap1(pos[3 * i + 2], box.L[2], box.invL[2]);
            p.w = q[i];
            posq_s[gid] = p;
            lj_s[gid] = make_double2(hsig[i], seps2[i]);
        }
        return;
    }
    const int wave = gid >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= ncell) return;
    const int b = start[wave];
    const int cnt = min(start[wave + 1] - b, capc);
    const int *mem = members + (size_t)wave * capc;
    // traversal order of the rank's rows (row_order): the slice's atoms WITH a Lennard-Jones site first, in slot order,
    // then the others.  ga / gi = number of such atoms in the slots below; the slice's own offsets come from the cell
    // that holds s_begin / s_end (binary search in the cell starts: only ranks of a multi-GPU run have s_begin > 0)
    const int lj_before = start_lj[wave], lj_here = start_lj[wave + 1] - lj_before;
    int ga0 = 0, ga1 = start_lj[ncell];
    if (row_order && (s_begin > 0 || s_end < n)) {
        int bounds[2] = {s_begin, s_end};
        int ga[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            int lo = 0, hi = ncell;                    // cell c with start[c] <= slot < start[c + 1] (slot == n: beyond the last)
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (start[mid] <= bounds[e]) lo = mid;
                else hi = mid;
            }
            const int o = bounds[e] - start[lo];
            ga[e] = bounds[e] >= n ? start_lj[ncell] : start_lj[lo] + min(o, start_lj[lo + 1] - start_lj[lo]);
        }
        ga0 = ga[0];
        ga1 = ga[1];
    }
    const int gi0 = s_begin - ga0;
    if (row_order && wave == 0 && lane == 0) n_lj_out[0] = ga1 - ga0;
    if (wave == 0 && lane == 0) n_lj_out[5] = n_lj_out[6] = n_lj_out[7] = 0;      // flags[8..10]: rows with entries of a filtered list (all / long / short), counted by the build that follows
    int sets_here = 0;                                // interaction-group force: which of the two sets have atoms in this cell
    for (int a0 = 0; a0 < cnt; a0 += 64) {
        const int a = a0 + lane;
        const int me = a < cnt ? mem[a] : 0;
        if (cell_sets) {
            const float code = a < cnt ? member[me] : 0.f;
            // (filter_mode 2, the rest part of a hybrid list: set 1 = the rest atoms (code 2), set 2 = every atom -- a rest atom is
            // in both, and the build's test "home has set 1 and the stencil set 2, or the other way round" becomes "the stencil holds a rest atom")
            sets_here |= (__builtin_amdgcn_ballot_w64(code == 1.0f) != 0ull ? 1 : 0) |
                         (__builtin_amdgcn_ballot_w64(code == 2.0f) != 0ull ? (filter_mode == 2 ? 3 : 2) : 0);
        }
        // sort key: class (atoms with a Lennard-Jones site first), then atom index -> deterministic whatever the atomics did
        const int key = a < cnt ? (me | (cls[me] << 30)) : 0x7fffffff;
        int rank = 0;
        for (int k0 = 0; k0 < cnt; k0 += 64) {
            const int kk = k0 + lane;
            int okey = 0x7fffffff;
            if (kk < cnt) {
                const int other = mem[kk];
                okey = other | (cls[other] << 30);
            }
            const int nk = min(64, cnt - k0);
            for (int jj = 0; jj < nk; ++jj) rank += __builtin_amdgcn_readlane(okey, jj) < key;
        }
        if (a >= cnt) continue;
        const int sl = b + rank;
        perm[sl] = me;
        double4 pd;
        pd.x = wrap1(pos[3 * me], box.L[0], box.invL[0]);
        pd.y = wrap1(pos[3 * me + 1], box.L[1], box.invL[1]);
        pd.z = wrap1(pos[3 * me + 2], box.L[2], box.invL[2]);
        float4 p;
        p.x = (float)pd.x;
        p.y = (float)pd.y;
        p.z = (float)pd.z;
        p.w = member ? member[me] : 0.f;       // set code of a filtered list (interaction group: the build keeps (1, 2) pairs only; hybrid rest part: pairs with a code-2 atom)
        pos4f_s[sl] = p;
        inv_perm[me] = sl;
        if (row_order && sl >= s_begin && sl < s_end) {
            const bool has_lj = rank < lj_here;
            const int ga = lj_before + (has_lj ? rank : lj_here);          // atoms with a site in the slots below sl
            row_order[has_lj ? ga - ga0 : (ga1 - ga0) + (sl - ga) - gi0] = sl;
        }
        if (posq_s) {
            pd.w = q[me];
            posq_s[sl] = pd;
            lj_s[sl] = make_double2(hsig[me], seps2[me]);
        }
    }
    if (cell_sets && lane == 0) cell_sets[wave] = sets_here;
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
// fp32 copy of the sorted, wrapped positions for a re-prune of the dual list (the cell sweep gets its copy
// from k_cell_sort)
__global__ void k_gather_f32(int n, const int *__restrict__ perm, const double *__restrict__ pos, Box box,
                             float4 *pos4f_s, int *inv_perm, double *xref_in, const int *flags, int which, int force) {
    if (!force && !flags[which]) return;
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    int i = perm[s];
    if (xref_in) {        // prune pass: these positions become the inner reference
        xref_in[3 * i] = pos[3 * i];
        xref_in[3 * i + 1] = pos[3 * i + 1];
        xref_in[3 * i + 2] = pos[3 * i + 2];
    }
    float4 p;
    p.x = (float)wrap1(pos[3 * i], box.L[0], box.invL[0]);
    p.y = (float)wrap1(pos[3 * i + 1], box.L[1], box.invL[1]);
    p.z = (float)wrap1(pos[3 * i + 2], box.L[2], box.invL[2]);
    p.w = pos4f_s[s].w;                        // the set code travels with the slot (same permutation)
    pos4f_s[s] = p;
    if (inv_perm) inv_perm[i] = s;
}

struct BoxF {
    float L[3], invL[3];
};

// one block: reduce the per-block list statistics of an outer build (which = 4) or of an inner build / prune
// (which = 0), re-arm the rebuild flag
__device__ void amm_finish_build_block(int *flags, unsigned long long *counters, const unsigned long long *blockstats,
                                       int nblocks, int count_only, int which) {
    __shared__ unsigned long long sh_sum[256];
    __shared__ unsigned long long sh_max[256];
    __shared__ unsigned long long sh_near[256];
    unsigned long long sum = 0, mx = 0, nr = 0;
    // the loads bypass the XCD's L2 (about a microsecond each): 8 blocks' worth in flight per thread, not one
    for (int b0 = threadIdx.x; b0 < nblocks; b0 += 8 * 256) {
        unsigned long long v[8][3];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int b = b0 + j * 256;
            const bool in = b < nblocks;
#pragma unroll
            for (int q = 0; q < 3; ++q) v[j][q] = in ? amm_ld_l2(&blockstats[3 * (in ? b : 0) + q]) : 0ull;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sum += v[j][0];
            mx = max(mx, v[j][1]);
            nr += v[j][2];
        }
    }
    sh_sum[threadIdx.x] = sum;
    sh_max[threadIdx.x] = mx;
    sh_near[threadIdx.x] = nr;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) {
            sh_sum[threadIdx.x] += sh_sum[threadIdx.x + off];
            sh_near[threadIdx.x] += sh_near[threadIdx.x + off];
            sh_max[threadIdx.x] = max(sh_max[threadIdx.x], sh_max[threadIdx.x + off]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (which == 4) {                  // outer list
            flags[5] = (int)sh_max[0];
            counters[3] = sh_sum[0];
            if (!count_only) {
                flags[4] = 0;
                flags[0] = 1;              // a fresh outer list must be pruned
                counters[4] += 1;
            }
        } else {                           // inner list
            flags[2] = (int)sh_max[0];
            counters[1] = sh_sum[0];
            counters[2] = sh_near[0];
            if (!count_only) {
                flags[0] = 0;
                counters[0] += 1;
            }
        }
    }
}

// neighbour-list build: one wavefront per (cell, part).  All i-atoms of a cell share the same candidates: the
// (2h+1)^2 (y,z) rows of the stencil, each one or two contiguous runs of the cell-sorted arrays (cells are
// x-fastest) with one periodic image per run.  The wave tabulates its runs once (start slot, exclusive prefix of
// the lengths, image shift; LDS, <= 50 runs) and then walks the CONCATENATED candidate stream in chunks of 128:
// lane l of a chunk finds its run by binary search of the prefix, loads the candidate (coalesced float4 within
// a run) and applies the image shift, so every chunk but the last is full whatever the run lengths are.  A chunk
// is tested against up to AMM_BATCH i-atoms of the cell, whose coordinates, exclusion slot ranges and running
// list lengths live in scalar registers (the t loop is unrolled).  Ordered ballot compaction keeps the list
// order -- and hence the force summation order -- deterministic.  Exclusions are looked up (in sorted-slot
// space, via inv_perm) only for the rare chunks that hold a slot inside the range spanned by the batch's atoms
// and their excluded partners.
#define AMM_BCHUNK 128
#define AMM_BATCH 8

template <bool COUNT_ONLY, bool RINT>
__global__ void __launch_bounds__(256) k_build_nlist(int s_begin, int s_end, int parts, const int *__restrict__ perm,
                              const int *__restrict__ inv_perm, const int *__restrict__ cell_start,
                              const float4 *__restrict__ pos4f_s, BoxF box, CellGrid g, float rlist2, float rnear2,
                              const int *__restrict__ excl_ptr, const int *__restrict__ excl_idx, int cap, int *nl,
                              int *nnb, int *nnb_near, int *flags, unsigned long long *blockstats,
                              unsigned long long *counters, int *ticket, int which, int force, int filtered, int *active, int active_cap, int active_size, const int *__restrict__ cell_sets,
                              const int *__restrict__ cell_start_lj, int *nnb_lj) {
    if (!force && !flags[which]) return;
    __shared__ int s_rstart[4][128];
    __shared__ int s_rpref[4][128];
    __shared__ float s_rshift[4][3][128];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    const int c = wave / parts, part = wave - c * parts;
    const unsigned long long below = (1ull << lane) - 1ull;
    const float FAR = 1.0e9f;
    unsigned long long wsum = 0, wnear = 0;
    int wmax = 0;
    int a_begin = 0, a_end = 0;
    if (c < g.ncell) {
        // this wave's share of the cell's atoms (even split over the parts), clipped to the rank's slice
        const int cb0 = __builtin_amdgcn_readfirstlane(cell_start[c]), cb1 = __builtin_amdgcn_readfirstlane(cell_start[c + 1]);
        const int per = (cb1 - cb0 + parts - 1) / parts;
        a_begin = max(cb0 + part * per, s_begin);
        a_end = min(min(cb0 + (part + 1) * per, cb1), s_end);
    }
    if (a_begin < a_end) {
        const int ncx = g.nc[0], ncy = g.nc[1], ncz = g.nc[2];
        const int cx = c % ncx, cy = (c / ncx) % ncy, cz = c / (ncx * ncy);
        // ---- cell table: a lane describes the stencil cells e = lane and lane + 64 (<= 125), e = (oz, oy, ox), x fastest ----
        // A cell's atoms are stored with the Lennard-Jones sites first (k_cell_sort_gather), so the candidates are walked as TWO
        // streams: the sites of all the stencil's cells, then the rest.  A row therefore comes out with the partners that have a
        // site first on either side (front: r < rnear; back: the others), and the traversal can keep the Lennard-Jones
        // arithmetic to the trips that need it -- in water one pair in nine.
        const int nsx = g.nstencil[0], nsy = g.nstencil[1], ne = nsx * nsy * g.nstencil[2];
        int ecs[2] = {0, 0}, elj[2] = {0, 0}, ecn[2] = {0, 0}, near_sets = 0;
        float esx[2] = {0.f, 0.f}, esy[2] = {0.f, 0.f}, esz[2] = {0.f, 0.f};
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int e = lane + 64 * hh;
            if (e < ne) {
                const int ox = e % nsx, oy = (e / nsx) % nsy, oz = e / (nsx * nsy);
                // fewer cells than the stencil is wide: visit every cell once (RINT build, minimum image in the test)
                int nz = ncz < 2 * g.h[2] + 1 ? oz : cz - g.h[2] + oz;
                esz[hh] = nz < 0 ? -box.L[2] : (nz >= ncz ? box.L[2] : 0.f);
                nz = nz < 0 ? nz + ncz : (nz >= ncz ? nz - ncz : nz);
                int ny = ncy < 2 * g.h[1] + 1 ? oy : cy - g.h[1] + oy;
                esy[hh] = ny < 0 ? -box.L[1] : (ny >= ncy ? box.L[1] : 0.f);
                ny = ny < 0 ? ny + ncy : (ny >= ncy ? ny - ncy : ny);
                int nx = ncx < 2 * g.h[0] + 1 ? ox : cx - g.h[0] + ox;
                esx[hh] = nx < 0 ? -box.L[0] : (nx >= ncx ? box.L[0] : 0.f);
                nx = nx < 0 ? nx + ncx : (nx >= ncx ? nx - ncx : nx);
                const int cc = (nz * ncy + ny) * ncx + nx;
                ecs[hh] = cell_start[cc];
                ecn[hh] = cell_start[cc + 1] - ecs[hh];
                elj[hh] = cell_start_lj[cc + 1] - cell_start_lj[cc];
                if (cell_sets) near_sets |= cell_sets[cc];
            }
        }
        if (cell_sets) {
            // interaction-group list: a cell whose atoms have no partner of the other set anywhere in the stencil gets empty rows
            // without walking the stream (a 30-atom solute in 249k atoms: 98 % of the wavefronts end here)
            const int home = cell_sets[c];
            const bool any1 = __builtin_amdgcn_ballot_w64((near_sets & 1) != 0) != 0ull, any2 = __builtin_amdgcn_ballot_w64((near_sets & 2) != 0) != 0ull;
            if (!(((home & 1) && any2) || ((home & 2) && any1))) {
                if (!COUNT_ONLY)
                    for (int a = a_begin + lane; a < a_end; a += 64) {
                        nnb[a - s_begin] = 0;
                        nnb_near[a - s_begin] = 0;
                        if (nnb_lj) nnb_lj[a - s_begin] = 0;
                    }
                a_end = a_begin;              // the batch loop below has nothing to do
            }
        }
        for (int tb = a_begin; tb < a_end; tb += AMM_BATCH) {
            const int nt = min(AMM_BATCH, a_end - tb);
            float4 my = make_float4(0.f, 0.f, 0.f, 0.f);
            int exlo = 0x7fffffff, exhi = -1;             // slot range of {self, excluded partners}
            int ex_p0 = 0, ex_n = 0;                      // lane t: first exclusion / number of exclusions of atom t
            if (lane < nt) {
                my = pos4f_s[tb + lane];
                const int i = perm[tb + lane];
                exlo = exhi = tb + lane;
                ex_p0 = excl_ptr[i];
                ex_n = excl_ptr[i + 1] - ex_p0;
                for (int k = ex_p0; k < ex_p0 + ex_n; ++k) {
                    const int es = inv_perm[excl_idx[k]];
                    exlo = min(exlo, es);
                    exhi = max(exhi, es);
                }
            }
            // excluded partners of the batch's atoms as sorted slots, in registers (exl[t], lane j = j-th partner of atom
            // t): the chunks that can hold one then need no memory at all -- four dependent loads per atom before, which a
            // wave that has the SIMD almost to itself (a rank's slice of a small box) waited out one by one
            int exl[AMM_BATCH];
#pragma unroll
            for (int t = 0; t < AMM_BATCH; ++t) {
                exl[t] = -1;
                const int p0 = __builtin_amdgcn_readlane(ex_p0, t), ne = __builtin_amdgcn_readlane(ex_n, t);
                if (lane < ne) exl[t] = inv_perm[excl_idx[p0 + lane]];
            }
            int blo = exlo, bhi = exhi;                   // batch-wide range
            for (int off = 32; off > 0; off >>= 1) {
                blo = min(blo, __shfl_xor(blo, off));
                bhi = max(bhi, __shfl_xor(bhi, off));
            }
            blo = __builtin_amdgcn_readfirstlane(blo);
            bhi = __builtin_amdgcn_readfirstlane(bhi);
            // per i-atom scalars: coordinates and the two running row lengths packed into one register
            // (front | back << 16; rows are shorter than 65536 -- checked on the host); the kernel lives on the
            // scalar-register budget, 8 atoms x 4 registers
            float px[AMM_BATCH], py[AMM_BATCH], pz[AMM_BATCH], pw[AMM_BATCH];
            int c2[AMM_BATCH], c2_sites[AMM_BATCH];
#pragma unroll
            for (int t = 0; t < AMM_BATCH; ++t) {
                pw[t] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my.w), t));
                px[t] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my.x), t));
                py[t] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my.y), t));
                pz[t] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my.z), t));
                c2[t] = 0;
                c2_sites[t] = 0;
            }
            // Row layout: entries with r < rnear fill the row from the front, the others from the back, so a
            // shorter-ranged force sharing this list walks only the front part.
            for (int phase = 0; phase < 2; ++phase) {
            // ---- candidate stream of this phase: the pieces (sites / the rest) of the stencil's cells, concatenated ----
            int total;
            {
                int len[2], inc[2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    len[hh] = phase ? ecn[hh] - elj[hh] : elj[hh];
                    inc[hh] = len[hh];
                    for (int off = 1; off < 64; off <<= 1) {
                        const int v = __shfl_up(inc[hh], off);
                        if (lane >= off) inc[hh] += v;
                    }
                }
                const int lower = __builtin_amdgcn_readlane(inc[0], 63);
                total = lower + __builtin_amdgcn_readlane(inc[1], 63);
                __builtin_amdgcn_wave_barrier();                 // the previous phase's look-ups are done
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    s_rstart[w][lane + 64 * hh] = ecs[hh] + (phase ? elj[hh] : 0);
                    s_rpref[w][lane + 64 * hh] = (hh ? lower : 0) + inc[hh] - len[hh];    // exclusive prefix; cells beyond the stencil: `total`
                    s_rshift[w][0][lane + 64 * hh] = esx[hh];
                    s_rshift[w][1][lane + 64 * hh] = esy[hh];
                    s_rshift[w][2][lane + 64 * hh] = esz[hh];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            for (int cb = 0; cb < total; cb += AMM_BCHUNK) {
                float4 cand[2];
                int js[2];
                bool inr = false;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int idx = cb + u * 64 + lane;
                    const bool in = idx < total;
                    int r = 0;                   // piece of the lane's candidate: the last one that starts at or below its stream index
#pragma unroll
                    for (int step = 64; step > 0; step >>= 1)
                        if (s_rpref[w][r + step] <= idx) r += step;
                    const int slot = in ? s_rstart[w][r] + idx - s_rpref[w][r] : 0;
                    float4 q = pos4f_s[slot];
                    if (!RINT) {             // image shift; lanes beyond the stream are parked far away
                        const float sx = s_rshift[w][0][r], sy = s_rshift[w][1][r], sz = s_rshift[w][2][r];
                        q.x = in ? q.x + sx : FAR;
                        q.y = in ? q.y + sy : FAR;
                        q.z = in ? q.z + sz : FAR;
                    }
                    cand[u] = q;
                    js[u] = in ? slot : -1;
                    inr = inr || (js[u] >= blo && js[u] <= bhi);
                }
                const bool special = __builtin_amdgcn_ballot_w64(inr) != 0ull;      // wave-uniform, rare
#pragma unroll
                for (int t = 0; t < AMM_BATCH; ++t) {
                    if (t >= nt) break;
                    int *row_out = nl + (size_t)(tb + t - s_begin) * cap;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        if (cb + u * 64 >= total) continue;                        // wave-uniform
                        float dx = px[t] - cand[u].x, dy = py[t] - cand[u].y, dz = pz[t] - cand[u].z;
                        if (RINT) {
                            dx -= box.L[0] * rintf(dx * box.invL[0]);
                            dy -= box.L[1] * rintf(dy * box.invL[1]);
                            dz -= box.L[2] * rintf(dz * box.invL[2]);
                        }
                        const float r2 = dx * dx + dy * dy + dz * dz;
                        // lane masks live in scalar registers: ballots of plain compares, combined with scalar logic
                        unsigned long long m_pass = __builtin_amdgcn_ballot_w64(r2 < rlist2);
                        if (RINT) m_pass &= __builtin_amdgcn_ballot_w64(js[u] >= 0);
                        if (filtered == 1) m_pass &= __builtin_amdgcn_ballot_w64(pw[t] * cand[u].w == 2.0f);   // (set 1, set 2) pairs only
                        else if (filtered == 2) m_pass &= __builtin_amdgcn_ballot_w64(pw[t] * cand[u].w >= 2.0f);   // codes 1 / 2: at least one rest atom
                        if (special) {       // wave-uniform branch, scalar loop over the atom's exclusions
                            const int st = tb + t;
                            unsigned long long m_excl = __builtin_amdgcn_ballot_w64(js[u] == st);
                            const int ne = __builtin_amdgcn_readlane(ex_n, t);
                            for (int k = 0; k < min(ne, 64); ++k)
                                m_excl |= __builtin_amdgcn_ballot_w64(js[u] == __builtin_amdgcn_readlane(exl[t], k));
                            if (ne > 64) {       // more partners than lanes: the rest from memory
                                const int p0 = __builtin_amdgcn_readlane(ex_p0, t);
                                for (int k = 64; k < ne; ++k)
                                    m_excl |= __builtin_amdgcn_ballot_w64(js[u] == inv_perm[excl_idx[p0 + k]]);
                            }
                            m_pass &= ~m_excl;
                        }
                        if (m_pass == 0ull) continue;        // wave-uniform: no candidate of this half chunk is in range of atom t
                        const unsigned long long m_near = m_pass & __builtin_amdgcn_ballot_w64(r2 < rnear2);
                        const int np_ = __popcll(m_pass), nn_ = __popcll(m_near);
                        const int cnt = c2[t] & 0xffff, cntf = (int)((unsigned)c2[t] >> 16);
                        if (!COUNT_ONLY) {
                            const int mp = __builtin_amdgcn_mbcnt_hi((unsigned)(m_pass >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m_pass, 0u));
                            const int mn = __builtin_amdgcn_mbcnt_hi((unsigned)(m_near >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m_near, 0u));
                            const int pos_near = cnt + mn, pos_far = (cap - 1 - cntf) - (mp - mn);
                            int pos_in;      // select by the scalar mask (the compiler would branch on a ternary)
                            asm volatile("v_cndmask_b32 %0, %1, %2, %3" : "=v"(pos_in) : "v"(pos_far), "v"(pos_near), "s"(m_near));
                            if (cnt + cntf + np_ <= cap) {                           // wave-uniform
                                if (__builtin_amdgcn_inverse_ballot_w64(m_pass)) row_out[pos_in] = js[u];
                            }
                        }
                        c2[t] += nn_ + ((np_ - nn_) << 16);
                    }
                }
            }
            if (phase == 0) {        // so far the row holds partners with a Lennard-Jones site only: remember how many, per side
#pragma unroll
                for (int t = 0; t < AMM_BATCH; ++t) c2_sites[t] = c2[t];
            }
            }      // phase
            int count = 0, countf = 0, sites = 0;
#pragma unroll
            for (int t = 0; t < AMM_BATCH; ++t) {
                count = (lane == t) ? (c2[t] & 0xffff) : count;
                countf = (lane == t) ? (int)((unsigned)c2[t] >> 16) : countf;
                sites = (lane == t) ? c2_sites[t] : sites;
            }
            if (lane < nt) {
                const int total_nb = count + countf;
                if (!COUNT_ONLY) {
                    const bool over = total_nb > cap;      // rows that overflow are flagged; amm_check() raises
                    nnb[tb + lane - s_begin] = over ? 0 : total_nb;
                    nnb_near[tb + lane - s_begin] = over ? 0 : count;
                    if (nnb_lj) nnb_lj[tb + lane - s_begin] = over ? 0 : sites;      // front | back << 16
                    if (over) flags[1] = 1;
                }
                wsum += (unsigned long long)total_nb;
                wnear += (unsigned long long)count;
                wmax = max(wmax, total_nb);
            }
            // filtered lists (interaction groups): most rows are empty -- the rows that hold entries are collected (any order:
            // a row has one producer whatever its place in the walk) so that the pair kernels visit only those; flags[8] counts
            // Two kinds of rows in the rest part of a hybrid list (filtered == 2): LONG ones (an atom outside the molecules against
            // everything around it: hundreds of entries) are filed from the front of `active`, SHORT ones (a molecule atom and its
            // few partners outside the molecules) from the back -- the traversal gives a long row a whole wavefront and packs eight
            // short ones into one (amm_active_row, k_pair_tab).  flags[9] / flags[10] count them; an interaction group's rows are all long.
            if (!COUNT_ONLY && active) {
                const bool holds = lane < nt && count + countf > 0 && count + countf <= cap;
                const bool is_long = holds && (filtered != 2 || my.w == 2.0f);
                const unsigned long long m_act = __builtin_amdgcn_ballot_w64(holds);
                if (m_act != 0ull) {
                    const unsigned long long m_long = __builtin_amdgcn_ballot_w64(is_long), m_short = m_act & ~m_long;
                    int base = 0, base_l = 0, base_s = 0;
                    if (lane == 0) {
                        base = atomicAdd(&flags[8], __popcll(m_act));
                        if (m_long != 0ull) base_l = atomicAdd(&flags[9], __popcll(m_long));
                        if (m_short != 0ull) base_s = atomicAdd(&flags[10], __popcll(m_short));
                    }
                    base = __builtin_amdgcn_readfirstlane(base);
                    base_l = __builtin_amdgcn_readfirstlane(base_l);
                    base_s = __builtin_amdgcn_readfirstlane(base_s);
                    const bool fits = base + __popcll(m_act) <= active_cap;
                    if (lane == 0 && !fits) flags[1] = 1;   // more rows than the pair kernels' grid covers
                    if (holds && fits) {
                        if (is_long) active[base_l + __popcll(m_long & below)] = tb + lane - s_begin;
                        else active[active_size - 1 - (base_s + __popcll(m_short & below))] = tb + lane - s_begin;
                    }
                }
            }
        }
    }
    // per-block (sum, max) of the list lengths -> blockstats; the last block reduces them.  (One same-address
    // atomic per atom serialises at L2: ~0.5 ms for 98k atoms -- measured -- so no atomics here.)
    for (int off = 32; off > 0; off >>= 1) {
        wsum += __shfl_xor(wsum, off);
        wnear += __shfl_xor(wnear, off);
        wmax = max(wmax, __shfl_xor(wmax, off));
    }
    __shared__ unsigned long long s_sum[4], s_near[4];
    __shared__ int s_max[4];
    if (lane == 0) {
        s_sum[threadIdx.x >> 6] = wsum;
        s_near[threadIdx.x >> 6] = wnear;
        s_max[threadIdx.x >> 6] = wmax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        amm_st_l2(&blockstats[3 * blockIdx.x], s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]);
        amm_st_l2(&blockstats[3 * blockIdx.x + 1], (unsigned long long)max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3])));
        amm_st_l2(&blockstats[3 * blockIdx.x + 2], s_near[0] + s_near[1] + s_near[2] + s_near[3]);
    }
    if (amm_last_block(ticket)) amm_finish_build_block(flags, counters, blockstats, (int)gridDim.x, COUNT_ONLY ? 1 : 0, which);
}

__global__ void k_finish_build(int *flags, unsigned long long *counters, const unsigned long long *blockstats,
                               int nblocks, int count_only, int which, int force) {
    if (!force && !flags[which]) return;
    amm_finish_build_block(flags, counters, blockstats, nblocks, count_only, which);
}

// Prune: inner list <- entries of the outer list within rlist_in of the CURRENT positions.  `lpp` lanes per atom
// stride through the outer row (coalesced), gather fp32 positions, and compact in order with group ballots; the
// inner row is partitioned front (r < rnear, walked by a guest force) / back (the rest).
template <bool COUNT_ONLY>
__global__ void __launch_bounds__(256) k_prune(int s_begin, int s_end, int lpp_shift, const float4 *__restrict__ pos4f_s,
                              BoxF box, float rlist2, float rnear2, const int *__restrict__ nl_out,
                              const int *__restrict__ nnb_out, int cap_out, int cap, int *nl, int *nnb, int *nnb_near,
                              int *flags, unsigned long long *blockstats, int force) {
    if (!force && !flags[0]) return;
    const int lpp = 1 << lpp_shift;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int a = tid >> lpp_shift;
    const int sub = tid & (lpp - 1);
    const int lane = threadIdx.x & 63;
    const int gbase = lane & ~(lpp - 1);
    const unsigned long long gmask = (lpp == 64 ? ~0ull : ((1ull << lpp) - 1ull)) << gbase;
    const unsigned long long below = (1ull << lane) - 1ull;
    const int s = s_begin + a;
    const bool valid = s < s_end;
    float4 pi = make_float4(0.f, 0.f, 0.f, 0.f);
    int nn = 0;
    if (valid) {
        pi = pos4f_s[s];
        nn = nnb_out[a];
    }
    int nmax = nn;
    for (int off = 32; off > 0; off >>= 1) nmax = max(nmax, __shfl_xor(nmax, off));
    const int *row_in = nl_out + (size_t)a * cap_out;
    int *row_out = nl + (size_t)a * cap;
    int cnt = 0, cntf = 0;
    for (int base = 0; base < nmax; base += 4 * lpp) {
        // four entries per trip, loads first: index -> position are two dependent round trips per entry otherwise
        int js[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = base + u * lpp + sub;
            ok[u] = k < nn;
            js[u] = ok[u] ? row_in[k] : s;
        }
        float4 pj[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) pj[u] = pos4f_s[js[u]];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float dx = pi.x - pj[u].x, dy = pi.y - pj[u].y, dz = pi.z - pj[u].z;
            dx -= box.L[0] * rintf(dx * box.invL[0]);
            dy -= box.L[1] * rintf(dy * box.invL[1]);
            dz -= box.L[2] * rintf(dz * box.invL[2]);
            const float r2 = dx * dx + dy * dy + dz * dz;
            const bool pass = ok[u] && (r2 < rlist2);
            const bool nearp = pass && (r2 < rnear2);
            const unsigned long long bal = __ballot(nearp) & gmask, balf = __ballot(pass && !nearp) & gmask;
            const int nb = __popcll(bal), nf = __popcll(balf);
            if (pass && !COUNT_ONLY) {
                const int pos_in = nearp ? cnt + __popcll(bal & below) : cap - 1 - (cntf + __popcll(balf & below));
                if (cnt + cntf + nb + nf <= cap) row_out[pos_in] = js[u];
            }
            cnt += nb;
            cntf += nf;
        }
    }
    const int total = cnt + cntf;
    if (valid && sub == 0 && !COUNT_ONLY) {
        const bool over = total > cap;
        nnb[a] = over ? 0 : total;
        nnb_near[a] = over ? 0 : cnt;
        if (over) flags[1] = 1;
    }
    unsigned long long wsum = (valid && sub == 0) ? (unsigned long long)total : 0ull;
    unsigned long long wnear = (valid && sub == 0) ? (unsigned long long)cnt : 0ull;
    int wmax = (valid && sub == 0) ? total : 0;
    for (int off = 32; off > 0; off >>= 1) {
        wsum += __shfl_xor(wsum, off);
        wnear += __shfl_xor(wnear, off);
        wmax = max(wmax, __shfl_xor(wmax, off));
    }
    __shared__ unsigned long long s_sum[4], s_near[4];
    __shared__ int s_max[4];
    if (lane == 0) {
        s_sum[threadIdx.x >> 6] = wsum;
        s_near[threadIdx.x >> 6] = wnear;
        s_max[threadIdx.x >> 6] = wmax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        blockstats[3 * blockIdx.x] = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
        blockstats[3 * blockIdx.x + 1] = (unsigned long long)max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
        blockstats[3 * blockIdx.x + 2] = s_near[0] + s_near[1] + s_near[2] + s_near[3];
    }
}

// ------------------------------------------------------------------------------------------------
// K2/K3: pair traversal.  lpa lanes per i-atom stride through the atom's neighbour row, UNR entries
// per trip with all loads issued up front (index -> position/charge -> sigma/eps gathers are the latency
// chain; independent chains per lane + 4-6 waves per SIMD hide it), branch-free arithmetic, wavefront-shuffle
// reduction.
struct PairArgs {
    int s_begin, s_end, lpa_shift, cap;
    const int *perm;
    const int *nl;
    const int *nnb;        // entries at the front of the row
    const int *nnb_total;  // if non-null: total entries; those beyond nnb[a] are stored from the back of the row
    const double4 *posq_s;
    const double2 *lj_s;
    double *force;     // original order [n][3]
    double *epart;     // per-block energy partials
    int accumulate;
    Box box;
    double *gforce;    // dual evaluation: force buffer of the guest force that shares this list (same particles)
    int gaccumulate;
    int sorted_out;    // exchange by all-gather: rows go to force[3 (s - s_begin)] (this rank's chunk of the exchange buffer)
    int gsame;         // the guest accumulates into the SAME rows as the host (fused FarNonbondedForce): one store of the sum
    const int *active;     // filtered lists: the rows that hold entries (slice-relative) ...
    const int *n_active;   // ... and their number (device); null: every row of the slice is walked
    const int *n_long;     // ... of which this many, filed from the front of `active`, are long rows; the others sit at the back
    int active_size;       //     of its active_size slots (amm_active_row)
    int long_shift;        // > 0: the long rows are walked with 1 << long_shift lanes each, before the others (k_pair_tab)
};


// row a (0 <= a < *n_active) of a filtered list's walk: the long rows from the front of `active`, the short ones from the back
__device__ __forceinline__ int amm_active_row(const PairArgs &A, int a) {
    const int nl = A.n_long ? *A.n_long : 0x7fffffff;
    return a < nl ? A.active[a] : A.active[A.active_size - 1 - (a - nl)];
}

__device__ double amm_erfcx_table_dev[AMM_ERFCX_NI * AMM_ERFCX_NC];
static bool g_erfcx_uploaded[64] = {false};   // per device: the table is a __device__ symbol of each device's code object

// GFAM >= 0: the guest force of a shared list (RESPA near force, same particles, shorter cutoff) is evaluated on the
// same pass into its own buffer: geometry, gathers, 1/r and the LJ / Coulomb pieces are common, so the guest costs a
// switching polynomial instead of a second traversal.
template <int FAM, int CMODE, bool GUARD, bool EN, int UNR, int GFAM, bool GROUPED = false>
__global__ void __launch_bounds__(256) k_pair_nlist(PairArgs A, PairConsts c, PairConsts gc) {
    const int lpa = 1 << A.lpa_shift;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    int a = tid >> A.lpa_shift;
    const int sub = tid & (lpa - 1);
    bool listed = true;
    if (A.active) {        // only the rows that hold entries; the caller has zeroed the others' outputs
        listed = a < *A.n_active;
        a = listed ? amm_active_row(A, a) : 0;
    }
    const int s = A.s_begin + a;
    const bool valid = listed && s < A.s_end;
    constexpr bool NEEDS_ERFC = (FAM == AMM_DAMPED) || (FAM == AMM_NONBONDED && CMODE == 1);
    __shared__ double s_tab[NEEDS_ERFC ? AMM_ERFCX_NI * AMM_ERFCX_NC : 1];
    if (NEEDS_ERFC) {
        for (int k = threadIdx.x; k < AMM_ERFCX_NI * AMM_ERFCX_NC; k += blockDim.x) s_tab[k] = amm_erfcx_table_dev[k];
        __syncthreads();
    }
    double fx = 0.0, fy = 0.0, fz = 0.0, esum = 0.0;
    double gx = 0.0, gy = 0.0, gz = 0.0;
    if (valid) {
        const double4 pi = A.posq_s[s];
        const double2 li = A.lj_s[s];
        const double qi = c.Kc * pi.w;
        const int nfront = A.nnb[a];
        const int nn = A.nnb_total ? A.nnb_total[a] : nfront;
        const int *row = A.nl + (size_t)a * A.cap;
        const int back = A.cap - 1 + nfront;
        const double guard2 = GUARD ? c.rc0 * c.rc0 : 0.0;
        for (int k0 = sub; k0 < nn; k0 += UNR * lpa) {
            int js[UNR];
            bool ok[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int k = k0 + u * lpa;
                ok[u] = k < nn;
                js[u] = ok[u] ? row[k < nfront ? k : back - k] : s;
            }
            double4 pj[UNR];
            double2 lj[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                // 32-bit byte offsets from the (scalar) array bases: one shift per load instead of a 64-bit index
                // extension and shift-add (slots < 2^26, checked when the force is created)
                pj[u] = *reinterpret_cast<const double4 *>(reinterpret_cast<const char *>(A.posq_s) + ((unsigned)js[u] << 5));
                lj[u] = *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(A.lj_s) + ((unsigned)js[u] << 4));
            }
            // the guest's candidates are the FRONT part of the row (walked first): once every row of this wavefront is
            // past its front part, the trips skip the guest arithmetic (its contribution there is an exact zero)
            const bool guest_trip = GFAM >= 0 && __builtin_amdgcn_ballot_w64(k0 < nfront) != 0ull;
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const double dx = amm_min_image(pi.x - pj[u].x, A.box.L[0], A.box.invL[0]);
                const double dy = amm_min_image(pi.y - pj[u].y, A.box.L[1], A.box.invL[1]);
                const double dz = amm_min_image(pi.z - pj[u].z, A.box.L[2], A.box.invL[2]);
                const double r2 = dx * dx + dy * dy + dz * dz;
                bool pass = ok[u] && (r2 < c.rc2);
                if (GUARD) pass = pass && (r2 <= guard2);          // step(rc0 - r)
                const double r2s = pass ? r2 : 1.0;
                // the guest first: it needs only what both share (1/r, the LJ and Coulomb pieces); the host's own part (its
                // table coefficients, exp, switch) is kept behind a scheduling barrier so that it is not live across the
                // guest block -- 142 -> see the kernel table in DESIGN.md for the register count
                if (GFAM >= 0) {
                    if (guest_trip) {
                        double eg, frg;
                        amm_pair_math<GFAM, 0, false, false>(gc, r2s, qi * pj[u].w, li.x + lj[u].x, li.y * lj[u].y, eg, frg, s_tab);
                        frg = (pass && r2 < gc.rc2) ? frg : 0.0;
                        gx += frg * dx;
                        gy += frg * dy;
                        gz += frg * dz;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                double e, fr;
                amm_pair_math<FAM, CMODE, false, EN, GROUPED>(c, r2s, qi * pj[u].w, li.x + lj[u].x, li.y * lj[u].y, e, fr, s_tab);
                fr = pass ? fr : 0.0;
                fx += fr * dx;
                fy += fr * dy;
                fz += fr * dz;
                if (EN) esum += pass ? e : 0.0;
            }
        }
    }
    // combine the lpa partial sums of each atom (fixed butterfly order -> deterministic)
    for (int off = lpa >> 1; off > 0; off >>= 1) {
        fx += __shfl_xor(fx, off);
        fy += __shfl_xor(fy, off);
        fz += __shfl_xor(fz, off);
        if (GFAM >= 0) {
            gx += __shfl_xor(gx, off);
            gy += __shfl_xor(gy, off);
            gz += __shfl_xor(gz, off);
        }
    }
    if (valid && sub == 0) {
        const int i = A.sorted_out ? s - A.s_begin : A.perm[s];
        if (GFAM >= 0) {
            if (A.gaccumulate) {
                A.gforce[3 * i] += gx;
                A.gforce[3 * i + 1] += gy;
                A.gforce[3 * i + 2] += gz;
            } else {
                A.gforce[3 * i] = gx;
                A.gforce[3 * i + 1] = gy;
                A.gforce[3 * i + 2] = gz;
            }
        }
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

template <int FAM, int CMODE, int UNR>
static void launch_pair_u(dim3 grid, dim3 block, hipStream_t st, bool guard, bool en, const PairArgs &A, const PairConsts &c) {
    if (guard) {
        if (en) hipLaunchKernelGGL((k_pair_nlist<FAM, CMODE, true, true, UNR, -1>), grid, block, 0, st, A, c, c);
        else hipLaunchKernelGGL((k_pair_nlist<FAM, CMODE, true, false, UNR, -1>), grid, block, 0, st, A, c, c);
    } else {
        if (en) hipLaunchKernelGGL((k_pair_nlist<FAM, CMODE, false, true, UNR, -1>), grid, block, 0, st, A, c, c);
        else hipLaunchKernelGGL((k_pair_nlist<FAM, CMODE, false, false, UNR, -1>), grid, block, 0, st, A, c, c);
    }
}

// list entries per lane and trip.  2: the erfc families fit 128 VGPRs = 4 waves/SIMD (218 VGPRs = 2 waves at 4:
// far kernel 295 -> 254 us at C3, near kernel unchanged); AMM_UNROLL / AMM_LPA are tuning knobs for experiments.
static int g_opt_unroll = 2, g_opt_dual_unroll = 2, g_opt_tab_bs = 0, g_opt_tab_dual_bs = 0;     // copied from the context by every evaluation
static int pair_unroll() { return g_opt_unroll; }

template <int FAM, int CMODE>
static void launch_pair(dim3 grid, dim3 block, hipStream_t st, bool guard, bool en, const PairArgs &A, const PairConsts &c) {
    const int unr = pair_unroll();
    if (unr == 1) launch_pair_u<FAM, CMODE, 1>(grid, block, st, guard, en, A, c);
    else if (unr == 2) launch_pair_u<FAM, CMODE, 2>(grid, block, st, guard, en, A, c);
    else if (unr == 3) launch_pair_u<FAM, CMODE, 3>(grid, block, st, guard, en, A, c);
    else launch_pair_u<FAM, CMODE, 4>(grid, block, st, guard, en, A, c);
}

// host force + guest force of the shared list in one pass (force only, no guard on either)
template <int FAM, int CMODE, int UNR>
static int launch_pair_dual_u(dim3 grid, dim3 block, hipStream_t st, int gfam, const PairArgs &A, const PairConsts &c,
                              const PairConsts &gc) {
    switch (gfam) {
    case AMM_NEAR_NONE: hipLaunchKernelGGL((k_pair_nlist<FAM, CMODE, false, false, UNR, AMM_NEAR_NONE>), grid, block, 0, st, A, c, gc); break;
    case AMM_NEAR_SHIFT: hipLaunchKernelGGL((k_pair_nlist<FAM, CMODE, false, false, UNR, AMM_NEAR_SHIFT>), grid, block, 0, st, A, c, gc); break;
    case AMM_NEAR_FSWITCH: hipLaunchKernelGGL((k_pair_nlist<FAM, CMODE, false, false, UNR, AMM_NEAR_FSWITCH>), grid, block, 0, st, A, c, gc); break;
    default: amm_set_error("dual evaluation: unsupported guest family"); return 1;
    }
    return 0;
}

// host force + guest force of the shared list in one pass (force only, no guard on either)
// interaction-group forces (AMM_GROUP_LJ / AMM_GROUP_Q): off the hot path, one entry per trip
template <int FAM>
static void launch_pair_grouped(dim3 grid, dim3 block, hipStream_t st, bool guard, bool en, const PairArgs &A, const PairConsts &c) {
    if (guard) {
        if (en) hipLaunchKernelGGL((k_pair_nlist<FAM, 0, true, true, 1, -1, true>), grid, block, 0, st, A, c, c);
        else hipLaunchKernelGGL((k_pair_nlist<FAM, 0, true, false, 1, -1, true>), grid, block, 0, st, A, c, c);
    } else {
        if (en) hipLaunchKernelGGL((k_pair_nlist<FAM, 0, false, true, 1, -1, true>), grid, block, 0, st, A, c, c);
        else hipLaunchKernelGGL((k_pair_nlist<FAM, 0, false, false, 1, -1, true>), grid, block, 0, st, A, c, c);
    }
}
template <int FAM, int CMODE>
static int launch_pair_dual(dim3 grid, dim3 block, hipStream_t st, int gfam, const PairArgs &A, const PairConsts &c,
                            const PairConsts &gc) {
    const int unr = g_opt_dual_unroll;
    if (unr == 1) return launch_pair_dual_u<FAM, CMODE, 1>(grid, block, st, gfam, A, c, gc);
    return launch_pair_dual_u<FAM, CMODE, 2>(grid, block, st, gfam, A, c, gc);
}

// ------------------------------------------------------------------------------------------------
// K2'/K3': force-only traversal with the Coulomb part read from the radial table of pair_tab.h.
//   * per pair: geometry (3 sub + r^2), the table look-up (2 integer ops for the interval, 5 fma) and three fma to
//     accumulate -- no 1/sqrt, erfc, exp or switching polynomial; the Lennard-Jones part (analytic, amm_lj_force) only in
//     wavefronts that hold a row with eps_i != 0 (rows are traversed in `row_order`: those rows first, so that all but one
//     wavefront are uniform); for TIP3P that is one row in three;
//   * rows whose atom is farther than `margin` from every face of the box skip the minimum-image arithmetic (wave-uniform);
//   * persistent wavefronts: the grid is sized to the chip (blocks per CU from the occupancy query), each block stages the
//     table(s) in LDS once and its wavefronts walk tasks (= 64 >> lpa_shift rows) with a stride; tasks are dealt so that
//     the blocks of one XCD (blockIdx & 7, MI355X_MICROARCH.md) own one contiguous eighth of the cell-sorted rows: the
//     j-records an XCD gathers are then one slab of the box, which its 4 MiB L2 holds;
//   * pairs closer than the table reaches take amm_pair_math under a wave-uniform branch (never in a liquid; lattice
//     starts and tests do meet them).
struct TabArgs {
    const double *host_tab, *guest_tab;   // nint x 6 doubles each (device)
    int host_bytes, guest_bytes;
    int need_erfcx;                       // the analytic path of an erfc family needs the erfcx table in LDS
    double margin;
    const int *row_order;                 // [nslice] sorted slots in traversal order, or null (slot order)
    const int *n_lj;                      // device: rows at the head of row_order that have a Lennard-Jones site (null: all)
    int ntask, nslice;
    const int *nnb_lj;                    // per row: entries that are partners WITH a site (front | back << 16), first on either side; null: unknown
    const int *nnb_all;                   // per row: total length (also when only the front part is walked: the stretches are cut the same way)
};

// NOQ: every charge of the force is zero (a Lennard-Jones fluid: config C2 of BASELINE.json): no Coulomb table -- no index
// arithmetic, no LDS look-up, no Horner chain, no charge product; the Lennard-Jones part is analytic as before and no pair is ever
// "below the table"
template <int FAM, int CMODE, int GFAM, bool INTERIOR, bool LJ, bool NOQ = false>
__device__ __forceinline__ void amm_walk_row_tab(const PairArgs &A, const PairConsts &c, const PairConsts &gc,
                                                 const char *tabh, const char *tabg, const double *s_erfcx, const double4 pi,
                                                 const double2 li, const int *row, int nfront, int nn, int sub, int lpa, int s,
                                                 double &fx, double &fy, double &fz, double &gx, double &gy, double &gz,
                                                 int kstart = 0, int kend = 0x7fffffff, int kcut = 0x7fffffff, int kgap = 0) {
    // The walk visits the row positions [kstart, kend), jumping over [kcut, kcut + kgap): one or two stretches of the rows.  All
    // four are wave-uniform multiples of the trip length and live in scalar registers (the trips are counted by a scalar): the
    // kernel gives the stretches whose partners may have a Lennard-Jones site to the walk with that arithmetic and the rest to
    // the walk without.
    constexpr int UNR = 2;
    const double qi = c.Kc * pi.w;
    const int back = A.cap - 1 + nfront;
    const int step = UNR * lpa;
    // Software pipeline, two trips deep: while trip t computes, the j-records of trip t + 1 are in flight (issued at the top of
    // the iteration) and so are the row entries of trip t + 2.  With half the arithmetic per pair of the analytic kernel the
    // wavefront must overlap its own memory latency -- there are too few wavefronts per SIMD (registers) to leave it to them.
    auto entry = [&](int kb, int k) { return (kb < kend && k < nn) ? row[k < nfront ? k : back - k] : s; };      // kb: the trip's first position (scalar)
    auto next_trip = [&](int kb) {
        const int n = kb + step;
        return __builtin_amdgcn_readfirstlane(n == kcut ? n + kgap : n);
    };
    auto any_left = [&](int kb) { return kb < kend && __builtin_amdgcn_ballot_w64(kb + sub < nn) != 0ull; };
    auto fetch = [&](int j, double4 &p, double2 &l) {
        p = *reinterpret_cast<const double4 *>(reinterpret_cast<const char *>(A.posq_s) + ((unsigned)j << 5));
        if (LJ) l = *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(A.lj_s) + ((unsigned)j << 4));
    };
    // one trip: UNR entries per lane, records already in registers.  Straight-line code for all entries (their dependent
    // chains -- table look-up, Horner steps, the Newton steps of 1/r -- interleave); pairs closer than a table reaches are
    // left out here and redone analytically below, in one rare branch that recomputes their geometry.
    auto geometry = [&](const double4 &pjv, double &dx, double &dy, double &dz) {
        dx = pi.x - pjv.x;
        dy = pi.y - pjv.y;
        dz = pi.z - pjv.z;
        if (!INTERIOR) {
            dx = amm_min_image(dx, A.box.L[0], A.box.invL[0]);
            dy = amm_min_image(dy, A.box.L[1], A.box.invL[1]);
            dz = amm_min_image(dz, A.box.L[2], A.box.invL[2]);
        }
        return dx * dx + dy * dy + dz * dz;
    };
    auto process = [&](const double4 (&pj)[UNR], const double2 (&lj)[UNR], const int (&jc)[UNR], int k0) {
        const bool guest_trip = GFAM >= 0 && __builtin_amdgcn_ballot_w64(k0 < nfront) != 0ull;
        bool any_low = false;
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const bool ok = k0 + u * lpa < nn;
            double dx, dy, dz;
            const double r2 = geometry(pj[u], dx, dy, dz);
            const bool pass = ok && (r2 < c.rc2);
            const double qq = NOQ ? 0.0 : qi * pj[u].w;
            double fr = NOQ ? 0.0 : qq * amm_tab_eval(tabh, c.tab, r2);
            double frg = 0.0;
            if (GFAM >= 0) {
                if (guest_trip) frg = qq * amm_tab_eval(tabg, gc.tab, r2);
            }
            if (LJ) {
                const double sig = li.x + lj[u].x, eps4 = li.y * lj[u].y;
                const LJCommon L = amm_lj_common(r2, sig, eps4);   // lanes out of range (r2 = 0 on padding) are selected away below
                fr += amm_lj_force<FAM, CMODE>(c, L, sig, eps4);
                if (GFAM >= 0) {
                    if (guest_trip) frg += amm_lj_force<GFAM, 0>(gc, L, sig, eps4);
                }
            }
            const bool gpass = GFAM >= 0 && pass && (r2 < gc.rc2);
            const bool low = !NOQ && pass && (r2 < c.tab.r2min), glow = gpass && (r2 < gc.tab.r2min);
            any_low = any_low || low || glow;
            fr = (pass && !low) ? fr : 0.0;
            fx += fr * dx;
            fy += fr * dy;
            fz += fr * dz;
            if (GFAM >= 0) {
                if (guest_trip) {
                    frg = (gpass && !glow) ? frg * gc.sign : 0.0;     // the table and amm_lj_force carry no sign (the discount: -1)
                    gx += frg * dx;
                    gy += frg * dy;
                    gz += frg * dz;
                }
            }
        }
#ifndef AMM_EXP_NOFALLBACK
        if (!NOQ && __builtin_amdgcn_ballot_w64(any_low) != 0ull) {     // closer than a table reaches: analytic (never in a liquid)
            for (int u = 0; u < UNR; ++u) {
                double dx, dy, dz;
                const double r2 = geometry(pj[u], dx, dy, dz);
                const bool pass = (k0 + u * lpa < nn) && (r2 < c.rc2);
                const bool low = pass && (r2 < c.tab.r2min), glow = GFAM >= 0 && pass && (r2 < gc.rc2) && (r2 < gc.tab.r2min);
                const double qq = qi * pj[u].w;
                const double2 lx = A.lj_s[jc[u]];
                const double sg = li.x + lx.x, e4 = li.y * lx.y;
                double e_, fr_;
                amm_pair_math<FAM, CMODE, false, false>(c, low ? r2 : 1.0, qq, sg, e4, e_, fr_, s_erfcx);
                fr_ = low ? fr_ : 0.0;
                fx += fr_ * dx;
                fy += fr_ * dy;
                fz += fr_ * dz;
                if (GFAM >= 0) {
                    amm_pair_math<GFAM, 0, false, false>(gc, glow ? r2 : 1.0, qq, sg, e4, e_, fr_, s_erfcx);
                    fr_ = glow ? fr_ : 0.0;
                    gx += fr_ * dx;
                    gy += fr_ * dy;
                    gz += fr_ * dz;
                }
            }
        }
#endif
    };
    // two register sets (a / b) alternate, so that no copy -- which would have to wait for the load -- sits between a
    // fetch and its use one iteration later
    int ja[UNR], jb[UNR];
    double4 pa[UNR], pb[UNR];
    double2 la[UNR], lb[UNR];
    int ka = __builtin_amdgcn_readfirstlane(kstart == kcut ? kstart + kgap : kstart), kb = next_trip(ka);
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
        ja[u] = entry(ka, ka + sub + u * lpa);
        jb[u] = entry(kb, kb + sub + u * lpa);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) fetch(ja[u], pa[u], la[u]);
    while (any_left(ka)) {
        int jc[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            fetch(jb[u], pb[u], lb[u]);                 // records of the next trip
            jc[u] = ja[u];
        }
        const int kc = next_trip(kb);
#pragma unroll
        for (int u = 0; u < UNR; ++u) ja[u] = entry(kc, kc + sub + u * lpa);   // entries of the trip after that
        process(pa, la, jc, ka + sub);
        if (!any_left(kb)) break;
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            fetch(ja[u], pa[u], la[u]);
            jc[u] = jb[u];
        }
        const int kd = next_trip(kc);
#pragma unroll
        for (int u = 0; u < UNR; ++u) jb[u] = entry(kd, kd + sub + u * lpa);
        process(pb, lb, jc, kb + sub);
        ka = kc;
        kb = kd;
    }
}

// Epilogue of the per-atom-row kernel (AEPI; the chargeless instantiations: config C2): the lane that has just stored the force on a
// row's atom applies the kicks and the move that follow the EVAL in the step program -- velocity Verlet: the closing half kick of
// this step and the opening half kick + move of the next (propagators.py:1136-1153 unrolled; k_kicks_move_atoms' arithmetic, per
// degree of freedom, FP contraction off: bit-identical) -- evaluates the lists' displacement triggers for the new position and
// writes the atom's record of the NEXT evaluation's sorted copy (a second buffer: other wavefronts still read this one).  A
// second loop over the wavefront's tasks, as cluster.hip's: nothing of the pair loop is live in it.
struct AtomEpiArgs {
    double *x, *v;
    const double *mass, *q;
    KickList K;
    int with_move;
    double dcoef;
    WatchArgs W;
    double4 *posq_next;
};

#ifndef AMM_TAB_WAVES_PER_EU
#define AMM_TAB_WAVES_PER_EU 1
#endif
// PH: the two-phase walk of a hybrid list's rest part (long rows, then short ones).  A compile-time switch: as a run-time one it cost
// the common kernels two registers -- 130 instead of 128, three wavefronts per SIMD instead of four (C2: 21 -> 27 us).
template <int FAM, int CMODE, int GFAM, int BS, bool PH = false, bool NOQ = false, bool AEPI = false>
__global__ void __launch_bounds__(BS) __attribute__((amdgpu_waves_per_eu(AMM_TAB_WAVES_PER_EU)))
k_pair_tab(PairArgs A, PairConsts c, PairConsts gc, TabArgs T, AtomEpiArgs E) {
    extern __shared__ __align__(16) char s_lds[];
    // stage the table(s): 16-byte pieces, coalesced
    for (int o = threadIdx.x * 16; o < T.host_bytes; o += BS * 16)
        *reinterpret_cast<double2 *>(s_lds + o) = *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(T.host_tab) + o);
    const char *tabh = s_lds, *tabg = s_lds + T.host_bytes;
    if (GFAM >= 0)
        for (int o = threadIdx.x * 16; o < T.guest_bytes; o += BS * 16)
            *reinterpret_cast<double2 *>(s_lds + T.host_bytes + o) = *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(T.guest_tab) + o);
    double *s_erfcx = reinterpret_cast<double *>(s_lds + T.host_bytes + (GFAM >= 0 ? T.guest_bytes : 0));
    if (T.need_erfcx)
        for (int k = threadIdx.x; k < AMM_ERFCX_NI * AMM_ERFCX_NC; k += BS) s_erfcx[k] = amm_erfcx_table_dev[k];
    __syncthreads();

    constexpr int WPB = BS / 64;
    const int lane = threadIdx.x & 63;
    // The rest part of a hybrid list is walked in two phases: its long rows (A.long_shift: a whole wavefront each), then its short
    // ones (A.lpa_shift) -- with one task size a wavefront that drew eight long rows walked 35 trips while most walked 6.
    constexpr int nphase = PH ? 2 : 1;
    const int nrows_all = A.active ? min(*A.n_active, T.nslice) : T.nslice;
    const int nlong_all = PH ? min(*A.n_long, nrows_all) : 0;
    for (int phase = 0; phase < nphase; ++phase) {
    const int lpa_shift = (PH && phase == 0) ? A.long_shift : A.lpa_shift;
    const int row0 = (PH && phase == 1) ? nlong_all : 0;
    const int lpa = 1 << lpa_shift;
    const int sub = lane & (lpa - 1);
    // Tasks: one wavefront's worth of rows (rpw = 64 >> lpa_shift), drawn from two pools -- rows with a Lennard-Jones site
    // (the first n_lj entries of row_order; about twice the arithmetic per pair) and rows without.  Each XCD
    // (blockIdx & 7) owns one contiguous eighth of either pool.
    const int xcd = blockIdx.x & 7, nwx = (gridDim.x >> 3) * WPB;
    const int rpw = 64 >> lpa_shift;
    // filtered lists (the rest part of a hybrid list): only the rows that hold entries, in the order the build collected them
    const int nrows = PH ? (phase == 0 ? nlong_all : nrows_all - nlong_all) : nrows_all;
    const int n_lj = T.n_lj ? min(*T.n_lj, nrows) : nrows;
    const int t_lj = (n_lj + rpw - 1) / rpw, t_h = (nrows - n_lj + rpw - 1) / rpw;
    // one contiguous eighth of either pool per XCD: consecutive cell-sorted rows, i.e. a slab of the box, whose gathers (the
    // slab + one list radius either side: 3.4 MB at 249k atoms, 2 MB at 98k) stay in the XCD's 4 MiB of L2.  (Two thinner
    // slabs per XCD -- to spread the face slabs, whose rows all need the minimum image -- gained nothing at 98k atoms and
    // overflowed the L2 at 249k.)
    const int per_lj = (t_lj + 7) >> 3, per_h = (t_h + 7) >> 3;
    const int lj0 = min(xcd * per_lj, t_lj), nlj_x = min(lj0 + per_lj, t_lj) - lj0;
    const int h0 = min(xcd * per_h, t_h), nh_x = min(h0 + per_h, t_h) - h0;
    const int ntask_x = nlj_x + nh_x;
    // static deal: position p of the XCD's task sequence goes to wavefront p mod nwx.  The sequence interleaves the two
    // pools in proportion (position p is a Lennard-Jones task iff floor((p + 1) nlj / n) > floor(p nlj / n)), so every
    // wavefront meets the same mix whatever the stride.  (A ticket counter costs more than it balances: the returning
    // atomic sits in front of the task's first loads in the in-order vmcnt queue.)
    for (int p = (int)(blockIdx.x >> 3) * WPB + (int)(threadIdx.x >> 6); p < ntask_x; p += nwx) {
        const int before = (int)(((long long)p * nlj_x) / ntask_x), upto = (int)(((long long)(p + 1) * nlj_x) / ntask_x);
        const bool lj_pool = upto > before;
        const int task = lj_pool ? before : nlj_x + (p - upto);
        const int a = lj_pool ? (lj0 + task) * rpw + (lane >> lpa_shift) : n_lj + (h0 + task - nlj_x) * rpw + (lane >> lpa_shift);
        const bool valid = lj_pool ? a < n_lj : a < nrows;
        int s = A.s_begin;
        double4 pi = make_double4(0.0, 0.0, 0.0, 0.0);
        double2 li = make_double2(0.0, 0.0);
        int nfront = 0, nn = 0, ntot = 0, sites_front = 0x7fffffff, sites_back = 0x7fffffff;
        const int *row = A.nl;
        if (valid) {
            s = A.active ? A.s_begin + (PH ? amm_active_row(A, row0 + a) : A.active[a]) : (T.row_order ? T.row_order[a] : A.s_begin + a);
            const int ra = s - A.s_begin;
            pi = A.posq_s[s];
            li = A.lj_s[s];
            nfront = A.nnb[ra];
            nn = A.nnb_total ? A.nnb_total[ra] : nfront;
            row = A.nl + (size_t)ra * A.cap;
            if (T.nnb_lj) {
                const int sites = T.nnb_lj[ra];
                sites_front = sites & 0xffff;
                sites_back = (int)((unsigned)sites >> 16);
                ntot = T.nnb_all[ra];
            }
        }
        double ev[3] = {0.0, 0.0, 0.0}, ex[3] = {0.0, 0.0, 0.0}, em = 1.0;       // AEPI: the row atom's state, fetched ahead of the walk
        if (AEPI && !PH && GFAM < 0 && valid && sub == 0) {
            const int ia = A.perm[s];
            em = E.mass[ia];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                ev[d] = E.v[3 * ia + d];
                ex[d] = E.x[3 * ia + d];
            }
        }
        const bool edge = valid && !(pi.x >= T.margin && pi.x <= A.box.L[0] - T.margin && pi.y >= T.margin && pi.y <= A.box.L[1] - T.margin &&
                                     pi.z >= T.margin && pi.z <= A.box.L[2] - T.margin);
        const bool interior = __builtin_amdgcn_ballot_w64(edge) == 0ull;
        const bool any_lj = __builtin_amdgcn_ballot_w64(valid && li.y != 0.0) != 0ull;
        double fx = 0.0, fy = 0.0, fz = 0.0, gx = 0.0, gy = 0.0, gz = 0.0;
#define AMM_WALK(IN, LJF, KS, KE, KC, KG) amm_walk_row_tab<FAM, CMODE, GFAM, IN, LJF, NOQ>(A, c, gc, tabh, tabg, s_erfcx, pi, li, row, nfront, nn, sub, lpa, s, fx, fy, fz, gx, gy, gz, KS, KE, KC, KG)
#define AMM_STRETCHES(LJF, KS, KE, KC, KG)                                                        \
        do {                                                                                      \
            if ((KE) > (KS) && __builtin_amdgcn_ballot_w64(valid && nn > (KS)) != 0ull) {         \
                if (interior) AMM_WALK(true, LJF, KS, KE, KC, KG);                                \
                else AMM_WALK(false, LJF, KS, KE, KC, KG);                                        \
            }                                                                                     \
        } while (0)
        // The build files the partners that have a Lennard-Jones site first on either side of a row, so only the first trips of the
        // front part and of the back part need the Lennard-Jones arithmetic (water: one pair in nine).  Cut at trip boundaries:
        // [0, f1) and [b0, b1) go to ONE walk with that arithmetic (it jumps from f1 to b0), [f1, b0) and [b1, ...) to one
        // walk without.  Tasks without a site take the second walk only, lists without the counts the first only.
        const int all = 0x7fffffff;
        int lj_end = 0, lj_cut = all, lj_gap = 0;              // walk with the Lennard-Jones part: [0, lj_end) less [lj_cut, lj_cut + lj_gap)
        int pl_start = 0, pl_end = all, pl_cut = all, pl_gap = 0;
        if (any_lj && !T.nnb_lj) {
            lj_end = all;
            pl_end = 0;
        } else if (any_lj) {
            const int trip_len = 2 << lpa_shift;
            const bool mine = valid && li.y != 0.0;
            // (from the row's full length even when only its front part is walked: the near force alone and as the guest of the
            // outer force's pass then add up a row in the same order -- bit for bit)
            const bool back = mine && ntot > nfront && sites_back > 0;
            int f1 = mine ? min(sites_front, nfront) : 0, b0 = back ? nfront : all, b1 = back ? nfront + sites_back : 0;
            for (int off = 32; off > 0; off >>= 1) {
                f1 = max(f1, __shfl_xor(f1, off));
                b0 = min(b0, __shfl_xor(b0, off));
                b1 = max(b1, __shfl_xor(b1, off));
            }
            f1 = __builtin_amdgcn_readfirstlane((f1 + trip_len - 1) / trip_len * trip_len);
            b0 = __builtin_amdgcn_readfirstlane(b0 == all ? all : b0 / trip_len * trip_len);
            b1 = __builtin_amdgcn_readfirstlane((b1 + trip_len - 1) / trip_len * trip_len);
            if (b1 <= b0) {                       // no partner with a site in any back part (or only front parts are walked)
                lj_end = f1;
                pl_start = f1;
            } else if (f1 >= b0) {                // the two stretches touch or overlap (rows of very different lengths in one wavefront)
                lj_end = max(f1, b1);       // (b1 alone would leave a long front part's partners to the walk without: tests seed 31)
                pl_start = lj_end;
            } else {
                lj_end = b1;
                lj_cut = f1;
                lj_gap = b0 - f1;
                pl_start = f1;
                pl_cut = b0;
                pl_gap = b1 - b0;
            }
        }
        AMM_STRETCHES(true, 0, lj_end, lj_cut, lj_gap);
        AMM_STRETCHES(false, pl_start, pl_end, pl_cut, pl_gap);
#undef AMM_STRETCHES
#undef AMM_WALK
        for (int off = lpa >> 1; off > 0; off >>= 1) {
            fx += __shfl_xor(fx, off);
            fy += __shfl_xor(fy, off);
            fz += __shfl_xor(fz, off);
            if (GFAM >= 0) {
                gx += __shfl_xor(gx, off);
                gy += __shfl_xor(gy, off);
                gz += __shfl_xor(gz, off);
            }
        }
        if (valid && sub == 0) {
            const int i = A.sorted_out ? s - A.s_begin : A.perm[s];
            if (GFAM >= 0) {
                if (A.gsame) {             // fused total + discount: one row, one store
                    fx += gx;
                    fy += gy;
                    fz += gz;
                } else if (A.gaccumulate) {
                    A.gforce[3 * i] += gx;
                    A.gforce[3 * i + 1] += gy;
                    A.gforce[3 * i + 2] += gz;
                } else {
                    A.gforce[3 * i] = gx;
                    A.gforce[3 * i + 1] = gy;
                    A.gforce[3 * i + 2] = gz;
                }
            }
            if (A.accumulate) {
                A.force[3 * i] += fx;
                A.force[3 * i + 1] += fy;
                A.force[3 * i + 2] += fz;
            } else {
                A.force[3 * i] = fx;
                A.force[3 * i + 1] = fy;
                A.force[3 * i + 2] = fz;
            }
            if (AEPI && !PH && GFAM < 0) {
                // the kicks (+ move) that follow the EVAL, for this row's atom: its velocity, position and mass were fetched before
                // the walk (a dependent load chain behind the walk cost more than the launch it replaced: 29.7 against 19.9 + 4.6 us)
                const double fnow[3] = {fx, fy, fz};
                double xn[3];
#pragma unroll
                for (int d = 0; d < 3; ++d) {
#pragma clang fp contract(off)
                    const int t = 3 * i + d;
                    double vt = ev[d];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (k < E.K.n) {
                            // (a kick with THIS launch's force takes it from the register; other buffers are read)
                            double ff = E.K.f[k] == A.force ? fnow[d] : E.K.f[k][t];
                            if (E.K.f2[k]) {
                                const double f2 = E.K.f2[k] == A.force ? fnow[d] : E.K.f2[k][t];
                                ff = E.K.plus[k] ? ff + f2 : ff - f2;
                            }
                            const double num = E.K.coef[k] * ff;
                            const double dv = num / em;
                            vt = vt + dv;
                        }
                    }
                    E.v[t] = vt;
                    xn[d] = ex[d];
                    if (E.with_move) {
                        const double dx = E.dcoef * vt;
                        xn[d] = xn[d] + dx;
                        E.x[t] = xn[d];
                    }
                }
                if (E.with_move) {
                    amm_watch_atom(E.W, i, xn);
                    if (E.posq_next)
                        E.posq_next[s] = make_double4(wrap1(xn[0], A.box.L[0], A.box.invL[0]), wrap1(xn[1], A.box.L[1], A.box.invL[1]),
                                                      wrap1(xn[2], A.box.L[2], A.box.invL[2]), E.q[i]);
                }
            }
        }
    }
    }
}


// one instantiation: dynamic LDS attribute + blocks per CU (cached), persistent grid (a multiple of 8 blocks)
template <int FAM, int CMODE, int GFAM, int BS, bool PH = false, bool NOQ = false, bool AEPI = false>
static int launch_pair_tab_i(hipStream_t st, const PairArgs &A, const PairConsts &c, const PairConsts &gc, TabArgs &T,
                             const AtomEpiArgs &E = AtomEpiArgs()) {
    // (the LDS attribute and the occupancy answer belong to a device: one slot per device, like the erfcx upload flags)
    static int bpc_dev[64], lds_set_dev[64], cu_dev[64];
    static bool init_dev[64];
    int dev = 0;
    AMM_HIP(hipGetDevice(&dev));
    dev &= 63;
    if (!init_dev[dev]) {
        init_dev[dev] = true;
        bpc_dev[dev] = -1;
        lds_set_dev[dev] = 0;
        AMM_HIP(hipDeviceGetAttribute(&cu_dev[dev], hipDeviceAttributeMultiprocessorCount, dev));
    }
    int &bpc = bpc_dev[dev], &lds_set = lds_set_dev[dev];
    const int g_num_cu = cu_dev[dev];
    const int lds = T.host_bytes + (GFAM >= 0 ? T.guest_bytes : 0) + AMM_ERFCX_NI * AMM_ERFCX_NC * 8;
    if (lds > 160 * 1024) {
        amm_set_error("tabulated pair kernel: the radial tables of the two forces do not fit LDS together");
        return 1;
    }
    auto kern = k_pair_tab<FAM, CMODE, GFAM, BS, PH, NOQ, AEPI>;
    if (lds > lds_set) {
        AMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        lds_set = lds;
        bpc = -1;
    }
    if (bpc < 0) {
        int nb = 0;
        AMM_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, BS, (size_t)lds));
        if (nb < 1) {
            amm_set_error("tabulated pair kernel does not fit on a CU (LDS)");
            return 1;
        }
        bpc = nb;
    }
    constexpr int WPB = BS / 64;
    long nblk = std::min((long)g_num_cu * bpc, ((long)T.ntask + WPB - 1) / WPB);
    nblk = std::max(8L, (nblk + 7) / 8 * 8);
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(BS), (size_t)lds, st, A, c, gc, T, E);
    return 0;
}

// block sizes (wavefronts per block a multiple of 4: an uneven deal over the 4 SIMDs strands registers).  Alone: 512 threads
// (2 per SIMD; 2 blocks per CU with <= 128 VGPRs) or 768 (3 per SIMD, 1 block); with a guest: 1024 or 768.  AMM_TAB_BS /
// AMM_TAB_DUAL_BS override for tuning.
static int tab_block_size(bool dual) {
    const int opt = dual ? g_opt_tab_dual_bs : g_opt_tab_bs;
    return opt > 0 ? opt : (dual ? 768 : 512);
}

template <int FAM, int CMODE, int GFAM>
static int launch_pair_tab_g(hipStream_t st, const PairArgs &A, const PairConsts &c, const PairConsts &gc, TabArgs &T) {
    const int bs = tab_block_size(GFAM >= 0);
    // (the rest part of a hybrid list, walked in two phases: the default block sizes only)
    if (A.active && A.n_long && A.long_shift > 0) return launch_pair_tab_i<FAM, CMODE, GFAM, (GFAM >= 0 ? 768 : 512), true>(st, A, c, gc, T);
    if (bs == 1024) return launch_pair_tab_i<FAM, CMODE, GFAM, 1024>(st, A, c, gc, T);
    if (bs == 768) return launch_pair_tab_i<FAM, CMODE, GFAM, 768>(st, A, c, gc, T);
    return launch_pair_tab_i<FAM, CMODE, GFAM, 512>(st, A, c, gc, T);
}

template <int FAM, int CMODE>
static int launch_pair_tab(hipStream_t st, int gfam, const PairArgs &A, const PairConsts &c, const PairConsts &gc, TabArgs &T) {
    switch (gfam) {
    case -1: return launch_pair_tab_g<FAM, CMODE, -1>(st, A, c, gc, T);
    case AMM_NEAR_NONE: return launch_pair_tab_g<FAM, CMODE, AMM_NEAR_NONE>(st, A, c, gc, T);
    case AMM_NEAR_SHIFT: return launch_pair_tab_g<FAM, CMODE, AMM_NEAR_SHIFT>(st, A, c, gc, T);
    case AMM_NEAR_FSWITCH: return launch_pair_tab_g<FAM, CMODE, AMM_NEAR_FSWITCH>(st, A, c, gc, T);
    default: amm_set_error("dual evaluation: unsupported guest family"); return 1;
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
        // cell edge >= rlist/2  ->  neighbours within +-2 cells; fewer than 5 cells: visit every cell once
        const double rgrid = pf->skin_out > pf->skin * (1 + 1e-9) ? pf->rlist_out_build : pf->rlist_build;
        int nc = (int)floor(L / (0.5 * rgrid));
        if (nc < 1) nc = 1;
        if (nc > 512) nc = 512;
        g.nc[k] = nc;
        g.h[k] = 2;
        g.nstencil[k] = nc >= 5 ? 5 : nc;
        g.cw[k] = L / nc;
        g.inv_cw[k] = nc / L;
        g.ncell *= nc;
    }
    return 0;
}

// cell list -> candidate sweep.  direct = false: OUTER list (radius rc + skin_out, conditional on flags[4]);
// direct = true (single-list mode, skin_out <= skin): straight into the traversed inner list (flags[0]).
static int cell_build_chain(amm_ctx *ctx, PairForce *pf, const double *d_pos, int force, bool count_only, bool direct,
                            PairForce *gather_for = nullptr, int copies_current = 0) {
    hipStream_t st = ctx->stream;
    const int n = pf->n;
    const int nb = (n + 255) / 256;
    const int which = direct ? 0 : 4;
    hipLaunchKernelGGL(k_cell_assign, dim3(nb), dim3(256), 0, st, n, d_pos, ctx->box, pf->grid, pf->d_cell_of,
                       pf->d_cell_count, pf->d_cell_start, pf->d_cell_members, pf->capc, direct ? pf->d_xref : pf->d_xref_out,
                       pf->d_flags, pf->d_ticket, which, force, pf->d_cls, pf->d_cell_count_lj, pf->d_cell_start_lj);
    if (!pf->d_cell_members) return 0;         // sizing pass of the first build: only the counts were wanted
    // gather_for: the sorted fp64 copies of the evaluation that follows ride on the same launch
    PairForce *gf = gather_for;
    const long sort_threads = std::max((long)pf->grid.ncell * 64, gf ? (long)n : 0L);
    hipLaunchKernelGGL(k_cell_sort_gather, dim3((unsigned)((sort_threads + 255) / 256)), dim3(256), 0, st, pf->grid.ncell, n,
                       pf->d_cell_start, pf->d_cell_members, pf->capc, pf->d_perm, d_pos, ctx->box, pf->d_pos4f_s,
                       pf->d_inv_perm, pf->d_flags, which, force, gf ? gf->d_q : nullptr, gf ? gf->d_hsig : nullptr,
                       gf ? gf->d_seps2 : nullptr, gf ? gf->d_posq_s : (double4 *)nullptr, gf ? gf->d_lj_s : (double2 *)nullptr,
                       pf->d_cls, pf->d_cell_start_lj, pf->d_row_order, pf->s_begin, pf->s_end, pf->d_flags + 3, pf->d_member,
                       pf->d_cell_sets, pf->hybrid_rest ? 2 : 1, copies_current);
    const long threads = (long)pf->grid.ncell * pf->parts * 64;   // one wavefront per (cell, part)
    dim3 grid((unsigned)((threads + 255) / 256));
    BoxF bf;
    for (int k = 0; k < 3; ++k) {
        bf.L[k] = (float)ctx->box.L[k];
        bf.invL[k] = (float)ctx->box.invL[k];
    }
    const double rl = direct ? pf->rlist_build : pf->rlist_out_build;
    const float rl2 = (float)(rl * rl);
    const float rn2 = (direct && pf->rnear_build > 0) ? (float)(pf->rnear_build * pf->rnear_build) : 3.0e38f;
    int *nl = direct ? pf->d_nl : pf->d_nl_out, *nnb = direct ? pf->d_nnb : pf->d_nnb_out;
    int *nnb_near = direct ? pf->d_nnb_near : pf->d_nnb_scratch;
    const int cap = direct ? pf->cap : pf->cap_out;
    const bool use_rint = pf->grid.nc[0] < 5 || pf->grid.nc[1] < 5 || pf->grid.nc[2] < 5;
#define AMM_LAUNCH_BUILD(CO, RI)                                                                                       \
    hipLaunchKernelGGL((k_build_nlist<CO, RI>), grid, dim3(256), 0, st, pf->s_begin, pf->s_end, pf->parts, pf->d_perm,  \
                       pf->d_inv_perm, pf->d_cell_start, pf->d_pos4f_s, bf, pf->grid, rl2, rn2, pf->d_excl_ptr,         \
                       pf->d_excl_idx, cap, nl, nnb, nnb_near, pf->d_flags, pf->d_blockstats, pf->d_counters,      \
                       pf->d_ticket + AMM_TICKET_INTS, which, force, pf->d_member ? (pf->hybrid_rest ? 2 : 1) : 0, direct ? pf->d_active : (int *)nullptr, pf->active_cap, pf->active_size, pf->d_cell_sets,        \
                       pf->d_cell_start_lj, direct ? pf->d_nnb_lj : (int *)nullptr)
    if (count_only) {
        if (use_rint) AMM_LAUNCH_BUILD(true, true);
        else AMM_LAUNCH_BUILD(true, false);
    } else {
        if (use_rint) AMM_LAUNCH_BUILD(false, true);
        else AMM_LAUNCH_BUILD(false, false);
    }
#undef AMM_LAUNCH_BUILD
    AMM_HIP(hipGetLastError());
    return 0;
}

// inner list: prune the outer rows with the current positions (conditional on flags[0] unless force)
static int prune_chain(amm_ctx *ctx, PairForce *pf, const double *d_pos, int force, bool count_only) {
    hipStream_t st = ctx->stream;
    const int n = pf->n;
    const int nb = (n + 255) / 256;
    hipLaunchKernelGGL(k_gather_f32, dim3(nb), dim3(256), 0, st, n, pf->d_perm, d_pos, ctx->box, pf->d_pos4f_s,
                       (int *)nullptr, count_only ? (double *)nullptr : pf->d_xref, pf->d_flags, 0, force);
    const int nslice = pf->s_end - pf->s_begin;
    const int lpp_shift = 4;                                      // 16 lanes per atom
    const long threads = (long)std::max(nslice, 1) << lpp_shift;
    dim3 grid((unsigned)((threads + 255) / 256));
    BoxF bf;
    for (int k = 0; k < 3; ++k) {
        bf.L[k] = (float)ctx->box.L[k];
        bf.invL[k] = (float)ctx->box.invL[k];
    }
    const float rl2 = (float)(pf->rlist_build * pf->rlist_build);
    const float rn2 = pf->rnear_build > 0 ? (float)(pf->rnear_build * pf->rnear_build) : 3.0e38f;
    if (count_only)
        hipLaunchKernelGGL((k_prune<true>), grid, dim3(256), 0, st, pf->s_begin, pf->s_end, lpp_shift, pf->d_pos4f_s, bf, rl2,
                           rn2, pf->d_nl_out, pf->d_nnb_out, pf->cap_out, pf->cap, pf->d_nl, pf->d_nnb, pf->d_nnb_near,
                           pf->d_flags, pf->d_blockstats, force);
    else
        hipLaunchKernelGGL((k_prune<false>), grid, dim3(256), 0, st, pf->s_begin, pf->s_end, lpp_shift, pf->d_pos4f_s, bf, rl2,
                           rn2, pf->d_nl_out, pf->d_nnb_out, pf->cap_out, pf->cap, pf->d_nl, pf->d_nnb, pf->d_nnb_near,
                           pf->d_flags, pf->d_blockstats, force);
    hipLaunchKernelGGL(k_finish_build, dim3(1), dim3(256), 0, st, pf->d_flags, pf->d_counters, pf->d_blockstats, (int)grid.x,
                       count_only ? 1 : 0, 0, force);
    AMM_HIP(hipGetLastError());
    return 0;
}

static int first_build(amm_ctx *ctx, PairForce *pf, const double *d_pos) {
    // slice of the sorted order owned by this rank
    const int n = pf->n;
    const int per = amm_slice_per(n, ctx->world);
    pf->s_begin = std::min(n, ctx->rank * per);
    pf->s_end = std::min(n, pf->s_begin + per);
    const int nslice = pf->s_end - pf->s_begin;
    const size_t ns = (size_t)std::max(nslice, 1);
    AMM_HIP(hipMalloc(&pf->d_nnb, sizeof(int) * ns));
    AMM_HIP(hipMalloc(&pf->d_nnb_near, sizeof(int) * ns));
    AMM_HIP(hipMalloc(&pf->d_nnb_out, sizeof(int) * ns));
    AMM_HIP(hipMalloc(&pf->d_nnb_scratch, sizeof(int) * ns));
    AMM_HIP(hipMalloc(&pf->d_row_order, sizeof(int) * ns));
    AMM_HIP(hipMalloc(&pf->d_nnb_lj, sizeof(int) * ns));
    AMM_HIP(hipMemset(pf->d_nnb_lj, 0, sizeof(int) * ns));
    if (pf->d_member && !(pf->skin_out > pf->skin * (1 + 1e-9))) AMM_HIP(hipMalloc(&pf->d_active, sizeof(int) * ns));
    pf->active_cap = (int)ns;
    pf->active_size = (int)ns;
    // lanes per atom: aim at >= 8 wavefronts per SIMD (1024 SIMDs) for latency hiding, but not beyond 16 lanes: longer
    // strides waste the tail of every row (measured on 1/8 slices of C3, scripts/probe_slices.py: dual pass 63.6 us
    // with 16 lanes, 71.9 us with 64)
    // never below 8: a wavefront's trip count is the maximum over its rows, and with 16 rows of 4 lanes the spread of the row
    // lengths costs more than the shorter tails save (249k atoms, dual pass: 837 us with 4 lanes, 657 with 8, 643 with 16;
    // 98k atoms: 229 us with 8, 240 with 16)
    int lpa = 8;
    while (lpa < 16 && (long)nslice * lpa < 64L * 1024 * 8) lpa <<= 1;
    if (ctx->opt_lpa > 0) lpa = ctx->opt_lpa;
    pf->lpa = lpa;
    int flags[8];
    {
        // member tables of the cell list: twice the fullest cell of the first configuration + 16
        pf->capc = 0;
        if (cell_build_chain(ctx, pf, d_pos, 1, true, true)) return 1;
        AMM_HIP(hipMemcpyAsync(flags, pf->d_flags, sizeof(flags), hipMemcpyDeviceToHost, ctx->stream));
        AMM_HIP(hipStreamSynchronize(ctx->stream));
        pf->capc = 2 * flags[6] + 16;
        AMM_HIP(hipMalloc(&pf->d_cell_members, sizeof(int) * (size_t)pf->grid.ncell * pf->capc));
        if (pf->d_member) AMM_HIP(hipMalloc(&pf->d_cell_sets, sizeof(int) * (size_t)pf->grid.ncell));
        // waves per cell: enough that even the fullest cell's share is one batch per wave (a second batch walks the
        // whole candidate stream again; measured at C3: 315 us with 4 parts, 331 us with the 3 that the mean suggests)
        pf->parts = std::max(1, std::min(8, (int)std::ceil(1.15 * flags[6] / AMM_BATCH)));
        if (ctx->opt_parts > 0) pf->parts = std::max(1, std::min(8, ctx->opt_parts));
        const long t1 = (long)pf->grid.ncell * pf->parts * 64, t2 = (long)ns * 16;
        const size_t nblk = (size_t)((std::max(t1, t2) + 255) / 256);
        AMM_HIP(hipMalloc(&pf->d_blockstats, sizeof(unsigned long long) * 3 * nblk));
    }
    pf->dual = pf->skin_out > pf->skin * (1 + 1e-9);
    if (pf->dual) {
        // outer list: count, size, build
        pf->cap_out = 0;
        if (cell_build_chain(ctx, pf, d_pos, 1, true, false)) return 1;
        AMM_HIP(hipMemcpyAsync(flags, pf->d_flags, sizeof(flags), hipMemcpyDeviceToHost, ctx->stream));
        AMM_HIP(hipStreamSynchronize(ctx->stream));
        pf->cap_out = ((int)(flags[5] * 1.5) + 32 + 15) / 16 * 16;   // head-room for density fluctuations between rebuilds
        if (pf->cap_out > 65535) {
            amm_set_error("neighbour rows longer than 65535 entries are not supported (cutoff too large for this density)");
            return 1;
        }
        AMM_HIP(hipMalloc(&pf->d_nl_out, sizeof(int) * ns * pf->cap_out));
        if (cell_build_chain(ctx, pf, d_pos, 1, false, false)) return 1;
        // inner list: count, size, prune
        pf->cap = 0;
        if (prune_chain(ctx, pf, d_pos, 1, true)) return 1;
        AMM_HIP(hipMemcpyAsync(flags, pf->d_flags, sizeof(flags), hipMemcpyDeviceToHost, ctx->stream));
        AMM_HIP(hipStreamSynchronize(ctx->stream));
        pf->cap = ((int)(flags[2] * 1.5) + 32 + 15) / 16 * 16;
        AMM_HIP(hipMalloc(&pf->d_nl, sizeof(int) * ns * pf->cap));
        if (prune_chain(ctx, pf, d_pos, 1, false)) return 1;
    } else {
        // single list: the cell sweep writes the traversed list directly.  The row capacity comes from the longest row of the
        // WHOLE box, not of this rank's slice: the slices follow the cell-sorted order, so a later rebuild can bring rows into
        // the slice that are far longer than any it held at first -- the rows of a filtered (interaction-group) list are a
        // few hundred entries for a solute atom and a handful for the solvent (found by the four-rank test of config C5)
        pf->cap = 0;
        {
            const int sb = pf->s_begin, se = pf->s_end;
            int *const row_order = pf->d_row_order;
            if (ctx->world > 1) {
                pf->s_begin = 0;
                pf->s_end = n;
                pf->d_row_order = nullptr;        // sized for the slice: the sort kernel must not fill it for the whole box
            }
            const int rc_count = cell_build_chain(ctx, pf, d_pos, 1, true, true);
            pf->s_begin = sb;
            pf->s_end = se;
            pf->d_row_order = row_order;
            if (rc_count) return 1;
        }
        AMM_HIP(hipMemcpyAsync(flags, pf->d_flags, sizeof(flags), hipMemcpyDeviceToHost, ctx->stream));
        AMM_HIP(hipStreamSynchronize(ctx->stream));
        pf->cap = ((int)(flags[2] * 1.5) + 32 + 15) / 16 * 16;
        if (pf->cap > 65535) {     // the build kernel packs the two running row lengths into 16 + 16 bits
            amm_set_error("neighbour rows longer than 65535 entries are not supported (cutoff too large for this density)");
            return 1;
        }
        AMM_HIP(hipMalloc(&pf->d_nl, sizeof(int) * ns * pf->cap));
        if (cell_build_chain(ctx, pf, d_pos, 1, false, true)) return 1;
    }
    if (pf->d_active) {
        // rows that hold entries: the pair kernels' grid covers twice the first build's number (+ 64); more than that later on
        // is reported by amm_check like a row overflow
        int f16[16];
        AMM_HIP(hipMemcpyAsync(f16, pf->d_flags, sizeof(f16), hipMemcpyDeviceToHost, ctx->stream));
        AMM_HIP(hipStreamSynchronize(ctx->stream));
        pf->active_cap = (int)std::min<size_t>(ns, (size_t)2 * f16[8] + 64);
        if (ctx->world > 1) pf->active_cap = (int)ns;      // a slice can come to hold any share of the rows with entries
        // the per-atom part of a hybrid list: how many molecule atoms have a partner outside the molecules follows the solute's shape
        // (a chain that unfolds): four times the first build's rows, and never an error short of every row
        if (pf->hybrid_rest && ctx->world == 1) pf->active_cap = (int)std::min<size_t>(ns, (size_t)4 * f16[8] + 1024);
    }
    pf->built = true;
    return 0;
}

// guest != nullptr: `pf` owns the list `guest` traverses, both forces act on the same particles (checked by
// amm_pair_can_eval_dual) and only forces are wanted: one pass writes pf's force to d_force and the guest's to g_force.
__global__ void k_unsort(int n, int per, int nf, const int *__restrict__ perm, const double *__restrict__ xchg, double *force,
                         double *gforce) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const int r = s / per, loc = s - r * per;
    const double *src = xchg + ((size_t)r * nf * per + loc) * 3;
    const int i = perm[s];
    force[3 * i] = src[0];
    force[3 * i + 1] = src[1];
    force[3 * i + 2] = src[2];
    if (nf == 2) {
        src += (size_t)per * 3;
        gforce[3 * i] = src[0];
        gforce[3 * i + 1] = src[1];
        gforce[3 * i + 2] = src[2];
    }
}

// second half of an exchanged evaluation: the gathered chunks -> the force buffers, in atom order
int amm_exchange_finish_impl(amm_ctx *ctx) {
    PendingExchange &pe = ctx->pending;
    if (!pe.active) {
        amm_set_error("amm_exchange_finish: no evaluation is waiting for its exchange");
        return 1;
    }
    if (pe.kind == 1) return amm_cluster_state_finish_impl(ctx);      // the chunks hold positions and velocities (cluster.hip)
    hipLaunchKernelGGL(k_unsort, dim3((ctx->n + 255) / 256), dim3(256), 0, ctx->stream, ctx->n, pe.per, pe.nf, pe.perm, ctx->d_xchg,
                       pe.force, pe.gforce);
    AMM_HIP(hipGetLastError());
    pe.active = false;
    return 0;
}

int amm_pair_eval_impl(amm_ctx *ctx, PairForce *pf, const double *d_pos, double *d_force, int accumulate,
                       double *d_energy, PairForce *guest, double *g_force, int g_accumulate, int exchange) {
    hipStream_t st = ctx->stream;
    const int n = pf->n;
    const int nb = (n + 255) / 256;
    if (!g_erfcx_uploaded[ctx->device & 63]) {
        AMM_HIP(hipMemcpyToSymbol(HIP_SYMBOL(amm_erfcx_table_dev), amm_erfcx_table_host, sizeof(amm_erfcx_table_host)));
        g_erfcx_uploaded[ctx->device & 63] = true;
    }
    // the neighbour list may belong to another, longer-ranged pair force (amm_pair_share_list)
    PairForce *L = pf->host ? pf->host : pf;
    // an interaction-group force whose smaller set is a few dozen atoms: evaluated directly, no list (group.hip)
    if (pf->small && ctx->opt_small_group && !pf->built && !guest && !exchange) {
        const int r = amm_small_group_eval_impl(ctx, pf, d_pos, d_force, accumulate, d_energy);
        if (r >= 0) return r;
    }
    g_opt_unroll = ctx->opt_unroll;
    g_opt_dual_unroll = ctx->opt_dual_unroll;
    g_opt_tab_bs = ctx->opt_tab_bs;
    g_opt_tab_dual_bs = ctx->opt_tab_dual_bs;
    // force-only evaluations of a water-like system walk molecule rows (cluster.hip): the same test as `tab_ok` below, made
    // before any per-atom list is touched -- that list is only built if an evaluation of the other kind asks for it
    {
        const bool guard0 = (pf->desc.flags & AMM_GUARD_RC0) != 0;
        // (the rows carry the list owner's site bits: a force that walks them must have its sites on the same atoms)
        if (pf->sites_match < 0) pf->sites_match = (pf == L || pf->h_cls == L->h_cls) ? 1 : 0;
        if (guest && guest->sites_match < 0) guest->sites_match = (guest == L || guest->h_cls == L->h_cls) ? 1 : 0;
        const bool cluster = ctx->opt_cluster && ctx->opt_tab && L->cluster_ok && pf->cluster_ok && !d_energy && !guard0 &&
                             (!L->hybrid || (ctx->opt_hybrid && pf->hybrid && (!guest || guest->hybrid))) &&
                             pf->sites_match == 1 && (!guest || guest->sites_match == 1) &&
                             pf->pc.tab.nint > 0 && pf->d_tab && pf->pc.sign == 1.0 && !(L->skin_out > L->skin * (1 + 1e-9)) &&
                             (!guest || (guest->cluster_ok && guest->pc.tab.nint > 0 && guest->d_tab));
        if (cluster && L->hybrid) {
            // hybrid list: the pairs of two three-site molecules walk molecule rows; every pair with an atom outside those molecules
            // is the child's (per-atom rows filtered to such pairs, pair.hip), which adds to the same buffers afterwards
            if (exchange) {
                amm_set_error("a hybrid list (molecule rows + per-atom rows) exchanges its forces by all-reduce, not by all-gather");
                return 1;
            }
            if (!pf->rest || (guest && !guest->rest)) {
                amm_set_error("hybrid list: the per-atom part of the force is missing");
                return 1;
            }
            L->last_kind = 2;
            if (ctx->world == 1 && L->n_rest > 0) {     // (world > 1: the molecule-row evaluation clears whole buffers)
                double *z0 = accumulate ? nullptr : d_force;
                double *z1 = (guest && !g_accumulate && g_force != d_force) ? g_force : nullptr;
                // (cleared by the molecule-row chain's sort / gather launch, which always runs: no launch of its own)
                if (z0 || z1) ctx->zero_rows = CZeroRows{L->n_rest, L->d_rest_idx, z0, z1};
            }
            const int rc_rows = amm_cluster_eval_impl(ctx, pf, d_pos, d_force, accumulate, guest, g_force, g_accumulate, 0);
            ctx->zero_rows = CZeroRows{0, nullptr, nullptr, nullptr};          // (consumed; cleared here too for the error paths)
            if (rc_rows) return 1;
            return amm_pair_eval_impl(ctx, pf->rest, d_pos, d_force, 1, nullptr, guest ? guest->rest : nullptr, g_force, 1, 0);
        }
        if (cluster) {
            L->last_kind = 1;
            return amm_cluster_eval_impl(ctx, pf, d_pos, d_force, accumulate, guest, g_force, g_accumulate, exchange);
        }
        L->last_kind = 0;
    }
    // the plan of an epilogue that amm_run_ops attached to this evaluation (kind 1: kicks + move of the rows' atoms); consumed here
    const EpiPlan *aplan = ctx->epi_request;
    ctx->epi_request = nullptr;
    ctx->epi_done = false;
    // the launch that moved the atoms to these positions wrote this force's sorted copies already?
    const int copies_current = (L->a_sorted_for == pf && L->a_sorted_epoch == ctx->pos_epoch && L->a_sorted_pos == d_pos && !L->dual) ? 1 : 0;
    if (copies_current) ctx->n_copies_current++;
    bool gathered = false;
    if (!L->built) {
        if (first_build(ctx, L, d_pos)) return 1;
    } else if (L->checked_epoch == ctx->pos_epoch && L->checked_pos == d_pos && !L->force_rebuild) {
        // positions unchanged since this list was last checked (e.g. the far force right after the near force
        // that shares the list): nothing to do
    } else {
        const double thr_in = 0.5 * L->skin, thr_out = L->dual ? 0.5 * (L->skin_out - L->skin) : 1.0e30;
        if (!(L->pre_epoch == ctx->pos_epoch && L->pre_pos == d_pos && !L->dual))   // else: the integration kernel already checked
            hipLaunchKernelGGL(k_check_displacement, dim3(nb), dim3(256), 0, st, n, d_pos, L->d_xref,
                               L->dual ? L->d_xref_out : L->d_xref, thr_in * thr_in, thr_out * thr_out, L->d_flags);
        const int forced = L->force_rebuild ? 1 : 0;       // the site pattern changed (amm_pair_set_params)
        L->force_rebuild = false;
        if (L->dual) {
            if (cell_build_chain(ctx, L, d_pos, forced, false, false)) return 1;
            if (prune_chain(ctx, L, d_pos, forced, false)) return 1;
        } else {
            if (cell_build_chain(ctx, L, d_pos, forced, false, true, pf, copies_current)) return 1;
            gathered = true;
        }
    }
    L->checked_epoch = ctx->pos_epoch;
    L->checked_pos = d_pos;
    if (!gathered && !copies_current)
        hipLaunchKernelGGL(k_gather_sorted, dim3(nb), dim3(256), 0, st, n, L->d_perm, d_pos, pf->d_q, pf->d_hsig,
                           pf->d_seps2, ctx->box, pf->d_posq_s, pf->d_lj_s);
    pf->s_begin = L->s_begin;
    pf->s_end = L->s_end;
    pf->lpa = L->lpa;
    pf->cap = L->cap;
    const int nslice = pf->s_end - pf->s_begin;
    const int per = amm_slice_per(n, ctx->world), nf = guest ? 2 : 1;
    double *out = d_force, *gout = g_force;
    if (exchange) {
        if (accumulate || g_accumulate || d_energy) {
            amm_set_error("exchanged evaluation: forces only, no accumulation");
            return 1;
        }
        if (ctx->pending.active) {
            amm_set_error("exchanged evaluation while the previous one still waits for amm_exchange_finish");
            return 1;
        }
        if (!ctx->d_xchg || ctx->xchg_doubles < (long long)ctx->world * 2 * per * 3) {
            amm_set_error("exchanged evaluation: bind an exchange buffer of world * 2 * ceil(n/world) * 3 doubles (amm_bind_exchange)");
            return 1;
        }
        out = ctx->d_xchg + (size_t)ctx->rank * nf * per * 3;       // this rank's chunk: [nf][per][3]
        gout = out + (size_t)per * 3;
    } else {
        if (!accumulate && ctx->world > 1) AMM_HIP(hipMemsetAsync(d_force, 0, sizeof(double) * 3 * (size_t)n, st));
        if (guest && !g_accumulate && ctx->world > 1) AMM_HIP(hipMemsetAsync(g_force, 0, sizeof(double) * 3 * (size_t)n, st));
    }
    if (nslice > 0) {
        PairArgs A;
        A.s_begin = pf->s_begin;
        A.s_end = pf->s_end;
        A.lpa_shift = ilog2(pf->lpa);
        A.cap = pf->cap;
        A.perm = L->d_perm;
        A.nl = L->d_nl;
        A.nnb = L->d_nnb_near;
        A.nnb_total = (pf == L && L->rnear_build > 0) ? L->d_nnb : nullptr;
        A.posq_s = pf->d_posq_s;
        A.lj_s = pf->d_lj_s;
        A.force = out;
        A.accumulate = accumulate;
        A.box = ctx->box;
        A.gforce = gout;
        A.gaccumulate = g_accumulate;
        A.gsame = (guest && g_force == d_force && !exchange) ? 1 : 0;
        A.sorted_out = exchange ? 1 : 0;
        A.active = nullptr;
        A.n_active = nullptr;
        A.n_long = nullptr;
        A.active_size = 0;
        A.long_shift = 0;
        int rows = nslice;
        if (L->d_active && !exchange && (!guest || L->hybrid_rest)) {
            // interaction-group force: walk the rows that hold entries, the others' forces are zero.  Those rows are few and
            // either long (a solute atom against all the solvent around it) or a handful of entries: a whole wavefront each.
            // The rest part of a hybrid list (rows of the atoms outside the molecules: long; rows of molecule atoms near them: their
            // few partners outside the molecules) is walked the same way with 8 lanes per row; it always adds to its parent's rows
            A.active = L->d_active;
            A.n_active = L->d_flags + 8;
            A.n_long = L->d_flags + 9;
            A.active_size = L->active_size;
            A.long_shift = (L->hybrid_rest && ctx->opt_row_phases) ? 6 : 0;
            if (!accumulate && ctx->world == 1) AMM_HIP(hipMemsetAsync(d_force, 0, sizeof(double) * 3 * (size_t)n, st));
            if (guest && !g_accumulate && g_force != d_force && ctx->world == 1) AMM_HIP(hipMemsetAsync(g_force, 0, sizeof(double) * 3 * (size_t)n, st));
            A.accumulate = 1;
            A.gaccumulate = 1;
            A.lpa_shift = L->hybrid_rest ? 3 : 6;
            rows = std::min(nslice, L->active_cap);
        }
        const long threads = (long)rows << A.lpa_shift;
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
        const bool timed = ctx->profile && (ctx->profile_only < 0 || ctx->profile_only == pf->id || ctx->profile_only == pf->profile_id);
        if (timed) {
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
        const int use_tab = ctx->opt_tab;
        // force-only, unguarded, ungrouped evaluations of the tabulated families: the kernel of pair_tab.h
        const bool tab_ok = use_tab && !en && !guard && !(pf->pc.flags & (AMM_GROUP_LJ | AMM_GROUP_Q)) && pf->pc.tab.nint > 0 && pf->d_tab &&
                            pf->pc.sign == 1.0 && (!guest || (guest->pc.tab.nint > 0 && guest->d_tab));
        if (tab_ok) {
            TabArgs T;
            T.host_tab = pf->d_tab;
            T.host_bytes = pf->pc.tab.nint * AMM_TAB_STRIDE;
            T.guest_tab = guest ? guest->d_tab : nullptr;
            T.guest_bytes = guest ? guest->pc.tab.nint * AMM_TAB_STRIDE : 0;
            T.need_erfcx = 1;
            T.margin = L->rlist_build + L->skin + 1e-6;
            T.row_order = A.active ? nullptr : L->d_row_order;
            T.n_lj = (L->d_row_order && !A.active) ? L->d_flags + 3 : nullptr;
            T.nslice = A.active ? rows : nslice;
            T.ntask = (int)((threads + 63) / 64);
            // the rows' site counts are the list owner's: a guest may cut its walk by them only if it has its sites on the same atoms
            if (pf->sites_match < 0) pf->sites_match = (pf == L || pf->h_cls == L->h_cls) ? 1 : 0;
            bool sites_ok = pf->sites_match == 1;
            if (guest) {
                if (guest->sites_match < 0) guest->sites_match = (guest == L || guest->h_cls == L->h_cls) ? 1 : 0;
                sites_ok = sites_ok && guest->sites_match == 1;
            }
            T.nnb_lj = (ctx->site_trips && !L->dual && sites_ok && !A.active) ? L->d_nnb_lj : nullptr;    // (the prune of a two-level list does not keep the counts)
            T.nnb_all = L->d_nnb;
            const int gfam = guest ? guest->desc.family : -1;
            PairConsts gpc = guest ? guest->pc : pf->pc;
            if (guest && (guest->desc.flags & AMM_GUARD_RC0)) gpc.rc2 = std::min(gpc.rc2, gpc.rc0 * gpc.rc0);   // step(rc0 - r)
            int rc_ = 0;
            // a force without charges (all q = 0: a Lennard-Jones fluid) on its own: the instantiation without the Coulomb table
            const bool noq = pf->all_q_zero && !guest && !A.active && tab_block_size(false) == 512 && ctx->opt_chargeless &&
                             pf->desc.family >= AMM_NEAR_NONE && pf->desc.family <= AMM_NEAR_FSWITCH;
            pf->last_chargeless = noq ? 1 : 0;
            // ... and the kicks (+ move) that follow the EVAL in the step program as the launch's epilogue (AtomEpiArgs), one rank
            bool aepi = false;
            AtomEpiArgs AE;
            if (noq && aplan && aplan->kind == 1 && ctx->opt_fuse_epilogue && ctx->world == 1 && !exchange && !accumulate && pf == L &&
                !L->dual && aplan->npre >= 1 && aplan->npre <= 4 && ctx->d_x == d_pos && ctx->d_v) {
                AE.x = ctx->d_x;
                AE.v = ctx->d_v;
                AE.mass = ctx->d_mass;
                AE.q = pf->d_q;
                AE.K.n = aplan->npre;
                aepi = true;
                for (int k = 0; k < 4; ++k) {
                    AE.K.f[k] = k < aplan->npre ? aplan->pre_a[k] : nullptr;
                    AE.K.f2[k] = k < aplan->npre ? aplan->pre_b[k] : nullptr;
                    AE.K.plus[k] = k < aplan->npre ? aplan->pre_plus[k] : 0;
                    AE.K.coef[k] = k < aplan->npre ? aplan->pre_coef[k] : 0.0;
                    if (k < aplan->npre && !AE.K.f[k]) aepi = false;
                }
                AE.with_move = aplan->with_move;
                AE.dcoef = aplan->dcoef;
                amm_collect_watches(ctx, AE.W);
                AE.posq_next = nullptr;
                if (aplan->with_move && aplan->next == pf) {
                    // (the next evaluation is this force's again: its positions go to the second copy, swapped in below; the
                    // parameter records stay -- the order of the rows does not change without a rebuild)
                    if (!pf->d_posq_alt) AMM_HIP(hipMalloc(&pf->d_posq_alt, sizeof(double4) * (size_t)n));
                    AE.posq_next = pf->d_posq_alt;
                }
            }
            if (noq) {
                T.host_bytes = 0;           // (nothing staged, no erfcx table needed: the analytic erfc families are not these)
                T.need_erfcx = 0;
                if (aepi) {
                    if (pf->desc.family == AMM_NEAR_NONE) rc_ = launch_pair_tab_i<AMM_NEAR_NONE, 0, -1, 512, false, true, true>(st, A, pf->pc, gpc, T, AE);
                    else if (pf->desc.family == AMM_NEAR_SHIFT) rc_ = launch_pair_tab_i<AMM_NEAR_SHIFT, 0, -1, 512, false, true, true>(st, A, pf->pc, gpc, T, AE);
                    else rc_ = launch_pair_tab_i<AMM_NEAR_FSWITCH, 0, -1, 512, false, true, true>(st, A, pf->pc, gpc, T, AE);
                    if (!rc_) {
                        ctx->epi_done = true;
                        ctx->n_epilogues++;
                        if (aplan->with_move) {
                            ctx->pos_epoch++;
                            amm_watch_moved(ctx);
                            if (AE.posq_next) {
                                std::swap(pf->d_posq_s, pf->d_posq_alt);
                                L->a_sorted_for = pf;
                                L->a_sorted_epoch = ctx->pos_epoch;
                                L->a_sorted_pos = ctx->d_x;
                            }
                        }
                    }
                } else if (pf->desc.family == AMM_NEAR_NONE) rc_ = launch_pair_tab_i<AMM_NEAR_NONE, 0, -1, 512, false, true>(st, A, pf->pc, gpc, T);
                else if (pf->desc.family == AMM_NEAR_SHIFT) rc_ = launch_pair_tab_i<AMM_NEAR_SHIFT, 0, -1, 512, false, true>(st, A, pf->pc, gpc, T);
                else rc_ = launch_pair_tab_i<AMM_NEAR_FSWITCH, 0, -1, 512, false, true>(st, A, pf->pc, gpc, T);
            } else
            switch (pf->desc.family) {
            case AMM_NEAR_NONE: rc_ = launch_pair_tab<AMM_NEAR_NONE, 0>(st, gfam, A, pf->pc, gpc, T); break;
            case AMM_NEAR_SHIFT: rc_ = launch_pair_tab<AMM_NEAR_SHIFT, 0>(st, gfam, A, pf->pc, gpc, T); break;
            case AMM_NEAR_FSWITCH: rc_ = launch_pair_tab<AMM_NEAR_FSWITCH, 0>(st, gfam, A, pf->pc, gpc, T); break;
            case AMM_DAMPED:
                if (pf->pc.degree == 1) rc_ = launch_pair_tab<AMM_DAMPED, 1>(st, gfam, A, pf->pc, gpc, T);
                else rc_ = launch_pair_tab<AMM_DAMPED, 0>(st, gfam, A, pf->pc, gpc, T);
                break;
            default:
                if (pf->pc.cmode == 1) rc_ = launch_pair_tab<AMM_NONBONDED, 1>(st, gfam, A, pf->pc, gpc, T);
                else if (pf->pc.cmode == 2) rc_ = launch_pair_tab<AMM_NONBONDED, 2>(st, gfam, A, pf->pc, gpc, T);
                else rc_ = launch_pair_tab<AMM_NONBONDED, 0>(st, gfam, A, pf->pc, gpc, T);
            }
            if (rc_) return 1;
            if (guest) {
                guest->n_evals++;
                guest->last_fused = 1;
            }
        } else if (guest) {
            if (A.gsame || (guest->desc.flags & AMM_GUARD_RC0) || guest->pc.sign != 1.0) {
                amm_set_error("dual evaluation of a guarded / signed guest needs the tabulated kernel (AMM_TAB=0?)");
                return 1;
            }
            int rc_ = 0;
            // DAMPED: CMODE 1 = the degree-1 specialisation (built-in switch in r: no power loop, no int -> double per pair)
            if (pf->desc.family == AMM_DAMPED && pf->pc.degree == 1) rc_ = launch_pair_dual<AMM_DAMPED, 1>(grid, block, st, guest->desc.family, A, pf->pc, guest->pc);
            else if (pf->desc.family == AMM_DAMPED) rc_ = launch_pair_dual<AMM_DAMPED, 0>(grid, block, st, guest->desc.family, A, pf->pc, guest->pc);
            else if (pf->desc.family == AMM_NONBONDED && pf->pc.cmode == 1)
                rc_ = launch_pair_dual<AMM_NONBONDED, 1>(grid, block, st, guest->desc.family, A, pf->pc, guest->pc);
            else if (pf->desc.family == AMM_NONBONDED && pf->pc.cmode == 2)
                rc_ = launch_pair_dual<AMM_NONBONDED, 2>(grid, block, st, guest->desc.family, A, pf->pc, guest->pc);
            else if (pf->desc.family == AMM_NONBONDED)
                rc_ = launch_pair_dual<AMM_NONBONDED, 0>(grid, block, st, guest->desc.family, A, pf->pc, guest->pc);
            else {
                amm_set_error("dual evaluation: unsupported host family");
                rc_ = 1;
            }
            if (rc_) return 1;
            guest->n_evals++;
            guest->last_fused = 1;
        } else if (pf->pc.flags & (AMM_GROUP_LJ | AMM_GROUP_Q)) {
            if (pf->desc.family == AMM_NEAR_FSWITCH) launch_pair_grouped<AMM_NEAR_FSWITCH>(grid, block, st, guard, en, A, pf->pc);
            else if (pf->desc.family == AMM_NONBONDED && pf->pc.cmode == 0) launch_pair_grouped<AMM_NONBONDED>(grid, block, st, false, en, A, pf->pc);
            else {
                amm_set_error("interaction groups (AMM_GROUP_LJ / AMM_GROUP_Q) are supported for the NEAR_FSWITCH and plain NONBONDED families");
                return 1;
            }
        } else
        switch (pf->desc.family) {
        case AMM_NEAR_NONE: launch_pair<AMM_NEAR_NONE, 0>(grid, block, st, guard, en, A, pf->pc); break;
        case AMM_NEAR_SHIFT: launch_pair<AMM_NEAR_SHIFT, 0>(grid, block, st, guard, en, A, pf->pc); break;
        case AMM_NEAR_FSWITCH: launch_pair<AMM_NEAR_FSWITCH, 0>(grid, block, st, guard, en, A, pf->pc); break;
        case AMM_DAMPED:
            if (pf->pc.degree == 1) launch_pair<AMM_DAMPED, 1>(grid, block, st, false, en, A, pf->pc);
            else launch_pair<AMM_DAMPED, 0>(grid, block, st, false, en, A, pf->pc);
            break;
        case AMM_SOFTCORE:
            if (pf->d_lambda_dev) {       // (amm_pair_set_lambda_dev: only the list-free kernel reads lambda from the device)
                amm_set_error("softcore force with lambda on the device evaluated through its neighbour list: call amm_pair_set_lambda first");
                return 1;
            }
            launch_pair<AMM_SOFTCORE, 0>(grid, block, st, false, en, A, pf->pc);
            break;
        case AMM_LJ_VIRIAL: launch_pair<AMM_LJ_VIRIAL, 0>(grid, block, st, false, en, A, pf->pc); break;
        case AMM_NONBONDED:
            if (pf->pc.cmode == 1) launch_pair<AMM_NONBONDED, 1>(grid, block, st, false, en, A, pf->pc);
            else if (pf->pc.cmode == 2) launch_pair<AMM_NONBONDED, 2>(grid, block, st, false, en, A, pf->pc);
            else launch_pair<AMM_NONBONDED, 0>(grid, block, st, false, en, A, pf->pc);
            break;
        default: amm_set_error("unknown pair family"); return 1;
        }
        if (timed) AMM_HIP(hipEventRecord(e1, st));
        AMM_HIP(hipGetLastError());
        if (en) {
            if (amm_reduce_add(ctx, pf->d_epart, nblk, 1.0, d_energy)) return 1;
        }
    }
    pf->n_evals++;
    if (exchange) {
        PendingExchange &pe = ctx->pending;
        pe.active = true;
        pe.per = per;
        pe.nf = nf;
        pe.perm = L->d_perm;
        pe.force = d_force;
        pe.gforce = g_force;
        // with a communicator of its own the library completes the exchange; otherwise the host gathers the chunks
        // (torch.distributed, MPI ...) and calls amm_exchange_finish
        if (ctx->comm) {
            if (amm_comm_allgather_impl(ctx, ctx->d_xchg, (size_t)nf * per * 3)) return 1;
            if (amm_exchange_finish_impl(ctx)) return 1;
        }
    }
    return 0;
}

// Can `guest` be evaluated on the pass of its list owner `host`?  Same particles (bitwise equal parameters), plain
// near family without the rc0 guard on the guest, unguarded DAMPED / NONBONDED host, guest cutoff inside the front
// part of the rows.
bool amm_pair_can_eval_dual(amm_ctx *ctx, PairForce *guest, PairForce *host) {
    if (!guest || !host || guest->host != host || host->host) return false;
    if (guest->dual_ok >= 0) return guest->dual_ok == 1;
    guest->dual_ok = 0;
    const int gf = guest->desc.family, hf = host->desc.family;
    if (!(gf == AMM_NEAR_NONE || gf == AMM_NEAR_SHIFT || gf == AMM_NEAR_FSWITCH)) return false;
    if (!(hf == AMM_DAMPED || hf == AMM_NONBONDED)) return false;
    if ((guest->desc.flags & AMM_GUARD_RC0) || (host->desc.flags & AMM_GUARD_RC0)) return false;
    if ((guest->desc.flags | host->desc.flags) & (AMM_GROUP_LJ | AMM_GROUP_Q)) return false;
    if (!(guest->desc.rc <= host->desc.rc)) return false;
    if (guest->pc.Kc != host->pc.Kc) return false;          // the pass forms Kc q_i q_j once, with the host's constant
    const size_t n = (size_t)ctx->n;
    std::vector<double> a(n), b(n);
    const double *ga[3] = {guest->d_q, guest->d_hsig, guest->d_seps2}, *ha[3] = {host->d_q, host->d_hsig, host->d_seps2};
    for (int k = 0; k < 3; ++k) {
        if (hipMemcpy(a.data(), ga[k], sizeof(double) * n, hipMemcpyDeviceToHost) != hipSuccess) return false;
        if (hipMemcpy(b.data(), ha[k], sizeof(double) * n, hipMemcpyDeviceToHost) != hipSuccess) return false;
        if (std::memcmp(a.data(), b.data(), sizeof(double) * n) != 0) return false;
    }
    guest->dual_ok = 1;
    return true;
}

// measurement helper: directed list entries within r_within of the CURRENT sorted positions (fp64, minimum image)
__global__ void __launch_bounds__(256) k_count_within(PairArgs A, double r2w, int nslice, unsigned long long *out) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int a = tid >> A.lpa_shift, sub = tid & ((1 << A.lpa_shift) - 1);
    unsigned long long cnt = 0;
    if (a < nslice) {
        const int s = A.s_begin + a;
        const double4 pi = A.posq_s[s];
        const int nfront = A.nnb[a], nn = A.nnb_total ? A.nnb_total[a] : nfront;
        const int *row = A.nl + (size_t)a * A.cap;
        const int back = A.cap - 1 + nfront;
        for (int k = sub; k < nn; k += 1 << A.lpa_shift) {
            const double4 pj = A.posq_s[row[k < nfront ? k : back - k]];
            const double dx = amm_min_image(pi.x - pj.x, A.box.L[0], A.box.invL[0]);
            const double dy = amm_min_image(pi.y - pj.y, A.box.L[1], A.box.invL[1]);
            const double dz = amm_min_image(pi.z - pj.z, A.box.L[2], A.box.invL[2]);
            cnt += (dx * dx + dy * dy + dz * dz < r2w) ? 1ull : 0ull;
        }
    }
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(out, cnt);
}

int amm_pair_count_within_impl(amm_ctx *ctx, PairForce *pf, const double *d_pos, double r_within, long long *count) {
    PairForce *L = pf->host ? pf->host : pf;
    if (!L->built) {
        amm_set_error("amm_pair_count_within: evaluate the force once first (the neighbour rows do not exist yet)");
        return 1;
    }
    hipStream_t st = ctx->stream;
    const int n = pf->n;
    hipLaunchKernelGGL(k_gather_sorted, dim3((n + 255) / 256), dim3(256), 0, st, n, L->d_perm, d_pos, pf->d_q, pf->d_hsig,
                       pf->d_seps2, ctx->box, pf->d_posq_s, pf->d_lj_s);
    PairArgs A;
    std::memset(&A, 0, sizeof(A));
    A.s_begin = L->s_begin;
    A.s_end = L->s_end;
    A.lpa_shift = 3;
    A.cap = L->cap;
    A.nl = L->d_nl;
    A.nnb = L->d_nnb_near;
    A.nnb_total = (pf == L && L->rnear_build > 0) ? L->d_nnb : nullptr;
    A.posq_s = pf->d_posq_s;
    A.box = ctx->box;
    const int nslice = L->s_end - L->s_begin;
    unsigned long long *d_cnt = L->d_counters + 7;
    AMM_HIP(hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long), st));
    if (nslice > 0)
        hipLaunchKernelGGL(k_count_within, dim3((unsigned)(((long)nslice * 8 + 255) / 256)), dim3(256), 0, st, A, r_within * r_within, nslice, d_cnt);
    unsigned long long h = 0;
    AMM_HIP(hipMemcpyAsync(&h, d_cnt, sizeof(h), hipMemcpyDeviceToHost, st));
    AMM_HIP(hipStreamSynchronize(st));
    *count = (long long)h;
    return 0;
}

// bump when a pair-traversal kernel changes: stored measurements (profiles/*_traffic.json) are matched against it
// A tuning or measurement build says so in its tag: scripts/build_variant.sh links a marker symbol into every variant library, and
// AMM_CLUSTER_TUNE is a variant by definition; atomsmm_amd/backend.py refuses a "-tune" library unless AMM_ALLOW_TUNE=1 (the probe
// scripts set it), so neither bench.py nor the tests can run on one by accident.
extern "C" __attribute__((weak)) const char *amm_variant_tag(void);
const char *amm_kernel_revision_impl() {
#ifdef AMM_CLUSTER_TUNE
    return "r05-epi5-tune";
#else
    return amm_variant_tag ? "r05-epi5-tune" : "r05-epi5";
#endif
}

// radial Coulomb table of the force-only traversal (pair_tab.h): built from the descriptor alone, once per pair force
int amm_pair_build_table(PairForce *pf) {
    std::vector<double> coef;
    // the site-site table (pair_tab.h) for molecule-row forces whose Lennard-Jones sites share one (sigma, eps, q)
    SiteTable ss;
    ss.want = pf->cluster_ok && pf->one_site_class && pf->site_one_charge && pf->site_q != 0.0 && pf->site_seps2 != 0.0;
    ss.sig = 2.0 * pf->site_hsig;
    ss.eps4 = pf->site_seps2 * pf->site_seps2;
    ss.QQ = pf->pc.Kc * pf->site_q * pf->site_q;
    pf->tab_error = amm_build_coulomb_table(pf->pc, coef, &ss);
    pf->ss_built_for[0] = pf->site_hsig;
    pf->ss_built_for[1] = pf->site_seps2;
    pf->ss_built_for[2] = ss.want ? pf->site_q : 0.0;
    if (pf->d_tab) {
        (void)hipFree(pf->d_tab);
        pf->d_tab = nullptr;
    }
    if (pf->d_tab_ss) {
        (void)hipFree(pf->d_tab_ss);
        pf->d_tab_ss = nullptr;
    }
    pf->ss_error = ss.error;
    // The bound is enforced, not assumed: the refinement stops at the LDS budget (or at its finest level), and a narrow
    // switching window or a high DAMPED degree can leave the table short of the arithmetic's accuracy.  Such a force keeps
    // the analytic kernels (no table: `tab_ok` is false for it, and a guest without a table disables the one-pass forms).
    if (pf->pc.tab.nint > 0 && !(pf->tab_error <= AMM_TAB_MAX_ERROR)) {
        pf->pc.tab.nint = 0;
        coef.clear();
    }
    if (pf->pc.tab.nint == 0 || ss.coef.empty() || !(ss.error <= AMM_TAB_SS_MAX_ERROR)) {
        pf->pc.tab.ss_first = -1;             // site pairs keep the analytic Lennard-Jones arithmetic
        ss.coef.clear();
    }
    if (pf->pc.tab.nint > 0) {
        AMM_HIP(hipMalloc(&pf->d_tab, sizeof(double) * coef.size()));
        AMM_HIP(hipMemcpy(pf->d_tab, coef.data(), sizeof(double) * coef.size(), hipMemcpyHostToDevice));
    }
    if (!ss.coef.empty()) {
        AMM_HIP(hipMalloc(&pf->d_tab_ss, sizeof(double) * ss.coef.size()));
        AMM_HIP(hipMemcpy(pf->d_tab_ss, ss.coef.data(), sizeof(double) * ss.coef.size(), hipMemcpyHostToDevice));
    }
    return 0;
}

// Can `guest` -- a guarded near force such as the discount of FarNonbondedForce -- ride on the pass of its list owner
// `host`, accumulating into the same buffer?  As amm_pair_can_eval_dual, with the guard allowed (it becomes the guest's
// cutoff) and any sign; needs the tabulated kernel on both.
bool amm_pair_can_fuse_discount(amm_ctx *ctx, PairForce *guest, PairForce *host) {
    if (!guest || !host || guest->host != host || host->host) return false;
    if (!(guest->desc.flags & AMM_GUARD_RC0)) return false;
    if (guest->fuse_ok >= 0) return guest->fuse_ok == 1;
    guest->fuse_ok = 0;
    if (!ctx->opt_tab) return false;
    const int gf = guest->desc.family, hf = host->desc.family;
    if (!(gf == AMM_NEAR_NONE || gf == AMM_NEAR_SHIFT || gf == AMM_NEAR_FSWITCH)) return false;
    if (!(hf == AMM_DAMPED || hf == AMM_NONBONDED)) return false;
    if ((host->desc.flags & AMM_GUARD_RC0) || host->pc.sign != 1.0) return false;
    if ((guest->desc.flags | host->desc.flags) & (AMM_GROUP_LJ | AMM_GROUP_Q)) return false;
    if (!(guest->pc.tab.nint > 0 && guest->d_tab && host->pc.tab.nint > 0 && host->d_tab)) return false;
    if (guest->pc.Kc != host->pc.Kc) return false;
    const size_t n = (size_t)ctx->n;
    std::vector<double> a(n), b(n);
    const double *ga[3] = {guest->d_q, guest->d_hsig, guest->d_seps2}, *ha[3] = {host->d_q, host->d_hsig, host->d_seps2};
    for (int k = 0; k < 3; ++k) {
        if (hipMemcpy(a.data(), ga[k], sizeof(double) * n, hipMemcpyDeviceToHost) != hipSuccess) return false;
        if (hipMemcpy(b.data(), ha[k], sizeof(double) * n, hipMemcpyDeviceToHost) != hipSuccess) return false;
        if (std::memcmp(a.data(), b.data(), sizeof(double) * n) != 0) return false;
    }
    guest->fuse_ok = 1;
    return true;
}

int amm_pair_free(PairForce *pf) {
    void *ptrs[] = {pf->d_q, pf->d_hsig, pf->d_seps2, pf->d_excl_ptr, pf->d_excl_idx, pf->d_cell_of, pf->d_cell_count,
                    pf->d_cell_start, pf->d_cell_members, pf->d_perm, pf->d_posq_s, pf->d_lj_s, pf->d_xref,
                    pf->d_nl, pf->d_nnb, pf->d_flags, pf->d_counters, pf->d_epart, pf->d_pos4f_s, pf->d_blockstats, pf->d_inv_perm, pf->d_nnb_near, pf->d_nl_out, pf->d_nnb_out,
                    pf->d_nnb_scratch, pf->d_xref_out, pf->d_ticket, pf->d_tab, pf->d_cls, pf->d_cell_count_lj, pf->d_cell_start_lj,
                    pf->d_row_order, pf->d_member, pf->d_active, pf->d_cell_sets, pf->d_nnb_lj, pf->d_mol_first, pf->d_rest_idx};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (hipEvent_t e : pf->ev) (void)hipEventDestroy(e);
    if (pf->cl) amm_cluster_free(pf->cl);
    pf->cl = nullptr;
    if (pf->small) amm_small_group_free(pf->small);
    pf->small = nullptr;
    return 0;
}

// exported for abi.hip
int amm_pair_setup_grid(amm_ctx *ctx, PairForce *pf) { return setup_grid(ctx, pf); }
