// atomsmm_amd/csrc/cluster.hip -- molecule-row neighbour lists and their force-only traversal (gfx950, fp64).  See cluster.h.
//
// Takes over, for the three-site molecules of any particle list (cluster.h), what OpenMM's neighbour search + per-pair evaluation do for the reference's
// CustomNonbondedForce / NonbondedForce objects on the RESPA hot path (forces.py:448-455, 539-567, 655-670, 710-724;
// systems.py:71-77): same pairs (exclusions = the three pairs inside each molecule), same per-pair arithmetic as pair.hip's
// tabulated kernel (pair_tab.h), another decomposition of the work.
//
// Data layout in HBM (per list owner; cluster c = sorted molecule, slot 3 c + a = its atom a):
//   cperm[c]             molecule index of sorted cluster c (cell-sorted by the molecule's first atom; within a cell by index)
//   pos4f[3 c + a]       fp32 position at the last build, molecule kept whole (image of its first atom); .w of atom 0: extent
//                        (largest distance of an atom from the first), .w of atom 1: bit b set = atom b has a Lennard-Jones site
//   posq_s / lj_s[3c+a]  fp64 sorted copies (x, y, z, q) and (sigma/2, 2 sqrt(eps)), refreshed before every evaluation
//   nl[row cap + k]      int32 entry = partner cluster | site bits of the partner << 29; partners whose closest atom pair is within
//                        the GUEST force's list radius fill the row from the front, the others from the back
// Work decomposition of the traversal: lpa (8) lanes share one row; a lane takes one partner molecule per trip, fetches its three
// records (96 contiguous bytes) once and evaluates the nine atom pairs against the row's three atoms held in registers; the 9 (+ 9
// for the guest force) partial force components are combined over the lanes with wavefront shuffles at the end of the row.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "amm_ctx.h"
#include "bonded_terms.h"
#include "cluster.h"
#include "device_utils.h"
#include "pair_math.h"
#include "pair_tab.h"

#ifndef AMM_EXP_NOLJ
#define AMM_EXP_NOLJ 0        // measurement only: no Lennard-Jones arithmetic at all (wrong forces)
#endif

__device__ double amm_erfcx_table_dev_c[AMM_ERFCX_NI * AMM_ERFCX_NC];
static bool g_erfcx_uploaded_c[64] = {false};

// ------------------------------------------------------------------------------------------------ classification (host)
// Three-site molecules of a force: atoms i, i + 1, i + 2 whose only exclusions are their own three pairs (the reference turns
// every exception into an exclusion, forces.py:310-312, so this is a flexible or rigid three-site water).  Every other atom --
// an ion, a solute, a chain, a four-site water -- is "rest": its pairs go through per-atom rows (the force's hidden child).
void amm_cluster_classify(int n, const std::vector<int> &ptr, const std::vector<int> &idx, std::vector<int> &mol_first, std::vector<int> &rest) {
    mol_first.clear();
    rest.clear();
    auto own_pairs_only = [&](int i, int i0) {
        if (ptr[i + 1] - ptr[i] != 2) return false;
        const int a = i - i0;
        const int p0 = i0 + (a == 0 ? 1 : 0), p1 = i0 + (a == 2 ? 1 : 2);
        const int e0 = idx[ptr[i]], e1 = idx[ptr[i] + 1];
        return (e0 == p0 && e1 == p1) || (e0 == p1 && e1 == p0);
    };
    int i = 0;
    while (i < n) {
        if (i + 2 < n && own_pairs_only(i, i) && own_pairs_only(i + 1, i) && own_pairs_only(i + 2, i)) {
            mol_first.push_back(i);
            i += 3;
        } else {
            rest.push_back(i);
            i += 1;
        }
    }
}

// ------------------------------------------------------------------------------------------------ cell list of molecules
__device__ __forceinline__ double cwrap1(double x, double L, double invL) {
    double w = x - L * floor(x * invL);
    if (w >= L) w -= L;
    if (w < 0.0) w = 0.0;
    return w;
}

__global__ void k_ccheck_displacement(int n, const double *__restrict__ pos, const double *__restrict__ xref, double thr2, int *flags) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double dx = pos[3 * i] - xref[3 * i], dy = pos[3 * i + 1] - xref[3 * i + 1], dz = pos[3 * i + 2] - xref[3 * i + 2];
    const double d2 = dx * dx + dy * dy + dz * dz;
    if (!(d2 <= thr2)) {                                             // benign race; NaN also triggers
        flags[0] = 1;
        if (!(d2 <= 4.0 * thr2)) flags[AMM_FLAG_FAR] = 1;
    }
}

// cell of every molecule (by its first atom) + per-cell counts; the arrival rank places the molecule in the cell's member table;
// the last block scans the counts (device_utils.h).  members == nullptr: sizing pass (counts only).
// first atom of molecule m: molecules are 3 m, 3 m + 1, 3 m + 2 in a pure water box (first == nullptr); a hybrid list
// (molecule rows for the three-site molecules, per-atom rows for the rest) carries the table
__device__ __forceinline__ int cfirst(const int *__restrict__ first, int m) { return first ? first[m] : 3 * m; }

__global__ void __launch_bounds__(256) k_cassign(int nc, const double *__restrict__ pos, Box box, CellGrid g, int *count, int *start,
                                                 int *members, int capc, double *xref, int *flags, int *ticket, int force,
                                                 const int *__restrict__ first, const int *__restrict__ rest, int nrest) {
    if (!force && !flags[0]) return;
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= nc && m < nc + nrest && xref) {       // hybrid list: the displacement trigger watches every atom of the force
        const int i = rest[m - nc];
#pragma unroll
        for (int k = 0; k < 3; ++k) xref[3 * i + k] = pos[3 * i + k];
    }
    if (m < nc) {
        const int i0 = cfirst(first, m);
        int cidx[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double w = cwrap1(pos[3 * i0 + k], box.L[k], box.invL[k]);
            const int ck = (w == w) ? (int)(w * g.inv_cw[k]) : 0;          // NaN-safe
            cidx[k] = ck >= g.nc[k] ? g.nc[k] - 1 : (ck < 0 ? 0 : ck);
        }
        // (the molecules' reference positions are written by the sort launch of the rebuild: k_csort_gather)
        const int cell = (cidx[2] * g.nc[1] + cidx[1]) * g.nc[0] + cidx[0];
        const int rank = atomicAdd(&count[cell], 1);
        if (members) {
            if (rank < capc) members[(size_t)cell * capc + rank] = m;
            else flags[7] = 1;
        }
    }
    if (!amm_last_block(ticket)) return;
    const int fullest = amm_block_scan_counts(g.ncell, count, start);
    if (threadIdx.x == 0) flags[6] = fullest;
}

// coordinate of an atom in its molecule's sorted copy: the image of the molecule's first atom (x0; sh = wrapped(x0) - x0).  An atom
// that was wrapped into the box on its own comes back to its molecule: the rint is 0 for a whole molecule and the sum then has the
// bits of x + sh.  ONE function for every writer of the sorted copies (the gather launch, the rebuild, the pair kernels' epilogue):
// they must agree bit for bit
__device__ __forceinline__ double csorted_image(double x, double x0, double sh, double L, double invL) {
    return x + (sh - L * rint((x - x0) * invL));
}

// sorted fp64 copies of one molecule (kept whole: every atom takes the image of the first)
__device__ __forceinline__ void cgather_one(int c, int i0, const double *__restrict__ pos, Box box, const double *__restrict__ q,
                                            const double *__restrict__ hsig, const double *__restrict__ seps2, double4 *posq_s, double2 *lj_s) {
    double sh[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) sh[k] = cwrap1(pos[3 * i0 + k], box.L[k], box.invL[k]) - pos[3 * i0 + k];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int i = i0 + a;
        double4 p;
        p.x = csorted_image(pos[3 * i], pos[3 * i0], sh[0], box.L[0], box.invL[0]);
        p.y = csorted_image(pos[3 * i + 1], pos[3 * i0 + 1], sh[1], box.L[1], box.invL[1]);
        p.z = csorted_image(pos[3 * i + 2], pos[3 * i0 + 2], sh[2], box.L[2], box.invL[2]);
        p.w = q[i];
        posq_s[3 * c + a] = p;
        lj_s[3 * c + a] = make_double2(hsig[i], seps2[i]);
    }
}

// rebuild: one wavefront per cell ranks the cell's molecules by index (deterministic whatever the atomics did), writes the
// permutation, the fp32 copies with extent and site bits, and the fp64 sorted copies of the evaluation that follows.
// No rebuild: only those fp64 copies.
__global__ void __launch_bounds__(256) k_csort_gather(int ncell, int nc, const int *__restrict__ start, const int *__restrict__ members,
                                                      int capc, int *cperm, int *aperm, const double *__restrict__ pos, Box box,
                                                      float4 *pos4f, const int *flags, int *wflags, int force, const double *__restrict__ q,
                                                      const double *__restrict__ hsig, const double *__restrict__ seps2, double4 *posq_s,
                                                      double2 *lj_s, float rext, const double *__restrict__ site_eps,
                                                      const int *__restrict__ first, CZeroRows Z, int copies_current, double *xref,
                                                      int c_begin, int c_end, int *slice_cells) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    // (hybrid lists: the force rows of the atoms outside the molecules start from zero; the molecule-row kernel writes the others
    // and the per-atom part adds to all of them -- a launch of its own before)
    if (gid < Z.n) {
        const int i = Z.idx[gid];
        if (Z.f0) Z.f0[3 * i] = Z.f0[3 * i + 1] = Z.f0[3 * i + 2] = 0.0;
        if (Z.f1) Z.f1[3 * i] = Z.f1[3 * i + 1] = Z.f1[3 * i + 2] = 0.0;
    }
    if (!force && !flags[0]) {
        // (copies_current: the launch that moved the atoms wrote the copies of these positions already -- cepi_rows)
        if (!copies_current && posq_s && gid < nc) cgather_one(gid, cfirst(first, cperm[gid]), pos, box, q, hsig, seps2, posq_s, lj_s);
        return;
    }
    const int wave = gid >> 6, lane = threadIdx.x & 63;
    if (wave >= ncell) return;
    const int b = start[wave];
    const int cnt = min(start[wave + 1] - b, capc);
    // the cells that hold the first and the last row of this rank's slice: the build launches blocks for those cells only
    if (lane == 0 && c_begin < c_end) {
        if (b <= c_begin && c_begin < start[wave + 1]) slice_cells[0] = wave;
        if (b <= c_end - 1 && c_end - 1 < start[wave + 1]) slice_cells[1] = wave;
    }
    const int *mem = members + (size_t)wave * capc;
    for (int a0 = 0; a0 < cnt; a0 += 64) {
        const int a = a0 + lane;
        const int me = a < cnt ? mem[a] : 0x7fffffff;
        int rank = 0;
        for (int k0 = 0; k0 < cnt; k0 += 64) {
            const int kk = k0 + lane;
            const int other = kk < cnt ? mem[kk] : 0x7fffffff;
            const int nk = min(64, cnt - k0);
            for (int jj = 0; jj < nk; ++jj) rank += __builtin_amdgcn_readlane(other, jj) < me;
        }
        if (a >= cnt) continue;
        const int sl = b + rank;
        cperm[sl] = me;
        const int i0 = cfirst(first, me);
        double sh[3], p0[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            p0[k] = pos[3 * i0 + k];
            sh[k] = cwrap1(p0[k], box.L[k], box.invL[k]) - p0[k];
        }
        float ext2 = 0.f;
        int sites = 0;
        float4 pf[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int i = i0 + t;
            aperm[3 * sl + t] = i;
            const double x = csorted_image(pos[3 * i], p0[0], sh[0], box.L[0], box.invL[0]);
            const double y = csorted_image(pos[3 * i + 1], p0[1], sh[1], box.L[1], box.invL[1]);
            const double z = csorted_image(pos[3 * i + 2], p0[2], sh[2], box.L[2], box.invL[2]);
            if (xref) {        // the displacement triggers' reference: the positions of this build
                xref[3 * i] = pos[3 * i];
                xref[3 * i + 1] = pos[3 * i + 1];
                xref[3 * i + 2] = pos[3 * i + 2];
            }
            pf[t] = make_float4((float)x, (float)y, (float)z, 0.f);
            const float dx = pf[t].x - pf[0].x, dy = pf[t].y - pf[0].y, dz = pf[t].z - pf[0].z;
            ext2 = fmaxf(ext2, dx * dx + dy * dy + dz * dz);
            if (site_eps[i] != 0.0) sites |= 1 << t;        // the LIST OWNER's site pattern (a guest must share it: checked by the caller)
            if (posq_s) {
                posq_s[3 * sl + t] = make_double4(x, y, z, q[i]);
                lj_s[3 * sl + t] = make_double2(hsig[i], seps2[i]);
            }
        }
        const float ext = sqrtf(ext2) * 1.000001f + 1e-6f;
        if (!(ext <= rext)) wflags[7] = 1;               // a molecule stretched beyond the bound the cells were sized for
        pf[0].w = ext;
        pf[1].w = __int_as_float(sites);
        pos4f[3 * sl] = pf[0];
        pos4f[3 * sl + 1] = pf[1];
        pos4f[3 * sl + 2] = pf[2];
    }
}

struct CBoxF {
    float L[3], invL[3];
};

// ------------------------------------------------------------------------------------------------ list build
// One wavefront per (cell, part); a batch of CB_BATCH row molecules lives in scalar registers.  Two passes over the candidates:
//   1. the stencil's molecules are walked as ONE concatenated stream (lane = candidate; piece table + binary search as in
//      pair.hip's build) and tested by their FIRST atoms against a sphere that is guaranteed to contain every partner
//      (|x0_i - x0_j| < rlist + ext_i + ext_j), two chunks of 64 side by side in packed fp32 (v_pk_*_f32): 9 instructions per 128
//      candidates and row molecule + the filing of the hits; the survivors (a fifth of the stream) are queued in LDS, per row
//      molecule (a ring of 256);
//   2. as soon as a ring holds 128 it is drained with ALL 64 lanes busy, two survivors per lane (packed again): three records per
//      candidate, the nine atom-pair distances, the smallest decides (< rlist: listed; < rnear: front part), ordered ballot
//      compaction into the row.
// A candidate's piece of the stream (which stencil cell, hence which periodic shift) comes from a coarse table (piece of every
// 8th stream position, built once per wavefront) + a short walk, not from a binary search per candidate.
// The old per-atom build spent 42 instructions per 64 ATOM-pair tests, mostly compaction; here the compaction is paid per
// MOLECULE pair and only for a stream that is already 80 % hits.
#ifndef CB_BATCH
#define CB_BATCH 5
#endif
#define CB_COARSE 256       // entries of the coarse piece table
#define CB_QCAP 256         // ring of survivors per row molecule: >= 127 + 128 (a drain leaves at most 127 behind, a chunk pair adds 128)
__device__ __forceinline__ int cq_wrap(int i) { return i & (CB_QCAP - 1); }
typedef float v2f __attribute__((ext_vector_type(2)));        // two fp32 per lane: v_pk_add / v_pk_mul / v_pk_fma_f32 at the rate of one

__device__ void cfinish_build_block(int *flags, unsigned long long *counters, const unsigned long long *blockstats, int nblocks, int count_only) {
    __shared__ unsigned long long sh_part[4][3];
    unsigned long long sum = 0, mx = 0, nr = 0;
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
    for (int off = 32; off > 0; off >>= 1) {
        sum += __shfl_xor(sum, off);
        nr += __shfl_xor(nr, off);
        mx = max(mx, __shfl_xor(mx, off));
    }
    if ((threadIdx.x & 63) == 0) {
        sh_part[threadIdx.x >> 6][0] = sum;
        sh_part[threadIdx.x >> 6][1] = mx;
        sh_part[threadIdx.x >> 6][2] = nr;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        flags[2] = (int)max(max(sh_part[0][1], sh_part[1][1]), max(sh_part[2][1], sh_part[3][1]));
        counters[1] = sh_part[0][0] + sh_part[1][0] + sh_part[2][0] + sh_part[3][0];
        counters[2] = sh_part[0][2] + sh_part[1][2] + sh_part[2][2] + sh_part[3][2];
        if (!count_only) {
            flags[0] = 0;
            flags[AMM_FLAG_FAR] = 0;
            counters[0] += 1;
        }
    }
}

// (the packed drain wants 104 registers = 4 wavefronts per SIMD; at 96 = 5 the rebuild of the 98 304-atom box is 10 us shorter)
#ifndef CB_WAVES
#define CB_WAVES 5
#endif
#define CB_OCC __attribute__((amdgpu_waves_per_eu(CB_WAVES)))
#ifdef AMM_CBS_TIMING          // measurement builds: clock sums of the phases over the first wavefronts of the busy blocks
__device__ unsigned long long g_cbs_t[8];
#define CBS_STAMP(i) do { const unsigned long long now_ = wall_clock64(); if (w == 0 && lane == 0) atomicAdd(&g_cbs_t[i], now_ - t_last); t_last = now_; } while (0)
#else
#define CBS_STAMP(i) do { } while (0)
#endif
// SLICE (a rank's slice of the rows): the grid covers the cells of the slice only -- k_csort_gather recorded the first and the last
// (slice_cells) -- with a stride over the (cell, part) units in case the slice spans more cells than the host allowed for; a grid
// over every cell of the box spent a third of the slice's build on dispatching blocks that return at once.
template <bool COUNT_ONLY, bool RINT, bool SLICE = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SLICE ? 3 : CB_WAVES))) k_cbuild(int c_begin, int c_end, int parts, const int *__restrict__ cell_start,
                                                const float4 *__restrict__ pos4f, CBoxF box, CellGrid g, float rlist, float rnear2, int cap,
                                                int *nl, int *nnb, int *nnb_near, int *flags, unsigned long long *blockstats,
                                                unsigned long long *counters, int *ticket, int force, const int *slice_cells) {
    if (!force && !flags[0]) return;
    __shared__ int s_rstart[4][128];
    __shared__ int s_rpref[4][128];
    __shared__ float s_rshift[4][3][128];
    __shared__ int s_q[4][CB_BATCH][CB_QCAP];        // per row molecule: ring of survivors of the sphere test (sorted slots)
    __shared__ unsigned char s_coarse[4][CB_COARSE]; // piece of stream position e << cshift
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    const float FAR = 1.0e9f;
    const float rlist2 = rlist * rlist;
    unsigned long long wsum = 0, wnear = 0;
    int wmax = 0;
    const int first_cell = SLICE ? __builtin_amdgcn_readfirstlane(slice_cells[0]) : 0;
    const int nunits = SLICE ? (__builtin_amdgcn_readfirstlane(slice_cells[1]) - first_cell + 1) * parts : g.ncell * parts;
    int unit = wave;
#ifdef AMM_CBS_TIMING
    unsigned long long t_last = wall_clock64(), t_drain = 0, n_drain = 0;
#endif
    do {
    const int c = unit < nunits ? first_cell + unit / parts : g.ncell, part = unit % parts;
    int a_begin = 0, a_end = 0;
    if (c < g.ncell) {
        const int cb0 = __builtin_amdgcn_readfirstlane(cell_start[c]), cb1 = __builtin_amdgcn_readfirstlane(cell_start[c + 1]);
        const int per = (cb1 - cb0 + parts - 1) / parts;
        a_begin = max(cb0 + part * per, c_begin);
        a_end = min(min(cb0 + (part + 1) * per, cb1), c_end);
    }
    if (a_begin < a_end) {
        const int ncx = g.nc[0], ncy = g.nc[1], ncz = g.nc[2];
        const int cx = c % ncx, cy = (c / ncx) % ncy, cz = c / (ncx * ncy);
        // ---- piece table: a lane describes the stencil cells e = lane and lane + 64 (<= 125), x fastest ----
        const int nsx = g.nstencil[0], nsy = g.nstencil[1], ne = nsx * nsy * g.nstencil[2];
        int total;
        {
            int ecs[2] = {0, 0}, len[2] = {0, 0}, inc[2];
            float esx[2] = {0.f, 0.f}, esy[2] = {0.f, 0.f}, esz[2] = {0.f, 0.f};
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int e = lane + 64 * hh;
                if (e < ne) {
                    const int ox = e % nsx, oy = (e / nsx) % nsy, oz = e / (nsx * nsy);
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
                    len[hh] = cell_start[cc + 1] - ecs[hh];
                }
                inc[hh] = len[hh];
                for (int off = 1; off < 64; off <<= 1) {
                    const int v = __shfl_up(inc[hh], off);
                    if (lane >= off) inc[hh] += v;
                }
            }
            const int lower = __builtin_amdgcn_readlane(inc[0], 63);
            total = lower + __builtin_amdgcn_readlane(inc[1], 63);
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                s_rstart[w][lane + 64 * hh] = ecs[hh];
                s_rpref[w][lane + 64 * hh] = (hh ? lower : 0) + inc[hh] - len[hh];       // exclusive prefix; beyond the stencil: `total`
                s_rshift[w][0][lane + 64 * hh] = esx[hh];
                s_rshift[w][1][lane + 64 * hh] = esy[hh];
                s_rshift[w][2][lane + 64 * hh] = esz[hh];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        // coarse table: the piece of every (1 << cshift)-th stream position (the last piece that starts at or before it)
        int cshift = 3;
        while (((total + 127) >> cshift) > CB_COARSE) ++cshift;
#pragma unroll
        for (int q = 0; q < CB_COARSE / 64; ++q) {
            const int idx = (lane + 64 * q) << cshift;
            int r = 0;
#pragma unroll
            for (int step = 64; step > 0; step >>= 1)
                if (s_rpref[w][r + step] <= idx) r += step;
            s_coarse[w][lane + 64 * q] = (unsigned char)r;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#ifdef AMM_CBS_TIMING
        if (SLICE && lane == 0) { atomicAdd(&g_cbs_t[7], 1ull); atomicAdd(&g_cbs_t[0], wall_clock64() - t_last); }
#endif
        for (int tb = a_begin; tb < a_end; tb += CB_BATCH) {
            const int nt = min(CB_BATCH, a_end - tb);
            // the batch's atoms: lane 3 t + a holds atom a of row molecule t; the loops over t below are REAL loops (t is a scalar
            // register): coordinates come by v_readlane, the per-row counters live in lane t of three registers -- one copy of the
            // test and of the drain code whatever the batch size
            float4 my = make_float4(0.f, 0.f, 0.f, 0.f);
            if (lane < 3 * nt) my = pos4f[3 * tb + lane];
            int row_cnt = 0;          // lane t: entries of row t so far (front | back << 16)
            int q_head = 0, q_cnt = 0;    // lane t: ring of row t (head index, entries waiting)
            auto rl = [&](float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
            // ---- pass 2: 128 queued survivors of row t (n of them valid), two per lane (lane, lane + 64) so that the arithmetic runs
            // on packed fp32: nine distances each, the smallest decides; the two halves are filed one after the other (ring order) ----
            auto drain = [&](int t, int n) {
#ifdef AMM_CBS_TIMING
                const unsigned long long t_d0 = wall_clock64();
                n_drain++;
#endif
                const int head = __builtin_amdgcn_readlane(q_head, t);
                int slot[2];
                bool v[2];
                float4 A[2][3];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const bool v0 = lane + 64 * u < n;
                    slot[u] = v0 ? s_q[w][t][cq_wrap(head + lane + 64 * u)] : tb + t;
                    v[u] = v0 && slot[u] != tb + t;          // (a molecule is not its own partner)
                }
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int b = 0; b < 3; ++b) A[u][b] = pos4f[3 * slot[u] + b];
                const int sites[2] = {__float_as_int(A[0][1].w), __float_as_int(A[1][1].w)};
                float px[3], py[3], pz[3];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    px[a] = rl(my.x, 3 * t + a);
                    py[a] = rl(my.y, 3 * t + a);
                    pz[a] = rl(my.z, 3 * t + a);
                }
                v2f ax[3], ay[3], az[3];
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    ax[b] = v2f{A[0][b].x, A[1][b].x};
                    ay[b] = v2f{A[0][b].y, A[1][b].y};
                    az[b] = v2f{A[0][b].z, A[1][b].z};
                }
                if (!RINT) {          // one periodic image per molecule pair, from the first atoms
                    const v2f sx = box.L[0] * __builtin_elementwise_rint((ax[0] - px[0]) * box.invL[0]);
                    const v2f sy = box.L[1] * __builtin_elementwise_rint((ay[0] - py[0]) * box.invL[1]);
                    const v2f sz = box.L[2] * __builtin_elementwise_rint((az[0] - pz[0]) * box.invL[2]);
#pragma unroll
                    for (int b = 0; b < 3; ++b) {
                        ax[b] -= sx;
                        ay[b] -= sy;
                        az[b] -= sz;
                    }
                }
                v2f m2 = v2f{3.0e38f, 3.0e38f};
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < 3; ++b) {
                        v2f dx = px[a] - ax[b], dy = py[a] - ay[b], dz = pz[a] - az[b];
                        if (RINT) {
                            dx -= box.L[0] * __builtin_elementwise_rint(dx * box.invL[0]);
                            dy -= box.L[1] * __builtin_elementwise_rint(dy * box.invL[1]);
                            dz -= box.L[2] * __builtin_elementwise_rint(dz * box.invL[2]);
                        }
                        v2f r2 = dx * dx;
                        r2 = __builtin_elementwise_fma(dy, dy, r2);
                        r2 = __builtin_elementwise_fma(dz, dz, r2);
                        m2 = __builtin_elementwise_min(m2, r2);
                    }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float m2u = u ? m2.y : m2.x;
                    const bool pass = v[u] && m2u < rlist2;
                    const unsigned long long m_pass = __builtin_amdgcn_ballot_w64(pass);
                    if (m_pass != 0ull) {
                        const bool is_near = pass && m2u < rnear2;
                        const unsigned long long m_near = __builtin_amdgcn_ballot_w64(is_near);
                        const int np_ = __popcll(m_pass), nn_ = __popcll(m_near);
                        const int c2 = __builtin_amdgcn_readlane(row_cnt, t);
                        const int cnt = c2 & 0xffff, cntf = (int)((unsigned)c2 >> 16);
                        if (!COUNT_ONLY) {
                            const int mp = __builtin_amdgcn_mbcnt_hi((unsigned)(m_pass >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m_pass, 0u));
                            const int mn = __builtin_amdgcn_mbcnt_hi((unsigned)(m_near >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m_near, 0u));
                            const int pos_near = cnt + mn, pos_far = (cap - 1 - cntf) - (mp - mn);
                            if (cnt + cntf + np_ <= cap) {
                                int *row_out = nl + (size_t)(tb + t - c_begin) * cap;
                                if (pass) row_out[is_near ? pos_near : pos_far] = slot[u] | (sites[u] << 29);
                            }
                        }
                        row_cnt = lane == t ? c2 + nn_ + ((np_ - nn_) << 16) : row_cnt;
                    }
                }
                q_head = lane == t ? cq_wrap(head + n) : q_head;
                q_cnt = lane == t ? q_cnt - n : q_cnt;
#ifdef AMM_CBS_TIMING
                t_drain += wall_clock64() - t_d0;
#endif
            };
            // ---- pass 1: the candidate stream, first atoms only; a row's ring is drained in FULL chunks as soon as it holds 64 ----
            // (the candidates of chunk pair i + 1 -- piece search in LDS, then a gather -- are fetched while pair i is tested)
            auto fetch = [&](int cb, float4 (&cand)[2], int (&js)[2]) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int idx = cb + u * 64 + lane;
                    const bool in = idx < total;
                    // (idx < total + 128 <= CB_COARSE << cshift; beyond the stream every piece "starts" at `total`: the walk stops)
                    int r = s_coarse[w][idx >> cshift];
                    while (r < 127 && s_rpref[w][r + 1] <= idx) ++r;
                    const int slot = in ? s_rstart[w][r] + idx - s_rpref[w][r] : 0;
                    float4 q = pos4f[3 * slot];
                    if (!RINT) {
                        const float sx = s_rshift[w][0][r], sy = s_rshift[w][1][r], sz = s_rshift[w][2][r];
                        q.x = in ? q.x + sx : FAR;
                        q.y = in ? q.y + sy : FAR;
                        q.z = in ? q.z + sz : FAR;
                    } else if (!in) {
                        q.w = -1.0e9f;            // beyond the stream: no sphere reaches it
                    }
                    cand[u] = q;
                    js[u] = slot;
                }
            };
            float4 cand[2], cand_n[2];
            int js[2], js_n[2];
            fetch(0, cand, js);
            // (lane 3 t: the radius of row t's sphere without the candidate's own extent)
            const float plim_l = my.w + rlist;
            for (int cb = 0; cb < total; cb += 128) {
                if (cb + 128 < total) fetch(cb + 128, cand_n, js_n);
                // the two chunks side by side in packed registers (a chunk beyond the stream holds far-away / never-reached candidates)
                const v2f cx = v2f{cand[0].x, cand[1].x}, cy = v2f{cand[0].y, cand[1].y}, cz = v2f{cand[0].z, cand[1].z};
                const v2f cw = v2f{cand[0].w, cand[1].w};
                for (int t = 0; t < nt; ++t) {
                    const float p0x = rl(my.x, 3 * t), p0y = rl(my.y, 3 * t), p0z = rl(my.z, 3 * t), plim = rl(plim_l, 3 * t);
                    int qn = __builtin_amdgcn_readlane(q_cnt, t);
                    const int head = __builtin_amdgcn_readlane(q_head, t);
                    v2f dx = p0x - cx, dy = p0y - cy, dz = p0z - cz;
                    if (RINT) {
                        dx -= box.L[0] * __builtin_elementwise_rint(dx * box.invL[0]);
                        dy -= box.L[1] * __builtin_elementwise_rint(dy * box.invL[1]);
                        dz -= box.L[2] * __builtin_elementwise_rint(dz * box.invL[2]);
                    }
                    v2f r2 = dx * dx;
                    r2 = __builtin_elementwise_fma(dy, dy, r2);
                    r2 = __builtin_elementwise_fma(dz, dz, r2);
                    const v2f lim = plim + cw;
                    const v2f lim2 = lim * lim;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        // (!RINT: a candidate beyond the stream sits 1e9 away; RINT: its extent is -1e9, lim < 0)
                        const bool pass = (u ? r2.y < lim2.y : r2.x < lim2.x) && (!RINT || (u ? lim.y : lim.x) > 0.f);
                        const unsigned long long m = __builtin_amdgcn_ballot_w64(pass);
                        if (m == 0ull) continue;
                        const int at = qn + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                        if (pass) s_q[w][t][cq_wrap(head + at)] = js[u];
                        qn += __popcll(m);
                    }
                    q_cnt = lane == t ? qn : q_cnt;
                    if (qn >= 128) {
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        drain(t, 128);
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    cand[u] = cand_n[u];
                    js[u] = js_n[u];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int t = 0; t < nt; ++t) {
                const int n = __builtin_amdgcn_readlane(q_cnt, t);
                if (n > 0) drain(t, n);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // the ring reads are done before the next batch refills them
            __builtin_amdgcn_wave_barrier();
            if (lane < nt) {
                const int count = row_cnt & 0xffff, countf = (int)((unsigned)row_cnt >> 16);
                const int total_nb = count + countf;
                if (!COUNT_ONLY) {
                    const bool over = total_nb > cap;
                    nnb[tb + lane - c_begin] = over ? 0 : total_nb;
                    nnb_near[tb + lane - c_begin] = over ? 0 : count;
                    if (over) flags[1] = 1;
                }
                wsum += (unsigned long long)total_nb;
                wnear += (unsigned long long)count;
                wmax = max(wmax, total_nb);
            }
        }
    }
#ifdef AMM_CBS_TIMING
    if (SLICE && lane == 0 && a_begin < a_end) { atomicAdd(&g_cbs_t[1], wall_clock64() - t_last); atomicAdd(&g_cbs_t[2], t_drain); atomicAdd(&g_cbs_t[3], n_drain); }
#endif
    unit += (int)(gridDim.x * (blockDim.x >> 6));
    } while (SLICE && unit < nunits);
    for (int off = 32; off > 0; off >>= 1) {
        wsum += __shfl_xor(wsum, off);
        wnear += __shfl_xor(wnear, off);
        wmax = max(wmax, __shfl_xor(wmax, off));
    }
    __shared__ unsigned long long s_sum[4], s_near[4];
    __shared__ int s_max[4];
    if (lane == 0) {
        s_sum[w] = wsum;
        s_near[w] = wnear;
        s_max[w] = wmax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        amm_st_l2(&blockstats[3 * blockIdx.x], s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]);
        amm_st_l2(&blockstats[3 * blockIdx.x + 1], (unsigned long long)max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3])));
        amm_st_l2(&blockstats[3 * blockIdx.x + 2], s_near[0] + s_near[1] + s_near[2] + s_near[3]);
    }
    if (amm_last_block(ticket)) cfinish_build_block(flags, counters, blockstats, (int)gridDim.x, COUNT_ONLY ? 1 : 0);
}

// ---- split-stream build: a rank's slice of the rows ----
// A slice gives too few (cell, part) units to fill the chip, and a wavefront's time in k_cbuild is its own chain of dependent reads
// over the cell's WHOLE candidate stream (~12 chunk pairs) plus ~3 drains per row, whatever its share of the rows.  Here a BLOCK takes
// the unit and both passes are shared out over its four wavefronts:
//   1. the stream goes in rounds of 4 x CBS_SEG candidates, wavefront w takes the w-th CBS_SEG of the round (whole chunk pairs) for
//      every row molecule of the batch and queues its survivors in a ring of its own (no drains: the ring holds a whole segment);
//   2. block barrier; the survivors of a row -- the four rings one behind the other, i.e. in stream order -- are cut into chunks of
//      128 and the chunks of all rows are dealt out to the wavefronts round robin (splitting the stream alone leaves the drains on
//      the two wavefronts whose segments cross the middle of the stencil: measured slower than the unsplit build); a chunk's hits
//      are filed in a chunk row in LDS (front part from the left, back part from the right, like the row itself);
//   3. block barrier; the chunk rows are copied into the row one behind the other, in chunk order = stream order.
// The rows are those of k_cbuild entry for entry (tests/test_gpu_abi_parity.py), so a rank's forces do not depend on the variant.
#ifndef CBS_BATCH
#define CBS_BATCH 3
#endif
#ifndef CBS_SEG
#define CBS_SEG 384             // candidates per wavefront and round (a multiple of 128)
#endif
#define CBS_MAXCH (4 * CBS_SEG / 128)       // chunks of 128 survivors a row can have in one round
template <bool RINT>
__global__ void __launch_bounds__(256) k_cbuild_split(int c_begin, int c_end, int parts, const int *__restrict__ cell_start,
                                                      const float4 *__restrict__ pos4f, CBoxF box, CellGrid g, float rlist, float rnear2, int cap,
                                                      int *nl, int *nnb, int *nnb_near, int *flags, unsigned long long *blockstats,
                                                      unsigned long long *counters, int *ticket, int force, const int *slice_cells) {
    if (!force && !flags[0]) return;
    __shared__ int s_rstart[128];
    __shared__ int s_rpref[128];
    __shared__ float s_rshift[3][128];
    __shared__ unsigned char s_coarse[CB_COARSE];
    __shared__ int s_total;
    __shared__ int s_ring[4][CBS_BATCH][CBS_SEG];           // survivors of the sphere test, per wavefront and row molecule
    __shared__ int s_cnt[4][CBS_BATCH];
    __shared__ int s_stage[CBS_BATCH * CBS_MAXCH][128];     // chunk rows
    __shared__ int s_ccnt[CBS_BATCH * CBS_MAXCH];           // their entries (front | back << 16)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float FAR = 1.0e9f;
    const float rlist2 = rlist * rlist;
    unsigned long long wsum = 0, wnear = 0;
    int wmax = 0;
    // the (cell, part) units of the slice's cells (k_csort_gather recorded the first and the last), a block each
    const int first_cell = __builtin_amdgcn_readfirstlane(slice_cells[0]);
    const int nunits = (__builtin_amdgcn_readfirstlane(slice_cells[1]) - first_cell + 1) * parts;
    for (int unit = (int)blockIdx.x; unit < nunits; unit += (int)gridDim.x) {
    const int c = first_cell + unit / parts, part = unit % parts;
    int a_begin = 0, a_end = 0;
    if (c < g.ncell) {
        const int cb0 = __builtin_amdgcn_readfirstlane(cell_start[c]), cb1 = __builtin_amdgcn_readfirstlane(cell_start[c + 1]);
        const int per = (cb1 - cb0 + parts - 1) / parts;
        a_begin = max(cb0 + part * per, c_begin);
        a_end = min(min(cb0 + (part + 1) * per, cb1), c_end);
    }
    if (a_begin < a_end) {          // (the same in every wavefront of the block: the barriers below are met by all four)
#ifdef AMM_CBS_TIMING
        unsigned long long t_last = wall_clock64();
        if (w == 0 && lane == 0) atomicAdd(&g_cbs_t[7], 1ull);
#endif
        const int ncx = g.nc[0], ncy = g.nc[1], ncz = g.nc[2];
        const int cx = c % ncx, cy = (c / ncx) % ncy, cz = c / (ncx * ncy);
        // ---- piece table of the block (first wavefront): a lane describes the stencil cells e = lane and lane + 64, x fastest ----
        if (w == 0) {
            const int nsx = g.nstencil[0], nsy = g.nstencil[1], ne = nsx * nsy * g.nstencil[2];
            int ecs[2] = {0, 0}, len[2] = {0, 0}, inc[2];
            float esx[2] = {0.f, 0.f}, esy[2] = {0.f, 0.f}, esz[2] = {0.f, 0.f};
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int e = lane + 64 * hh;
                if (e < ne) {
                    const int ox = e % nsx, oy = (e / nsx) % nsy, oz = e / (nsx * nsy);
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
                    len[hh] = cell_start[cc + 1] - ecs[hh];
                }
                inc[hh] = len[hh];
                for (int off = 1; off < 64; off <<= 1) {
                    const int v = __shfl_up(inc[hh], off);
                    if (lane >= off) inc[hh] += v;
                }
            }
            const int lower = __builtin_amdgcn_readlane(inc[0], 63);
            if (lane == 0) s_total = lower + __builtin_amdgcn_readlane(inc[1], 63);
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                s_rstart[lane + 64 * hh] = ecs[hh];
                s_rpref[lane + 64 * hh] = (hh ? lower : 0) + inc[hh] - len[hh];       // exclusive prefix; beyond the stencil: the total
                s_rshift[0][lane + 64 * hh] = esx[hh];
                s_rshift[1][lane + 64 * hh] = esy[hh];
                s_rshift[2][lane + 64 * hh] = esz[hh];
            }
        }
        __syncthreads();
        const int total = s_total;
        int cshift = 3;
        while (((total + 127) >> cshift) > CB_COARSE) ++cshift;
        {       // coarse table: the piece of every (1 << cshift)-th stream position, one entry per thread
            const int idx = (int)threadIdx.x << cshift;
            int r = 0;
#pragma unroll
            for (int step = 64; step > 0; step >>= 1)
                if (s_rpref[r + step] <= idx) r += step;
            s_coarse[threadIdx.x] = (unsigned char)r;
        }
        __syncthreads();
        CBS_STAMP(0);
        // rounds of equal length: a wavefront's segment is the round's quarter in whole chunk pairs (<= CBS_SEG)
        const int rounds = (total + 4 * CBS_SEG - 1) / (4 * CBS_SEG);
        const int seg = (((total + rounds - 1) / rounds + 511) >> 9) << 7;
        auto rl = [&](float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
        for (int tb = a_begin; tb < a_end; tb += CBS_BATCH) {
            const int nt = min(CBS_BATCH, a_end - tb);
            float4 my = make_float4(0.f, 0.f, 0.f, 0.f);
            if (lane < 3 * nt) my = pos4f[3 * tb + lane];
            const float plim_l = my.w + rlist;
            int row_cnt = 0;          // lane t: entries of row t so far (front | back << 16) -- the same in every wavefront
            for (int r0 = 0; r0 < total; r0 += 4 * seg) {
                // ---- pass 1: this wavefront's segment of the round, first atoms only ----
                const int lo = min(r0 + w * seg, total), hi = min(lo + seg, total);
                int q_cnt = 0;        // lane t: survivors of row t in this wavefront's ring
                auto fetch = [&](int cb, float4 (&cand)[2], int (&js)[2]) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int idx = cb + u * 64 + lane;
                        const bool in = idx < hi;
                        int r = s_coarse[idx >> cshift];
                        while (r < 127 && s_rpref[r + 1] <= idx) ++r;
                        const int slot = in ? s_rstart[r] + idx - s_rpref[r] : 0;
                        float4 q = pos4f[3 * slot];
                        if (!RINT) {
                            const float sx = s_rshift[0][r], sy = s_rshift[1][r], sz = s_rshift[2][r];
                            q.x = in ? q.x + sx : FAR;
                            q.y = in ? q.y + sy : FAR;
                            q.z = in ? q.z + sz : FAR;
                        } else if (!in) {
                            q.w = -1.0e9f;            // beyond the segment: no sphere reaches it
                        }
                        cand[u] = q;
                        js[u] = slot;
                    }
                };
                if (lo < hi) {
                    // (the whole segment -- at most CBS_SEG / 128 chunk pairs -- is in flight before the first is tested)
                    float4 pc[CBS_SEG / 128][2];
                    int pj[CBS_SEG / 128][2];
#pragma unroll
                    for (int st = 0; st < CBS_SEG / 128; ++st)
                        if (lo + 128 * st < hi) fetch(lo + 128 * st, pc[st], pj[st]);
#pragma unroll
                    for (int st = 0; st < CBS_SEG / 128; ++st) {
                        if (lo + 128 * st >= hi) break;
                        const float4 (&cand)[2] = pc[st];
                        const int (&js)[2] = pj[st];
                        const v2f cx2 = v2f{cand[0].x, cand[1].x}, cy2 = v2f{cand[0].y, cand[1].y}, cz2 = v2f{cand[0].z, cand[1].z};
                        const v2f cw = v2f{cand[0].w, cand[1].w};
                        for (int t = 0; t < nt; ++t) {
                            const float p0x = rl(my.x, 3 * t), p0y = rl(my.y, 3 * t), p0z = rl(my.z, 3 * t), plim = rl(plim_l, 3 * t);
                            int qn = __builtin_amdgcn_readlane(q_cnt, t);
                            v2f dx = p0x - cx2, dy = p0y - cy2, dz = p0z - cz2;
                            if (RINT) {
                                dx -= box.L[0] * __builtin_elementwise_rint(dx * box.invL[0]);
                                dy -= box.L[1] * __builtin_elementwise_rint(dy * box.invL[1]);
                                dz -= box.L[2] * __builtin_elementwise_rint(dz * box.invL[2]);
                            }
                            v2f r2 = dx * dx;
                            r2 = __builtin_elementwise_fma(dy, dy, r2);
                            r2 = __builtin_elementwise_fma(dz, dz, r2);
                            const v2f lim = plim + cw;
                            const v2f lim2 = lim * lim;
#pragma unroll
                            for (int u = 0; u < 2; ++u) {
                                const bool pass = (u ? r2.y < lim2.y : r2.x < lim2.x) && (!RINT || (u ? lim.y : lim.x) > 0.f);
                                const unsigned long long m = __builtin_amdgcn_ballot_w64(pass);
                                if (m == 0ull) continue;
                                const int at = qn + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                                if (pass) s_ring[w][t][at] = js[u];          // (at < CBS_SEG: a segment has no more candidates)
                                qn += __popcll(m);
                            }
                            q_cnt = lane == t ? qn : q_cnt;
                        }
                    }
                }
                CBS_STAMP(1);
                if (lane < CBS_BATCH) s_cnt[w][lane] = q_cnt;
                __syncthreads();
                CBS_STAMP(2);
                // ---- pass 2: the rows' survivors in chunks of 128, dealt out round robin; two per lane, packed fp32 (as k_cbuild's drain) ----
                int j0 = 0;       // number of the row's first chunk
                for (int t = 0; t < nt; ++t) {
                    const int c0 = __builtin_amdgcn_readfirstlane(s_cnt[0][t]), c1 = c0 + __builtin_amdgcn_readfirstlane(s_cnt[1][t]);
                    const int c2 = c1 + __builtin_amdgcn_readfirstlane(s_cnt[2][t]), T = c2 + __builtin_amdgcn_readfirstlane(s_cnt[3][t]);
                    const int nch = (T + 127) >> 7;
                    for (int k = 0; k < nch; ++k) {
                        const int j = j0 + k;
                        if ((j & 3) != w) continue;
                        const int n = min(128, T - 128 * k);
                        int slot[2];
                        bool v[2];
                        float4 A[2][3];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int gi = 128 * k + lane + 64 * u;
                            const bool v0 = lane + 64 * u < n;
                            const int ww = (gi >= c0 ? 1 : 0) + (gi >= c1 ? 1 : 0) + (gi >= c2 ? 1 : 0);
                            const int off = gi - (ww == 0 ? 0 : (ww == 1 ? c0 : (ww == 2 ? c1 : c2)));
                            slot[u] = v0 ? s_ring[ww][t][off] : tb + t;
                            v[u] = v0 && slot[u] != tb + t;          // (a molecule is not its own partner)
                        }
#pragma unroll
                        for (int u = 0; u < 2; ++u)
#pragma unroll
                            for (int b = 0; b < 3; ++b) A[u][b] = pos4f[3 * slot[u] + b];
                        const int sites[2] = {__float_as_int(A[0][1].w), __float_as_int(A[1][1].w)};
                        float px[3], py[3], pz[3];
#pragma unroll
                        for (int a = 0; a < 3; ++a) {
                            px[a] = rl(my.x, 3 * t + a);
                            py[a] = rl(my.y, 3 * t + a);
                            pz[a] = rl(my.z, 3 * t + a);
                        }
                        v2f ax[3], ay[3], az[3];
#pragma unroll
                        for (int b = 0; b < 3; ++b) {
                            ax[b] = v2f{A[0][b].x, A[1][b].x};
                            ay[b] = v2f{A[0][b].y, A[1][b].y};
                            az[b] = v2f{A[0][b].z, A[1][b].z};
                        }
                        if (!RINT) {          // one periodic image per molecule pair, from the first atoms
                            const v2f sx = box.L[0] * __builtin_elementwise_rint((ax[0] - px[0]) * box.invL[0]);
                            const v2f sy = box.L[1] * __builtin_elementwise_rint((ay[0] - py[0]) * box.invL[1]);
                            const v2f sz = box.L[2] * __builtin_elementwise_rint((az[0] - pz[0]) * box.invL[2]);
#pragma unroll
                            for (int b = 0; b < 3; ++b) {
                                ax[b] -= sx;
                                ay[b] -= sy;
                                az[b] -= sz;
                            }
                        }
                        v2f m2 = v2f{3.0e38f, 3.0e38f};
#pragma unroll
                        for (int a = 0; a < 3; ++a)
#pragma unroll
                            for (int b = 0; b < 3; ++b) {
                                v2f dx = px[a] - ax[b], dy = py[a] - ay[b], dz = pz[a] - az[b];
                                if (RINT) {
                                    dx -= box.L[0] * __builtin_elementwise_rint(dx * box.invL[0]);
                                    dy -= box.L[1] * __builtin_elementwise_rint(dy * box.invL[1]);
                                    dz -= box.L[2] * __builtin_elementwise_rint(dz * box.invL[2]);
                                }
                                v2f r2 = dx * dx;
                                r2 = __builtin_elementwise_fma(dy, dy, r2);
                                r2 = __builtin_elementwise_fma(dz, dz, r2);
                                m2 = __builtin_elementwise_min(m2, r2);
                            }
                        int cnt = 0, cntf = 0;        // the chunk row so far
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const float m2u = u ? m2.y : m2.x;
                            const bool pass = v[u] && m2u < rlist2;
                            const unsigned long long m_pass = __builtin_amdgcn_ballot_w64(pass);
                            if (m_pass != 0ull) {
                                const bool is_near = pass && m2u < rnear2;
                                const unsigned long long m_near = __builtin_amdgcn_ballot_w64(is_near);
                                const int np_ = __popcll(m_pass), nn_ = __popcll(m_near);
                                const int mp = __builtin_amdgcn_mbcnt_hi((unsigned)(m_pass >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m_pass, 0u));
                                const int mn = __builtin_amdgcn_mbcnt_hi((unsigned)(m_near >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m_near, 0u));
                                if (pass) s_stage[j][is_near ? cnt + mn : 127 - cntf - (mp - mn)] = slot[u] | (sites[u] << 29);
                                cnt += nn_;
                                cntf += np_ - nn_;
                            }
                        }
                        if (lane == 0) s_ccnt[j] = cnt | (cntf << 16);
                    }
                    j0 += nch;
                }
                CBS_STAMP(3);
                __syncthreads();
                CBS_STAMP(4);
                // ---- the chunk rows into the rows, one behind the other (every wavefront walks the counts, the one that made a chunk copies it) ----
                j0 = 0;
                for (int t = 0; t < nt; ++t) {
                    const int T = __builtin_amdgcn_readfirstlane(s_cnt[0][t] + s_cnt[1][t] + s_cnt[2][t] + s_cnt[3][t]);
                    const int nch = (T + 127) >> 7;
                    const int have = __builtin_amdgcn_readlane(row_cnt, t);
                    int nf = have & 0xffff, nb_ = (int)((unsigned)have >> 16);
                    int *row_out = nl + (size_t)(tb + t - c_begin) * cap;
                    for (int k = 0; k < nch; ++k) {
                        const int j = j0 + k;
                        const int cc = __builtin_amdgcn_readfirstlane(s_ccnt[j]);
                        const int cn = cc & 0xffff, cf = (int)((unsigned)cc >> 16);
                        if ((j & 3) == w && nf + nb_ + cn + cf <= cap) {
                            for (int i = lane; i < cn; i += 64) row_out[nf + i] = s_stage[j][i];
                            for (int i = lane; i < cf; i += 64) row_out[cap - 1 - nb_ - i] = s_stage[j][127 - i];
                        }
                        nf = min(nf + cn, 0xffff);        // (a row that overflows is reported below; the counters must not wrap)
                        nb_ = min(nb_ + cf, 0xffff);
                    }
                    row_cnt = lane == t ? (nf | (nb_ << 16)) : row_cnt;
                    j0 += nch;
                }
                CBS_STAMP(5);
                __syncthreads();          // rings, counts and chunk rows are free for the next round
                CBS_STAMP(6);
            }
            if (w == 0 && lane < nt) {
                const int count = row_cnt & 0xffff, countf = (int)((unsigned)row_cnt >> 16);
                const int total_nb = count + countf;
                const bool over = total_nb > cap;
                nnb[tb + lane - c_begin] = over ? 0 : total_nb;
                nnb_near[tb + lane - c_begin] = over ? 0 : count;
                if (over) flags[1] = 1;
                wsum += (unsigned long long)total_nb;
                wnear += (unsigned long long)count;
                wmax = max(wmax, total_nb);
            }
        }
    }
    __syncthreads();          // (a second unit of this block rebuilds the piece table)
    }
    for (int off = 32; off > 0; off >>= 1) {
        wsum += __shfl_xor(wsum, off);
        wnear += __shfl_xor(wnear, off);
        wmax = max(wmax, __shfl_xor(wmax, off));
    }
    if (w == 0 && lane == 0) {
        amm_st_l2(&blockstats[3 * blockIdx.x], wsum);
        amm_st_l2(&blockstats[3 * blockIdx.x + 1], (unsigned long long)wmax);
        amm_st_l2(&blockstats[3 * blockIdx.x + 2], wnear);
    }
    if (amm_last_block(ticket)) cfinish_build_block(flags, counters, blockstats, (int)gridDim.x, 0);
}

// ------------------------------------------------------------------------------------------------ traversal
#define AMM_CPHASES 6
struct CPairArgs {
    int c_begin, nrows, lpa_shift, cap;
    const int *aperm;
    const int *nl, *nnb, *nnb_total;
    const double4 *posq;
    const double2 *lj;
    double *force;
    int accumulate, sorted_out;
    Box box;
    const double *host_tab;
    int host_bytes;
    double margin;
    int ntask;
    int per_pair_image;
    // the rows of an XCD (rpx consecutive rows each) are walked in phases: phase p covers the XCD's rows from ph_off[p] on as
    // ph_ntask[p] wavefront tasks of 64 >> ph_shift[p] rows (cpair_plan: whole rounds of big tasks, the remainder in smaller ones)
    int rpx, nphase;
    int ph_off[AMM_CPHASES], ph_shift[AMM_CPHASES], ph_ntask[AMM_CPHASES];
#ifdef AMM_CPAIR_TIMING           // measurement builds (scripts/build_variant.sh): wall clock of every wavefront
    unsigned long long *wave_times;      // [wavefront][4]: kernel entry, tables staged, tasks done, flags (interior tasks << 8 | tasks)
#endif
    // site-site tables (pair_tab.h: SiteTable; kernels with SS): LDS byte offset FROM THE FORCE'S COULOMB TABLE to the place
    // interval 0 of its site-site table would have, the bytes it really holds, and the site class for the analytic fallback
    const double *host_tab_ss;
    int host_ss_bytes, host_ss_off, site_atoms;      // site_atoms: bit a set when atom a of some molecule is a site
    double hsig_site, seps2_site;
    // the guest force of a fused pass: its tables, its output and its factors relative to the host's
    const double *guest_tab, *guest_tab_ss;
    int guest_bytes, guest_ss_bytes, guest_ss_off, g_accumulate;
    double *gforce;
    double gfac, gsr;                  // (Kc sign)_guest / (Kc sign)_host ; sign_guest / sign_host
};

// One walk of a wavefront's rows.  IMG: 0 interior rows (no periodic image), 1 one image per molecule pair (from the first
// atoms), 2 minimum image per atom pair.  GFAM >= 0: the fused step-boundary pass (below).  SS: site-site tables.
//
// One trip = one partner molecule per lane = nine atom pairs, walked partner atom by partner atom (b outer, a inner).  Per partner
// atom: geometry, index arithmetic and LDS reads of its three pairs are pinned AHEAD of the three Horner chains (a scheduling
// barrier: left to itself the compiler waits for each pair's reads right behind them, s_waitcnt lgkmcnt(0) nine times per trip),
// then the Lennard-Jones part, then the accumulation.  When the three pairs of partner atom b are done its record is dead and the
// same registers take the record of the NEXT trip's partner: the loads of trip t + 1 are in flight behind ~1 000 cycles of
// arithmetic without a second register set.
//
// Lennard-Jones part.  SS (every site of the force has the same sigma, eps and charge -- water): a pair of two sites reads the
// site-site table instead of the Coulomb table -- the same interval, `ss_off` bytes further -- and that is all: no 1/r, no switch
// in r, no parameter records (measured on the 98 304-atom box: near 68 -> 58 us, fused pass 174 -> 141 us with the arithmetic
// removed; the table costs 4 integer instructions per partner atom).  Otherwise: analytically under a wave-uniform branch on the
// partner's site bit, the row atoms' parameters in an LDS strip.
//
// Fused pass (GFAM >= 0): the step boundary of RESPA needs the outer force over the whole rows AND the near force over their
// front parts.  As two launches the front entries are gathered and their geometry formed twice; here the guest rides on the
// host's walk: per partner atom the host's three look-ups, Horner chains and accumulation come first, then -- only on trips where
// some row of the wavefront is still in its front part (wave-uniform) -- the guest's three look-ups into its own tables, into a
// second set of nine accumulators.  The scheduling barriers keep the two forces' look-up registers from being live together
// (the forms with the guest inside the host's nine-pair body all spilled: 72 to 436 bytes of scratch per lane).  Product and sum
// of the guest are rounded as its stand-alone launch rounds them: the same bits either way.
template <int FAM, int CMODE, int GFAM, int IMG, int SMASK>
__device__ __forceinline__ void cwalk_rows(const CPairArgs &A, const PairConsts &c, const PairConsts &g, const char *tabh, const char *tabg,
                                           const double *erfcx, const double4 (&pi)[3], const double2 *li, int i_sites, const int (&so)[3],
                                           const int (&sog)[3], double sign_lj, const int *row, int nfront, int nn, int sub, int lpa,
                                           int self, double (&f)[9], double (&fg)[9]) {
    constexpr bool DUAL = GFAM >= 0;
    constexpr bool SS = SMASK != 0;          // SMASK: the row atoms that can be sites (1: only the first -- water; 7: any)
    const int back = A.cap - 1 + nfront;
    // closer pairs are left out of the main path and redone analytically below (never in a liquid)
    const double r2low = DUAL ? fmax(c.tab.r2min, g.tab.r2min) : c.tab.r2min;
    const double r2low_ss = DUAL ? fmax(c.tab.ss_r2min, g.tab.ss_r2min) : c.tab.ss_r2min;
    auto entry = [&](int k) { return k < nn ? row[k < nfront ? k : back - k] : self; };
    auto load_pos = [&](int e, int b) {
        return *reinterpret_cast<const double4 *>(reinterpret_cast<const char *>(A.posq) + (size_t)((unsigned)e & 0x1fffffffu) * 96u + 32 * b);
    };
    auto load_lj = [&](int e, int b) {
        // the Lennard-Jones record only of partner atoms that have a site in some entry of this trip (wave-uniform)
        double2 l = make_double2(0.0, 0.0);
        if (i_sites != 0 && __builtin_amdgcn_ballot_w64((((unsigned)e >> 29) >> b) & 1u) != 0ull)
            l = *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(A.lj) + (size_t)((unsigned)e & 0x1fffffffu) * 48u + 16 * b);
        return l;
    };
    double4 pj[3];
    double2 lj[3];
    int k = sub;
    int e = entry(k), en = entry(k + lpa);
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        pj[b] = load_pos(e, b);
        if (!SS) lj[b] = load_lj(e, b);
    }
    while (__builtin_amdgcn_ballot_w64(k < nn) != 0ull) {
        const bool ok = k < nn;
        const bool front = k < nfront;
        const bool any_front = DUAL && __builtin_amdgcn_ballot_w64(front) != 0ull;
        const unsigned bits = (unsigned)e >> 29;
        const int e2 = entry(k + 2 * lpa);           // the entry after next: its index is there when the next trip starts
        double sx = 0.0, sy = 0.0, sz = 0.0;
        if (IMG == 1) {
            sx = A.box.L[0] * rint((pj[0].x - pi[0].x) * A.box.invL[0]);
            sy = A.box.L[1] * rint((pj[0].y - pi[0].y) * A.box.invL[1]);
            sz = A.box.L[2] * rint((pj[0].z - pi[0].z) * A.box.invL[2]);
        }
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const double xb = pj[b].x - sx, yb = pj[b].y - sy, zb = pj[b].z - sz, qb = pj[b].w;
            // SS: all ones where the partner atom is a site (bit field extract with sign extension)
            const int mb = SS ? -(int)((bits >> b) & 1u) : 0;
            double dx[3], dy[3], dz[3], r2[3], fr[3], gl[3];
            {
                TabLookup th[3];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    dx[a] = pi[a].x - xb;
                    dy[a] = pi[a].y - yb;
                    dz[a] = pi[a].z - zb;
                    if (IMG == 2) {
                        dx[a] = amm_min_image(dx[a], A.box.L[0], A.box.invL[0]);
                        dy[a] = amm_min_image(dy[a], A.box.L[1], A.box.invL[1]);
                        dz[a] = amm_min_image(dz[a], A.box.L[2], A.box.invL[2]);
                    }
                    r2[a] = dx[a] * dx[a] + dy[a] * dy[a] + dz[a] * dz[a];
                    th[a] = amm_tab_fetch(tabh, c.tab, r2[a], ((SMASK >> a) & 1) ? (unsigned)(so[a] & mb) : 0u);
                    if (DUAL) gl[a] = 0.0;
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = 0; a < 3; ++a) fr[a] = (pi[a].w * qb) * amm_tab_horner(th[a]);
            }
            // Lennard-Jones part, analytic: only where two sites can meet (wave-uniform); the other lanes add an exact zero (eps4 = 0)
            if (!SS && !AMM_EXP_NOLJ && i_sites != 0 && __builtin_amdgcn_ballot_w64(ok && ((bits >> b) & 1u)) != 0ull) {
#pragma unroll
                for (int a = 0; a < 3; ++a)
                    if ((i_sites >> a) & 1) {
                        const double2 la = li[64 * a];
                        const double sig = la.x + lj[b].x, eps4 = la.y * lj[b].y;
                        const LJCommon L = amm_lj_common(r2[a], sig, eps4);
                        fr[a] = amm_sum_unfused(fr[a], amm_lj_force<FAM, CMODE>(c, L, sig, eps4));
                        if (DUAL && any_front) {     // the guest's Lennard-Jones force from the same 1/r, (sigma/r)^6 ... (a sign apart)
                            LJCommon Lg = L;
                            Lg.dlj_r *= A.gsr;
                            gl[a] = amm_lj_force<(DUAL ? GFAM : FAM), 0>(g, Lg, sig, eps4 * A.gsr);
                        }
                    }
            }
            bool any_low = false;
            bool lowp[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                bool low = r2[a] < r2low;
                // (a site pair below its own table; no branch here: a branch in this loop costs more than it skips)
                if ((SMASK >> a) & 1) low = low || ((so[a] & mb) != 0 && r2[a] < r2low_ss);
                lowp[a] = low;
                const bool pass = ok && (r2[a] < c.rc2);
                any_low = any_low || (ok && low);
                const double fh = (pass && !low) ? fr[a] : 0.0;
                f[3 * a] += fh * dx[a];
                f[3 * a + 1] += fh * dy[a];
                f[3 * a + 2] += fh * dz[a];
            }
            if (DUAL) {
                __builtin_amdgcn_sched_barrier(0);
                if (any_front) {
                    TabLookup tg[3];
#pragma unroll
                    for (int a = 0; a < 3; ++a) tg[a] = amm_tab_fetch(tabg, g.tab, r2[a], ((SMASK >> a) & 1) ? (unsigned)(sog[a] & mb) : 0u);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        const double gr = amm_sum_unfused(((pi[a].w * qb) * A.gfac) * amm_tab_horner(tg[a]), gl[a]);
                        const bool pass = front && (r2[a] < g.rc2) && !lowp[a];
                        const double gh = pass ? gr : 0.0;
                        fg[3 * a] += gh * dx[a];
                        fg[3 * a + 1] += gh * dy[a];
                        fg[3 * a + 2] += gh * dz[a];
                    }
                }
            }
            if (__builtin_amdgcn_ballot_w64(any_low) != 0ull) {       // closer than a table reaches: analytic
                for (int a = 0; a < 3; ++a) {
                    const bool low = ok && lowp[a];
                    double sg, e4;
                    if (SS) {
                        const bool both = ((SMASK >> a) & 1) && (so[a] & mb) != 0;
                        sg = 2.0 * A.hsig_site;
                        e4 = both ? A.seps2_site * A.seps2_site * sign_lj : 0.0;
                    } else {
                        const double2 lx = A.lj[3 * ((unsigned)e & 0x1fffffffu) + b];
                        const double2 la = li[64 * a];
                        sg = la.x + lx.x;
                        e4 = la.y * lx.y;
                    }
                    double e_, fr_;
                    // (pi.w and the epsilons carry the sign already: the math runs with sign 1)
                    amm_pair_math<FAM, CMODE, false, false>(c, low ? r2[a] : 1.0, pi[a].w * qb, sg, e4, e_, fr_, erfcx);
                    fr_ = (low && r2[a] < c.rc2) ? fr_ : 0.0;
                    f[3 * a] += fr_ * dx[a];
                    f[3 * a + 1] += fr_ * dy[a];
                    f[3 * a + 2] += fr_ * dz[a];
                    if (DUAL) {
                        double gr_;
                        amm_pair_math<(DUAL ? GFAM : FAM), 0, false, false>(g, low ? r2[a] : 1.0, (pi[a].w * qb) * A.gfac, sg, e4 * A.gsr, e_, gr_, erfcx);
                        gr_ = (low && front && r2[a] < g.rc2) ? gr_ : 0.0;
                        fg[3 * a] += gr_ * dx[a];
                        fg[3 * a + 1] += gr_ * dy[a];
                        fg[3 * a + 2] += gr_ * dz[a];
                    }
                }
            }
            // this partner atom's record is dead: its registers take the next trip's (the image shift of THIS trip is in sx, sy, sz)
            pj[b] = load_pos(en, b);
            if (!SS) lj[b] = load_lj(en, b);
        }
        e = en;
        en = e2;
        k += lpa;
    }
}

// ------------------------------------------------------------------------------------------------ epilogue: the inner RESPA loop
// A box of flexible three-site molecules (BondedSet::mol3_ok: the innermost force group is one bond-list set whose components are the
// molecules) couples nothing between molecules inside the inner RESPA loop
//     [kicks with the forces of this launch / of earlier ones]  n0 x { v += c1 f0/m ; x += d v ; f0 = bonded(x) ; v += c2 f0/m }
// (propagators.py:933-973 unrolled; bonded.hip: k_inner_lanes runs it as a launch of its own).  The wavefront that has just summed
// the rows of its molecules holds everything the loop needs that this launch produced, so it runs the loop right there: the first
// four lanes of a row's lanes take atom l / term l of the molecule (k_inner_lanes' TERMS scheme: lane l evaluates term l once, every
// atom adds up its records in record order -- the same numbers in the same order, bit-identical), positions and term forces travel
// between the four lanes by wavefront shuffles (no LDS: the fused pass uses 155 of the CU's 160 KB), and the launch also writes what
// the NEXT pair evaluation reads -- the sorted fp64 copies of the new positions, through the current permutation -- and evaluates the
// lists' displacement triggers.  Per outer step of RESPA [4, 2, 1] that is two launches of k_inner_lanes and two gather launches
// less, and the work runs where the pair kernel leaves the chip half empty (the older wavefront of every SIMD finishes a quarter
// of the kernel's time before the younger one: profiles/r04_wave_times.txt).  The pair kernel never reads x, v or the force
// buffers of other groups (only the sorted copies, which this launch does not write: the next evaluation's are another buffer).
struct CEpiPre {
    const double *a, *b;       // v += coef (a -/+ b) / m ; b may be null (the buffers may be those this launch writes)
    double coef;
    int plus;
};
struct CEpiArgs {
    int niter, npre;
    int recompute_f0;                          // the innermost forces at the start are evaluated here, not taken from f0
    double *x, *v, *f0;
    const double *mass;
    double c1, d, c2;
    CEpiPre pre[AMM_MAX_PRE];
    const int *cperm;                          // sorted cluster -> molecule = component of the bond-list set
    const int4 *term_l;                        // [molecule * 4 + lane] BondedSet::d_term_l / d_term_q / d_atom_recs
    const double4 *term_q;
    const unsigned long long *atom_recs;
    // cells of the moved molecules, filed ahead of a possible rebuild (ClusterList::d_spec_count): count / members / start as
    // k_cassign's, the array to clear for the next launch, the list's flags, a ticket for the last block; null: not this launch
    int *spec_count, *spec_clear, *spec_members, *spec_start, *spec_flags, *spec_ticket;
    int spec_capc;
    CellGrid grid;
    // multi-rank: this rank's chunk of the exchange buffer takes the new state of its molecules, [x: per x 3][v: per x 3] by sorted
    // slot relative to the slice (null: single rank)
    double *xchg_x, *xchg_v;
    int c_first;                               // first sorted cluster of the slice
    double4 *posq_next;                        // sorted copies of the next pair evaluation (null: none)
    double2 *lj_next;                          // (null: that force's are in place already)
    const double *q_next, *hsig_next, *seps2_next;
    WatchArgs W;
};

#ifndef CEPI_INLINE
#define CEPI_INLINE __forceinline__
#endif

struct CEpiArgs;
// value of lane (quad base + Q) of the caller's quad, for all four lanes of the quad: a DPP move (VALU rate, no LDS round trip -- the
// ds_bpermute form of the epilogue spent a dozen serialized LDS latencies per inner iteration on its shuffles)
template <int Q>
__device__ __forceinline__ double cquad_bcast(double v) {
    constexpr int ctrl = Q | (Q << 2) | (Q << 4) | (Q << 6);       // quad_perm: [Q, Q, Q, Q]
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_mov_dpp((int)(b & 0xffffffffll), ctrl, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), ctrl, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// positions of the (up to three) atoms of the lane's term, fetched from their lanes before the term code runs (the term code
// branches on the kind of term per lane, and a shuffle reads nothing from a lane that is not executing it)
struct PosTermRegs {
    double p[3][3];
    __device__ __forceinline__ double get(int r, int k) const { return p[r][k]; }
};

// the part of the epilogue that depends on the final positions alone (all lanes call: the quad broadcast of the first atom)
__device__ __forceinline__ void cepi_positions_final(const CEpiArgs &E, const Box &box, int cs, int a, int l, int mol, bool has, const double (&x)[3],
                                                     double q_nx, double hs_nx, double se_nx) {
    double x0[3];          // the molecule's first atom decides the image of the sorted copy
#pragma unroll
    for (int k = 0; k < 3; ++k) x0[k] = cquad_bcast<0>(x[k]);
    if (!has) return;
    if (l == 0 && E.spec_count) {
        // the molecule's cell at its new position, as k_cassign would find it (same wrap, same clamps)
        int cidx[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double w = cwrap1(x[k], box.L[k], box.invL[k]);
            const int ck = (w == w) ? (int)(w * E.grid.inv_cw[k]) : 0;
            cidx[k] = ck >= E.grid.nc[k] ? E.grid.nc[k] - 1 : (ck < 0 ? 0 : ck);
        }
        const int cell = (cidx[2] * E.grid.nc[1] + cidx[1]) * E.grid.nc[0] + cidx[0];
        const int rank = atomicAdd(&E.spec_count[cell], 1);
        if (rank < E.spec_capc) E.spec_members[(size_t)cell * E.spec_capc + rank] = mol;
        else E.spec_flags[7] = 1;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) E.x[3 * a + j] = x[j];
    if (E.xchg_x) {
        const size_t sl = 3 * (size_t)(cs - E.c_first) + l;
#pragma unroll
        for (int j = 0; j < 3; ++j) E.xchg_x[3 * sl + j] = x[j];
    }
    // the lists' displacement triggers (amm_watch_atom), the flags stored past this XCD's L2: the kernel's last block -- on another
    // XCD, perhaps -- decides on them whether the cells' counts become a cell table
    for (int q = 0; q < E.W.n; ++q) {
        const double dx = x[0] - E.W.xref[q][3 * a], dy = x[1] - E.W.xref[q][3 * a + 1], dz = x[2] - E.W.xref[q][3 * a + 2];
        const double d2 = dx * dx + dy * dy + dz * dz;
        if (!(d2 <= E.W.thr2[q])) {
            amm_st_l2(&E.W.flags[q][0], 1);
            if (!(d2 <= 4.0 * E.W.thr2[q])) amm_st_l2(&E.W.flags[q][AMM_FLAG_FAR], 1);
        }
    }
    if (E.posq_next) {
        double4 pq;
        pq.x = csorted_image(x[0], x0[0], cwrap1(x0[0], box.L[0], box.invL[0]) - x0[0], box.L[0], box.invL[0]);
        pq.y = csorted_image(x[1], x0[1], cwrap1(x0[1], box.L[1], box.invL[1]) - x0[1], box.L[1], box.invL[1]);
        pq.z = csorted_image(x[2], x0[2], cwrap1(x0[2], box.L[2], box.invL[2]) - x0[2], box.L[2], box.invL[2]);
        pq.w = q_nx;
        E.posq_next[3 * cs + l] = pq;
        if (E.lj_next) E.lj_next[3 * cs + l] = make_double2(hs_nx, se_nx);
    }
}

// NOT inlined, and called when the wavefront has walked ALL of its rows (a second loop over its tasks): inlined behind a task's
// walk, its arguments and temporaries entered the register allocation of the pair loop (the near kernel went from no scratch to 63
// scratch accesses inside the loop, the fused pass from 4 to 223), and as a call behind every task the values the pair loop keeps
// across tasks competed for the callee-saved registers (19 scratch accesses in the fused pass's loop).  Here nothing of the pair loop
// is live any more.  This launch's forces on the molecule are read back from the buffers the row's first lane stored them to
// (the caller orders those stores before these loads: same wavefront, same L1).
// (the box BY VALUE: a reference into the kernel's CPairArgs makes the compiler keep a private copy of the whole argument block in
// scratch, and the pair loop then reads A.box, A.margin ... from there on every trip)
#ifdef AMM_CPAIR_TIMING
#define CEPI_STAMP(k) (tstamp[k] = wall_clock64())
__device__ CEPI_INLINE void cepi_rows(const CEpiArgs &E, const Box box, int cs, bool valid, int sub, unsigned long long (&tstamp)[3]) {
#else
#define CEPI_STAMP(k)
__device__ CEPI_INLINE void cepi_rows(const CEpiArgs &E, const Box box, int cs, bool valid, int sub) {
#endif
    const int lane = threadIdx.x & 63;
    const int l = sub & 3;
    const int qb = lane - sub;                 // first lane of the row's lanes (a multiple of 4)
    const bool has = valid && sub < 3;         // this lane owns atom l of the row's molecule ...
    const bool tlane = valid && sub < 4;       // ... and evaluates its term l
    const int mol = E.cperm[cs];               // (rows beyond the slice alias a valid molecule and never store)
    const int a = 3 * mol + (l < 3 ? l : 0);
    const double m = E.mass[a];
    double x[3], v[3], f[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        x[j] = E.x[3 * a + j];
        v[j] = E.v[3 * a + j];
        f[j] = E.f0[3 * a + j];
    }
    int4 my_tl = make_int4(-1, -1, -1, -1);
    double4 my_tq = make_double4(0.0, 0.0, 0.0, 0.0);
    unsigned long long my_recs = 0ull;
    if (tlane) {
        my_tl = E.term_l[4 * mol + l];
        my_tq = E.term_q[4 * mol + l];
    }
    if (has) my_recs = E.atom_recs[a];
    // what the END of the loop needs from memory is fetched now: the records of the next sorted copy (charge; sigma/2, 2 sqrt(eps))
    double q_nx = 0.0, hs_nx = 0.0, se_nx = 0.0;
    if (has && E.posq_next && E.niter > 0) {
        q_nx = E.q_next[a];
        if (E.lj_next) {
            hs_nx = E.hsig_next[a];
            se_nx = E.seps2_next[a];
        }
    }
    const double rm = 1.0 / m;
    const bool rok = (__double_as_longlong(m) & 0xFFFFFFFFFFFFFll) != 0xFFFFFFFFFFFFFll && m > 1e-200 && m < 1e200;
    // the forces this launch stored (by the first lane of each row) are read back by the molecule's lanes from here on: the stores
    // have left the wavefront before these loads are issued, one L1 serves both.  (Behind the loads above, which do not wait for them.)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    CEPI_STAMP(0);
    {
#pragma clang fp contract(off)
        // the forces the first four kicks read are fetched in one go (a RESPA boundary has four: as loads inside the loop each kick
        // waited for its own: 0.6 us a kick for a wavefront that runs alone)
        double pa[4][3], pb[4][3];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const bool on = p < E.npre;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                pa[p][j] = on ? E.pre[p].a[3 * a + j] : 0.0;
                pb[p][j] = (on && E.pre[p].b) ? E.pre[p].b[3 * a + j] : 0.0;
            }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (p < E.npre) {
                const CEpiPre pk = E.pre[p];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    double ff = pa[p][j];
                    if (pk.b) ff = pk.plus ? ff + pb[p][j] : ff - pb[p][j];
                    const double num = pk.coef * ff;
                    const double dv = amm_div_mass(num, m, rm, rok);
                    v[j] = v[j] + dv;
                }
            }
        }
        for (int p = 4; p < E.npre; ++p) {
            const CEpiPre pk = E.pre[p];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                double ff = pk.a[3 * a + j];
                if (pk.b) ff = pk.plus ? ff + pk.b[3 * a + j] : ff - pk.b[3 * a + j];
                const double num = pk.coef * ff;
                const double dv = amm_div_mass(num, m, rm, rok);
                v[j] = v[j] + dv;
            }
        }
    }
    CEPI_STAMP(1);
    BondedArgs BA;                             // (harmonic bonds and angles read its box alone)
    BA.box = box;
    const int rec_n = (int)(my_recs >> 60);
    const long long code = __double_as_longlong(my_tq.w);
    const int kind = (int)(code & 1), periodic = (int)((code >> 5) & 1);
    const int ix[4] = {0, 1, 2, 3};
    const double par[3] = {my_tq.x, my_tq.y, my_tq.z};
    // slots (lanes of the quad) the term's atoms live in; a molecule has three atoms: slots 0 .. 2
    const int s0 = my_tl.x & 3, s1 = my_tl.y & 3, s2 = my_tl.z & 3;
    // (recompute_f0: pass -1 evaluates the terms at the positions as they are -- several ranks: a rebuild may have made this rank the
    // owner of molecules whose innermost forces another rank held; bonded(x) is the same number whoever computes it)
    for (int it = E.recompute_f0 ? -1 : 0; it < E.niter; ++it) {
        if (it >= 0) {
#pragma clang fp contract(off)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double num = E.c1 * f[j];
                const double dv = amm_div_mass(num, m, rm, rok);
                v[j] = v[j] + dv;
                const double dx = E.d * v[j];
                x[j] = x[j] + dx;
            }
        }
        // the positions are final after the last move: everything that depends on them alone -- the stores, the molecule's cell, the
        // triggers, the next sorted copy -- is issued here, behind the last evaluation of the terms and the last kick (an atomic with
        // a return value and a dozen stores: their latency was the tail of the kernel)
        if (it == E.niter - 1) cepi_positions_final(E, box, cs, a, l, mol, has, x, q_nx, hs_nx, se_nx);
        // the three atoms' new positions on all four lanes of the quad (DPP broadcasts), each term picks its own
        PosTermRegs pos;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double a0 = cquad_bcast<0>(x[k]), a1 = cquad_bcast<1>(x[k]), a2 = cquad_bcast<2>(x[k]);
            pos.p[0][k] = s0 == 0 ? a0 : (s0 == 1 ? a1 : a2);
            pos.p[1][k] = s1 == 0 ? a0 : (s1 == 1 ? a1 : a2);
            pos.p[2][k] = s2 == 0 ? a0 : (s2 == 1 ? a1 : a2);
        }
        double fo[4][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
        if (my_tl.x >= 0) {
            double e;
            bonded_term_forces(BA, pos, ix, par, kind, periodic, fo, e);
        }
        f[0] = f[1] = f[2] = 0.0;
        // the atom's records in order: (term slot, role) -> that term's force on that role.  An atom's records name the terms in
        // ascending slot order (BondedSet: terms and records are laid down in one sweep), a term at most once: so the four terms'
        // forces are broadcast over the quad one after the other (DPP) and an atom adds the role its next record names when the
        // record's term is the one on the air -- the sums of k_inner_lanes, in its order
        {
            int t_next = 0;                           // the atom's next record
#define AMM_CEPI_TERM(TL)                                                                                           \
            {                                                                                                       \
                const int rcode = (int)((my_recs >> (5 * t_next)) & 31ull);                                         \
                const bool mine = t_next < rec_n && (rcode & 7) == TL;                                               \
                const int role = rcode >> 3;                                                                        \
                _Pragma("unroll") for (int xx = 0; xx < 3; ++xx) {                                                  \
                    const double r0 = cquad_bcast<TL>(fo[0][xx]), r1 = cquad_bcast<TL>(fo[1][xx]), r2 = cquad_bcast<TL>(fo[2][xx]); \
                    const double add = role == 0 ? r0 : (role == 1 ? r1 : r2);                                       \
                    f[xx] = mine ? f[xx] + add : f[xx];                                                              \
                }                                                                                                   \
                t_next += mine ? 1 : 0;                                                                             \
            }
            AMM_CEPI_TERM(0)
            AMM_CEPI_TERM(1)
            AMM_CEPI_TERM(2)
            AMM_CEPI_TERM(3)
#undef AMM_CEPI_TERM
        }
        if (it >= 0) {
#pragma clang fp contract(off)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double num = E.c2 * f[j];
                const double dv = amm_div_mass(num, m, rm, rok);
                v[j] = v[j] + dv;
            }
        }
    }
    CEPI_STAMP(2);
    if (has) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (E.niter == 0) E.x[3 * a + j] = x[j];
            E.v[3 * a + j] = v[j];
            E.f0[3 * a + j] = f[j];
        }
        if (E.xchg_x) {                        // the other ranks take the molecule's state from here (k_state_scatter)
            const size_t sl = 3 * (size_t)(cs - E.c_first) + l;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (E.niter == 0) E.xchg_x[3 * sl + j] = x[j];
                E.xchg_v[3 * sl + j] = v[j];
            }
        }
    }
}

// exclusive scan of count[0 .. ncell) -> start[0 .. ncell] by ONE block of BS threads (wavefront shuffles: 64 bytes of LDS -- the
// fused pass has 5 KB to spare); the counts are left as they are (the next launch clears them); returns the largest count
template <int BS>
__device__ __forceinline__ int cscan_counts(int ncell, const int *count, int *start) {
    __shared__ int s_part[BS / 64], s_most[BS / 64];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int per = (ncell + BS - 1) / BS;
    const int c0 = min(t * per, ncell), c1 = min(c0 + per, ncell);
    int sum = 0, most = 0;
    for (int c = c0; c < c1; ++c) {
        const int v = amm_ld_l2(&count[c]);
        sum += v;
        most = max(most, v);
    }
    int incl = sum;
    for (int off = 1; off < 64; off <<= 1) {
        const int up = __shfl_up(incl, off);
        if (lane >= off) incl += up;
        most = max(most, __shfl_xor(most, off));
    }
    if (lane == 63) {
        s_part[w] = incl;
        s_most[w] = most;
    }
    __syncthreads();
    int base = 0, all_most = 0;
#pragma unroll
    for (int k = 0; k < BS / 64; ++k) {
        if (k < w) base += s_part[k];
        all_most = max(all_most, s_most[k]);
    }
    int run = base + incl - sum;
    for (int c = c0; c < c1; ++c) {
        start[c] = run;
        run += amm_ld_l2(&count[c]);
    }
    if (t == BS - 1) start[ncell] = run;
    return all_most;
}

#ifndef AMM_CBS_SINGLE
#define AMM_CBS_SINGLE 512
#endif

#ifndef AMM_CBS_NEAR          // one force per pass (the fused pass keeps AMM_CBS_SINGLE)
#define AMM_CBS_NEAR 512
#endif
#ifndef AMM_CTAB_WAVES_PER_EU
#define AMM_CTAB_WAVES_PER_EU 1
#endif
// LDS: [host Coulomb table][guest Coulomb table][host site-site table][guest site-site table][erfcx table][parameter strips];
// kernels with site-site tables need neither strips nor -- in LDS -- the erfcx table (their rare analytic path reads it from HBM)
template <int FAM, int CMODE, int GFAM, int BS, int SMASK, bool EPI = false>
__global__ void __launch_bounds__(BS) __attribute__((amdgpu_waves_per_eu(AMM_CTAB_WAVES_PER_EU)))
k_cpair(CPairArgs A, PairConsts c, PairConsts g, CEpiArgs E) {
    constexpr bool DUAL = GFAM >= 0;
    constexpr bool SS = SMASK != 0;
    extern __shared__ __align__(16) char s_lds[];
    if (EPI && E.spec_clear) {      // the cell counts the NEXT launch's epilogue files its molecules in: cleared here, a launch ahead
        for (int cidx = blockIdx.x * BS + threadIdx.x; cidx <= E.grid.ncell; cidx += gridDim.x * BS) E.spec_clear[cidx] = 0;
    }
#ifdef AMM_CPAIR_TIMING
    const unsigned long long t_entry = wall_clock64();
    unsigned long long t_staged = 0;
    int n_tasks_done = 0, n_interior = 0;
    unsigned long long t_epi[3] = {0, 0, 0};      // epilogue: forces visible / preceding kicks done / loop done (last task)
#endif
    auto stage = [&](int at, const double *src, int bytes) {
        for (int o = threadIdx.x * 16; o < bytes; o += BS * 16)
            *reinterpret_cast<double2 *>(s_lds + at + o) = *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(src) + o);
    };
    const int gbytes = DUAL ? A.guest_bytes : 0;
    stage(0, A.host_tab, A.host_bytes);
    if (DUAL) stage(A.host_bytes, A.guest_tab, gbytes);
    int used = A.host_bytes + gbytes;
    if (SS) {
        stage(used, A.host_tab_ss, A.host_ss_bytes);
        used += A.host_ss_bytes;
        if (DUAL) {
            stage(used, A.guest_tab_ss, A.guest_ss_bytes);
            used += A.guest_ss_bytes;
        }
    }
#if defined(AMM_EXP_TAB_GLOBAL)           // measurement only: the table is read through the vector-memory path instead of LDS
    const char *tabh = reinterpret_cast<const char *>(A.host_tab);
#else
    const char *tabh = s_lds;
#endif
    const char *tabg = s_lds + A.host_bytes;
    const double *erfcx = amm_erfcx_table_dev_c;
    double2 *s_li = nullptr;
    if (!SS) {
        double *s_erfcx = reinterpret_cast<double *>(s_lds + used);
        for (int k = threadIdx.x; k < AMM_ERFCX_NI * AMM_ERFCX_NC; k += BS) s_erfcx[k] = amm_erfcx_table_dev_c[k];
        erfcx = s_erfcx;
        // Lennard-Jones parameters of the rows' atoms: [wave][atom][lane] (read by the site-site pairs only)
        s_li = reinterpret_cast<double2 *>(s_erfcx + AMM_ERFCX_NI * AMM_ERFCX_NC) + (threadIdx.x >> 6) * 192;
    }
    __syncthreads();
#ifdef AMM_CPAIR_TIMING
    t_staged = wall_clock64();
#endif

    constexpr int WPB = BS / 64;
    const int lane = threadIdx.x & 63;
    const double sign = c.sign;
    PairConsts c1 = c, g1 = g;
    c1.sign = 1.0;          // the sign travels with the row atoms' charges and epsilons (every family is linear in both)
    g1.sign = 1.0;
    // one contiguous eighth of the rows per XCD (blockIdx & 7): consecutive cell-sorted rows = one slab of the box per L2
    const int xcd = blockIdx.x & 7, nwx = (gridDim.x >> 3) * WPB;
    const int row_end = min((xcd + 1) * A.rpx, A.nrows);
    for (int phase = 0; phase < A.nphase; ++phase) {
    const int shift = A.ph_shift[phase], row0 = xcd * A.rpx + A.ph_off[phase], ntask = A.ph_ntask[phase];
    const int lpa = 1 << shift;
    const int sub = lane & (lpa - 1);
    const int rpw = 64 >> shift;
#ifdef AMM_CPAIR_SWAP          // measurement: which tasks the last four wavefronts of a block take (is their lag theirs or their rows'?)
    for (int task = (int)(blockIdx.x >> 3) * WPB + ((int)(threadIdx.x >> 6) ^ 4); task < ntask; task += nwx) {
#else
    for (int task = (int)(blockIdx.x >> 3) * WPB + (int)(threadIdx.x >> 6); task < ntask; task += nwx) {
#endif
        const int a = row0 + task * rpw + (lane >> shift);
        const bool valid = a < row_end;
        const int cs = A.c_begin + (valid ? a : 0);
        double4 pi[3];
        int i_sites = 0;
        int so[3] = {0, 0, 0}, sog[3] = {0, 0, 0};
        if (!SS) __builtin_amdgcn_wave_barrier();        // the previous task's reads of the strip are done
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            pi[t] = A.posq[3 * cs + t];
            pi[t].w *= c.Kc * sign;
            double2 l = A.lj[3 * cs + t];
            l.y *= sign;
            if (!SS) s_li[64 * t + lane] = l;
            const bool site = valid && l.y != 0.0;
            if ((SMASK >> t) & 1) {
                so[t] = site ? A.host_ss_off : 0;
                if (DUAL) sog[t] = site ? A.guest_ss_off : 0;
            }
            if (__builtin_amdgcn_ballot_w64(site) != 0ull) i_sites |= 1 << t;
        }
        if (!SS) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        const double2 *li = SS ? nullptr : s_li + lane;  // li[64 a]
        const int nfront = valid ? A.nnb[a] : 0;
        const int nn = valid ? (A.nnb_total ? A.nnb_total[a] : nfront) : 0;
        const int *row = A.nl + (size_t)(valid ? a : 0) * A.cap;
        const bool edge = valid && !(pi[0].x >= A.margin && pi[0].x <= A.box.L[0] - A.margin && pi[0].y >= A.margin &&
                                     pi[0].y <= A.box.L[1] - A.margin && pi[0].z >= A.margin && pi[0].z <= A.box.L[2] - A.margin);
        const bool interior = __builtin_amdgcn_ballot_w64(edge) == 0ull;
#ifdef AMM_CPAIR_TIMING
        ++n_tasks_done;
        n_interior += interior ? 1 : 0;
#endif
        double f[9], fg[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) f[k] = fg[k] = 0.0;
        if (A.per_pair_image) cwalk_rows<FAM, CMODE, GFAM, 2, SMASK>(A, c1, g1, tabh, tabg, erfcx, pi, li, i_sites, so, sog, sign, row, nfront, nn, sub, lpa, cs, f, fg);
        else if (interior) cwalk_rows<FAM, CMODE, GFAM, 0, SMASK>(A, c1, g1, tabh, tabg, erfcx, pi, li, i_sites, so, sog, sign, row, nfront, nn, sub, lpa, cs, f, fg);
        else cwalk_rows<FAM, CMODE, GFAM, 1, SMASK>(A, c1, g1, tabh, tabg, erfcx, pi, li, i_sites, so, sog, sign, row, nfront, nn, sub, lpa, cs, f, fg);
        for (int off = lpa >> 1; off > 0; off >>= 1) {
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                f[k] += __shfl_xor(f[k], off);
                if (DUAL) fg[k] += __shfl_xor(fg[k], off);
            }
        }
        if (valid && sub == 0) {
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int i = A.sorted_out ? 3 * a + t : A.aperm[3 * cs + t];
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    // (host first: when both forces go to the same buffer the guest adds to what the host just wrote)
                    if (A.accumulate) A.force[3 * i + d] += f[3 * t + d];
                    else A.force[3 * i + d] = f[3 * t + d];
                    if (DUAL) {
                        if (A.g_accumulate) A.gforce[3 * i + d] += fg[3 * t + d];
                        else A.gforce[3 * i + d] = fg[3 * t + d];
                    }
                }
            }
        }
        // the inner RESPA loop of the rows' molecules (cepi_rows): the row's sums are in f / fg on all of its lanes
    }
    }
#ifdef AMM_CPAIR_TIMING
    if (EPI) t_staged = wall_clock64();        // (measurement builds, kernels with the epilogue: the second clock = rows walked)
#endif
    if (EPI) {
        // the inner RESPA loop of the molecules whose rows this wavefront has just summed (cepi_rows): the same tasks again.  The
        // forces were stored by the first lane of each row; the molecule's lanes read them back (wavefront-scope ordering: the
        // stores have left the wavefront before the loads are issued; one L1 serves both)
        // (the ordering fence sits inside cepi_rows, behind the loads that do not depend on this launch's forces)
        for (int phase = 0; phase < A.nphase; ++phase) {
            const int shift = A.ph_shift[phase], row0 = xcd * A.rpx + A.ph_off[phase], ntask = A.ph_ntask[phase];
            const int rpw = 64 >> shift;
            for (int task = (int)(blockIdx.x >> 3) * WPB + (int)(threadIdx.x >> 6); task < ntask; task += nwx) {
                const int a = row0 + task * rpw + (lane >> shift);
                const bool valid = a < row_end;
#ifdef AMM_CPAIR_TIMING
                cepi_rows(E, A.box, A.c_begin + (valid ? a : 0), valid, lane & ((1 << shift) - 1), t_epi);
#else
                cepi_rows(E, A.box, A.c_begin + (valid ? a : 0), valid, lane & ((1 << shift) - 1));
#endif
            }
        }
        // the molecules are filed in their new cells (cepi_rows).  If some trigger now asks for a rebuild, the last block to get here
        // turns the counts into the cells' start offsets: the next evaluation's chain begins at the sort, no assign launch
        if (E.spec_count && amm_last_block(E.spec_ticket)) {
            if (amm_ld_l2(&E.spec_flags[0])) {
                const int fullest = cscan_counts<BS>(E.grid.ncell, E.spec_count, E.spec_start);
                if (threadIdx.x == 0) E.spec_flags[6] = fullest;
            }
        }
    }
#ifdef AMM_CPAIR_TIMING
    if (A.wave_times && (threadIdx.x & 63) == 0) {
        unsigned long long *o = A.wave_times + 4 * ((size_t)blockIdx.x * (BS / 64) + (threadIdx.x >> 6));
        o[0] = t_entry;
        o[1] = t_staged;
        o[2] = wall_clock64();
        // (kernels with the epilogue: the fourth word holds three 20-bit offsets from "rows walked", in clock ticks of 10 ns)
        o[3] = EPI ? (((t_epi[0] - t_staged) & 0xfffffull) | (((t_epi[1] - t_staged) & 0xfffffull) << 20) | (((t_epi[2] - t_staged) & 0xfffffull) << 40))
                   : (((unsigned long long)n_interior << 8) | (unsigned long long)n_tasks_done);
    }
#endif
}

// How the rows of one XCD are shared out among its `waves` resident wavefronts.  A task of 64 >> s rows costs a wavefront about
// AMM_PHASE_COST[s] (relative to 16 rows at 4 lanes each: more lanes per row = shorter rows, but the row's prologue and the
// reduction weigh more; from the per-rank table of profiles/, launch and table staging taken off).  With one task size the
// slowest wavefront walks ceil(tasks / waves) of them -- 3 where the mean is 2.5 at 82 015 rows: a sixth of the chip idle.
// Here: whole rounds of the biggest tasks, then the remainder in whole rounds of a smaller size ..., at every level the cheaper
// of "finish with this size" and "a smaller size for what is left".  (The order in which a row's entries are summed depends on
// its lanes per row, i.e. on the phase the row falls in; the plan is a function of (rows, waves, first size) alone, so a system
// gets the same sums on every evaluation and every run.)  At most 5 sizes (4 ... 64 lanes per row) < AMM_CPHASES phases.
static const double AMM_PHASE_COST[7] = {0, 0, 1.0, 0.55, 0.32, 0.18, 0.11};
static double cpair_plan_rec(int rows, int waves, int s, int off, CPairArgs *P) {
    const int rpw = 64 >> s, per_round = waves * rpw;
    const int full = rows / per_round, left = rows - full * per_round;
    const double finish = (double)((rows + per_round - 1) / per_round) * AMM_PHASE_COST[s];
    auto put = [&](int ntask) {
        if (P && ntask > 0) {
            P->ph_off[P->nphase] = off;
            P->ph_shift[P->nphase] = s;
            P->ph_ntask[P->nphase] = ntask;
            ++P->nphase;
        }
    };
    int s2 = 6;
    double split = 1e300;
    for (int t = s + 1; t <= 6 && left > 0; ++t) {           // the next size: the one that finishes the remainder cheapest
        const double v = cpair_plan_rec(left, waves, t, 0, nullptr);
        if (v < split) {
            split = v;
            s2 = t;
        }
    }
    split += full * AMM_PHASE_COST[s];
    if (left == 0 || s == 6 || finish <= split) {
        put((rows + rpw - 1) / rpw);
        return finish;
    }
    put(full * waves);
    cpair_plan_rec(left, waves, s2, off + full * per_round, P);
    return split;
}
static void cpair_plan(CPairArgs &P, int waves, int enabled) {
    P.rpx = (P.nrows + 7) / 8;
    P.nphase = 0;
    if (enabled) {
        cpair_plan_rec(P.rpx, waves, P.lpa_shift, 0, &P);
    } else {
        const int rpw = 64 >> P.lpa_shift;
        P.ph_off[0] = 0;
        P.ph_shift[0] = P.lpa_shift;
        P.ph_ntask[0] = (P.rpx + rpw - 1) / rpw;
        P.nphase = 1;
    }
}

// per (device, kernel) launch configuration: dynamic LDS attribute + blocks per CU from the occupancy query
struct CLaunchCfg {
    int lds_set = 0, bpc = -1;
};
static int g_num_cu_c[64] = {0};

// 2 wavefronts per SIMD (187 registers one force, 240 fused): one block of 512 per CU; 8 rows per wavefront then deal 2 tasks to
// every wavefront at 98 304 atoms (768 threads: 1.33 -- a third of the chip idles in the tail)
template <int FAM, int CMODE, int GFAM, int SMASK, bool EPI = false>
static int launch_cpair_t(amm_ctx *ctx, const CPairArgs &A, const PairConsts &c, const PairConsts &g, const CEpiArgs &E = CEpiArgs()) {
    constexpr bool DUAL = GFAM >= 0;
    constexpr int BS = DUAL ? AMM_CBS_SINGLE : AMM_CBS_NEAR;
    constexpr bool SS = SMASK != 0;
    static CLaunchCfg cfg[64];
    CLaunchCfg &k = cfg[ctx->device & 63];
    int lds = A.host_bytes + (DUAL ? A.guest_bytes : 0);
    if (SS) lds += A.host_ss_bytes + (DUAL ? A.guest_ss_bytes : 0);
    else lds += AMM_ERFCX_NI * AMM_ERFCX_NC * 8 + (BS / 64) * 192 * 16;
    auto kern = k_cpair<FAM, CMODE, GFAM, BS, SMASK, EPI>;
    if (lds > k.lds_set) {
        AMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        k.lds_set = lds;
        k.bpc = -1;
    }
    if (k.bpc < 0) {
        int nb = 0;
        AMM_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, BS, (size_t)lds));
        if (nb < 1) {
            amm_set_error("molecule-row pair kernel does not fit on a CU (LDS)");
            return 1;
        }
        k.bpc = nb;
    }
    int &ncu = g_num_cu_c[ctx->device & 63];
    if (!ncu) AMM_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device));
    constexpr int WPB = BS / 64;
    long nblk = std::min((long)ncu * k.bpc, ((long)A.ntask + WPB - 1) / WPB);
    nblk = std::max(8L, (nblk + 7) / 8 * 8);
    CPairArgs P = A;
    cpair_plan(P, (int)(nblk >> 3) * WPB, ctx->opt_row_phases);
#ifdef AMM_CPAIR_TIMING
    // the AMM_WAVE_TIMES-th launch of this kernel writes its wavefronts' clocks to $AMM_WAVE_TIMES_OUT.<fused|single> (measurement builds)
    static int launches = 0;
    static unsigned long long *d_times = nullptr;
    const char *which = std::getenv("AMM_WAVE_TIMES");
    P.wave_times = nullptr;
    if (which && ++launches == std::atoi(which)) {
        const size_t nw = (size_t)nblk * WPB;
        AMM_HIP(hipMalloc(&d_times, sizeof(unsigned long long) * 4 * nw));
        P.wave_times = d_times;
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(BS), (size_t)lds, ctx->stream, P, c, g, E);
        std::vector<unsigned long long> h(4 * nw);
        AMM_HIP(hipMemcpy(h.data(), d_times, sizeof(unsigned long long) * 4 * nw, hipMemcpyDeviceToHost));
        std::string path = std::string(std::getenv("AMM_WAVE_TIMES_OUT") ? std::getenv("AMM_WAVE_TIMES_OUT") : "/tmp/wave_times") + (DUAL ? ".fused" : ".single") + (EPI ? "_epi" : "");
        if (FILE *fp = std::fopen(path.c_str(), "wb")) {
            std::fwrite(h.data(), sizeof(unsigned long long), h.size(), fp);
            std::fclose(fp);
        }
        return 0;
    }
#endif
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(BS), (size_t)lds, ctx->stream, P, c, g, E);
    return 0;
}

#define AMM_CPAIR_LDS_LIMIT (160 * 1024)
// site-site tables are used when the force has one (A.host_ss_bytes > 0; for a fused pass: both forces) and everything fits LDS
// E: the inner RESPA loop to run as the kernel's epilogue (cepi_rows; kernels with site-site tables only -- without one the caller
// is told, *epi_done stays false, and runs the loop as launches of its own)
template <int FAM, int CMODE, int GFAM>
static int launch_cpair_s(amm_ctx *ctx, CPairArgs A, const PairConsts &c, const PairConsts &g, const CEpiArgs *E = nullptr, bool *epi_done = nullptr) {
    constexpr bool DUAL = GFAM >= 0;
    bool ss = ctx->opt_site_tab && A.host_ss_bytes > 0 && (!DUAL || A.guest_ss_bytes > 0);
    if (ss && A.host_bytes + A.host_ss_bytes + (DUAL ? A.guest_bytes + A.guest_ss_bytes : 0) > AMM_CPAIR_LDS_LIMIT) ss = false;
#ifndef AMM_CLUSTER_TUNE
    if (ss && E) {
        if (epi_done) *epi_done = true;
        return A.site_atoms == 1 ? launch_cpair_t<FAM, CMODE, GFAM, 1, true>(ctx, A, c, g, *E) : launch_cpair_t<FAM, CMODE, GFAM, 7, true>(ctx, A, c, g, *E);
    }
#endif
    // (A.site_atoms: the row atoms that are sites in some molecule -- 1 for three-site water: only the first)
    if (ss) return A.site_atoms == 1 ? launch_cpair_t<FAM, CMODE, GFAM, 1>(ctx, A, c, g) : launch_cpair_t<FAM, CMODE, GFAM, 7>(ctx, A, c, g);
    PairConsts c0 = c, g0 = g;
    c0.tab.ss_first = g0.tab.ss_first = -1;
    A.host_ss_bytes = A.guest_ss_bytes = 0;
    return launch_cpair_t<FAM, CMODE, GFAM, 0>(ctx, A, c0, g0);
}

// fused pass: which (host, guest) pairs have a kernel.  Returns -1 when there is none (the caller launches the two forces one
// after the other), 0 / 1 as the launch functions
static int launch_cdual(amm_ctx *ctx, const CPairArgs &A, const PairConsts &c, const PairConsts &g, const CEpiArgs *E = nullptr, bool *epi_done = nullptr) {
    if (g.family != AMM_NEAR_FSWITCH) return -1;
    if (c.family == AMM_DAMPED && c.degree == 1) return launch_cpair_s<AMM_DAMPED, 1, AMM_NEAR_FSWITCH>(ctx, A, c, g, E, epi_done);
#ifndef AMM_CLUSTER_TUNE
    if (c.family == AMM_DAMPED) return launch_cpair_s<AMM_DAMPED, 0, AMM_NEAR_FSWITCH>(ctx, A, c, g, E, epi_done);
    if (c.family == AMM_NONBONDED) {
        if (c.cmode == 1) return launch_cpair_s<AMM_NONBONDED, 1, AMM_NEAR_FSWITCH>(ctx, A, c, g, E, epi_done);
        if (c.cmode == 2) return launch_cpair_s<AMM_NONBONDED, 2, AMM_NEAR_FSWITCH>(ctx, A, c, g, E, epi_done);
        return launch_cpair_s<AMM_NONBONDED, 0, AMM_NEAR_FSWITCH>(ctx, A, c, g, E, epi_done);
    }
#endif
    return -1;
}

static int launch_cpair(amm_ctx *ctx, const CPairArgs &A, const PairConsts &c, const CEpiArgs *E = nullptr, bool *epi_done = nullptr) {
#ifdef AMM_CLUSTER_TUNE      // kernel tuning builds (scripts/build_variant.sh cluster ...): the two instantiations of the bench only
    if (c.family == AMM_NEAR_FSWITCH) return launch_cpair_s<AMM_NEAR_FSWITCH, 0, -1>(ctx, A, c, c);
    if (c.family == AMM_DAMPED && c.degree == 1) return launch_cpair_s<AMM_DAMPED, 1, -1>(ctx, A, c, c);
    amm_set_error("this is a kernel-tuning build (AMM_CLUSTER_TUNE): it holds the near force-switch and degree-1 damped kernels only");
    return 1;
#else
    switch (c.family) {
    case AMM_NEAR_NONE: return launch_cpair_s<AMM_NEAR_NONE, 0, -1>(ctx, A, c, c, E, epi_done);
    case AMM_NEAR_SHIFT: return launch_cpair_s<AMM_NEAR_SHIFT, 0, -1>(ctx, A, c, c, E, epi_done);
    case AMM_NEAR_FSWITCH: return launch_cpair_s<AMM_NEAR_FSWITCH, 0, -1>(ctx, A, c, c, E, epi_done);
    case AMM_DAMPED:
        if (c.degree == 1) return launch_cpair_s<AMM_DAMPED, 1, -1>(ctx, A, c, c, E, epi_done);
        return launch_cpair_s<AMM_DAMPED, 0, -1>(ctx, A, c, c, E, epi_done);
    default:
        if (c.cmode == 1) return launch_cpair_s<AMM_NONBONDED, 1, -1>(ctx, A, c, c, E, epi_done);
        if (c.cmode == 2) return launch_cpair_s<AMM_NONBONDED, 2, -1>(ctx, A, c, c, E, epi_done);
        return launch_cpair_s<AMM_NONBONDED, 0, -1>(ctx, A, c, c, E, epi_done);
    }
#endif
}

// ------------------------------------------------------------------------------------------------ exchange of the molecules' state
// Multi-rank, owner-integrates: the rank that walks a molecule's rows also runs its inner RESPA loop (cepi_rows)
// and leaves the molecule's new positions and velocities in its chunk of the exchange buffer; the chunks are all-gathered and this
// kernel -- one thread per molecule of the OTHER ranks -- spreads them to the atom-order arrays, evaluates the lists' displacement
// triggers for them (every rank then holds the same flags: its own molecules' from the epilogue, the others' from here) and writes
// their records of the next pair evaluation's sorted copies, as the epilogue did for the rank's own.  Replaces, per evaluation, the
// unsort of the gathered forces, the redundant inner loop over all atoms and the gather of the sorted copies.
__global__ void __launch_bounds__(256) k_state_scatter(int nc, int per_c, int c_begin, int c_end, const int *__restrict__ aperm,
                                                       const double *__restrict__ xchg, double *x, double *v, Box box, WatchArgs W,
                                                       double4 *posq_next, double2 *lj_next, const double *__restrict__ q_next,
                                                       const double *__restrict__ hsig_next, const double *__restrict__ seps2_next) {
    // one thread per ATOM of the other ranks' molecules (a thread per molecule left the launch latency bound: 12.6 us for 98 304 atoms,
    // 87 us for 786 432); the chunk reads are contiguous, the atom-order writes are the scattered part
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = s / 3, a = s - 3 * c;
    if (c >= nc || (c >= c_begin && c < c_end)) return;
    const int r = c / per_c, loc = c - r * per_c;
    const size_t per = 3 * (size_t)per_c;
    const double *sx = xchg + ((size_t)r * 2 * per + 3 * (size_t)loc) * 3, *sv = sx + per * 3;
    const int i = aperm[s];
    double p[3], p0[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        p[k] = sx[3 * a + k];
        p0[k] = sx[k];
        x[3 * i + k] = p[k];
        v[3 * i + k] = sv[3 * a + k];
    }
    amm_watch_atom(W, i, p);
    if (posq_next) {
        posq_next[s] = make_double4(csorted_image(p[0], p0[0], cwrap1(p0[0], box.L[0], box.invL[0]) - p0[0], box.L[0], box.invL[0]),
                                    csorted_image(p[1], p0[1], cwrap1(p0[1], box.L[1], box.invL[1]) - p0[1], box.L[1], box.invL[1]),
                                    csorted_image(p[2], p0[2], cwrap1(p0[2], box.L[2], box.invL[2]) - p0[2], box.L[2], box.invL[2]), q_next[i]);
        if (lj_next) lj_next[s] = make_double2(hsig_next[i], seps2_next[i]);
    }
}

int amm_cluster_state_finish_impl(amm_ctx *ctx) {
    PendingExchange &pe = ctx->pending;
    ClusterList *cl = pe.cl;
    WatchArgs W;
    amm_collect_watches(ctx, W);
    PairForce *nx = pe.next;
    hipLaunchKernelGGL(k_state_scatter, dim3((3 * cl->nc + 255) / 256), dim3(256), 0, ctx->stream, cl->nc, pe.per / 3, cl->c_begin, cl->c_end,
                       cl->d_aperm, ctx->d_xchg, ctx->d_x, ctx->d_v, ctx->box, W, nx ? nx->d_posq_s : (double4 *)nullptr,
                       nx ? nx->d_lj_s : (double2 *)nullptr, nx ? nx->d_q : nullptr, nx ? nx->d_hsig : nullptr, nx ? nx->d_seps2 : nullptr);
    AMM_HIP(hipGetLastError());
    amm_watch_moved(ctx);              // (the triggers of these positions are evaluated now: the owners' in the epilogue, the others' here)
    if (nx) {
        cl->sorted_for = nx;
        cl->sorted_epoch = ctx->pos_epoch;
        cl->sorted_pos = ctx->d_x;
    }
    pe.active = false;
    pe.kind = 0;
    return 0;
}

// ------------------------------------------------------------------------------------------------ host orchestration
static int cluster_setup_grid(amm_ctx *ctx, ClusterList *cl, double rc) {
    CellGrid &g = cl->grid;
    g.ncell = 1;
    const double reach = cl->rlist_build + 2.0 * cl->rext;     // first atoms of two listed molecules are closer than this
    for (int k = 0; k < 3; ++k) {
        const double L = ctx->box.L[k];
        int nc = (int)floor(L / (0.5 * reach));
        if (nc < 1) nc = 1;
        if (nc > 512) nc = 512;
        g.nc[k] = nc;
        g.h[k] = 2;
        g.nstencil[k] = nc >= 5 ? 5 : nc;
        g.cw[k] = L / nc;
        g.inv_cw[k] = nc / L;
        g.ncell *= nc;
        // one image per molecule pair is the nearest image of every atom pair within the cutoff only if rc + 2 rext < L / 2
        if (!(rc + cl->skin + 2.0 * cl->rext < 0.5 * L * (1 - 1e-9))) cl->per_pair_image = true;
    }
    return 0;
}

static int cluster_chain(amm_ctx *ctx, PairForce *L, ClusterList *cl, const double *d_pos, int force, bool count_only, PairForce *gather_for,
                         CZeroRows Z = CZeroRows{0, nullptr, nullptr, nullptr}, int copies_current = 0, bool skip_assign = false) {
    hipStream_t st = ctx->stream;
    const int nc = cl->nc;
    // (skip_assign: the launch that moved the atoms filed every molecule in its cell and, if a rebuild is due, scanned the counts)
    if (!skip_assign)
        hipLaunchKernelGGL(k_cassign, dim3((nc + cl->nrest + 255) / 256), dim3(256), 0, st, nc, d_pos, ctx->box, cl->grid, cl->d_cell_count, cl->d_cell_start,
                           cl->d_cell_members, cl->capc, count_only ? (double *)nullptr : cl->d_xref, cl->d_flags, cl->d_ticket, force, cl->d_first,
                           cl->d_rest, cl->nrest);
    if (!cl->d_cell_members) return 0;
    PairForce *gf = gather_for;
    const long sort_threads = std::max(std::max((long)cl->grid.ncell * 64, gf ? (long)nc : 0L), (long)Z.n);
    hipLaunchKernelGGL(k_csort_gather, dim3((unsigned)((sort_threads + 255) / 256)), dim3(256), 0, st, cl->grid.ncell, nc, cl->d_cell_start,
                       cl->d_cell_members, cl->capc, cl->d_cperm, cl->d_aperm, d_pos, ctx->box, cl->d_pos4f, cl->d_flags, cl->d_flags, force,
                       gf ? gf->d_q : nullptr, gf ? gf->d_hsig : nullptr, gf ? gf->d_seps2 : nullptr, gf ? gf->d_posq_s : (double4 *)nullptr,
                       gf ? gf->d_lj_s : (double2 *)nullptr, (float)cl->rext, L->d_seps2, cl->d_first, Z, copies_current,
                       count_only ? (double *)nullptr : cl->d_xref, cl->c_begin, cl->c_end, cl->d_slice_cells);
    // A rank's slice (rows c_begin .. c_end of the sorted order): blocks for the cells of the slice only (a quarter more than its
    // share of the cells + 4; the kernels stride over the units should the slice span more), and -- k_cbuild_split -- a block per
    // (cell, part) with both passes shared out over its wavefronts
    const bool slice = cl->c_end - cl->c_begin < cl->nc;
    const bool split = cl->split_parts > 0 && !count_only;
    const int ncell_grid = slice ? (int)std::min<long>(cl->grid.ncell, (5L * cl->grid.ncell) / (4L * std::max(1, ctx->world)) + 4) : cl->grid.ncell;
    const long threads = (long)ncell_grid * cl->parts * 64;
    dim3 grid(split ? (unsigned)(ncell_grid * cl->split_parts) : (unsigned)((threads + 255) / 256));
    CBoxF bf;
    for (int k = 0; k < 3; ++k) {
        bf.L[k] = (float)ctx->box.L[k];
        bf.invL[k] = (float)ctx->box.invL[k];
    }
    const float rl = (float)cl->rlist_build;
    const float rn2 = cl->rnear_build > 0 ? (float)(cl->rnear_build * cl->rnear_build) : 3.0e38f;
    const bool use_rint = cl->grid.nc[0] < 5 || cl->grid.nc[1] < 5 || cl->grid.nc[2] < 5;
#define AMM_LAUNCH_CBUILD(CO, RI)                                                                                                   \
    do {                                                                                                                            \
        if (slice)                                                                                                                  \
            hipLaunchKernelGGL((k_cbuild<CO, RI, true>), grid, dim3(256), 0, st, cl->c_begin, cl->c_end, cl->parts, cl->d_cell_start,    \
                               cl->d_pos4f, bf, cl->grid, rl, rn2, cl->cap, cl->d_nl, cl->d_nnb, cl->d_nnb_near, cl->d_flags,            \
                               cl->d_blockstats, cl->d_counters, cl->d_ticket + AMM_TICKET_INTS, force, cl->d_slice_cells);              \
        else                                                                                                                        \
            hipLaunchKernelGGL((k_cbuild<CO, RI, false>), grid, dim3(256), 0, st, cl->c_begin, cl->c_end, cl->parts, cl->d_cell_start,   \
                               cl->d_pos4f, bf, cl->grid, rl, rn2, cl->cap, cl->d_nl, cl->d_nnb, cl->d_nnb_near, cl->d_flags,            \
                               cl->d_blockstats, cl->d_counters, cl->d_ticket + AMM_TICKET_INTS, force, cl->d_slice_cells);              \
    } while (0)
#define AMM_LAUNCH_CBUILD_SPLIT(RI)                                                                                                  \
    hipLaunchKernelGGL((k_cbuild_split<RI>), grid, dim3(256), 0, st, cl->c_begin, cl->c_end, cl->split_parts, cl->d_cell_start,     \
                       cl->d_pos4f, bf, cl->grid, rl, rn2, cl->cap, cl->d_nl, cl->d_nnb, cl->d_nnb_near, cl->d_flags, cl->d_blockstats,  \
                       cl->d_counters, cl->d_ticket + AMM_TICKET_INTS, force, cl->d_slice_cells)
    if (count_only) {
        if (use_rint) AMM_LAUNCH_CBUILD(true, true);
        else AMM_LAUNCH_CBUILD(true, false);
    } else if (split) {
        if (use_rint) AMM_LAUNCH_CBUILD_SPLIT(true);
        else AMM_LAUNCH_CBUILD_SPLIT(false);
    } else {
        if (use_rint) AMM_LAUNCH_CBUILD(false, true);
        else AMM_LAUNCH_CBUILD(false, false);
    }
#undef AMM_LAUNCH_CBUILD
#undef AMM_LAUNCH_CBUILD_SPLIT
    AMM_HIP(hipGetLastError());
    return 0;
}

// largest distance of an atom from the first atom of its molecule (host, first build only)
// (minimum image: a configuration whose atoms were wrapped into the box one by one still has whole molecules)
static double cluster_extent_host(amm_ctx *ctx, const double *d_pos, int n, int nc, const std::vector<int> &first) {
    std::vector<double> x(3 * (size_t)n);
    if (hipMemcpyAsync(x.data(), d_pos, sizeof(double) * 3 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return -1.0;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return -1.0;
    double worst = 0.0;
    for (int m = 0; m < nc; ++m) {
        const size_t i0 = first.empty() ? 3 * (size_t)m : (size_t)first[m];
        for (int a = 1; a < 3; ++a) {
            double d2 = 0.0;
            for (int k = 0; k < 3; ++k) {
                double d = x[3 * (i0 + a) + k] - x[3 * i0 + k];
                d -= ctx->box.L[k] * std::nearbyint(d * ctx->box.invL[k]);
                d2 += d * d;
            }
            worst = std::max(worst, d2);
        }
    }
    return std::sqrt(worst);
}

static int cluster_first_build(amm_ctx *ctx, PairForce *L, const double *d_pos) {
    ClusterList *cl = new ClusterList();
    L->cl = cl;
    const int n = L->n, nc = L->n_mol;
    cl->nc = nc;
    cl->d_first = L->d_mol_first;
    cl->d_rest = L->d_rest_idx;
    cl->nrest = L->n_rest;
    cl->skin = L->skin;
    cl->rlist_build = L->rlist_build;
    cl->rnear_build = L->rnear_build;
    const double ext = cluster_extent_host(ctx, d_pos, n, nc, L->h_mol_first);
    if (!(ext >= 0.0) || !(ext == ext)) {
        amm_set_error("molecule-row list: cannot read the positions (or NaN positions)");
        return 1;
    }
    // bound on the molecules' extent: the cells and the interior margin are sized for it; a molecule that stretches beyond it
    // is reported by amm_check (flags[7])
    cl->rext = std::max(1.5 * ext, ext + 0.05);
    if (cluster_setup_grid(ctx, cl, L->desc.rc)) return 1;
    const int per = (nc + ctx->world - 1) / ctx->world;
    cl->c_begin = std::min(nc, ctx->rank * per);
    cl->c_end = std::min(nc, cl->c_begin + per);
    const size_t ns = (size_t)std::max(cl->c_end - cl->c_begin, 1);
    const int ncell = cl->grid.ncell;
    AMM_HIP(hipMalloc(&cl->d_cell_count, sizeof(int) * (ncell + 1)));
    AMM_HIP(hipMemset(cl->d_cell_count, 0, sizeof(int) * (ncell + 1)));
    AMM_HIP(hipMalloc(&cl->d_cell_start, sizeof(int) * (ncell + 1)));
    AMM_HIP(hipMalloc(&cl->d_cperm, sizeof(int) * nc));
    AMM_HIP(hipMalloc(&cl->d_aperm, sizeof(int) * 3 * (size_t)nc));
    AMM_HIP(hipMalloc(&cl->d_pos4f, sizeof(float4) * 3 * (size_t)nc));
    AMM_HIP(hipMalloc(&cl->d_xref, sizeof(double) * 3 * (size_t)n));
    AMM_HIP(hipMalloc(&cl->d_nnb, sizeof(int) * ns));
    AMM_HIP(hipMalloc(&cl->d_nnb_near, sizeof(int) * ns));
    AMM_HIP(hipMalloc(&cl->d_flags, sizeof(int) * 16));
    AMM_HIP(hipMemset(cl->d_flags, 0, sizeof(int) * 16));
    AMM_HIP(hipMalloc(&cl->d_ticket, sizeof(int) * 4 * AMM_TICKET_INTS));
    AMM_HIP(hipMemset(cl->d_ticket, 0, sizeof(int) * 4 * AMM_TICKET_INTS));
    AMM_HIP(hipMalloc(&cl->d_counters, sizeof(unsigned long long) * 8));
    AMM_HIP(hipMemset(cl->d_counters, 0, sizeof(unsigned long long) * 8));
    AMM_HIP(hipMalloc(&cl->d_slice_cells, sizeof(int) * 2));
    AMM_HIP(hipMemset(cl->d_slice_cells, 0, sizeof(int) * 2));
    // lanes per row: 16 rows per wavefront on a whole box; a rank's slice has fewer rows than the persistent grid has wavefronts
    // (256 CUs x 8), so rows are shared out over more lanes until every wavefront has one task (measured on slices of the
    // 98 304-atom box, dual pass: 2 ranks 113 us with 8 lanes; 4 ranks 68 with 16 (107 with 8); 8 ranks 44 with 32 (105 with 8))
    // (whole boxes: 4 lanes = 16 rows per wavefront, one task per wavefront at 98 304 atoms: near / outer / fused 58 / 118 / 153 us
    // against 60 / 121 / 156 with 8; at 255 552 atoms 142 / 314 / 416 against 150 / 319 / 427)
    int lpa = 4;
    while (lpa < 64 && (long)(cl->c_end - cl->c_begin) * lpa < 64L * 2048) lpa <<= 1;
    if (ctx->opt_lpa > 0) lpa = ctx->opt_lpa;
    cl->lpa = lpa;
    int flags[8];
    // member tables of the cell list: twice the fullest cell of the first configuration + 16
    cl->capc = 0;
    if (cluster_chain(ctx, L, cl, d_pos, 1, true, nullptr)) return 1;
    AMM_HIP(hipMemcpyAsync(flags, cl->d_flags, sizeof(flags), hipMemcpyDeviceToHost, ctx->stream));
    AMM_HIP(hipStreamSynchronize(ctx->stream));
    cl->capc = 2 * flags[6] + 16;
    AMM_HIP(hipMalloc(&cl->d_cell_members, sizeof(int) * (size_t)ncell * cl->capc));
    // wavefronts per cell: a wave walks the cell's whole candidate stream once per batch of CB_BATCH row molecules, so batches
    // should be full -- but the build is a chain of dependent reads (piece search, gather, ring, gather) that only more
    // wavefronts hide: 1.5 x the parts that would just fill the batches at the MEAN occupancy measured fastest
    // (98 304 atoms, 11.5 molecules per cell, us per rebuild: batch 8 x 2 parts 150, 6 x 3 127, 5 x 4 125, 4 x 4 131, 3 x 6 138).
    // A rank's slice covers 1 / world of the cells: its rows are spread over more waves, up to one row each (slice of 1/8: 99 us
    // with 2 parts, 67 with 5-6)
    cl->parts = std::max(1, std::min(16, (int)std::ceil(1.5 * (double)nc / ncell / CB_BATCH)));
    if (ctx->world > 1) {
        const int active_cells = std::max(1, ncell / ctx->world);
        const int want = (4096 + active_cells - 1) / active_cells, most = std::max(1, (int)std::ceil((double)nc / ncell));
        cl->parts = std::max(cl->parts, std::min(want, std::min(most, 6)));      // (1/8 slice, us: 3 parts 87, 4 73, 5 67, 6 68, 8 79, 12 83)
    }
    if (ctx->opt_parts > 0) cl->parts = std::max(1, std::min(16, ctx->opt_parts));
    // split-stream build of a slice (k_cbuild_split; option build_split: -1 = this choice, k = k blocks per cell; default 0 = off -- on
    // the 1/8 slice of the 98 304-atom box it takes 67 us for the whole chain against 65 of k_cbuild<.., SLICE>):
    // blocks per cell so that the slice's cells give about one round of resident blocks (four per CU: 40 KB of LDS each), and no
    // more than give every block a full batch
    cl->split_parts = 0;
    if (ctx->world > 1 && ctx->opt_build_split < 0) {
        const int active_cells = std::max(1, ncell / ctx->world);
        const int most = std::max(1, (int)std::ceil((double)nc / ncell / CBS_BATCH));
        cl->split_parts = std::max(1, std::min(std::min(16, most), (1024 + active_cells / 2) / active_cells));
    }
    if (ctx->opt_build_split == 0) cl->split_parts = 0;
    if (ctx->opt_build_split > 0) cl->split_parts = std::min(16, ctx->opt_build_split);
    {
        const long t1 = (long)ncell * cl->parts * 64;
        const size_t nblocks = std::max((size_t)((t1 + 255) / 256), (size_t)ncell * (size_t)cl->split_parts);
        AMM_HIP(hipMalloc(&cl->d_blockstats, sizeof(unsigned long long) * 3 * nblocks));
    }
    // row capacity from the longest row of the whole box (a later rebuild can bring any row into this rank's slice)
    cl->cap = 0;
    {
        const int cb = cl->c_begin, ce = cl->c_end;
        cl->c_begin = 0;
        cl->c_end = nc;
        const int rc_count = cluster_chain(ctx, L, cl, d_pos, 1, true, nullptr);
        cl->c_begin = cb;
        cl->c_end = ce;
        if (rc_count) return 1;
    }
    AMM_HIP(hipMemcpyAsync(flags, cl->d_flags, sizeof(flags), hipMemcpyDeviceToHost, ctx->stream));
    AMM_HIP(hipStreamSynchronize(ctx->stream));
    if (flags[7]) {
        amm_set_error("molecule-row list: cell table overflow at the first build");
        return 1;
    }
    cl->cap = ((int)(flags[2] * 1.5) + 32 + 15) / 16 * 16;
    if (cl->cap > 65535 || nc >= (1 << 29)) {
        amm_set_error("molecule-row list: rows longer than 65535 entries or more than 2^29 molecules are not supported");
        return 1;
    }
    AMM_HIP(hipMalloc(&cl->d_nl, sizeof(int) * ns * cl->cap));
    if (cluster_chain(ctx, L, cl, d_pos, 1, false, nullptr)) return 1;
    cl->built = true;
    return 0;
}

int amm_cluster_eval_impl(amm_ctx *ctx, PairForce *pf, const double *d_pos, double *d_force, int accumulate, PairForce *guest,
                          double *g_force, int g_accumulate, int exchange) {
    hipStream_t st = ctx->stream;
    const int n = pf->n;
    if (!g_erfcx_uploaded_c[ctx->device & 63]) {
        AMM_HIP(hipMemcpyToSymbol(HIP_SYMBOL(amm_erfcx_table_dev_c), amm_erfcx_table_host, sizeof(amm_erfcx_table_host)));
        g_erfcx_uploaded_c[ctx->device & 63] = true;
    }
    PairForce *L = pf->host ? pf->host : pf;
    bool gathered = false;
    // (a hybrid list's request to clear the rows of the atoms outside the molecules: served by the first sort / gather launch below)
    const CZeroRows Z = ctx->zero_rows;
    ctx->zero_rows = CZeroRows{0, nullptr, nullptr, nullptr};
    if (!L->cl || !L->cl->built) {
        if (L->cl) {
            amm_set_error("molecule-row list: an earlier first build failed");
            return 1;
        }
        if (cluster_first_build(ctx, L, d_pos)) return 1;
    }
    ClusterList *cl = L->cl;
    // the plan of an epilogue (cepi_rows) that amm_run_ops attached to this evaluation; consumed here whatever happens to it
    const EpiPlan *plan = ctx->epi_request;
    ctx->epi_request = nullptr;
    ctx->epi_done = false;
    // the launch that moved the atoms to these positions wrote this force's sorted copies already (an earlier epilogue)?
    const int copies_current = (cl->sorted_for == pf && cl->sorted_epoch == ctx->pos_epoch && cl->sorted_pos == d_pos && Z.n == 0) ? 1 : 0;
    if (copies_current) ctx->n_copies_current++;
    if (cl->checked_epoch == ctx->pos_epoch && cl->checked_pos == d_pos && !L->force_rebuild_c) {
        // positions unchanged since this list was last checked
    } else {
        if (!(cl->pre_epoch == ctx->pos_epoch && cl->pre_pos == d_pos))
            hipLaunchKernelGGL(k_ccheck_displacement, dim3((n + 255) / 256), dim3(256), 0, st, n, d_pos, cl->d_xref, 0.25 * cl->skin * cl->skin,
                               cl->d_flags);
        const int forced = L->force_rebuild_c ? 1 : 0;
        L->force_rebuild_c = false;
        const bool assigned = !forced && cl->assigned_epoch == ctx->pos_epoch && cl->assigned_pos == d_pos;
        if (cluster_chain(ctx, L, cl, d_pos, forced, false, pf, Z, copies_current, assigned)) return 1;
        gathered = true;
    }
    cl->checked_epoch = ctx->pos_epoch;
    cl->checked_pos = d_pos;
    if (!gathered && !copies_current)
        hipLaunchKernelGGL(k_csort_gather, dim3((std::max(cl->nc, Z.n) + 255) / 256), dim3(256), 0, st, cl->grid.ncell, cl->nc, cl->d_cell_start,
                           cl->d_cell_members, cl->capc, cl->d_cperm, cl->d_aperm, d_pos, ctx->box, cl->d_pos4f, cl->d_flags + 8, cl->d_flags, 0,
                           pf->d_q, pf->d_hsig, pf->d_seps2, pf->d_posq_s, pf->d_lj_s, (float)cl->rext, L->d_seps2, cl->d_first, Z, 0, (double *)nullptr, 0, 0, (int *)nullptr);     // flags[8] stays 0: copies only
    // (the copies in place are this force's at these positions from here on, whoever wrote them)
    cl->sorted_for = pf;
    cl->sorted_epoch = ctx->pos_epoch;
    cl->sorted_pos = d_pos;
    const int nrows = cl->c_end - cl->c_begin;
    const int per_c = (cl->nc + ctx->world - 1) / ctx->world, per = 3 * per_c, nf = guest ? 2 : 1;
    double *out = d_force, *gout = g_force;
    // Multi-rank with a plan for the launch's epilogue: STATE exchange -- this rank's launch integrates the molecules whose rows it
    // walks and the ranks all-gather positions and velocities instead of forces (k_state_scatter).  Decided from quantities every
    // rank holds alike (the plan, the families, the sizes): all ranks take the same path.
    auto ss_bytes_of = [](const PairForce *p) { return (p->d_tab_ss && p->pc.tab.ss_first >= 0) ? (p->pc.tab.nint - p->pc.tab.ss_first) * AMM_TAB_STRIDE : 0; };
    auto plan_fits = [&]() {
        if (!(plan && plan->kind == 0 && ctx->opt_fuse_epilogue && !accumulate && !cl->d_first && cl->nrest == 0 && plan->bs && plan->bs->mol3_ok &&
              plan->bs->finalized && plan->bs->ncomp == cl->nc && plan->f0 && plan->npre <= AMM_MAX_PRE && ctx->d_x == d_pos && ctx->d_v &&
              plan->f0 != d_force && plan->f0 != g_force)) return false;
        if (guest && !(!g_accumulate && g_force != d_force && ctx->opt_fuse_rows && pf == L && cl->rnear_build > 0)) return false;
        return true;
    };
    bool state_mode = false;
    if (exchange && ctx->world > 1 && ctx->opt_state_exchange && plan_fits() && cl->nc - per_c * (ctx->world - 1) > 0) {
        // (the kernels that carry an epilogue: site-site tables, and for two forces the fused pass of launch_cdual)
        bool kernel = ctx->opt_site_tab && ss_bytes_of(pf) > 0 && (!guest || ss_bytes_of(guest) > 0);
        const int lds = pf->pc.tab.nint * AMM_TAB_STRIDE + ss_bytes_of(pf) + (guest ? guest->pc.tab.nint * AMM_TAB_STRIDE + ss_bytes_of(guest) : 0);
        kernel = kernel && lds <= AMM_CPAIR_LDS_LIMIT;
        if (guest) kernel = kernel && guest->pc.family == AMM_NEAR_FSWITCH && (pf->pc.family == AMM_DAMPED || pf->pc.family == AMM_NONBONDED);
        state_mode = kernel;
    }
    if (state_mode) {
        if (ctx->pending.active) {
            amm_set_error("exchanged evaluation while the previous one still waits for amm_exchange_finish");
            return 1;
        }
        if (!ctx->d_xchg || ctx->xchg_doubles < (long long)ctx->world * 2 * per * 3) {
            amm_set_error("exchanged evaluation: bind an exchange buffer of world * 2 * per * 3 doubles, per = amm_exchange_per()");
            return 1;
        }
        // (the forces of this rank's rows go straight to the groups' buffers, in atom order: only this rank's epilogue reads them)
    } else if (exchange) {
        if (accumulate || g_accumulate) {
            amm_set_error("exchanged evaluation: forces only, no accumulation");
            return 1;
        }
        if (ctx->pending.active) {
            amm_set_error("exchanged evaluation while the previous one still waits for amm_exchange_finish");
            return 1;
        }
        if (!ctx->d_xchg || ctx->xchg_doubles < (long long)ctx->world * 2 * per * 3) {
            amm_set_error("exchanged evaluation: bind an exchange buffer of world * 2 * per * 3 doubles, per = amm_exchange_per()");
            return 1;
        }
        out = ctx->d_xchg + (size_t)ctx->rank * nf * per * 3;
        gout = out + (size_t)per * 3;
    } else {
        if (!accumulate && ctx->world > 1) AMM_HIP(hipMemsetAsync(d_force, 0, sizeof(double) * 3 * (size_t)n, st));
        if (guest && !g_accumulate && ctx->world > 1 && g_force != d_force) AMM_HIP(hipMemsetAsync(g_force, 0, sizeof(double) * 3 * (size_t)n, st));
    }
    if (nrows > 0) {
        CPairArgs A;
        A.c_begin = cl->c_begin;
        A.nrows = nrows;
        int sh = 0;
        while ((1 << sh) < cl->lpa) ++sh;
        A.lpa_shift = sh;
        A.cap = cl->cap;
        A.aperm = cl->d_aperm;
        A.nl = cl->d_nl;
        A.nnb = cl->d_nnb_near;
        A.nnb_total = (pf == L && cl->rnear_build > 0) ? cl->d_nnb : nullptr;
        if (pf == L && !(cl->rnear_build > 0)) {
            A.nnb = cl->d_nnb;            // no guest radius: the whole row is its "front part"
            A.nnb_total = nullptr;
        }
        A.posq = pf->d_posq_s;
        A.lj = pf->d_lj_s;
        A.force = out;
        A.accumulate = accumulate;
        A.sorted_out = (exchange && !state_mode) ? 1 : 0;
        A.box = ctx->box;
        A.host_tab = pf->d_tab;
        A.host_bytes = pf->pc.tab.nint * AMM_TAB_STRIDE;
        // site-site table right behind the Coulomb table in LDS (fused pass: behind both Coulomb tables, the host's first)
        auto ss_bytes = [](const PairForce *p) { return (p->d_tab_ss && p->pc.tab.ss_first >= 0) ? (p->pc.tab.nint - p->pc.tab.ss_first) * AMM_TAB_STRIDE : 0; };
        A.host_tab_ss = pf->d_tab_ss;
        A.host_ss_bytes = ss_bytes(pf);
        A.host_ss_off = A.host_bytes - pf->pc.tab.ss_first * AMM_TAB_STRIDE;
        A.site_atoms = pf->site_atoms;
        A.guest_tab = A.guest_tab_ss = nullptr;
        A.guest_bytes = A.guest_ss_bytes = A.guest_ss_off = A.g_accumulate = 0;
        A.gforce = nullptr;
        A.gfac = A.gsr = 1.0;
        A.margin = cl->rlist_build + cl->skin + 2.0 * cl->rext + 1e-6;
        const int rpw = 64 >> A.lpa_shift;
        A.ntask = (nrows + rpw - 1) / rpw;
        A.per_pair_image = cl->per_pair_image ? 1 : 0;
        A.hsig_site = pf->site_hsig;
        A.seps2_site = pf->site_seps2;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        const bool timed = ctx->profile && (ctx->profile_only < 0 || ctx->profile_only == pf->id);
        if (timed) {
            if (pf->ev_used + 2 > pf->ev.size())
                for (int k = 0; k < 64; ++k) {
                    hipEvent_t ev;
                    AMM_HIP(hipEventCreate(&ev));
                    pf->ev.push_back(ev);
                }
            e0 = pf->ev[pf->ev_used++];
            e1 = pf->ev[pf->ev_used++];
            AMM_HIP(hipEventRecord(e0, st));
        }
        int rc_ = 0;
        bool fused = false;
        if (guest) guest->last_fused = 0;
        // ---- epilogue: the inner RESPA loop of the rows' molecules on this launch (cepi_rows) ----
        CEpiArgs E;
        const CEpiArgs *Ep = nullptr;
        bool epi_launched = false;
        PairForce *enext = nullptr;
        bool enext_alias = false;
        if (((ctx->world == 1 && !exchange) || state_mode) && plan_fits()) {
            E.niter = plan->niter;
            E.npre = plan->npre;
            E.recompute_f0 = (state_mode && plan->niter > 0) ? 1 : 0;
            E.x = ctx->d_x;
            E.v = ctx->d_v;
            E.f0 = plan->f0;
            E.mass = ctx->d_mass;
            E.c1 = plan->c1;
            E.d = plan->d;
            E.c2 = plan->c2;
            bool ok = true;
            E.xchg_x = state_mode ? ctx->d_xchg + (size_t)ctx->rank * 2 * per * 3 : nullptr;
            E.xchg_v = state_mode ? E.xchg_x + (size_t)per * 3 : nullptr;
            E.c_first = cl->c_begin;
            for (int p = 0; p < AMM_MAX_PRE; ++p) {
                CEpiPre &k = E.pre[p];
                k.a = p < plan->npre ? plan->pre_a[p] : nullptr;
                k.b = p < plan->npre ? plan->pre_b[p] : nullptr;
                k.coef = p < plan->npre ? plan->pre_coef[p] : 0.0;
                k.plus = p < plan->npre ? plan->pre_plus[p] : 0;
                if (p < plan->npre && !k.a) ok = false;
            }
            E.cperm = cl->d_cperm;
            E.term_l = plan->bs->d_term_l;
            E.term_q = plan->bs->d_term_q;
            E.atom_recs = plan->bs->d_atom_recs;
            E.posq_next = nullptr;
            E.lj_next = nullptr;
            E.q_next = E.hsig_next = E.seps2_next = nullptr;
            // the sorted copies the next pair evaluation reads: written through the permutation this launch walks (should the
            // triggers ask for a rebuild, that evaluation's chain writes them again in the new order)
            PairForce *nx = plan->niter > 0 ? plan->next : nullptr;
            if (nx && nx->cluster_ok && (nx == L || nx->host == L) && nx->n == n) {
                enext = nx;
                enext_alias = nx->d_posq_s == pf->d_posq_s;
                if (enext_alias && !nx->d_posq_alt) AMM_HIP(hipMalloc(&nx->d_posq_alt, sizeof(double4) * (size_t)n));
                E.posq_next = enext_alias ? nx->d_posq_alt : nx->d_posq_s;
                E.lj_next = enext_alias ? nullptr : nx->d_lj_s;      // (the same force: its parameter records are in place)
                E.q_next = nx->d_q;
                E.hsig_next = nx->d_hsig;
                E.seps2_next = nx->d_seps2;
            }
            amm_collect_watches(ctx, E.W);
            // the moved molecules' cells, ahead of a possible rebuild (needs this list's own trigger among the watched ones: it is)
            E.spec_count = E.spec_clear = E.spec_members = E.spec_start = E.spec_flags = E.spec_ticket = nullptr;
            E.spec_capc = 0;
            E.grid = cl->grid;
            if (plan->niter > 0 && ctx->opt_spec_assign && ctx->world == 1) {
                if (!cl->d_spec_count[0]) {
                    for (int k = 0; k < 2; ++k) {
                        AMM_HIP(hipMalloc(&cl->d_spec_count[k], sizeof(int) * (cl->grid.ncell + 1)));
                        AMM_HIP(hipMemsetAsync(cl->d_spec_count[k], 0, sizeof(int) * (cl->grid.ncell + 1), st));
                    }
                    AMM_HIP(hipMalloc(&cl->d_spec_ticket, sizeof(int) * AMM_TICKET_INTS));
                    AMM_HIP(hipMemsetAsync(cl->d_spec_ticket, 0, sizeof(int) * AMM_TICKET_INTS, st));
                }
                E.spec_count = cl->d_spec_count[cl->spec_parity];
                E.spec_clear = cl->d_spec_count[cl->spec_parity ^ 1];
                E.spec_members = cl->d_cell_members;
                E.spec_start = cl->d_cell_start;
                E.spec_flags = cl->d_flags;
                E.spec_ticket = cl->d_spec_ticket;
                E.spec_capc = cl->capc;
            }
            if (ok) Ep = &E;
        }
        if (guest && ctx->opt_fuse_rows && A.nnb_total) {
            // host and guest in ONE walk of the rows when a fused kernel exists for the two families
            CPairArgs D = A;
            D.guest_tab = guest->d_tab;
            D.guest_bytes = guest->pc.tab.nint * AMM_TAB_STRIDE;
            D.guest_tab_ss = guest->d_tab_ss;
            D.guest_ss_bytes = ss_bytes(guest);
            D.host_ss_off = D.host_bytes + D.guest_bytes - pf->pc.tab.ss_first * AMM_TAB_STRIDE;
            D.guest_ss_off = D.guest_bytes + D.host_ss_bytes - guest->pc.tab.ss_first * AMM_TAB_STRIDE;
            D.gforce = gout;
            D.g_accumulate = (g_force == d_force && !(exchange && !state_mode)) ? 1 : g_accumulate;
            D.gfac = (guest->pc.Kc * guest->pc.sign) / (pf->pc.Kc * pf->pc.sign);
            D.gsr = guest->pc.sign / pf->pc.sign;
            PairConsts gpc = guest->pc;
            if (guest->desc.flags & AMM_GUARD_RC0) gpc.rc2 = std::min(gpc.rc2, gpc.rc0 * gpc.rc0);      // step(rc0 - r)
            const int r = launch_cdual(ctx, D, pf->pc, gpc, Ep, &epi_launched);
            if (r > 0) return 1;
            fused = r == 0;
            if (fused) guest->last_fused = 1;
            else epi_launched = false;
        }
        if (!fused) rc_ = launch_cpair(ctx, A, pf->pc, guest ? nullptr : Ep, &epi_launched);
        if (timed) AMM_HIP(hipEventRecord(e1, st));          // (the guest's launch below is timed under the guest's own id)
        if (!rc_ && guest && !fused) {
            // the guest force of the shared list (same particles, bitwise equal parameters: the host's sorted copies serve): a
            // second launch over the FRONT parts of the same rows, into its own buffer -- or, for the discount of
            // FarNonbondedForce, added to the host's (sign -1 and the step(rc0 - r) guard travel in its constants)
            CPairArgs G = A;
            G.nnb = cl->d_nnb_near;
            G.nnb_total = nullptr;
            G.force = gout;
            G.accumulate = (g_force == d_force && !exchange) ? 1 : g_accumulate;
            G.host_tab = guest->d_tab;
            G.host_bytes = guest->pc.tab.nint * AMM_TAB_STRIDE;
            G.host_tab_ss = guest->d_tab_ss;
            G.host_ss_bytes = ss_bytes(guest);
            G.host_ss_off = G.host_bytes - guest->pc.tab.ss_first * AMM_TAB_STRIDE;
            G.site_atoms = guest->site_atoms;
            G.hsig_site = guest->site_hsig;
            G.seps2_site = guest->site_seps2;
            PairConsts gpc = guest->pc;
            if (guest->desc.flags & AMM_GUARD_RC0) gpc.rc2 = std::min(gpc.rc2, gpc.rc0 * gpc.rc0);      // step(rc0 - r)
            const bool gtimed = ctx->profile && (ctx->profile_only < 0 || ctx->profile_only == guest->id);
            hipEvent_t g0 = nullptr, g1 = nullptr;
            if (gtimed) {
                if (guest->ev_used + 2 > guest->ev.size())
                    for (int k = 0; k < 64; ++k) {
                        hipEvent_t ev;
                        AMM_HIP(hipEventCreate(&ev));
                        guest->ev.push_back(ev);
                    }
                g0 = guest->ev[guest->ev_used++];
                g1 = guest->ev[guest->ev_used++];
                AMM_HIP(hipEventRecord(g0, st));
            }
            rc_ = launch_cpair(ctx, G, gpc);
            if (gtimed) AMM_HIP(hipEventRecord(g1, st));
        }
        if (rc_) return 1;
        AMM_HIP(hipGetLastError());
        if (guest) guest->n_evals++;
        if (state_mode && !epi_launched) {
            amm_set_error("internal: state exchange planned but the launch carried no epilogue");
            return 1;
        }
        if (epi_launched && state_mode) {
            // this rank's molecules are integrated; the others' state comes with the exchange (k_state_scatter), which also evaluates
            // their displacement triggers and writes their part of the next evaluation's sorted copies
            ctx->epi_done = true;
            ctx->n_epilogues++;
            ctx->n_state_exchanges++;
            if (plan->niter > 0) ctx->pos_epoch++;
            if (enext && enext_alias) std::swap(enext->d_posq_s, enext->d_posq_alt);
            for (const double *b : {(const double *)d_force, (const double *)(guest ? g_force : nullptr), (const double *)plan->f0})
                if (b && std::find(ctx->own_only.begin(), ctx->own_only.end(), b) == ctx->own_only.end()) ctx->own_only.push_back(b);
            PendingExchange &pe = ctx->pending;
            pe.active = true;
            pe.kind = 1;
            pe.per = per;
            pe.nf = 2;
            pe.cl = cl;
            pe.next = plan->niter > 0 ? enext : nullptr;
            pe.perm = cl->d_aperm;
            pe.force = pe.gforce = nullptr;
            if (ctx->comm) {
                if (amm_comm_allgather_impl(ctx, ctx->d_xchg, (size_t)2 * per * 3)) return 1;
                if (amm_exchange_finish_impl(ctx)) return 1;
            }
        } else if (epi_launched) {
            // the launch moved the atoms: new positions epoch, their displacement triggers are evaluated, and (when the next force
            // was known) its sorted copies are those of the new positions
            ctx->epi_done = true;
            ctx->n_epilogues++;
            if (plan->niter > 0) {
                ctx->pos_epoch++;
                amm_watch_moved(ctx);
                if (E.spec_count) {
                    cl->spec_parity ^= 1;
                    cl->assigned_epoch = ctx->pos_epoch;
                    cl->assigned_pos = ctx->d_x;
                }
                if (enext) {
                    if (enext_alias) std::swap(enext->d_posq_s, enext->d_posq_alt);
                    cl->sorted_for = enext;
                    cl->sorted_epoch = ctx->pos_epoch;
                    cl->sorted_pos = ctx->d_x;
                }
            }
        }
    }
    pf->n_evals++;
    if (exchange && !state_mode) {
        PendingExchange &pe = ctx->pending;
        pe.active = true;
        pe.kind = 0;
        pe.per = per;
        pe.nf = nf;
        pe.perm = cl->d_aperm;
        pe.force = d_force;
        pe.gforce = g_force;
        if (ctx->comm) {
            if (amm_comm_allgather_impl(ctx, ctx->d_xchg, (size_t)nf * per * 3)) return 1;
            if (amm_exchange_finish_impl(ctx)) return 1;
        }
    }
    return 0;
}

// measurement helper: directed ATOM-pair entries of the molecule rows within r_within of the current sorted positions
__global__ void __launch_bounds__(256) k_ccount_within(CPairArgs A, double r2w, unsigned long long *out) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int a = tid >> 3, sub = tid & 7;
    unsigned long long cnt = 0;
    if (a < A.nrows) {
        const int cs = A.c_begin + a;
        const int nfront = A.nnb[a], nn = A.nnb_total ? A.nnb_total[a] : nfront;
        const int *row = A.nl + (size_t)a * A.cap;
        const int back = A.cap - 1 + nfront;
        for (int k = sub; k < nn; k += 8) {
            const unsigned j = (unsigned)row[k < nfront ? k : back - k] & 0x1fffffffu;
            for (int t = 0; t < 3; ++t)
                for (int b = 0; b < 3; ++b) {
                    const double4 pi = A.posq[3 * cs + t], pj = A.posq[3 * j + b];
                    const double dx = amm_min_image(pi.x - pj.x, A.box.L[0], A.box.invL[0]);
                    const double dy = amm_min_image(pi.y - pj.y, A.box.L[1], A.box.invL[1]);
                    const double dz = amm_min_image(pi.z - pj.z, A.box.L[2], A.box.invL[2]);
                    cnt += (dx * dx + dy * dy + dz * dz < r2w) ? 1ull : 0ull;
                }
        }
    }
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(out, cnt);
}

int amm_cluster_count_within_impl(amm_ctx *ctx, PairForce *pf, const double *d_pos, double r_within, long long *count) {
    PairForce *L = pf->host ? pf->host : pf;
    ClusterList *cl = L->cl;
    hipStream_t st = ctx->stream;
    hipLaunchKernelGGL(k_csort_gather, dim3((cl->nc + 255) / 256), dim3(256), 0, st, cl->grid.ncell, cl->nc, cl->d_cell_start, cl->d_cell_members,
                       cl->capc, cl->d_cperm, cl->d_aperm, d_pos, ctx->box, cl->d_pos4f, cl->d_flags + 8, cl->d_flags, 0, pf->d_q, pf->d_hsig,
                       pf->d_seps2, pf->d_posq_s, pf->d_lj_s, (float)cl->rext, L->d_seps2, cl->d_first, CZeroRows{0, nullptr, nullptr, nullptr}, 0, (double *)nullptr, 0, 0, (int *)nullptr);
    if (cl->sorted_for == pf) {        // (this force's copies in place are those of d_pos now)
        cl->sorted_epoch = ctx->pos_epoch;
        cl->sorted_pos = d_pos;
    }
    CPairArgs A;
    std::memset(&A, 0, sizeof(A));
    A.c_begin = cl->c_begin;
    A.nrows = cl->c_end - cl->c_begin;
    A.cap = cl->cap;
    A.nl = cl->d_nl;
    A.nnb = (pf == L) ? cl->d_nnb : cl->d_nnb_near;
    A.nnb_total = nullptr;
    if (pf == L && cl->rnear_build > 0) {
        A.nnb = cl->d_nnb_near;
        A.nnb_total = cl->d_nnb;
    }
    A.posq = pf->d_posq_s;
    A.box = ctx->box;
    unsigned long long *d_cnt = cl->d_counters + 7;
    AMM_HIP(hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long), st));
    if (A.nrows > 0)
        hipLaunchKernelGGL(k_ccount_within, dim3((unsigned)(((long)A.nrows * 8 + 255) / 256)), dim3(256), 0, st, A, r_within * r_within, d_cnt);
    unsigned long long h = 0;
    AMM_HIP(hipMemcpyAsync(&h, d_cnt, sizeof(h), hipMemcpyDeviceToHost, st));
    AMM_HIP(hipStreamSynchronize(st));
    *count = (long long)h;
    return 0;
}

// measurement helper: how much of the traversal's lane-trips holds an entry.  A wavefront walks 64 / lpa rows at once, lpa lanes per
// row, until the LONGEST of them is done: out[0] = lane-trips executed (64 x trips, summed over the wavefront tasks), out[1] = entries.
__global__ void __launch_bounds__(256) k_crow_padding(int nrows, int lpa_shift, const int *__restrict__ nnb, unsigned long long *out) {
    const int task = blockIdx.x * blockDim.x + threadIdx.x;
    const int rpw = 64 >> lpa_shift, lpa = 1 << lpa_shift;
    unsigned long long slots = 0, entries = 0;
    if (task * rpw < nrows) {
        int longest = 0;
        for (int r = task * rpw; r < min(nrows, (task + 1) * rpw); ++r) {
            const int nn = nnb[r];
            entries += (unsigned long long)nn;
            longest = max(longest, (nn + lpa - 1) >> lpa_shift);
        }
        slots = 64ull * (unsigned long long)longest;
    }
    for (int off = 32; off > 0; off >>= 1) {
        slots += __shfl_xor(slots, off);
        entries += __shfl_xor(entries, off);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out[0], slots);
        atomicAdd(&out[1], entries);
    }
}

int amm_cluster_row_padding_impl(amm_ctx *ctx, PairForce *pf, long long out[2]) {
    PairForce *L = pf->host ? pf->host : pf;
    ClusterList *cl = L->cl;
    hipStream_t st = ctx->stream;
    const int nrows = cl->c_end - cl->c_begin;
    int sh = 0;
    while ((1 << sh) < cl->lpa) ++sh;
    unsigned long long *d_out = cl->d_counters + 5;
    AMM_HIP(hipMemsetAsync(d_out, 0, 2 * sizeof(unsigned long long), st));
    const int ntask = (nrows + (64 >> sh) - 1) / (64 >> sh);
    if (ntask > 0)
        hipLaunchKernelGGL(k_crow_padding, dim3((unsigned)((ntask + 255) / 256)), dim3(256), 0, st, nrows, sh, (pf == L) ? cl->d_nnb : cl->d_nnb_near, d_out);
    unsigned long long h[2] = {0, 0};
    AMM_HIP(hipMemcpyAsync(h, d_out, sizeof(h), hipMemcpyDeviceToHost, st));
    AMM_HIP(hipStreamSynchronize(st));
    out[0] = (long long)h[0];
    out[1] = (long long)h[1];
    return 0;
}

int amm_cluster_free(ClusterList *cl) {
#ifdef AMM_CBS_TIMING
    {
        unsigned long long t[8];
        // (measurement builds: the clocks of the slice's list build -- the two kernels use the slots differently)
        const bool got = hipMemcpyFromSymbol(t, HIP_SYMBOL(g_cbs_t), sizeof(t)) == hipSuccess && t[7];
        if (got && cl->split_parts == 0)
            fprintf(stderr, "k_cbuild<SLICE>: %llu busy wavefronts, mean: prologue %.2f us, whole %.2f us, in drains %.2f us (%.2f drains)\n", t[7], t[0] / 100.0 / t[7],
                    t[1] / 100.0 / t[7], t[2] / 100.0 / t[7], (double)t[3] / t[7]);
        if (got && cl->split_parts > 0)
            fprintf(stderr, "k_cbuild_split, first wavefront of %llu busy blocks, mean clocks (100 MHz): prologue %.2f | pass 1 %.2f wait %.2f | pass 2 %.2f wait %.2f | copy %.2f wait %.2f us\n",
                    t[7], t[0] / 100.0 / t[7], t[1] / 100.0 / t[7], t[2] / 100.0 / t[7], t[3] / 100.0 / t[7], t[4] / 100.0 / t[7], t[5] / 100.0 / t[7], t[6] / 100.0 / t[7]);
    }
#endif
    for (void *q : {(void *)cl->d_spec_count[0], (void *)cl->d_spec_count[1], (void *)cl->d_spec_ticket})
        if (q) (void)hipFree(q);
    if (cl->d_slice_cells) (void)hipFree(cl->d_slice_cells);
    void *ptrs[] = {cl->d_cell_count, cl->d_cell_start, cl->d_cell_members, cl->d_cperm, cl->d_aperm, cl->d_pos4f, cl->d_xref, cl->d_nl,
                    cl->d_nnb, cl->d_nnb_near, cl->d_flags, cl->d_counters, cl->d_blockstats, cl->d_ticket};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    delete cl;
    return 0;
}
