// atomsmm_amd/csrc/bonded.hip -- bond-list forces of one force group, owner-computes (gfx950, fp64).
//
// Takes over OpenMM's CustomBondForce / HarmonicBondForce / HarmonicAngleForce / PeriodicTorsionForce
// evaluation for the objects the reference builds or keeps in group 0 (forces.py:326-407, 673-680;
// systems.py:113-119) and the exclusion term of the group-2 NonbondedForce (forces.py:181-183).
// One thread per atom walks a CSR of (kind, term, role) references and recomputes each term's force
// on ITS atom: no atomics, so the summation order -- and therefore every bit of the result -- is the
// same on every launch and on every rank (ranks integrate all atoms redundantly and must stay in
// lock-step).  In RESPA this kernel runs n0*n1*n2 times per outer step: it is latency-, not flop-bound.
#include <algorithm>
#include <cmath>
#include <type_traits>
#include <cstring>

#include "amm_ctx.h"
#include "cluster.h"
#include "expr_vm.h"
#include "pair_math.h"

#include "bonded_terms.h"

template <class P, bool LOCAL = false>
__device__ __forceinline__ void bonded_atom(const BondedArgs &A, const P &pos, int iglobal, int i, double *f, double &esum) {
    const int rb = A.ref_ptr[iglobal], re = A.ref_ptr[iglobal + 1];
    for (int r = rb; r < re; ++r) {
        const int4 at = LOCAL ? A.rec_l[r] : A.rec_a[r];
        const double4 q = A.rec_q[r];
        const long long code = __double_as_longlong(q.w);
        const int kind = (int)(code & 7), role = (int)((code >> 3) & 3), periodic = (int)((code >> 5) & 1);
        const int ix[4] = {at.x, at.y, at.z, at.w};
        const double p[3] = {q.x, q.y, q.z};
        double fo[4][3], e;
        bonded_term_forces(A, pos, ix, p, kind, periodic, fo, e);
#pragma unroll
        for (int x = 0; x < 3; ++x) f[x] += role == 0 ? fo[0][x] : (role == 1 ? fo[1][x] : (role == 2 ? fo[2][x] : fo[3][x]));
        if (role == 0) esum += e;
    }
}

__global__ void __launch_bounds__(256) k_bonded(BondedArgs A) {
    const int i = A.row_begin + blockIdx.x * blockDim.x + threadIdx.x;
    double f[3] = {0.0, 0.0, 0.0};
    double esum = 0.0;
    if (i < A.row_end) {
        PosPlain pos{A.pos};
        bonded_atom(A, pos, i, i, f, esum);
        if (A.accumulate) {
            A.force[3 * i] += f[0]; A.force[3 * i + 1] += f[1]; A.force[3 * i + 2] += f[2];
        } else {
            A.force[3 * i] = f[0]; A.force[3 * i + 1] = f[1]; A.force[3 * i + 2] = f[2];
        }
    }
    if (A.want_energy) {
        __shared__ double red[4];
        for (int off = 32; off > 0; off >>= 1) esum += __shfl_xor(esum, off);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = esum;
        __syncthreads();
        if (threadIdx.x == 0) A.epart[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
    }
}

// Term-parallel evaluation (force only) for sets with many terms -- a solvated chain, a protein: one thread per TERM runs
// bonded_term_forces once (k_bonded runs it once per atom of the term, and its wavefronts mix all the kinds; terms are stored
// kind by kind, so these wavefronts are uniform) and parks the forces of the term's roles; one thread per atom then adds up
// its records from there in record order: the same numbers in the same order as k_bonded, bit for bit.
__global__ void __launch_bounds__(256) k_terms_eval(BondedArgs A, int nterms, const int4 *__restrict__ gt_a,
                                                    const double4 *__restrict__ gt_q, double *__restrict__ tf,
                                                    const int *__restrict__ list = nullptr) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nterms) return;
    const int t = list ? list[k] : k;          // (mixed sets: only the terms of the big components)
    const int4 at = gt_a[t];
    const double4 q = gt_q[t];
    const long long code = __double_as_longlong(q.w);
    const int kind = (int)(code & 7), periodic = (int)((code >> 5) & 1);
    const int ix[4] = {at.x, at.y, at.z, at.w};
    const double p[3] = {q.x, q.y, q.z};
    double fo[4][3], e;
    PosPlain pos{A.pos};
    bonded_term_forces(A, pos, ix, p, kind, periodic, fo, e);
    double *out = tf + (size_t)t * 12;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int x = 0; x < 3; ++x) out[3 * r + x] = fo[r][x];
}

__global__ void __launch_bounds__(256) k_terms_gather(BondedArgs A, const int *__restrict__ rec_src, const double *__restrict__ tf) {
    const int i = A.row_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.row_end) return;
    double f[3] = {0.0, 0.0, 0.0};
    const int rb = A.ref_ptr[i], re = A.ref_ptr[i + 1];
    for (int r0 = rb; r0 < re; r0 += 4) {          // (four records in flight, added in record order: see k_terms_gather_kicks)
        int src[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) src[u] = r0 + u < re ? rec_src[r0 + u] : -1;
        double g[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int x = 0; x < 3; ++x) g[u][x] = src[u] >= 0 ? tf[(size_t)src[u] * 3 + x] : 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (src[u] >= 0) {
#pragma unroll
                for (int x = 0; x < 3; ++x) f[x] += g[u][x];
            }
    }
    if (A.accumulate) {
        A.force[3 * i] += f[0]; A.force[3 * i + 1] += f[1]; A.force[3 * i + 2] += f[2];
    } else {
        A.force[3 * i] = f[0]; A.force[3 * i + 1] = f[1]; A.force[3 * i + 2] = f[2];
    }
}

// k_terms_gather + the kicks (and the move) that follow the EVAL in the step program (propagators.py:933-973: `v <- v +
// (c2) f0 / m` closes an inner iteration, `v <- v + (c1) f0 / m ; x <- x + (d) v` opens the next): per degree of freedom the
// operations of k_kicks_move in the same order (mul, div, add rounded separately), on the force row this thread has just
// written -- a kick that reads this group's buffer takes the value from the register, the same number.  Positions are not
// read here (the terms' forces are parked in tf), so moving them is safe.
__global__ void __launch_bounds__(256) k_terms_gather_kicks(BondedArgs A, const int *__restrict__ rec_src, const double *__restrict__ tf,
                                                            KickList K, double *__restrict__ x, double *__restrict__ v,
                                                            const double *__restrict__ mass, int with_move, double dcoef, WatchArgs W) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.n) return;
    double f[3] = {0.0, 0.0, 0.0};
    const int rb = A.ref_ptr[i], re = A.ref_ptr[i + 1];
    // four records at a time: their indices, then their parked forces, are fetched together (one record after the other the
    // kernel was a chain of 2 x records dependent loads per atom at 4 wavefronts per SIMD); added in record order as before
    for (int r0 = rb; r0 < re; r0 += 4) {
        int src[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) src[u] = r0 + u < re ? rec_src[r0 + u] : -1;
        double g[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < 3; ++c) g[u][c] = src[u] >= 0 ? tf[(size_t)src[u] * 3 + c] : 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (src[u] >= 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c) f[c] += g[u][c];
            }
    }
    if (A.accumulate) {
#pragma unroll
        for (int c = 0; c < 3; ++c) f[c] = A.force[3 * i + c] + f[c];
    }
    const double m = mass[i];
    double xn[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int t = 3 * i + c;
        A.force[t] = f[c];
        double vt = v[t];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < K.n) {
                double ff = K.f[k] == A.force ? f[c] : K.f[k][t];
                if (K.f2[k]) {
                    const double g2 = K.f2[k] == A.force ? f[c] : K.f2[k][t];
                    ff = K.plus[k] ? ff + g2 : ff - g2;
                }
                const double num = K.coef[k] * ff;
                const double dv = num / m;
                vt = vt + dv;
            }
        }
        v[t] = vt;
        if (with_move) {
            const double dx = dcoef * vt;
            xn[c] = x[t] + dx;
            x[t] = xn[c];
        }
    }
    if (with_move) amm_watch_atom(W, i, xn);
}

// EVAL + kicks + move of a MIXED bond-list set (BondedSet::mixed_ok: waters next to a chain).  Blocks [0, sc_blocks): four lanes per
// small component -- lane l holds atom l, the positions are exchanged through a wavefront-private LDS strip, lane l evaluates the
// component's term l once (all roles) and parks its forces in LDS, every atom adds up its records from there in record order
// (k_inner_lanes' TERMS scheme: the numbers and the order of the term-parallel gather, bit for bit) -- no parked forces in HBM, no
// three-level gather for 98 % of the atoms.  Blocks beyond: one thread per atom of the big components, from the parked forces of
// k_terms_eval (k_terms_gather_kicks' loop).  Then, for both: add the pair force already in the row, store the row, kicks, move.
struct MixedArgs {
    int n_sc, sc_blocks, n_big;
    const int4 *sc_atoms, *sc_term_l;
    const double4 *sc_term_q;
    const unsigned long long *sc_recs;
    const int *big_atoms;
};

__global__ void __launch_bounds__(256) k_mixed_eval_kicks(BondedArgs A, MixedArgs M, const int *__restrict__ rec_src,
                                                          const double *__restrict__ tf, KickList K, double *__restrict__ x,
                                                          double *__restrict__ v, const double *__restrict__ mass, int with_move, double dcoef,
                                                          WatchArgs W, const double *__restrict__ pair_rows) {
#pragma clang fp contract(off)
    __shared__ double s_x[3][256];
    __shared__ double s_out[12][256];
    int a = -1;
    double f[3] = {0.0, 0.0, 0.0};
    const bool small_part = (int)blockIdx.x < M.sc_blocks;
    if (small_part) {
        const int tid = blockIdx.x * 256 + threadIdx.x;
        if ((tid >> 2) < M.n_sc) a = (&M.sc_atoms[tid >> 2].x)[tid & 3];
    } else {
        const int k = ((int)blockIdx.x - M.sc_blocks) * 256 + threadIdx.x;
        if (k < M.n_big) a = M.big_atoms[k];
    }
    // everything the tail needs is fetched NOW, behind the term arithmetic: the row the pair force left, mass, velocity, the other
    // buffers the kicks read (left where they were used, the loads sat behind the LDS exchanges: a fourth dependent round trip)
    const int ah = a >= 0 ? a : 0;
    double f_row[3] = {0.0, 0.0, 0.0}, v_in[3], kf[4][3], kf2[4][3];
    const double m = mass[ah];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (A.accumulate) f_row[c] = pair_rows ? pair_rows[3 * ah + c] : A.force[3 * ah + c];
        v_in[c] = v[3 * ah + c];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            kf[k][c] = (k < K.n && K.f[k] != A.force) ? K.f[k][3 * ah + c] : 0.0;
            kf2[k][c] = (k < K.n && K.f2[k] && K.f2[k] != A.force) ? K.f2[k][3 * ah + c] : 0.0;
        }
    }
    if (small_part) {
        const int tid = blockIdx.x * 256 + threadIdx.x;
        const int c = tid >> 2, l = tid & 3;
        const bool cvalid = c < M.n_sc;
        const int al = ah;
        s_x[0][threadIdx.x] = A.pos[3 * al];
        s_x[1][threadIdx.x] = A.pos[3 * al + 1];
        s_x[2][threadIdx.x] = A.pos[3 * al + 2];
        int4 my_tl = make_int4(-1, -1, -1, -1);
        double4 my_tq = make_double4(0.0, 0.0, 0.0, 0.0);
        unsigned long long my_recs = 0ull;
        if (cvalid) {
            my_tl = M.sc_term_l[tid];
            my_tq = M.sc_term_q[tid];
            my_recs = M.sc_recs[tid];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int gbase = (int)threadIdx.x - l;
        PosLds pos{&s_x[0][gbase], &s_x[1][gbase], &s_x[2][gbase]};
        double fo[4][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
        if (my_tl.x >= 0) {
            const long long code = __double_as_longlong(my_tq.w);
            const int ix[4] = {my_tl.x, my_tl.y, my_tl.z, my_tl.w};
            const double p[3] = {my_tq.x, my_tq.y, my_tq.z};
            double e;
            bonded_term_forces(A, pos, ix, p, (int)(code & 7), (int)((code >> 5) & 1), fo, e);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int xx = 0; xx < 3; ++xx) s_out[r * 3 + xx][threadIdx.x] = fo[r][xx];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int rec_n = (int)(my_recs >> 60);
        unsigned long long rr = my_recs;
        for (int t = 0; t < rec_n; ++t) {
            const int rcode = (int)(rr & 31ull);
            rr >>= 5;
            const int src = gbase + (rcode & 7), row = 3 * (rcode >> 3);
            f[0] += s_out[row][src];
            f[1] += s_out[row + 1][src];
            f[2] += s_out[row + 2][src];
        }
    } else {
        if (a >= 0) {
            const int rb = A.ref_ptr[a], re = A.ref_ptr[a + 1];
            for (int r0 = rb; r0 < re; r0 += 4) {
                int src[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) src[u] = r0 + u < re ? rec_src[r0 + u] : -1;
                double g[4][3];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int c = 0; c < 3; ++c) g[u][c] = src[u] >= 0 ? tf[(size_t)src[u] * 3 + c] : 0.0;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (src[u] >= 0) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) f[c] += g[u][c];
                    }
            }
        }
    }
    if (a < 0) return;
    if (A.accumulate) {
#pragma unroll
        for (int c = 0; c < 3; ++c) f[c] = f_row[c] + f[c];
    }
    double xn[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int t = 3 * a + c;
        A.force[t] = f[c];
        double vt = v_in[c];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < K.n) {
                double ff = K.f[k] == A.force ? f[c] : kf[k][c];
                if (K.f2[k]) {
                    const double g2 = K.f2[k] == A.force ? f[c] : kf2[k][c];
                    ff = K.plus[k] ? ff + g2 : ff - g2;
                }
                const double num = K.coef[k] * ff;
                const double dv = num / m;
                vt = vt + dv;
            }
        }
        v[t] = vt;
        if (with_move) {
            const double dx = dcoef * vt;
            xn[c] = x[t] + dx;
            x[t] = xn[c];
        }
    }
    if (with_move) amm_watch_atom(W, a, xn);
}

// Fused inner RESPA iteration (propagators.py:940-973, innermost level):
//     v <- v + c1*f0/m ;  x <- x + d*v ;  f0 <- bonded(x) ;  v <- v + c2*f0/m
// in ONE launch.  Each thread advances its own atom and, redundantly, the few atoms it shares bond-list terms
// with (same arithmetic, same rounding, so every thread sees identical new positions); inputs and outputs are
// separate buffers (ping-pong), so no grid-wide synchronisation is needed between the move and the force.
struct FusedArgs {
    const double *x_in, *v_in, *f_in, *mass;
    double *x_out, *v_out, *f_out;
    double c1, d, c2;
};

__global__ void __launch_bounds__(256) k_fused_inner(BondedArgs A, FusedArgs F) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.n) return;
    PosAdvanced pos{F.x_in, F.v_in, F.f_in, F.mass, F.c1, F.d};
    double f[3] = {0.0, 0.0, 0.0};
    double esum = 0.0;
    bonded_atom(A, pos, i, i, f, esum);
    {
#pragma clang fp contract(off)
        const double mi = F.mass[i];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double v1 = pos.vel(i, k);
            F.x_out[3 * i + k] = pos.get(i, k);
            const double num = F.c2 * f[k];
            const double dv = num / mi;
            F.v_out[3 * i + k] = v1 + dv;
            F.f_out[3 * i + k] = f[k];
        }
    }
}

// Component-parallel inner loop.  The bond-list terms of group 0 only couple atoms of the same connected component
// (a water molecule, a small solute), so a component can be integrated through ALL n0 inner RESPA iterations
//     [pre-kicks]  n0 x { v += c1 f0/m ; x += d v ; f0 = bonded(x) ; v += c2 f0/m }
// without looking at any other: one launch instead of 4*n0, same arithmetic and rounding as the separate kernels
// (bit-identical).  Used when every component has at most 8 atoms.
struct PreKick {
    const double *a, *b;   // v += coef*(a -/+ b)/m ; b may be null
    double coef;
    int plus;
};

struct CompArgs {
    const int *comp_ptr, *comp_atoms;
    const int4 *term_l;             // term-parallel variant (TERMS): one term per lane, see BondedSet
    const double4 *term_q;
    const unsigned long long *atom_recs;
    int ncomp, niter, npre;
    double *x, *v, *f0;
    const double *mass;
    double c1, d, c2;
    // BATH: kick ; move(d) ; Ornstein-Uhlenbeck step ; move(d2) ; forces ; kick  (Langevin_R 'middle' scheme)
    double d2, bath_z, bath_kT;
    int iso;                        // isokinetic mode (SIN(R)): kicks are amm_iso_kick on (v, v1), v1 in iso_v1
    double iso_LkT, iso_Q1;
    double *iso_v1;
    int bath_kind;                  // 0 Ornstein-Uhlenbeck, 1 Nose-Hoover-Langevin, 2 stochastic-isokinetic (thermostat velocities in bath_w)
    double bath_h, bath_Q, bath_friction;
    double *bath_w;
    unsigned long long seed, counter0;
    PreKick pre[AMM_MAX_PRE];
    // displacement watchers: neighbour lists whose rebuild trigger this kernel evaluates for the positions it
    // writes (saves the separate k_check_displacement launch before the next pair-force evaluation)
    int nwatch;
    const double *wref[AMM_MAX_WATCH];
    double wthr2[AMM_MAX_WATCH];
    int *wflags[AMM_MAX_WATCH];
};

// G lanes (4 or 8, a power of two dividing the wavefront) share one component, lane l owns the component's atom l.  Each lane kicks/moves its own atom, publishes the new position in
// an LDS strip of its group (same wavefront: no block barrier), and walks ITS OWN (atom, term) records exactly like
// k_bonded does (owner-computes, same order, same bonded_term_forces) -- hence bit-identical -- reading the partner
// atoms from the strip.  (A first version ran one THREAD per component with everything in registers: 0.5 wavefronts
// per SIMD at C3 and select chains over register slots, 50 us per 4 iterations; this one: 2 wavefronts per SIMD, 40 us.)
// (PosLds: bonded_terms.h)

// TERMS: when no component has more terms than lanes, lane l evaluates the component's term l ONCE per iteration (all
// roles), parks the forces in LDS, and every atom adds up ITS records from there in the same order as before -- the
// same numbers in the same order, so still bit-identical to k_bonded, with each term computed once instead of once per
// atom and a dependency chain of one term instead of the atom's whole record list (a water: 1 bond + 1 angle path per
// iteration instead of 2 + 2).
// HONLY: the set holds harmonic bonds and angles only (a flexible water model): the other kinds' code is compiled out.
// ISO: the isokinetic mode of SIN(R) exists in the kernel (its extra registers and tests are compiled out otherwise).
template <int G, bool BATH, bool TERMS, bool HONLY, bool ISO>
__global__ void __launch_bounds__(256) k_inner_lanes(BondedArgs A, CompArgs C) {
    __shared__ double s_x[3][256];
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = tid / G, l = tid % G;
    const bool cvalid = c < C.ncomp;
    const int cb = cvalid ? C.comp_ptr[c] : 0, n = cvalid ? C.comp_ptr[c + 1] - cb : 0;
    const bool has = l < n;
    const int a = C.comp_atoms[has ? cb + l : 0];      // idle lanes alias a valid atom and never store
    const double m = C.mass[a];
    double x[3], v[3], f[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        x[j] = C.x[3 * a + j];
        v[j] = C.v[3 * a + j];
        f[j] = C.f0[3 * a + j];
    }
    double w[3] = {0.0, 0.0, 0.0};       // thermostat velocities of a Nose-Hoover-Langevin / stochastic-isokinetic bath
    double u1[3] = {0.0, 0.0, 0.0};      // isokinetic mode: the thermostat velocity coupled to each velocity component
    if (BATH) {
        if (C.bath_kind >= 1) {
#pragma unroll
            for (int j = 0; j < 3; ++j) w[j] = C.bath_w[3 * a + j];
        }
    }
    if (ISO && C.iso) {
#pragma unroll
        for (int j = 0; j < 3; ++j) u1[j] = C.iso_v1[3 * a + j];
    }
    const double rm = 1.0 / m;
    const bool rok = (__double_as_longlong(m) & 0xFFFFFFFFFFFFFll) != 0xFFFFFFFFFFFFFll && m > 1e-200 && m < 1e200;
    {
#pragma clang fp contract(off)
        for (int p = 0; p < C.npre; ++p) {
            const PreKick pk = C.pre[p];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                double ff = pk.a[3 * a + j];
                if (pk.b) ff = pk.plus ? ff + pk.b[3 * a + j] : ff - pk.b[3 * a + j];
                if (ISO && C.iso) {
                    amm_iso_kick(v[j], u1[j], ff, m, pk.coef, C.iso_LkT, C.iso_Q1);
                    continue;
                }
                const double num = pk.coef * ff;
                const double dv = amm_div_mass(num, m, rm, rok);
                v[j] = v[j] + dv;
            }
        }
    }
    // this atom's term records stay in LDS across the iterations (first MAXR; the rest is re-read): the record loop
    // below is a real loop with ONE inlined copy of the term code, so the records need a dynamically indexed home
    constexpr int MAXR = TERMS ? 1 : 4;
    __shared__ int4 s_tl[TERMS ? 1 : MAXR][TERMS ? 1 : 256];
    __shared__ double4 s_tq[TERMS ? 1 : MAXR][TERMS ? 1 : 256];
    __shared__ double s_out[TERMS ? 12 : 1][TERMS ? 256 : 1];      // [role*3 + xyz][lane]: forces of the lane's term
    const int rb = has ? A.ref_ptr[a] : 0, nrec = has ? A.ref_ptr[a + 1] - rb : 0;
    int4 my_tl = make_int4(-1, -1, -1, -1);
    double4 my_tq = make_double4(0.0, 0.0, 0.0, 0.0);
    unsigned long long my_recs = 0ull;
    if (TERMS) {
        if (cvalid) {
            my_tl = C.term_l[tid];
            my_tq = C.term_q[tid];
        }
        if (has) my_recs = C.atom_recs[a];
    } else {
#pragma unroll
        for (int t = 0; t < MAXR; ++t) {
            const int r = t < nrec ? rb + t : 0;      // record 0 always exists (the buffers hold at least one)
            s_tl[t][threadIdx.x] = A.rec_l[r];
            s_tq[t][threadIdx.x] = A.rec_q[r];
        }
    }
    const int gbase = (int)threadIdx.x - l;
    PosLds pos{&s_x[0][gbase], &s_x[1][gbase], &s_x[2][gbase]};
    // TERMS: where the atom's first four records sit in s_out (element index of the x component)
    const int rec_n = (int)(my_recs >> 60);
    int rec_at[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int rcode = (int)((my_recs >> (5 * t)) & 31ull);
        rec_at[t] = 3 * (rcode >> 3) * 256 + gbase + (rcode & 7);
    }
    for (int it = 0; it < C.niter; ++it) {
        {
#pragma clang fp contract(off)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (ISO && C.iso) {
                    amm_iso_kick(v[j], u1[j], f[j], m, C.c1, C.iso_LkT, C.iso_Q1);
                } else {
                    const double num = C.c1 * f[j];
                    const double dv = amm_div_mass(num, m, rm, rok);
                    v[j] = v[j] + dv;
                }
                const double dx = C.d * v[j];
                x[j] = x[j] + dx;
            }
        }
        if (BATH) {
            // the same amm_ou_step / random stream as a separate AMM_OP_BATH launch of this iteration would use
            const unsigned long long counter = (1ull << 63) | (C.counter0 + (unsigned long long)it + 1ull);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double g = amm_gaussian(C.seed, counter, (unsigned)(3 * a + j));
                if (ISO && C.bath_kind == 2) amm_sin_bath_step(v[j], u1[j], w[j], m, C.bath_h, C.bath_z, C.bath_kT, C.bath_Q, C.bath_friction, C.iso_Q1, C.iso_LkT, g);
                else if (C.bath_kind == 1) amm_nhl_step(v[j], w[j], m, C.bath_h, C.bath_z, C.bath_kT, C.bath_Q, C.bath_friction, g);
                else v[j] = amm_ou_step(v[j], m, C.bath_z, C.bath_kT, g);
            }
            {
#pragma clang fp contract(off)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const double dx = C.d2 * v[j];
                    x[j] = x[j] + dx;
                }
            }
        }
        // publish the new position to the group (wavefront-synchronous: the group never spans two wavefronts)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // earlier reads of the strip are done
        __builtin_amdgcn_wave_barrier();
        s_x[0][threadIdx.x] = x[0];
        s_x[1][threadIdx.x] = x[1];
        s_x[2][threadIdx.x] = x[2];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        f[0] = f[1] = f[2] = 0.0;
        auto do_rec = [&](const int4 al, const double4 q) {
            const long long code = __double_as_longlong(q.w);
            const int kind = HONLY ? (int)(code & 1) : (int)(code & 7), role = (int)((code >> 3) & 3), periodic = (int)((code >> 5) & 1);
            const int ix[4] = {al.x, al.y, al.z, al.w};
            const double p[3] = {q.x, q.y, q.z};
            double fo[4][3], e;
            bonded_term_forces(A, pos, ix, p, kind, periodic, fo, e);
#pragma unroll
            for (int xx = 0; xx < 3; ++xx)
                f[xx] += role == 0 ? fo[0][xx] : (role == 1 ? fo[1][xx] : (role == 2 ? fo[2][xx] : fo[3][xx]));
        };
        if (TERMS) {
            double fo[4][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
            if (my_tl.x >= 0) {
                const long long code = __double_as_longlong(my_tq.w);
                const int ix[4] = {my_tl.x, my_tl.y, my_tl.z, my_tl.w};
                const double p[3] = {my_tq.x, my_tq.y, my_tq.z};
                double e;
                bonded_term_forces(A, pos, ix, p, HONLY ? (int)(code & 1) : (int)(code & 7), (int)((code >> 5) & 1), fo, e);
            }
#pragma unroll
            for (int r = 0; r < (HONLY ? 3 : 4); ++r)          // bonds and angles have no fourth atom
#pragma unroll
                for (int xx = 0; xx < 3; ++xx) s_out[TERMS ? r * 3 + xx : 0][TERMS ? threadIdx.x : 0] = fo[r][xx];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // the atom's records in order: the first four from the table decoded before the loop, the rest from the word
            const double *so = &s_out[0][0];
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (t < rec_n) {
                    f[0] += so[TERMS ? rec_at[t] : 0];
                    f[1] += so[TERMS ? rec_at[t] + 256 : 0];
                    f[2] += so[TERMS ? rec_at[t] + 512 : 0];
                }
            if (rec_n > 4) {
                unsigned long long rr = my_recs >> 20;
                for (int t = 4; t < rec_n; ++t) {
                    const int rcode = (int)(rr & 31ull);
                    rr >>= 5;
                    const int src = gbase + (rcode & 7), row = 3 * (rcode >> 3);
                    f[0] += s_out[TERMS ? row : 0][TERMS ? src : 0];
                    f[1] += s_out[TERMS ? row + 1 : 0][TERMS ? src : 0];
                    f[2] += s_out[TERMS ? row + 2 : 0][TERMS ? src : 0];
                }
            }
        } else
        // ONE inlined copy of the term code (it covers every bond-list kind: unrolling this loop over a register
        // cache multiplied the kernel to 10 k instructions, beyond the instruction cache)
        for (int t = 0; t < nrec; ++t) {
            const bool cached = t < MAXR;
            const int4 al = cached ? s_tl[cached ? t : 0][threadIdx.x] : A.rec_l[rb + t];
            const double4 q = cached ? s_tq[cached ? t : 0][threadIdx.x] : A.rec_q[rb + t];
            do_rec(al, q);
        }
        {
#pragma clang fp contract(off)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (ISO && C.iso) {
                    amm_iso_kick(v[j], u1[j], f[j], m, C.c2, C.iso_LkT, C.iso_Q1);
                    continue;
                }
                const double num = C.c2 * f[j];
                const double dv = amm_div_mass(num, m, rm, rok);
                v[j] = v[j] + dv;
            }
        }
    }
    if (has) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            C.x[3 * a + j] = x[j];
            C.v[3 * a + j] = v[j];
            C.f0[3 * a + j] = f[j];
        }
        if (BATH) {
            if (C.bath_kind >= 1) {
#pragma unroll
                for (int j = 0; j < 3; ++j) C.bath_w[3 * a + j] = w[j];
            }
        }
        if (ISO && C.iso) {
#pragma unroll
            for (int j = 0; j < 3; ++j) C.iso_v1[3 * a + j] = u1[j];
        }
        for (int q = 0; q < C.nwatch; ++q) {
            const double dx = x[0] - C.wref[q][3 * a], dy = x[1] - C.wref[q][3 * a + 1], dz = x[2] - C.wref[q][3 * a + 2];
            const double d2 = dx * dx + dy * dy + dz * dz;
            if (!(d2 <= C.wthr2[q])) {                                               // benign race (NaN also triggers)
                C.wflags[q][0] = 1;
                if (!(d2 <= 4.0 * C.wthr2[q])) C.wflags[q][AMM_FLAG_FAR] = 1;
            }
        }
    }
}

static const int kArity[8] = {2, 3, 2, 2, 4, 2, 2, 2};
static const int kNpar[8] = {2, 2, 3, 3, 3, 1, 2, 3};

int amm_bonded_finalize_impl(amm_ctx *ctx, BondedSet *bs) {
    const int n = ctx->n;
    std::vector<int> cnt(n + 1, 0);
    for (int kind = 0; kind < 8; ++kind) {
        const int nt = (int)bs->h_idx[kind].size() / kArity[kind];
        bs->n_terms[kind] = nt;
        for (int t = 0; t < nt; ++t)
            for (int r = 0; r < kArity[kind]; ++r) {
                int a = bs->h_idx[kind][t * kArity[kind] + r];
                if (a < 0 || a >= n) {
                    amm_set_error("bonded term references an atom index out of range");
                    return 1;
                }
                cnt[a + 1]++;
            }
        if (nt >= (1 << 26)) {
            amm_set_error("too many bonded terms of one kind");
            return 1;
        }
    }
    for (int i = 0; i < n; ++i) cnt[i + 1] += cnt[i];
    const size_t nref = (size_t)cnt[n];
    std::vector<int4> rec_a(nref);
    std::vector<double4> rec_q(nref);
    std::vector<int> rec_src(nref);
    std::vector<int> fill(cnt.begin(), cnt.end() - 1);
    int gterm = 0;
    for (int kind = 0; kind < 8; ++kind)
        for (int t = 0; t < bs->n_terms[kind]; ++t, ++gterm)
            for (int r = 0; r < kArity[kind]; ++r) {
                const int a = bs->h_idx[kind][t * kArity[kind] + r];
                const size_t slot = (size_t)fill[a]++;
                rec_src[slot] = gterm * 4 + r;
                int at[4] = {-1, -1, -1, -1};
                for (int k = 0; k < kArity[kind]; ++k) at[k] = bs->h_idx[kind][t * kArity[kind] + k];
                rec_a[slot] = make_int4(at[0], at[1], at[2], at[3]);
                double pr[3] = {0.0, 0.0, 0.0};
                for (int k = 0; k < kNpar[kind]; ++k) pr[k] = bs->h_par[kind][(size_t)t * kNpar[kind] + k];
                const long long code = (long long)kind | ((long long)r << 3) | ((long long)(bs->periodic[kind] ? 1 : 0) << 5);
                double w;
                std::memcpy(&w, &code, sizeof(w));
                rec_q[slot] = make_double4(pr[0], pr[1], pr[2], w);
            }
    // connected components of the term graph (union-find); component kernel if all are small
    std::vector<int> parent(n);
    for (int i = 0; i < n; ++i) parent[i] = i;
    auto find = [&](int a) {
        while (parent[a] != a) {
            parent[a] = parent[parent[a]];
            a = parent[a];
        }
        return a;
    };
    for (int kind = 0; kind < 8; ++kind)
        for (int t = 0; t < bs->n_terms[kind]; ++t)
            for (int r = 1; r < kArity[kind]; ++r) {
                int a = find(bs->h_idx[kind][t * kArity[kind]]), b = find(bs->h_idx[kind][t * kArity[kind] + r]);
                if (a != b) parent[std::max(a, b)] = std::min(a, b);
            }
    std::vector<int> comp_of(n), csize;
    std::vector<int> root_to_comp(n, -1);
    for (int i = 0; i < n; ++i) {           // components numbered by their smallest atom -> ascending, deterministic
        int r = find(i);
        if (root_to_comp[r] < 0) {
            root_to_comp[r] = (int)csize.size();
            csize.push_back(0);
        }
        comp_of[i] = root_to_comp[r];
        csize[comp_of[i]]++;
    }
    const int ncomp = (int)csize.size();
    int maxc = 0;
    for (int c2 = 0; c2 < ncomp; ++c2) maxc = std::max(maxc, csize[c2]);
    std::vector<int> comp_ptr(ncomp + 1, 0), comp_atoms(n), local_of(n);
    for (int c2 = 0; c2 < ncomp; ++c2) comp_ptr[c2 + 1] = comp_ptr[c2] + csize[c2];
    {
        std::vector<int> fillc(comp_ptr.begin(), comp_ptr.end() - 1);
        for (int i = 0; i < n; ++i) {
            local_of[i] = fillc[comp_of[i]] - comp_ptr[comp_of[i]];
            comp_atoms[fillc[comp_of[i]]++] = i;
        }
    }
    std::vector<int4> rec_l(nref);
    for (size_t r = 0; r < nref; ++r) {
        const int4 a = rec_a[r];
        rec_l[r] = make_int4(a.x >= 0 ? local_of[a.x] : -1, a.y >= 0 ? local_of[a.y] : -1, a.z >= 0 ? local_of[a.z] : -1,
                             a.w >= 0 ? local_of[a.w] : -1);
    }
    bs->ncomp = ncomp;
    bs->max_comp = maxc;
    // term tables of the term-parallel inner loop: terms numbered within their component in the order the records were
    // laid down above (kind, term), so that an atom's packed (term slot, role) list repeats its record order
    bs->terms_ok = maxc <= 8 && nref > 0;
    if (bs->terms_ok) {
        const int G = maxc <= 4 ? 4 : 8;
        std::vector<int> nterm(ncomp, 0);
        std::vector<int4> term_l((size_t)ncomp * G, make_int4(-1, -1, -1, -1));
        std::vector<double4> term_q((size_t)ncomp * G, make_double4(0.0, 0.0, 0.0, 0.0));
        std::vector<unsigned long long> atom_recs(n, 0ull);
        std::vector<int> nrec_of(n, 0);
        for (int kind = 0; kind < 8 && bs->terms_ok; ++kind)
            for (int t = 0; t < bs->n_terms[kind] && bs->terms_ok; ++t) {
                const int *at = &bs->h_idx[kind][(size_t)t * kArity[kind]];
                const int c2 = comp_of[at[0]];
                const int tl = nterm[c2]++;
                if (tl >= G) {
                    bs->terms_ok = false;
                    break;
                }
                int lo[4] = {-1, -1, -1, -1};
                for (int k = 0; k < kArity[kind]; ++k) lo[k] = local_of[at[k]];
                term_l[(size_t)c2 * G + tl] = make_int4(lo[0], lo[1], lo[2], lo[3]);
                double pr[3] = {0.0, 0.0, 0.0};
                for (int k = 0; k < kNpar[kind]; ++k) pr[k] = bs->h_par[kind][(size_t)t * kNpar[kind] + k];
                const long long code = (long long)kind | ((long long)(bs->periodic[kind] ? 1 : 0) << 5);
                double w;
                std::memcpy(&w, &code, sizeof(w));
                term_q[(size_t)c2 * G + tl] = make_double4(pr[0], pr[1], pr[2], w);
                for (int r = 0; r < kArity[kind]; ++r) {
                    const int a = at[r];
                    if (nrec_of[a] >= 12) {
                        bs->terms_ok = false;
                        break;
                    }
                    atom_recs[a] |= (unsigned long long)(tl | (r << 3)) << (5 * nrec_of[a]);
                    nrec_of[a]++;
                }
            }
        if (bs->terms_ok) {
            // one component per three-site molecule, slots in atom order, harmonic bonds / angles only?  (cluster.hip's epilogue)
            bool mol3 = G == 4 && 3 * (long)ncomp == (long)n;
            for (int c2 = 0; c2 < ncomp && mol3; ++c2)
                mol3 = csize[c2] == 3 && comp_atoms[comp_ptr[c2]] == 3 * c2 && comp_atoms[comp_ptr[c2] + 1] == 3 * c2 + 1 &&
                       comp_atoms[comp_ptr[c2] + 2] == 3 * c2 + 2;
            for (int kd = 2; kd < 8; ++kd) mol3 = mol3 && bs->n_terms[kd] == 0;
            for (int i = 0; i < n && mol3; ++i) mol3 = nrec_of[i] <= 4;
            bs->mol3_ok = mol3;
            for (int i = 0; i < n; ++i) atom_recs[i] |= (unsigned long long)nrec_of[i] << 60;
            AMM_HIP(hipMalloc(&bs->d_term_l, sizeof(int4) * term_l.size()));
            AMM_HIP(hipMalloc(&bs->d_term_q, sizeof(double4) * term_q.size()));
            AMM_HIP(hipMalloc(&bs->d_atom_recs, sizeof(unsigned long long) * n));
            AMM_HIP(hipMemcpy(bs->d_term_l, term_l.data(), sizeof(int4) * term_l.size(), hipMemcpyHostToDevice));
            AMM_HIP(hipMemcpy(bs->d_term_q, term_q.data(), sizeof(double4) * term_q.size(), hipMemcpyHostToDevice));
            AMM_HIP(hipMemcpy(bs->d_atom_recs, atom_recs.data(), sizeof(unsigned long long) * n, hipMemcpyHostToDevice));
        }
    }
    AMM_HIP(hipMalloc(&bs->d_comp_ptr, sizeof(int) * (ncomp + 1)));
    AMM_HIP(hipMemcpy(bs->d_comp_ptr, comp_ptr.data(), sizeof(int) * (ncomp + 1), hipMemcpyHostToDevice));
    AMM_HIP(hipMalloc(&bs->d_comp_atoms, sizeof(int) * n));
    AMM_HIP(hipMemcpy(bs->d_comp_atoms, comp_atoms.data(), sizeof(int) * n, hipMemcpyHostToDevice));
    AMM_HIP(hipMalloc(&bs->d_rec_l, sizeof(int4) * std::max<size_t>(nref, 1)));
    if (nref) AMM_HIP(hipMemcpy(bs->d_rec_l, rec_l.data(), sizeof(int4) * nref, hipMemcpyHostToDevice));
    // term tables of the term-parallel evaluation: worth two launches from a few thousand terms on
    const int terms_from = ctx->opt_terms_from;
    bs->n_gterms = 0;
    if (gterm >= terms_from) {
        std::vector<int4> gt_a((size_t)gterm);
        std::vector<double4> gt_q((size_t)gterm);
        int g = 0;
        for (int kind = 0; kind < 8; ++kind)
            for (int t = 0; t < bs->n_terms[kind]; ++t, ++g) {
                int at[4] = {-1, -1, -1, -1};
                for (int k = 0; k < kArity[kind]; ++k) at[k] = bs->h_idx[kind][(size_t)t * kArity[kind] + k];
                double pr[3] = {0.0, 0.0, 0.0};
                for (int k = 0; k < kNpar[kind]; ++k) pr[k] = bs->h_par[kind][(size_t)t * kNpar[kind] + k];
                const long long code = (long long)kind | ((long long)(bs->periodic[kind] ? 1 : 0) << 5);
                double w;
                std::memcpy(&w, &code, sizeof(w));
                gt_a[g] = make_int4(at[0], at[1], at[2], at[3]);
                gt_q[g] = make_double4(pr[0], pr[1], pr[2], w);
            }
        AMM_HIP(hipMalloc(&bs->d_gt_a, sizeof(int4) * (size_t)gterm));
        AMM_HIP(hipMalloc(&bs->d_gt_q, sizeof(double4) * (size_t)gterm));
        AMM_HIP(hipMalloc(&bs->d_rec_src, sizeof(int) * nref));
        AMM_HIP(hipMalloc(&bs->d_tf, sizeof(double) * 12 * (size_t)gterm));
        AMM_HIP(hipMemcpy(bs->d_gt_a, gt_a.data(), sizeof(int4) * (size_t)gterm, hipMemcpyHostToDevice));
        AMM_HIP(hipMemcpy(bs->d_gt_q, gt_q.data(), sizeof(double4) * (size_t)gterm, hipMemcpyHostToDevice));
        AMM_HIP(hipMemcpy(bs->d_rec_src, rec_src.data(), sizeof(int) * nref, hipMemcpyHostToDevice));
        bs->n_gterms = gterm;
    }
    // mixed sets: small components next to big ones (see BondedSet::mixed_ok)
    bs->mixed_ok = false;
    if (bs->n_gterms > 0 && !bs->terms_ok) {
        std::vector<int> nterm(ncomp, 0), nrec_of(n, 0);
        std::vector<char> small(ncomp, 0);
        for (int c2 = 0; c2 < ncomp; ++c2) small[c2] = csize[c2] <= 4;
        for (int kind = 0; kind < 8; ++kind)
            for (int t = 0; t < bs->n_terms[kind]; ++t) {
                const int *at = &bs->h_idx[kind][(size_t)t * kArity[kind]];
                if (++nterm[comp_of[at[0]]] > 4) small[comp_of[at[0]]] = 0;
                for (int r = 0; r < kArity[kind]; ++r)
                    if (++nrec_of[at[r]] > 12) small[comp_of[at[r]]] = 0;
            }
        std::vector<int> sc_of(ncomp, -1);
        int nsc = 0;
        long small_atoms = 0;
        for (int c2 = 0; c2 < ncomp; ++c2)
            if (small[c2]) {
                sc_of[c2] = nsc++;
                small_atoms += csize[c2];
            }
        if (nsc > 0 && 2 * small_atoms >= n) {           // worth it when most atoms sit in small components
            std::vector<int4> sc_atoms(nsc, make_int4(-1, -1, -1, -1)), term_l((size_t)nsc * 4, make_int4(-1, -1, -1, -1));
            std::vector<double4> term_q((size_t)nsc * 4, make_double4(0.0, 0.0, 0.0, 0.0));
            std::vector<unsigned long long> recs((size_t)nsc * 4, 0ull);
            std::vector<int> big_atoms, big_terms, filled(nsc, 0), nr(n, 0);
            for (int i = 0; i < n; ++i) {
                const int k = sc_of[comp_of[i]];
                if (k < 0) big_atoms.push_back(i);
                else (&sc_atoms[k].x)[local_of[i]] = i;
            }
            int g = 0;
            for (int kind = 0; kind < 8; ++kind)
                for (int t = 0; t < bs->n_terms[kind]; ++t, ++g) {
                    const int *at = &bs->h_idx[kind][(size_t)t * kArity[kind]];
                    const int k = sc_of[comp_of[at[0]]];
                    if (k < 0) {
                        big_terms.push_back(g);
                        continue;
                    }
                    const int tl = filled[k]++;                 // terms numbered within the component in (kind, term) order: the record order
                    int lo[4] = {-1, -1, -1, -1};
                    for (int q = 0; q < kArity[kind]; ++q) lo[q] = local_of[at[q]];
                    term_l[(size_t)k * 4 + tl] = make_int4(lo[0], lo[1], lo[2], lo[3]);
                    double pr[3] = {0.0, 0.0, 0.0};
                    for (int q = 0; q < kNpar[kind]; ++q) pr[q] = bs->h_par[kind][(size_t)t * kNpar[kind] + q];
                    const long long code = (long long)kind | ((long long)(bs->periodic[kind] ? 1 : 0) << 5);
                    double w;
                    std::memcpy(&w, &code, sizeof(w));
                    term_q[(size_t)k * 4 + tl] = make_double4(pr[0], pr[1], pr[2], w);
                    for (int r = 0; r < kArity[kind]; ++r) {
                        const int a = at[r];
                        recs[(size_t)k * 4 + local_of[a]] |= (unsigned long long)(tl | (r << 3)) << (5 * nr[a]);
                        nr[a]++;
                    }
                }
            for (int i = 0; i < n; ++i) {
                const int k = sc_of[comp_of[i]];
                if (k >= 0) recs[(size_t)k * 4 + local_of[i]] |= (unsigned long long)nr[i] << 60;
            }
            auto up = [&](auto **dst, const auto &v) -> int {
                using T = typename std::remove_reference<decltype(v[0])>::type;
                AMM_HIP(hipMalloc(dst, sizeof(T) * std::max<size_t>(v.size(), 1)));
                if (!v.empty()) AMM_HIP(hipMemcpy(*dst, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
                return 0;
            };
            if (up(&bs->d_sc_atoms, sc_atoms) || up(&bs->d_sc_term_l, term_l) || up(&bs->d_sc_term_q, term_q) || up(&bs->d_sc_recs, recs) ||
                up(&bs->d_big_atoms, big_atoms) || up(&bs->d_big_terms, big_terms)) return 1;
            bs->n_sc = nsc;
            bs->n_big = (int)big_atoms.size();
            bs->n_bigterms = (int)big_terms.size();
            bs->mixed_ok = true;
        }
    }
    AMM_HIP(hipMalloc(&bs->d_ref_ptr, sizeof(int) * (n + 1)));
    AMM_HIP(hipMemcpy(bs->d_ref_ptr, cnt.data(), sizeof(int) * (n + 1), hipMemcpyHostToDevice));
    AMM_HIP(hipMalloc(&bs->d_rec_a, sizeof(int4) * std::max<size_t>(nref, 1)));
    AMM_HIP(hipMalloc(&bs->d_rec_q, sizeof(double4) * std::max<size_t>(nref, 1)));
    if (nref) {
        AMM_HIP(hipMemcpy(bs->d_rec_a, rec_a.data(), sizeof(int4) * nref, hipMemcpyHostToDevice));
        AMM_HIP(hipMemcpy(bs->d_rec_q, rec_q.data(), sizeof(double4) * nref, hipMemcpyHostToDevice));
    }
    bs->n_epart = (n + 255) / 256;
    AMM_HIP(hipMalloc(&bs->d_epart, sizeof(double) * bs->n_epart));
    bs->finalized = true;
    return 0;
}

int amm_bonded_eval_impl(amm_ctx *ctx, BondedSet *bs, const double *d_pos, double *d_force, int accumulate,
                         double *d_energy) {
    if (!bs->finalized) {
        amm_set_error("bonded set evaluated before amm_bonded_finalize");
        return 1;
    }
    const int n = ctx->n;
    BondedArgs A;
    A.n = n;
    A.row_begin = 0;
    A.row_end = n;
    if (bs->sliced && ctx->world > 1) {
        const int per = (n + ctx->world - 1) / ctx->world;
        A.row_begin = std::min(n, ctx->rank * per);
        A.row_end = std::min(n, A.row_begin + per);
        if (!accumulate) AMM_HIP(hipMemsetAsync(d_force, 0, sizeof(double) * 3 * (size_t)n, ctx->stream));
    }
    A.ref_ptr = bs->d_ref_ptr;
    A.rec_a = bs->d_rec_a;
    A.rec_q = bs->d_rec_q;
    A.rec_l = bs->d_rec_l;
    A.pos = d_pos;
    A.force = d_force;
    A.epart = bs->d_epart;
    A.accumulate = accumulate;
    A.want_energy = d_energy != nullptr;
    A.box = ctx->box;
    A.near_pc = bs->near_pc;
    A.ewald_alpha = bs->ewald_alpha;
    A.ewald_tasp = bs->ewald_alpha * 1.1283791670955125739;
    A.Kc_ljc = bs->ljc_Kc;
    const int rows = A.row_end - A.row_begin;
    const int nblk = std::max(1, (rows + 255) / 256);
    if (bs->n_gterms > 0 && !d_energy) {
        // every term once (all of them on every rank: they are cheap next to the pair forces), then the atoms of the rank's rows
        hipLaunchKernelGGL(k_terms_eval, dim3((bs->n_gterms + 255) / 256), dim3(256), 0, ctx->stream, A, bs->n_gterms, bs->d_gt_a,
                           bs->d_gt_q, bs->d_tf);
        hipLaunchKernelGGL(k_terms_gather, dim3(nblk), dim3(256), 0, ctx->stream, A, bs->d_rec_src, bs->d_tf);
        AMM_HIP(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(k_bonded, dim3(nblk), dim3(256), 0, ctx->stream, A);
    AMM_HIP(hipGetLastError());
    if (d_energy) return amm_reduce_add(ctx, bs->d_epart, nblk, 1.0, d_energy);
    return 0;
}

static int terms_args(amm_ctx *ctx, BondedSet *bs, const double *d_pos, double *d_force, int accumulate, BondedArgs &A) {
    if (!bs->finalized || bs->n_gterms <= 0 || (bs->sliced && ctx->world > 1)) {
        amm_set_error("term-parallel evaluation: needs a finalized, unsliced bond-list set with parked-force buffers");
        return 1;
    }
    const int n = ctx->n;
    A.n = n;
    A.row_begin = 0;
    A.row_end = n;
    A.ref_ptr = bs->d_ref_ptr;
    A.rec_a = bs->d_rec_a;
    A.rec_q = bs->d_rec_q;
    A.rec_l = bs->d_rec_l;
    A.pos = d_pos;
    A.force = d_force;
    A.epart = bs->d_epart;
    A.accumulate = accumulate;
    A.want_energy = 0;
    A.box = ctx->box;
    A.near_pc = bs->near_pc;
    A.ewald_alpha = bs->ewald_alpha;
    A.ewald_tasp = bs->ewald_alpha * 1.1283791670955125739;
    A.Kc_ljc = bs->ljc_Kc;
    return 0;
}

int amm_bonded_terms_work(amm_ctx *ctx, BondedSet *bs, const double *d_pos, BondedArgs *A, int *nterms, const int4 **gt_a,
                          const double4 **gt_q, double **tf, const int **list) {
    if (terms_args(ctx, bs, d_pos, nullptr, 0, *A)) return 1;
    // (mixed sets: the small components are evaluated by the gather launch itself; only the big components' terms are parked)
    *nterms = (bs->mixed_ok && ctx->opt_mixed_terms) ? bs->n_bigterms : bs->n_gterms;
    *list = (bs->mixed_ok && ctx->opt_mixed_terms) ? bs->d_big_terms : nullptr;
    *gt_a = bs->d_gt_a;
    *gt_q = bs->d_gt_q;
    *tf = bs->d_tf;
    return 0;
}

int amm_bonded_eval_kicks_impl(amm_ctx *ctx, BondedSet *bs, const double *d_pos, double *d_force, int accumulate, const KickList &K,
                               int with_move, double dcoef, int terms_done, const double *pair_rows) {
    BondedArgs A;
    if (terms_args(ctx, bs, d_pos, d_force, accumulate, A)) return 1;
    const int n = ctx->n;
    WatchArgs W;
    W.n = 0;
    if (with_move) amm_collect_watches(ctx, W);       // (the caller bumps pos_epoch and calls amm_watch_moved)
    if (bs->mixed_ok && ctx->opt_mixed_terms) {
        if (!terms_done && bs->n_bigterms > 0)
            hipLaunchKernelGGL(k_terms_eval, dim3((bs->n_bigterms + 255) / 256), dim3(256), 0, ctx->stream, A, bs->n_bigterms, bs->d_gt_a,
                               bs->d_gt_q, bs->d_tf, bs->d_big_terms);
        MixedArgs M;
        M.n_sc = bs->n_sc;
        M.sc_blocks = (4 * bs->n_sc + 255) / 256;
        M.n_big = bs->n_big;
        M.sc_atoms = bs->d_sc_atoms;
        M.sc_term_l = bs->d_sc_term_l;
        M.sc_term_q = bs->d_sc_term_q;
        M.sc_recs = bs->d_sc_recs;
        M.big_atoms = bs->d_big_atoms;
        hipLaunchKernelGGL(k_mixed_eval_kicks, dim3(M.sc_blocks + (bs->n_big + 255) / 256), dim3(256), 0, ctx->stream, A, M, bs->d_rec_src,
                           bs->d_tf, K, ctx->d_x, ctx->d_v, ctx->d_mass, with_move, dcoef, W, pair_rows);
        AMM_HIP(hipGetLastError());
        return 0;
    }
    if (pair_rows) {
        amm_set_error("fused evaluation + kicks: pair rows in a buffer of their own need the mixed kernel");
        return 1;
    }
    if (!terms_done)
        hipLaunchKernelGGL(k_terms_eval, dim3((bs->n_gterms + 255) / 256), dim3(256), 0, ctx->stream, A, bs->n_gterms, bs->d_gt_a, bs->d_gt_q,
                           bs->d_tf);
    hipLaunchKernelGGL(k_terms_gather_kicks, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, A, bs->d_rec_src, bs->d_tf, K, ctx->d_x,
                       ctx->d_v, ctx->d_mass, with_move, dcoef, W);
    AMM_HIP(hipGetLastError());
    return 0;
}

int amm_fused_inner_impl(amm_ctx *ctx, BondedSet *bs, const double *x_in, const double *v_in, const double *f_in,
                         double *x_out, double *v_out, double *f_out, double c1, double d, double c2) {
    const int n = ctx->n;
    BondedArgs A;
    A.n = n;
    A.row_begin = 0;
    A.row_end = n;
    A.ref_ptr = bs->d_ref_ptr;
    A.rec_a = bs->d_rec_a;
    A.rec_q = bs->d_rec_q;
    A.rec_l = bs->d_rec_l;
    A.pos = nullptr;
    A.force = nullptr;
    A.epart = nullptr;
    A.accumulate = 0;
    A.want_energy = 0;
    A.box = ctx->box;
    A.near_pc = bs->near_pc;
    A.ewald_alpha = bs->ewald_alpha;
    A.ewald_tasp = bs->ewald_alpha * 1.1283791670955125739;
    A.Kc_ljc = bs->ljc_Kc;
    FusedArgs F{x_in, v_in, f_in, ctx->d_mass, x_out, v_out, f_out, c1, d, c2};
    hipLaunchKernelGGL(k_fused_inner, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, A, F);
    AMM_HIP(hipGetLastError());
    return 0;
}

int amm_inner_components_impl(amm_ctx *ctx, BondedSet *bs, double *x, double *v, double *f0, int npre, const double *const *pre_a,
                              const double *const *pre_b, const double *pre_coef, const int *pre_plus, double c1, double d,
                              double c2, int niter, const BathDef *bath, double d2) {
    BondedArgs A;
    A.n = ctx->n;
    A.row_begin = 0;
    A.row_end = ctx->n;
    A.ref_ptr = bs->d_ref_ptr;
    A.rec_a = bs->d_rec_a;
    A.rec_q = bs->d_rec_q;
    A.rec_l = bs->d_rec_l;
    A.pos = nullptr;
    A.force = nullptr;
    A.epart = nullptr;
    A.accumulate = 0;
    A.want_energy = 0;
    A.box = ctx->box;
    A.near_pc = bs->near_pc;
    A.ewald_alpha = bs->ewald_alpha;
    A.ewald_tasp = bs->ewald_alpha * 1.1283791670955125739;
    A.Kc_ljc = bs->ljc_Kc;
    CompArgs C;
    C.comp_ptr = bs->d_comp_ptr;
    C.comp_atoms = bs->d_comp_atoms;
    C.ncomp = bs->ncomp;
    C.niter = niter;
    C.npre = npre;
    C.x = x;
    C.v = v;
    C.f0 = f0;
    C.mass = ctx->d_mass;
    {
        WatchArgs W;
        amm_collect_watches(ctx, W);
        C.nwatch = W.n;
        for (int q = 0; q < AMM_MAX_WATCH; ++q) {
            C.wref[q] = W.xref[q];
            C.wthr2[q] = W.thr2[q];
            C.wflags[q] = W.flags[q];
        }
    }
    C.c1 = c1;
    C.d = d;
    C.c2 = c2;
    for (int p = 0; p < AMM_MAX_PRE; ++p) {
        C.pre[p].a = p < npre ? pre_a[p] : nullptr;
        C.pre[p].b = p < npre ? pre_b[p] : nullptr;
        C.pre[p].coef = p < npre ? pre_coef[p] : 0.0;
        C.pre[p].plus = p < npre ? pre_plus[p] : 0;
    }
    dim3 block(256);
    const int G = bs->max_comp <= 4 ? 4 : 8;
    dim3 grid((unsigned)(((long)bs->ncomp * G + 255) / 256));
    C.d2 = d2;
    C.bath_z = bath ? bath->z : 1.0;
    C.bath_kT = bath ? bath->kT : 0.0;
    C.bath_kind = bath ? bath->kind : 0;
    C.bath_h = bath ? bath->h : 0.0;
    C.bath_Q = bath ? bath->Q : 1.0;
    C.bath_friction = bath ? bath->friction : 1.0;
    C.bath_w = (bath && bath->kind >= 1 && bath->slot >= 0 && bath->slot < AMM_MAX_SLOTS) ? ctx->slots[bath->slot] : nullptr;
    if (bath && bath->kind >= 1 && !C.bath_w) {
        amm_set_error("Nose-Hoover-Langevin / stochastic-isokinetic bath: the thermostat-velocity buffer is not bound");
        return 1;
    }
    C.iso = ctx->iso.on ? 1 : 0;
    C.iso_LkT = ctx->iso.LkT;
    C.iso_Q1 = ctx->iso.Q1;
    C.iso_v1 = (ctx->iso.on && ctx->iso.slot >= 0 && ctx->iso.slot < AMM_MAX_SLOTS) ? ctx->slots[ctx->iso.slot] : nullptr;
    if ((ctx->iso.on && !C.iso_v1) || (bath && bath->kind == 2 && !ctx->iso.on)) {
        amm_set_error("isokinetic mode: the thermostat-velocity buffer is not bound (or a stochastic-isokinetic bath without the mode)");
        return 1;
    }
    C.seed = ctx->expr_seed;
    C.counter0 = ctx->expr_counter;
    const bool no_terms = ctx->opt_no_term_lanes != 0;     // tuning option (A/B)
    const bool terms = bs->terms_ok && !no_terms;
    C.term_l = bs->d_term_l;
    C.term_q = bs->d_term_q;
    C.atom_recs = bs->d_atom_recs;
    if (bath) ctx->expr_counter += (unsigned long long)niter;      // one BATH op per iteration, as the unfused sequence counts
    bool honly = true;
    for (int kd = 2; kd < 8; ++kd) honly = honly && bs->n_terms[kd] == 0;
#define AMM_LAUNCH_INNER(GG, BB, TT)                                                                          \
    do {                                                                                                        \
        if (C.iso) {                                                                                            \
            if (honly) hipLaunchKernelGGL((k_inner_lanes<GG, BB, TT, true, true>), grid, block, 0, ctx->stream, A, C);    \
            else hipLaunchKernelGGL((k_inner_lanes<GG, BB, TT, false, true>), grid, block, 0, ctx->stream, A, C);        \
        } else {                                                                                                \
            if (honly) hipLaunchKernelGGL((k_inner_lanes<GG, BB, TT, true, false>), grid, block, 0, ctx->stream, A, C);   \
            else hipLaunchKernelGGL((k_inner_lanes<GG, BB, TT, false, false>), grid, block, 0, ctx->stream, A, C);       \
        }                                                                                                       \
    } while (0)
    if (bath) {
        if (G == 4) { if (terms) AMM_LAUNCH_INNER(4, true, true); else AMM_LAUNCH_INNER(4, true, false); }
        else { if (terms) AMM_LAUNCH_INNER(8, true, true); else AMM_LAUNCH_INNER(8, true, false); }
    } else {
        if (G == 4) { if (terms) AMM_LAUNCH_INNER(4, false, true); else AMM_LAUNCH_INNER(4, false, false); }
        else { if (terms) AMM_LAUNCH_INNER(8, false, true); else AMM_LAUNCH_INNER(8, false, false); }
    }
#undef AMM_LAUNCH_INNER
    AMM_HIP(hipGetLastError());
    return 0;
}

// lists whose rebuild trigger a launch that moves the atoms evaluates: the molecule rows first (the hot path's), then single per-atom
// lists (a hybrid list's per-atom part is one of those)
void amm_collect_watches(amm_ctx *ctx, WatchArgs &W) {
    W.n = 0;
    for (int q = 0; q < AMM_MAX_WATCH; ++q) {
        W.xref[q] = nullptr;
        W.thr2[q] = 0.0;
        W.flags[q] = nullptr;
    }
    auto watch = [&](const double *xref, double skin, int *flags, long *pre_epoch, const double **pre_pos) {
        if (W.n >= AMM_MAX_WATCH) return;
        W.xref[W.n] = xref;
        W.thr2[W.n] = 0.25 * skin * skin;
        W.flags[W.n] = flags;
        ListWatch &w = ctx->watched[W.n];
        w.xref = xref;
        w.thr2 = W.thr2[W.n];
        w.flags = flags;
        w.pre_epoch = pre_epoch;
        w.pre_pos = pre_pos;
        W.n++;
    };
    for (auto &fo : ctx->forces)
        if (fo.type == 1 && fo.pair->cl && fo.pair->cl->built && fo.pair->last_kind >= 1)
            watch(fo.pair->cl->d_xref, fo.pair->cl->skin, fo.pair->cl->d_flags, &fo.pair->cl->pre_epoch, &fo.pair->cl->pre_pos);
    for (auto &fo : ctx->forces)
        if (fo.type == 1 && fo.pair->built && !fo.pair->host && !fo.pair->dual && !(fo.pair->cl && fo.pair->last_kind >= 1))
            watch(fo.pair->d_xref, fo.pair->skin, fo.pair->d_flags, &fo.pair->pre_epoch, &fo.pair->pre_pos);
    ctx->n_watched = W.n;
}

// the positions the last collected watches were evaluated for are the context's current ones (call after pos_epoch was bumped)
void amm_watch_moved(amm_ctx *ctx) {
    for (int w = 0; w < ctx->n_watched; ++w) {
        *ctx->watched[w].pre_epoch = ctx->pos_epoch;
        *ctx->watched[w].pre_pos = ctx->d_x;
    }
}

int amm_bonded_free(BondedSet *bs) {
    if (bs->d_ref_ptr) (void)hipFree(bs->d_ref_ptr);
    if (bs->d_rec_a) (void)hipFree(bs->d_rec_a);
    if (bs->d_rec_q) (void)hipFree(bs->d_rec_q);
    if (bs->d_rec_l) (void)hipFree(bs->d_rec_l);
    for (void *ptr : {(void *)bs->d_gt_a, (void *)bs->d_gt_q, (void *)bs->d_rec_src, (void *)bs->d_tf})
        if (ptr) (void)hipFree(ptr);
    if (bs->d_term_l) (void)hipFree(bs->d_term_l);
    if (bs->d_term_q) (void)hipFree(bs->d_term_q);
    if (bs->d_atom_recs) (void)hipFree(bs->d_atom_recs);
    for (void *ptr : {(void *)bs->d_sc_atoms, (void *)bs->d_sc_term_l, (void *)bs->d_sc_term_q, (void *)bs->d_sc_recs, (void *)bs->d_big_atoms,
                      (void *)bs->d_big_terms})
        if (ptr) (void)hipFree(ptr);
    if (bs->d_comp_ptr) (void)hipFree(bs->d_comp_ptr);
    if (bs->d_comp_atoms) (void)hipFree(bs->d_comp_atoms);
    if (bs->d_epart) (void)hipFree(bs->d_epart);
    return 0;
}

int amm_bonded_arity(int kind) { return kArity[kind]; }
int amm_bonded_npar(int kind) { return kNpar[kind]; }
