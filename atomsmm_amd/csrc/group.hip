// atomsmm_amd/csrc/group.hip -- interaction-group pair forces with ONE SMALL SET, without a neighbour list (gfx950, fp64).
//
// The reference restricts some CustomNonbondedForce objects to an interaction group (set 1 x set 2): the softcore solute-solvent
// force of SolvationSystem (systems.py:266-272), the solute-solvent Lennard-Jones / Coulomb forces of AlchemicalRespaSystem
// (systems.py:696-708, 739-772).  One of the two sets is a solute: a few dozen atoms.  Such a force needs no cell list, no
// Verlet rows and none of the launches that maintain them (displacement check, cell assign, sort, conditional build: four
// launches per evaluation, and RESPASystem leaves these forces in group 0, i.e. on the INNERMOST loop -- config C5 paid them
// 32 times per AFED step): every atom of the large set tests the small set's atoms directly.
//
// Work decomposition: four lanes per atom j of the large set (grid stride over the atoms); the small set's positions and parameters
// sit in LDS; the four lanes split its atoms between them (a fixed order of summation) and keep the force on j.  The reaction
// forces on the small set's atoms are turned to 64-bit FIXED POINT (2^-40 kJ/mol/nm) lane by lane, summed over the wavefront
// through LDS (four small atoms at a time) -- only for the few (wavefront, small atoms) combinations that hold a pair inside the
// cutoff -- and added to accumulators with device-scope integer atomics: integer additions commute, so the sums -- and every bit
// of the forces -- are the same on every launch whatever the order the wavefronts arrive in and whichever atoms share a wavefront
// (the PME spread of pme.hip does the same); the LAST block (ticket) turns the accumulators into force rows and clears them.
// (A first version staged per-block floating-point partial sums for the last block to add in block order: the ~100 blocks'
// partials sit behind loads that bypass the XCD's L2 -- 17 us of a 38 us kernel at 249 075 atoms.)  Resolution 9e-13 against forces
// of 1e2..1e4: at the level of the fp64 rounding of the sums themselves.  Same pair arithmetic (amm_pair_math) and the same
// arguments as the list-based kernel k_pair_nlist.
// HBM traffic: positions + parameters + force rows once (~56 B per atom), 7.5 M distance tests at 249 075 atoms x 30 -- unless the
// launch walks CANDIDATES only (SmallArgs: the fused inner loop of amm_run_ops; ~1 700 atoms at config C5).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "amm_ctx.h"
#include "bonded_terms.h"
#include "cluster.h"
#include "device_utils.h"
#include "pair_math.h"

#define AMM_SMALL_MAX 128
#ifndef AMM_SG_BPC
#define AMM_SG_BPC 4        // blocks per CU at most (grid stride in the kernel)
#endif
#ifndef AMM_SG_EXP
#define AMM_SG_EXP 0        // measurement variants (scripts/build_variant.sh group ...): wrong forces
#endif
#define AMM_FIX_SCALE 1099511627776.0        // 2^40
#define AMM_FIX_MAX 1.0e6                    // a wavefront's share of a reaction force stays below this (a pair's beyond 1/16 of it, kJ/mol/nm, is reported): the sums hold +-8.4e6


struct SmallArgs {
    int n, j0, j1;                 // atoms; this rank's block of them
    int ns;                        // atoms of the small set
    const int *small;              // their indices (ascending)
    float small_code;              // set code of the small set (1 or 2); partners carry 3 - small_code
    const double *pos;             // [n][3], original order, not wrapped
    const double *q, *hsig, *seps2;
    const float *member;           // set code of every atom (0: in neither set)
    double *force;
    int accumulate;                // rows of the large set: += or =
    int small_accumulate;          // rows of the small set (written by the last block): += or =
    Box box;
    unsigned long long *acc;       // [ns][3] fixed-point sums of the reaction forces (zero between launches)
    int *overflow;                 // set when a wavefront's contribution leaves the fixed-point range (overlapping atoms): amm_check reports it
    int *ticket;
    double *epart;                 // [nblocks] energies (EN)
    // ---- candidate rows (inner loops: the force is evaluated far more often than atoms change neighbourhoods) ----
    // A companion neighbour list of the context keeps, for every atom, its position at the list's last build (xref) and two flags
    // that the kernels which move the atoms raise as soon as one of them is farther than skin / 2 (the list wants a rebuild) or
    // than the whole skin (AMM_FLAG_FAR) from it.  While the second flag is down a pair inside this force's cutoff was closer than
    // rc + 2 skin in the xref positions: the launch that follows a build of the companion (its build counter changed) walks every
    // atom as usual and also lists the atoms within rc + 2 skin of a small atom (`cand`, any order: no sum depends on it, see the
    // fixed-point remark above); later launches walk that list only -- 2 000 atoms instead of 249 075 at config C5 -- and leave
    // the other rows of `force` alone: they are zero and stay zero (`force` is then a buffer of this force's own:
    // SmallGroup::d_fpair).  (The doubled margin: the companion is rebuilt at its next evaluation, a few inner iterations AFTER the
    // first flag went up; with a margin of one skin those iterations would have to walk every atom.)
    int cand_on;                   // 0: plain launch (every atom, no list kept)
    int cand_trust;                // host: the companion's flag was evaluated for THESE positions
    int cand_may_list;             // this launch writes every row of the force's own buffer, so it may replace the list (a launch whose
                                   // rows go elsewhere must not: rows of atoms that drop out of the list would keep their last values)
    const double *xref;            // companion: [n][3]
    const int *lflags;             // companion: [AMM_FLAG_FAR] an atom left the doubled margin
    const unsigned long long *lcounters;     // companion: [0] builds
    double r2cand;                 // (rc + 2 skin)^2
    int *cand;                     // [n]
    int *cstate;                   // [0] candidates [1] companion build they belong to [2] append cursor (zero between launches)
    double *eout;                  // EN: the last block adds the blocks' energies to *eout (the order of k_reduce_add: no launch of its own)
    const double *lambda_dev;      // softcore family: lambda (PairConsts::alpha) is this device scalar, not the launch argument's (or nullptr)
};

// Term evaluation of a bond-list set that shares the force group with this pair force (amm_run_ops: the group of the innermost RESPA
// loop at config C5 = bond lists + softcore force): the blocks of this launch evaluate the terms too, with a grid stride -- the
// work of bonded.hip's k_terms_eval (same function, same parked forces), without a launch of its own.  nterms == 0: none.
struct TermsWork {
    int nterms, nblocks;      // nblocks: the first blocks of the grid, which do nothing else
    const int *list;          // the terms to evaluate (indices into gt_a / gt_q / tf); nullptr: all of 0 .. nterms - 1
    const int4 *gt_a;
    const double4 *gt_q;
    double *tf;
    BondedArgs A;
};

template <int FAM, bool GUARD, bool EN, bool GROUPED>
__global__ void __launch_bounds__(256) k_small_group(SmallArgs A, PairConsts c_launch, TermsWork T) {
    PairConsts c = c_launch;
    if (FAM == AMM_SOFTCORE && A.lambda_dev) c.alpha = A.lambda_dev[0];       // (amm_pair_set_lambda_dev: an AFED step left it on the device)
    __shared__ double4 s_pos[AMM_SMALL_MAX];          // x, y, z, Kc q
    __shared__ double2 s_lj[AMM_SMALL_MAX];
    __shared__ long long s_tr[4][12][65];             // per wavefront: the twelve reaction components of a trip, [value][lane], fixed point
    __shared__ double s_ref[AMM_SMALL_MAX][3];        // the small atoms in the companion list's reference positions (candidate builds)
    static_assert(FAM == AMM_SOFTCORE || FAM == AMM_NEAR_FSWITCH || FAM == AMM_NONBONDED, "families of the reference's interaction groups");
    const double *s_tab = nullptr;          // (none of them evaluates erfc: NONBONDED here is the plain-Coulomb instantiation, CMODE 0)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // Block roles: the first T.nblocks blocks evaluate the bond-list terms (nothing else), the others walk atoms -- two chains of
    // dependent round trips side by side instead of one behind the other.
    const int pb = (int)blockIdx.x - T.nblocks, npb = (int)gridDim.x - T.nblocks;        // this block among the atom walkers
    // (a candidate slot read before anyone knows whether the candidates are walked: one round trip less on that chain)
    const int spec = (A.cand_on && pb >= 0) ? A.cand[min(pb * 64 + (int)(threadIdx.x >> 2), A.n - 1)] : 0;
    // walk every atom (and, `listing`, keep the candidates), or the candidates only: the same decision in every block -- the state
    // changes in the LAST block's tail only, when every other block has read it
    bool full = true, listing = false;
    int ncand = 0;
    if (A.cand_on && A.cand_trust && A.lflags[AMM_FLAG_FAR] == 0) {
        if ((int)A.lcounters[0] == A.cstate[1]) {
            full = false;
            ncand = A.cstate[0];
        } else {
            listing = A.cand_may_list != 0;
        }
    }
    // a candidate walk needs few of the blocks; the others leave at once and the ticket counts the rest
    const int walkers = full ? npb : max(1, min(npb, (ncand + 63) >> 6));
    if (pb >= walkers) {
        if (EN && threadIdx.x == 0) amm_st_l2((unsigned long long *)&A.epart[blockIdx.x], 0ull);
        return;
    }
    const int ticket_n = full ? (int)gridDim.x : T.nblocks + walkers;
    if (pb < 0) {
        for (int kt = blockIdx.x * 256 + threadIdx.x; kt < T.nterms; kt += T.nblocks * 256) {
            const int t = T.list ? T.list[kt] : kt;
            const int4 at = T.gt_a[t];
            const double4 q = T.gt_q[t];
            const long long code = __double_as_longlong(q.w);
            const int kind = (int)(code & 7), periodic = (int)((code >> 5) & 1);
            const int ix[4] = {at.x, at.y, at.z, at.w};
            const double p[3] = {q.x, q.y, q.z};
            double fo[4][3], e;
            PosPlain pos{T.A.pos};
            bonded_term_forces(T.A, pos, ix, p, kind, periodic, fo, e);
            double *out = T.tf + (size_t)t * 12;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int x = 0; x < 3; ++x) out[3 * r + x] = fo[r][x];
        }
        if (EN && threadIdx.x == 0) amm_st_l2((unsigned long long *)&A.epart[blockIdx.x], 0ull);
    }
    const double guard2 = GUARD ? c.rc0 * c.rc0 : 0.0;
    double esum = 0.0;
    // Four lanes share an atom j of the large set: lane `sub` of them takes the small atoms sub, sub + 4, sub + 8 ... -- four per
    // trip, as independent chains.  The wavefronts next to the solute meet EVERY small atom and are the kernel's critical path:
    // sixteen atoms per wavefront instead of sixty-four cuts it four times (30 small atoms: 2 trips of pair arithmetic per
    // wavefront instead of 8; measured 31 -> see DESIGN.md).  A block walks its share of the atoms with a grid stride.
    const int sub = lane & 3;
    // the next atom's record is fetched while this one is tested (all loads at once, whatever the atom's set): the kernel is a chain
    // of dependent round trips at four wavefronts per SIMD, and a trip of the loop below is shorter than one of them
    struct Rec {
        int j;
        float code;
        double px, py, pz, q;
        double2 lj;
    };
    // (it: position in the walk -- an atom of this rank's block, or a slot of the candidate list)
    const int walk_n = full ? A.j1 - A.j0 : ncand;
    auto atom_of = [&](int it) {
        const int t = min(max(it, 0), max(walk_n - 1, 0));
        return full ? A.j0 + t : A.cand[t];
    };
    auto fetch_atom = [&](int jl) {
        Rec r;
        r.j = jl;
        r.code = A.member[jl];
        r.px = A.pos[3 * jl];
        r.py = A.pos[3 * jl + 1];
        r.pz = A.pos[3 * jl + 2];
        r.q = A.q[jl];
        r.lj = make_double2(A.hsig[jl], A.seps2[jl]);
        return r;
    };
    auto fetch = [&](int itb) {
        const int jl = walk_n > 0 ? atom_of(itb + (int)(threadIdx.x >> 2)) : 0;
        return fetch_atom(jl);
    };
    const int jstride = npb * 64;
    // the first records are on their way (the candidate slot came with the flags) while the small set is staged
    Rec nxt;
    if (pb >= 0) {
        const int it0 = pb * 64 + (int)(threadIdx.x >> 2);
        nxt = fetch_atom(full ? min(A.j0 + it0, A.j1 - 1) : spec);       // (every slot of `cand` holds an atom, also beyond the list)
        for (int k = threadIdx.x; k < A.ns; k += 256) {
            const int i = A.small[k];
            s_pos[k] = make_double4(A.pos[3 * i], A.pos[3 * i + 1], A.pos[3 * i + 2], A.q[i]);
            s_lj[k] = make_double2(A.hsig[i], A.seps2[i]);
            if (listing) {
                s_ref[k][0] = A.xref[3 * i];
                s_ref[k][1] = A.xref[3 * i + 1];
                s_ref[k][2] = A.xref[3 * i + 2];
            }
        }
        __syncthreads();
    }
    for (int itb = pb * 64; pb >= 0 && itb < walk_n; itb += jstride) {
        const Rec cur = nxt;
        if (itb + jstride < walk_n) nxt = fetch(itb + jstride);
        const bool in = itb + (int)(threadIdx.x >> 2) < walk_n;
        const int j = cur.j;
        const float code = in ? cur.code : 0.f;
        const double px = cur.px, py = cur.py, pz = cur.pz;
        const double qj = c.Kc * cur.q;
        const double2 lj = cur.lj;
        const bool partner = in && code != 0.f && code != A.small_code;       // an atom of the other (large) set
        double fx = 0.0, fy = 0.0, fz = 0.0;
        if (listing) {
            // candidate test in the companion's reference positions (lane `sub` takes every fourth small atom, as below)
            const double rx = A.xref[3 * j], ry = A.xref[3 * j + 1], rz = A.xref[3 * j + 2];
            bool near = false;
            for (int kk = sub; kk < A.ns; kk += 4) {
                const double ddx = amm_min_image(rx - s_ref[kk][0], A.box.L[0], A.box.invL[0]);
                const double ddy = amm_min_image(ry - s_ref[kk][1], A.box.L[1], A.box.invL[1]);
                const double ddz = amm_min_image(rz - s_ref[kk][2], A.box.L[2], A.box.invL[2]);
                near = near || (ddx * ddx + ddy * ddy + ddz * ddz < A.r2cand);
            }
            const unsigned long long m = __builtin_amdgcn_ballot_w64(near && partner);
            const unsigned int four = (unsigned int)(m >> (lane & ~3)) & 15u;          // the four lanes of this atom
            if (sub == 0 && four != 0u) A.cand[atomicAdd(&A.cstate[2], 1)] = j;
        }
        // (a wavefront without an atom of the large set has nothing to do: wave-uniform)
        if (AMM_SG_EXP != 3 && __builtin_amdgcn_ballot_w64(partner) != 0ull) {
            for (int k0 = 0; k0 < A.ns; k0 += 16) {
                double dx[4], dy[4], dz[4], r2[4];
                bool pass[4];
                bool any = false;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int kk = k0 + 4 * u + sub;
                    const double4 pk = s_pos[min(kk, A.ns - 1)];
                    dx[u] = amm_min_image(px - pk.x, A.box.L[0], A.box.invL[0]);
                    dy[u] = amm_min_image(py - pk.y, A.box.L[1], A.box.invL[1]);
                    dz[u] = amm_min_image(pz - pk.z, A.box.L[2], A.box.invL[2]);
                    r2[u] = dx[u] * dx[u] + dy[u] * dy[u] + dz[u] * dz[u];
                    pass[u] = partner && (kk < A.ns) && (r2[u] < c.rc2);
                    if (GUARD) pass[u] = pass[u] && (r2[u] <= guard2);
                    any = any || pass[u];
                }
                if (__builtin_amdgcn_ballot_w64(any) == 0ull) continue;            // wave-uniform: almost always
#if AMM_SG_EXP == 1        // measurement only: no pair arithmetic at all
                continue;
#endif
                double rr[4][3];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int kk = min(k0 + 4 * u + sub, A.ns - 1);
                    const double4 pk = s_pos[kk];
                    const double2 lk = s_lj[kk];
                    double e, fr;
                    amm_pair_math<FAM, 0, false, EN, GROUPED>(c, pass[u] ? r2[u] : 1.0, qj * pk.w, lj.x + lk.x, lj.y * lk.y, e, fr, s_tab);
                    fr = pass[u] ? fr : 0.0;
                    const double gx = fr * dx[u], gy = fr * dy[u], gz = fr * dz[u];
                    fx += gx;
                    fy += gy;
                    fz += gz;
                    if (EN) esum += pass[u] ? e : 0.0;
                    rr[u][0] = -gx;
                    rr[u][1] = -gy;
                    rr[u][2] = -gz;
                }
                // Reactions on the small atoms k0 + 4 u + s (u, s = 0..3), three components each: 48 sums over the 16 lanes with
                // sub = s.  Through LDS: every lane parks its twelve values ([value][lane], rows padded against bank conflicts),
                // lane 4 v + s adds value v of the lanes s, s + 4, ... in lane order -- a fixed order, and a tenth of the
                // cross-lane traffic of the six-round butterflies that stood here first (17 us of this kernel then)
                long long(*tr)[65] = s_tr[w];
                __builtin_amdgcn_wave_barrier();                  // the previous trip's reads are done
                bool big = false;
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        // fixed point BEFORE the lanes are added: integers commute, so the sum does not depend on which atoms
                        // share a wavefront (the candidate walk groups them differently from the walk over every atom)
                        big = big || !(fabs(rr[u][d]) < AMM_FIX_MAX / 16);     // (NaN too; sixteen lanes add up below)
                        tr[3 * u + d][lane] = __double2ll_rn(rr[u][d] * AMM_FIX_SCALE);
                    }
                if (big) A.overflow[0] = 1;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int v = min(lane >> 2, 11);
                long long part = 0;
#pragma unroll
                for (int i = 0; i < 16; ++i) part += tr[v][sub + 4 * i];
                // (small atom of this sum: k0 + 4 (v / 3) + sub, component v % 3; skipped when no lane holds a pair with it)
                const int ku = v / 3, kd = v - 3 * ku, ksum = k0 + 4 * ku + sub;
                bool mine = false;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(pass[u]);       // lanes (any sub) with a pair in slot u
                    // the lanes with this lane's sub sit at bit positions == sub (mod 4)
                    const bool hit = (m & (0x1111111111111111ull << sub)) != 0ull;
                    mine = mine || (u == ku && hit);
                }
                if (lane < 48 && mine && ksum < A.ns) atomicAdd(&A.acc[3 * ksum + kd], (unsigned long long)part);
            }
        }
        // the four lanes' shares of the force on j
        fx += __shfl_xor(fx, 1);
        fy += __shfl_xor(fy, 1);
        fz += __shfl_xor(fz, 1);
        fx += __shfl_xor(fx, 2);
        fy += __shfl_xor(fy, 2);
        fz += __shfl_xor(fz, 2);
        if (sub == 0) {
            if (partner) {
                if (A.accumulate) {
                    A.force[3 * j] += fx;
                    A.force[3 * j + 1] += fy;
                    A.force[3 * j + 2] += fz;
                } else {
                    A.force[3 * j] = fx;
                    A.force[3 * j + 1] = fy;
                    A.force[3 * j + 2] = fz;
                }
            } else if (in && !A.accumulate && code != A.small_code) {
                A.force[3 * j] = A.force[3 * j + 1] = A.force[3 * j + 2] = 0.0;       // an atom of neither set
            }
        }
    }
    if (EN && pb >= 0) {
        __shared__ double red[4];
        for (int off = 32; off > 0; off >>= 1) esum += __shfl_xor(esum, off);
        if (lane == 0) red[w] = esum;
        __syncthreads();
        // (through L2: the last block reads it)
        if (threadIdx.x == 0) amm_st_l2((unsigned long long *)&A.epart[blockIdx.x], (unsigned long long)__double_as_longlong(((red[0] + red[1]) + red[2]) + red[3]));       // every pair once: no factor 1/2
    }
#if AMM_SG_EXP == 2            // measurement only: no ticket, no tail
    return;
#endif
    if (!(full ? amm_last_block(A.ticket) : amm_last_of(A.ticket, ticket_n))) return;
    if (EN && A.eout) {
        // the blocks' energies, added up as k_reduce_add does (256 strided partial sums, then a tree): blocks that did not walk hold 0
        __shared__ double sh[256];
        double s = 0.0;
        for (int i = threadIdx.x; i < (int)gridDim.x; i += 256)
            s += i < ticket_n ? __longlong_as_double((long long)amm_ld_l2((const unsigned long long *)&A.epart[i])) : 0.0;
        sh[threadIdx.x] = s;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x == 0) A.eout[0] += sh[0];
    }
    if (!full && threadIdx.x == 0) A.cstate[3] += 1;
    if (listing && threadIdx.x == 0) {         // the list this launch made serves from the next launch on
        A.cstate[0] = amm_ld_l2(&A.cstate[2]);
        A.cstate[1] = (int)A.lcounters[0];
        amm_st_l2(&A.cstate[2], 0);
    }
    // ---- last block: accumulators -> force rows of the small set; the accumulators are cleared for the next launch ----
    for (int t = threadIdx.x; t < A.ns * 3; t += 256) {
        const long long fixed = (long long)amm_ld_l2(&A.acc[t]);
        amm_st_l2(&A.acc[t], 0ull);
        const double sum = (double)fixed * (1.0 / AMM_FIX_SCALE);
        const int k = t / 3, d = t - 3 * k, i = A.small[k];
        if (A.small_accumulate) A.force[3 * i + d] += sum;
        else A.force[3 * i + d] = sum;
    }
}

struct SmallGroup {
    int ns = 0;
    float code = 0.f;
    int *d_small = nullptr;
    unsigned long long *d_acc = nullptr;
    int *d_ticket = nullptr, *d_overflow = nullptr;
    double *d_epart = nullptr;
    int nblocks = 0;
    // candidate rows (SmallArgs): the force's own rows, the list and its state
    double *d_fpair = nullptr;
    int *d_cand = nullptr, *d_cstate = nullptr;
    const void *companion = nullptr;       // the list the candidates were made against (another one: the list starts over)
};

int amm_small_group_free(SmallGroup *sg) {
    if (!sg) return 0;
    void *ptrs[] = {sg->d_small, sg->d_acc, sg->d_ticket, sg->d_epart, sg->d_overflow, sg->d_fpair, sg->d_cand, sg->d_cstate};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    delete sg;
    return 0;
}

// (re)decide whether `pf` takes this path: set codes on the host, the exclusion CSR of the force
int amm_small_group_setup(amm_ctx *ctx, PairForce *pf, const std::vector<float> &member) {
    if (pf->small) {
        amm_small_group_free(pf->small);
        pf->small = nullptr;
    }
    const int n = pf->n;
    std::vector<int> set1, set2;
    for (int i = 0; i < n; ++i) {
        if (member[i] == 1.0f) set1.push_back(i);
        else if (member[i] == 2.0f) set2.push_back(i);
        else if (member[i] != 0.0f) return 0;           // codes other than 0, 1, 2: the list path's product test decides
    }
    const bool first = set1.size() <= set2.size();
    const std::vector<int> &small = first ? set1 : set2;
    if (small.empty() || (int)small.size() > AMM_SMALL_MAX) return 0;
    // an excluded (set 1, set 2) pair would have to be skipped per pair: left to the list path (none in the reference's systems)
    for (int i : small)
        for (int e = pf->h_excl_ptr[i]; e < pf->h_excl_ptr[i + 1]; ++e) {
            const float other = member[pf->h_excl_idx[e]];
            if (other != 0.0f && other != member[i]) return 0;
        }
    SmallGroup *sg = new SmallGroup();
    sg->ns = (int)small.size();
    sg->code = first ? 1.0f : 2.0f;
    sg->nblocks = (n + 63) / 64;            // (a rank's block of atoms needs fewer; the slice may be set after this)
    AMM_HIP(hipMalloc(&sg->d_small, sizeof(int) * sg->ns));
    AMM_HIP(hipMemcpy(sg->d_small, small.data(), sizeof(int) * sg->ns, hipMemcpyHostToDevice));
    AMM_HIP(hipMalloc(&sg->d_acc, sizeof(unsigned long long) * sg->ns * 3));
    AMM_HIP(hipMemset(sg->d_acc, 0, sizeof(unsigned long long) * sg->ns * 3));
    AMM_HIP(hipMalloc(&sg->d_ticket, sizeof(int) * AMM_TICKET_INTS));
    AMM_HIP(hipMemset(sg->d_ticket, 0, sizeof(int) * AMM_TICKET_INTS));
    AMM_HIP(hipMalloc(&sg->d_epart, sizeof(double) * (sg->nblocks + 64)));
    AMM_HIP(hipMalloc(&sg->d_overflow, sizeof(int)));
    AMM_HIP(hipMemset(sg->d_overflow, 0, sizeof(int)));
    pf->small = sg;
    return 0;
}

template <int FAM, bool GROUPED>
static void launch_small(hipStream_t st, int nblocks, bool guard, bool en, const SmallArgs &A, const PairConsts &c, const TermsWork &T) {
    dim3 grid(nblocks), block(256);
    if (guard) {
        if (en) hipLaunchKernelGGL((k_small_group<FAM, true, true, GROUPED>), grid, block, 0, st, A, c, T);
        else hipLaunchKernelGGL((k_small_group<FAM, true, false, GROUPED>), grid, block, 0, st, A, c, T);
    } else {
        if (en) hipLaunchKernelGGL((k_small_group<FAM, false, true, GROUPED>), grid, block, 0, st, A, c, T);
        else hipLaunchKernelGGL((k_small_group<FAM, false, false, GROUPED>), grid, block, 0, st, A, c, T);
    }
}

// same contract as amm_pair_eval_impl (no guest, no exchange); returns -1 when the force's family has no instantiation here
// the families k_small_group is instantiated for (every caller that cannot fall back to the list path asks first)
bool amm_small_group_supported(const PairForce *pf) {
    if (!pf || !pf->small) return false;
    const int fam = pf->desc.family;
    const bool grouped = (pf->pc.flags & (AMM_GROUP_LJ | AMM_GROUP_Q)) != 0;
    return (fam == AMM_SOFTCORE && !grouped) || (grouped && (fam == AMM_NEAR_FSWITCH || (fam == AMM_NONBONDED && pf->pc.cmode == 0)));
}

int amm_small_group_eval_impl(amm_ctx *ctx, PairForce *pf, const double *d_pos, double *d_force, int accumulate, double *d_energy,
                              BondedSet *carry_terms, const double **own_rows, int rows_unused) {
    SmallGroup *sg = pf->small;
    if (own_rows) *own_rows = nullptr;
    hipStream_t st = ctx->stream;
    const int fam = pf->desc.family;
    const bool grouped = (pf->pc.flags & (AMM_GROUP_LJ | AMM_GROUP_Q)) != 0;
    if (!amm_small_group_supported(pf)) return -1;
    const int n = pf->n;
    const int per = (n + ctx->world - 1) / ctx->world;
    SmallArgs A;
    A.n = n;
    A.j0 = std::min(n, ctx->rank * per);
    A.j1 = std::min(n, A.j0 + per);
    A.ns = sg->ns;
    A.small = sg->d_small;
    A.small_code = sg->code;
    A.pos = d_pos;
    A.q = pf->d_q;
    A.hsig = pf->d_hsig;
    A.seps2 = pf->d_seps2;
    A.member = pf->d_member;
    A.force = d_force;
    A.accumulate = accumulate;
    A.small_accumulate = accumulate;
    if (ctx->world > 1 && !accumulate) {          // the rows outside this rank's block are zero: the group is summed over the ranks
        AMM_HIP(hipMemsetAsync(d_force, 0, sizeof(double) * 3 * (size_t)n, st));
        A.accumulate = A.small_accumulate = 1;
    }
    A.box = ctx->box;
    A.acc = sg->d_acc;
    A.overflow = sg->d_overflow;
    A.ticket = sg->d_ticket;
    A.epart = sg->d_epart;
    A.cand_on = A.cand_trust = A.cand_may_list = 0;
    A.xref = nullptr;
    A.lflags = nullptr;
    A.lcounters = nullptr;
    A.r2cand = 0.0;
    A.cand = nullptr;
    A.cstate = nullptr;
    A.lambda_dev = pf->d_lambda_dev;
    A.eout = d_energy;
    // candidates: when the caller takes the rows from the force's own buffer (the fused inner loop), or when it wants the energy
    // alone (rows_unused: deriv(energy, lambda) -- d_force is scratch, only the candidates' rows are written)
    if ((own_rows ? !d_energy : (rows_unused && d_energy)) && ctx->opt_group_candidates && ctx->world == 1 && !accumulate) {
        // the companion: a molecule-row list of the context (its reference positions and its flag cover EVERY atom, the atoms
        // outside the molecules too); the caller reads this force's rows from the force's own buffer
        const ClusterList *cl = nullptr;
        for (auto &fo : ctx->forces)
            if (fo.type == 1 && fo.pair->cl && fo.pair->cl->built && fo.pair->last_kind >= 1 && !fo.pair->host) {
                cl = fo.pair->cl;
                break;
            }
        if (cl) {
            if (!sg->d_fpair) {
                AMM_HIP(hipMalloc(&sg->d_fpair, sizeof(double) * 3 * (size_t)n));
                AMM_HIP(hipMalloc(&sg->d_cand, sizeof(int) * (size_t)n));
                AMM_HIP(hipMemset(sg->d_cand, 0, sizeof(int) * (size_t)n));
                AMM_HIP(hipMalloc(&sg->d_cstate, sizeof(int) * 4));
                const int init[4] = {0, -1, 0, 0};
                AMM_HIP(hipMemcpy(sg->d_cstate, init, sizeof(init), hipMemcpyHostToDevice));
            }
            if (sg->companion != (const void *)cl) {
                // (build counters of two lists are not comparable: forget the candidates; ordered on the stream)
                static const int fresh[4] = {0, -1, 0, 0};
                AMM_HIP(hipMemcpyAsync(sg->d_cstate, fresh, sizeof(int) * 3, hipMemcpyHostToDevice, st));
                sg->companion = cl;
            }
            A.cand_on = 1;
            A.cand_may_list = own_rows ? 1 : 0;
            A.cand_trust = ((cl->pre_epoch == ctx->pos_epoch && cl->pre_pos == d_pos) || (cl->checked_epoch == ctx->pos_epoch && cl->checked_pos == d_pos)) ? 1 : 0;
            A.xref = cl->d_xref;
            A.lflags = cl->d_flags;
            A.lcounters = cl->d_counters;
            const double reach = std::sqrt(pf->pc.rc2) + 2.0 * cl->skin;
            A.r2cand = reach * reach;
            A.cand = sg->d_cand;
            A.cstate = sg->d_cstate;
            if (own_rows) {
                A.force = sg->d_fpair;
                *own_rows = sg->d_fpair;
            }
        }
    }
    // AMM_SG_BPC blocks per CU at most (grid stride in the kernel)
    static int ncu_dev[64] = {0};
    int &ncu = ncu_dev[ctx->device & 63];
    if (!ncu) AMM_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device));
    const int nblocks = std::max(1, std::min((A.j1 - A.j0 + 63) / 64, AMM_SG_BPC * std::max(ncu, 1)));
    const bool guard = (pf->desc.flags & AMM_GUARD_RC0) != 0, en = d_energy != nullptr;
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
    TermsWork T;
    T.nblocks = 0;
    T.nterms = 0;
    T.list = nullptr;
    T.gt_a = nullptr;
    T.gt_q = nullptr;
    T.tf = nullptr;
    if (carry_terms) {
        if (amm_bonded_terms_work(ctx, carry_terms, d_pos, &T.A, &T.nterms, &T.gt_a, &T.gt_q, &T.tf, &T.list)) return 1;
    } else {
        std::memset(&T.A, 0, sizeof(T.A));
    }
    T.nblocks = std::min(64, (T.nterms + 255) / 256);
    const int grid = nblocks + T.nblocks;
    if (fam == AMM_SOFTCORE) launch_small<AMM_SOFTCORE, false>(st, grid, false, en, A, pf->pc, T);
    else if (fam == AMM_NEAR_FSWITCH) launch_small<AMM_NEAR_FSWITCH, true>(st, grid, guard, en, A, pf->pc, T);
    else launch_small<AMM_NONBONDED, true>(st, grid, false, en, A, pf->pc, T);
    if (timed) AMM_HIP(hipEventRecord(e1, st));
    AMM_HIP(hipGetLastError());
    pf->n_evals++;
    pf->last_kind = 3;
    return 0;
}

// out[0] = candidates of the current list, out[1] = evaluations that walked candidates only (the caller has synchronised)
int amm_small_group_stats(SmallGroup *sg, int out[2]) {
    out[0] = out[1] = 0;
    if (!sg->d_cstate) return 0;
    int st[4];
    if (hipMemcpy(st, sg->d_cstate, sizeof(st), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    out[0] = st[0];
    out[1] = st[3];
    return 0;
}

// pending device-side error of a list-free group force (amm_check; the caller has synchronised): 1 = a reaction force left the
// fixed-point range of the accumulators
int amm_small_group_failed(SmallGroup *sg) {
    int flag = 0;
    if (hipMemcpy(&flag, sg->d_overflow, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return flag;
}
