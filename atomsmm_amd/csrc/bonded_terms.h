// atomsmm_amd/csrc/bonded_terms.h -- one bond-list term: argument block, position accessors and the forces of a term on all of
// its atoms (bonded_term_forces).  Shared by bonded.hip (owner-computes / term-parallel / component kernels) and group.hip (whose
// list-free interaction-group launch carries the term evaluation of the same force group: one launch instead of two per inner
// RESPA iteration of config C5).
#pragma once
#include "amm_ctx.h"
#include "pair_math.h"

struct BondedArgs {
    int n, row_begin, row_end;
    const int *ref_ptr;
    // one packed record per (atom, term) reference: the term's atoms and parameters inline, so a thread needs
    // only two dependent load levels (ref_ptr -> records -> positions) instead of four
    const int4 *rec_a;      // atom indices of the term (-1 padded)
    const double4 *rec_q;   // p0, p1, p2, and kind | role<<3 | periodic<<5 in the bits of .w
    const int4 *rec_l;      // the same atoms as slots inside their connected component (component kernel)
    const double *pos;
    double *force;
    double *epart;
    int accumulate, want_energy;
    Box box;
    PairConsts near_pc;
    double ewald_alpha, ewald_tasp;
    double Kc_ljc;
};

// position accessors: plain array, or "state advanced by kick+move" (fused inner RESPA iteration)
struct PosPlain {
    const double *pos;
    __device__ __forceinline__ double get(int a, int k) const { return pos[3 * a + k]; }
};
struct PosAdvanced {
    // x_new = x + d*(v + (c1*f)/m): exactly the arithmetic (and rounding sequence) of k_kick followed by k_move
    const double *x, *v, *f, *m;
    double c1, d;
    __device__ __forceinline__ double vel(int a, int k) const {
#pragma clang fp contract(off)
        const double num = c1 * f[3 * a + k];
        const double dv = num / m[a];
        return v[3 * a + k] + dv;
    }
    __device__ __forceinline__ double get(int a, int k) const {
#pragma clang fp contract(off)
        const double dx = d * vel(a, k);
        return x[3 * a + k] + dx;
    }
};

// displacement triggers of the watched lists for ONE atom at its new position (movers call this for every atom they move)
__device__ __forceinline__ void amm_watch_atom(const WatchArgs &W, int a, const double *xn) {
    for (int q = 0; q < W.n; ++q) {
        const double dx = xn[0] - W.xref[q][3 * a], dy = xn[1] - W.xref[q][3 * a + 1], dz = xn[2] - W.xref[q][3 * a + 2];
        const double d2 = dx * dx + dy * dy + dz * dz;
        if (!(d2 <= W.thr2[q])) {                                              // benign race (NaN also triggers)
            W.flags[q][0] = 1;
            if (!(d2 <= 4.0 * W.thr2[q])) W.flags[q][AMM_FLAG_FAR] = 1;
        }
    }
}

// num / m, correctly rounded, from a precomputed r = RN(1/m) (Markstein 1990): q = RN(num r), e = num - q m exactly
// (fma), result RN(q + e r) -- identical to the IEEE quotient unless m's significand is all ones, where RN(1/m) is
// not within half an ulp; those masses (never seen in a force field) take the hardware division.  3 instructions
// instead of the ~30 of the fp64 division sequence; the inner loop does 18 of them per iteration.
__device__ __forceinline__ double amm_div_mass(double num, double m, double r, bool exact_r) {
    const double q = num * r;
    const double e = fma(-q, m, num);
    const double fast = fma(e, r, q);
    // (the IEEE division only when some lane of the wavefront needs it: as a select both were computed -- 3 x ~30 instructions per
    // kick in a loop that is a chain of dependent latencies; the branch is wave-uniform, the result the same)
    if (__builtin_amdgcn_ballot_w64(!exact_r) == 0ull) return fast;
    return exact_r ? fast : num / m;
}

// positions of a component's atoms in a wavefront-private LDS strip, indexed by the atom's slot in its component
struct PosLds {
    const double *sx, *sy, *sz;     // strip of this lane's group
    __device__ __forceinline__ double get(int slot, int k) const { return k == 0 ? sx[slot] : (k == 1 ? sy[slot] : sz[slot]); }
};

template <class P>
__device__ __forceinline__ void delta3(const P &pos, int a, int b, const Box &box, int periodic, double *d) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double v = pos.get(a, k) - pos.get(b, k);
        if (periodic) v = amm_min_image(v, box.L[k], box.invL[k]);
        d[k] = v;
    }
}
__device__ __forceinline__ double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ void cross3(const double *a, const double *b, double *c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

// Forces of ONE bond-list term on all of its atoms (fo[role][xyz]) and its energy.  Every path (owner-computes
// per atom, fused inner iteration, component kernel) goes through this one function, so they agree bit for bit;
// roles are related by exact IEEE symmetries (x_j - x_i = -(x_i - x_j), rint odd), e.g. f_1 = -f_0 for a bond.
template <class P>
__device__ __forceinline__ void bonded_term_forces(const BondedArgs &A, const P &pos, const int *ix, const double *p, int kind,
                                                   int periodic, double fo[4][3], double &e) {
    switch (kind) {
    case AMM_BOND_HARMONIC: {
        double d[3];
        delta3(pos, ix[0], ix[1], A.box, periodic, d);
        const double r2 = dot3(d, d);
        const double rinv = amm_rsqrt(r2);         // reciprocal-sqrt + Newton: no IEEE sqrt/divide sequences
        const double dr = r2 * rinv - p[0];
        const double fr = -p[1] * dr * rinv;
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            fo[0][x] = fr * d[x];
            fo[1][x] = -fo[0][x];
        }
        e = 0.5 * p[1] * dr * dr;
    } break;
    case AMM_ANGLE_HARMONIC: {
        double d1[3], d2[3];
        delta3(pos, ix[0], ix[1], A.box, periodic, d1);
        delta3(pos, ix[2], ix[1], A.box, periodic, d2);
        const double i1 = amm_rsqrt(dot3(d1, d1)), i2 = amm_rsqrt(dot3(d2, d2));
        double c = dot3(d1, d2) * i1 * i2;
        c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
        const double th = acos(c), dth = th - p[0];
        double s2 = 1.0 - c * c;
        if (s2 < 1e-24) s2 = 1e-24;
        const double g = p[1] * dth * amm_rsqrt(s2);
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            const double u1 = d1[x] * i1, u2 = d2[x] * i2;      // unit vectors
            const double fi = g * (u2 - c * u1) * i1;
            const double fk = g * (u1 - c * u2) * i2;
            fo[0][x] = fi;
            fo[2][x] = fk;
            fo[1][x] = -(fi + fk);
        }
        e = 0.5 * p[1] * dth * dth;
    } break;
    case AMM_BOND_LJC:
    case AMM_BOND_NEAR: {
        double d[3];
        delta3(pos, ix[0], ix[1], A.box, periodic, d);
        const double r2 = dot3(d, d);
        double fr;
        if (kind == AMM_BOND_LJC) {   // forces.py:406
            const double rinv2 = 1.0 / r2, rinv = sqrt(rinv2);
            const double s2 = p[1] * p[1] * rinv2, x6 = s2 * s2 * s2;
            e = 4.0 * p[2] * x6 * (x6 - 1.0) + A.Kc_ljc * p[0] * rinv;
            fr = (4.0 * p[2] * (12.0 * x6 * x6 - 6.0 * x6) + A.Kc_ljc * p[0] * rinv) * rinv2;
        } else {
            amm_pair_math_rt(A.near_pc, r2, A.near_pc.Kc * p[0], p[1], 4.0 * p[2], e, fr);
        }
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            fo[0][x] = fr * d[x];
            fo[1][x] = -fo[0][x];
        }
    } break;
    case AMM_BOND_EWALD_EXCL: {
        const double qq = p[0];
        double d[3];
        delta3(pos, ix[0], ix[1], A.box, 1, d);
        const double r2 = dot3(d, d), rr = sqrt(r2), ar = A.ewald_alpha * rr;
        const double er = erf(ar);
        // E = -qq erf(ar)/r ;  -dE/dr = qq [ tasp exp(-a^2 r^2)/r - erf(ar)/r^2 ]
        const double fr = qq * (A.ewald_tasp * exp(-ar * ar) / rr - er / r2) / rr;
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            fo[0][x] = fr * d[x];
            fo[1][x] = -fo[0][x];
        }
        e = -qq * er / rr;
    } break;
    case AMM_BOND_VIRIAL_HARMONIC:
    case AMM_BOND_VIRIAL_LJ: {       // virial contributions as energies (ComputingSystem, systems.py:894-915)
        double d[3];
        delta3(pos, ix[0], ix[1], A.box, periodic, d);
        const double r2 = dot3(d, d), r = sqrt(r2);
        double fr;
        if (kind == AMM_BOND_VIRIAL_HARMONIC) {
            e = -p[1] * r * (r - p[0]);
            fr = p[1] * (2.0 * r - p[0]) / r;                      // -(dE/dr)/r
        } else {
            const double s2 = p[1] * p[1] / r2, x = s2 * s2 * s2;
            e = 24.0 * p[2] * (2.0 * x * x - x);
            fr = 24.0 * p[2] * (24.0 * x * x - 6.0 * x) / r2;
        }
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            fo[0][x] = fr * d[x];
            fo[1][x] = -fo[0][x];
        }
    } break;
    case AMM_TORSION_PERIODIC: {
        double F[3], G[3], H[3], Av[3], Bv[3], BA[3];
        delta3(pos, ix[0], ix[1], A.box, periodic, F);
        delta3(pos, ix[1], ix[2], A.box, periodic, G);
        delta3(pos, ix[3], ix[2], A.box, periodic, H);
        cross3(F, G, Av);
        cross3(H, G, Bv);
        cross3(Bv, Av, BA);
        const double Gn = sqrt(dot3(G, G));
        const double phi = atan2(dot3(BA, G) / Gn, dot3(Av, Bv));
        const double nper = p[0];
        const double dEdphi = -p[2] * nper * sin(nper * phi - p[1]);
        const double A2 = dot3(Av, Av), B2 = dot3(Bv, Bv), FG = dot3(F, G), HG = dot3(H, G);
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            const double gi = -Gn / A2 * Av[x], gl = Gn / B2 * Bv[x];
            const double gj = Gn / A2 * Av[x] + FG / (A2 * Gn) * Av[x] - HG / (B2 * Gn) * Bv[x];
            const double gk = -Gn / B2 * Bv[x] - FG / (A2 * Gn) * Av[x] + HG / (B2 * Gn) * Bv[x];
            fo[0][x] = -(dEdphi * gi);
            fo[1][x] = -(dEdphi * gj);
            fo[2][x] = -(dEdphi * gk);
            fo[3][x] = -(dEdphi * gl);
        }
        e = p[2] * (1.0 + cos(nper * phi - p[1]));
    } break;
    default: e = 0.0; break;
    }
}

// force on atom i (and, for role-0 references, the energy) of every bond-list term that contains it

// the argument block and arrays of a term-parallel evaluation of `bs` at d_pos (bonded.hip), for a launch of another file that
// carries it (group.hip)
int amm_bonded_terms_work(amm_ctx *ctx, BondedSet *bs, const double *d_pos, BondedArgs *A, int *nterms, const int4 **gt_a,
                          const double4 **gt_q, double **tf, const int **list);
