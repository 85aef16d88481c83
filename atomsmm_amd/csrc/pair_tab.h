// atomsmm_amd/csrc/pair_tab.h -- radial Coulomb tables and LJ-only force terms of the force-only traversal (gfx950, fp64).
//
// Every pair family of the reference is  E = E_LJ(r; sigma, eps) + qq * E_C(r)  (forces.py:406, 448-455, 539-567,
// 134-190): the Coulomb part depends on the pair only through the factor qq = Kc q_i q_j.  Its force over r,
//     B(r^2) = (-dE_C/dr) / r        (switching functions, erfc damping, reaction field ... all inside),
// is therefore ONE radial function per pair force.  The force-only kernels read it from a table instead of evaluating
// erfc / exp / 1/sqrt / switching polynomials per pair:
//   * abscissa w = r^2 * scale with scale = 1 / r_kink^2 (r_kink = the radius where the switching function starts, so
//     that the C2 kink of the reference's S(u) is an interval boundary; rc for kink-free families);
//   * intervals = octaves of w split in 128 (the exponent and the top 7 mantissa bits of w ARE the interval index: two
//     integer instructions, no divide, relative width 2^-7 whatever r is -- the 1/r^3 growth at short range needs that);
//   * per interval a degree-5 polynomial in t = w - centre (Chebyshev-node interpolation of the exact long-double function,
//     relative error < 1e-15 -- checked at build time), 6 coefficients = 48 B: three ds_read_b128 whose bank quads
//     (3 idx + k) mod 16 spread over all 16 quads (no padding needed);
//   * pairs closer than the table's lower end (r < r_kink / 8 ... / 11) take the analytic path under a wave-uniform branch.
// With the table the water hydrogens (eps = 0) need no 1/r at all.  The Lennard-Jones part stays analytic
// (amm_lj_force below: the same expressions as amm_pair_math with qq = 0) and is skipped by wavefronts whose rows all have
// eps_i = 0.
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

#include <cstring>

#include "amm_ctx.h"

#define AMM_TAB_SHIFT 13                 // hi32(w) >> 13 = exponent | top 7 mantissa bits
#define AMM_TAB_PER_OCTAVE 128
#define AMM_TAB_STRIDE 48                // bytes per interval: c0..c5
#define AMM_TAB_MAX_ERROR 1e-13          // largest relative interpolation error a table may show at its check points

// ------------------------------------------------------------------------------------------------ host: build
// exact radial Coulomb force over r, per unit qq, in long double (the formulas of amm_pair_math, Coulomb terms only)
static inline long double amm_coul_radial_exact(const PairConsts &c, long double r) {
    const long double rinv = 1.0L / r, rinv2 = rinv * rinv;
    auto S = [](long double u) { return 1.0L + u * u * u * (15.0L * u - 6.0L * u * u - 10.0L); };
    auto dS = [](long double u) { const long double w = u * (1.0L - u); return -30.0L * w * w; };
    switch (c.family) {
    case AMM_NEAR_NONE:
    case AMM_NEAR_SHIFT:
    case AMM_NEAR_FSWITCH: {
        const long double du = r - (long double)c.rs0;
        const long double u = du >= 0 ? du * (long double)c.inv_dr0 : 0.0L;
        if (c.family == AMM_NEAR_FSWITCH) return S(u) * rinv2 * rinv;
        const long double V = c.family == AMM_NEAR_SHIFT ? rinv - (long double)c.inv_rc0 : rinv;
        return S(u) * rinv2 * rinv - dS(u) * (long double)c.inv_dr0 * V * rinv;
    }
    case AMM_DAMPED: {
        const long double ar = (long double)c.alpha * r;
        const long double ec = erfcl(ar), ex = expl(-ar * ar);
        const long double V = ec * rinv;
        const long double mdV_r = ec * rinv2 * rinv + (long double)c.two_alpha_over_sqrtpi * ex * rinv2;
        const int d = c.degree;
        const long double rd1 = powl(r, d - 1);
        const long double du = rd1 * r - (long double)c.rswitch_d;
        const long double u = du >= 0 ? du * (long double)c.inv_sw_den : 0.0L;
        return S(u) * mdV_r - dS(u) * d * rd1 * (long double)c.inv_sw_den * V * rinv;
    }
    case AMM_NONBONDED: {
        if (c.cmode == 1) {
            const long double ar = (long double)c.alpha * r;
            return erfcl(ar) * rinv2 * rinv + (long double)c.two_alpha_over_sqrtpi * expl(-ar * ar) * rinv2;
        }
        if (c.cmode == 2) return rinv2 * rinv - 2.0L * (long double)c.krf;
        return rinv2 * rinv;
    }
    default: return 0.0L;
    }
}

static inline bool amm_family_has_table(int family) {
    return family == AMM_NEAR_NONE || family == AMM_NEAR_SHIFT || family == AMM_NEAR_FSWITCH || family == AMM_DAMPED ||
           family == AMM_NONBONDED;
}

// Interval j of a table: zone A (w < 1, below the kink) has 128 intervals per octave, zone B (w >= 1: the switching region,
// where the reference's quintic S(u) varies on the scale rc - rs, far shorter than r) has 128 << fine per octave.
static inline void amm_tab_interval(const PairTab &T, int j, long double &lo, long double &width) {
    unsigned I;
    int nb;
    if (j < T.nA) {
        I = (unsigned)T.baseA + (unsigned)j;
        nb = 20 - AMM_TAB_SHIFT;
    } else {
        I = (unsigned)(T.rawB_minus_nA + j);
        nb = 20 - T.shiftB;
    }
    const int e = (int)(I >> nb) - 1023;
    const unsigned m = I & ((1u << nb) - 1u);
    lo = ldexpl(1.0L + (long double)m / (long double)(1u << nb), e);
    width = ldexpl(1.0L / (long double)(1u << nb), e);
}

// Degree-5 Chebyshev-node interpolation of `exact(r)` on the intervals [j0, T.nint) of T; coefficients of interval j at
// coef[(j - j0) * 6 ..].  Returns the largest relative error met between the nodes (in zone B only if `zone_b_err` is given: there).
template <class F>
static inline double amm_fit_radial_table(const PairConsts &pc, const PairTab &T, std::vector<double> &coef, F exact, int j0 = 0,
                                          double *zone_b_err = nullptr) {
    // a force guarded by step(rc0 - r) is never looked up beyond rc0, whatever its nominal cutoff (the discount of FarNonbondedForce)
    const double reach = ((pc.flags & AMM_GUARD_RC0) && pc.rc0 > 0.0) ? std::min(pc.rc, pc.rc0) : pc.rc;
    coef.assign((size_t)(T.nint - j0) * 6, 0.0);
    long double nodes[6];
    for (int k = 0; k < 6; ++k) nodes[k] = cosl((2 * k + 1) * 3.14159265358979323846264338327950288L / 12.0L);
    double worst = 0.0, worst_b = 0.0;
    for (int j = j0; j < T.nint; ++j) {
        long double lo, width;
        amm_tab_interval(T, j, lo, width);
        const long double h = 0.5L * width, centre = lo + h;
        double *cj = &coef[(size_t)(j - j0) * 6];
        // interpolate at the Chebyshev nodes: solve V a = f with V[k][m] = s_k^m (6 x 6, long double, partial pivoting)
        long double A[6][7];
        for (int k = 0; k < 6; ++k) {
            long double p = 1.0L;
            for (int q = 0; q < 6; ++q) {
                A[k][q] = p;
                p *= nodes[k];
            }
            A[k][6] = exact(sqrtl((centre + h * nodes[k]) / (long double)T.scale));
        }
        for (int col = 0; col < 6; ++col) {
            int piv = col;
            for (int r = col + 1; r < 6; ++r)
                if (fabsl(A[r][col]) > fabsl(A[piv][col])) piv = r;
            for (int q = 0; q < 7; ++q) std::swap(A[col][q], A[piv][q]);
            for (int r = col + 1; r < 6; ++r) {
                const long double f = A[r][col] / A[col][col];
                for (int q = col; q < 7; ++q) A[r][q] -= f * A[col][q];
            }
        }
        long double a[6];
        for (int r = 5; r >= 0; --r) {
            long double sum = A[r][6];
            for (int q = r + 1; q < 6; ++q) sum -= A[r][q] * a[q];
            a[r] = sum / A[r][r];
        }
        long double hp = 1.0L;
        for (int q = 0; q < 6; ++q) {
            cj[q] = (double)(a[q] / hp);
            hp *= h;
        }
        // check between the nodes, with the double coefficients and double Horner arithmetic
        for (int k = 0; k < 7; ++k) {
            const double t = (double)(h * (-1.0L + k / 3.0L));
            double p = cj[5];
            for (int q = 4; q >= 0; --q) p = fma(p, t, cj[q]);
            const long double r = sqrtl((centre + (long double)t) / (long double)T.scale);
            if (r >= (long double)reach) continue;
            const long double ex = exact(r);
            // where a switching function takes the force through zero the error is measured against the unswitched 1/r^3
            const long double denom = fmaxl(fabsl(ex), 1e-3L / (r * r * r));
            const double e = (double)(fabsl((long double)p - ex) / denom);
            worst = fmax(worst, e);
            if (j >= T.nA) worst_b = fmax(worst_b, e);
        }
    }
    if (zone_b_err) *zone_b_err = worst_b;
    return worst;
}

// exact Lennard-Jones force over r of one pair (sig, eps4 = 4 eps) in long double: the formulas of amm_lj_force below
static inline long double amm_lj_radial_exact(const PairConsts &c, long double sig, long double eps4, long double r) {
    const long double rinv = 1.0L / r, rinv2 = rinv * rinv;
    const long double s2 = sig * sig * rinv2, s6 = s2 * s2 * s2, s12 = s6 * s6;
    const long double dlj_r = eps4 * (12.0L * s12 - 6.0L * s6) * rinv2;
    auto S = [](long double u) { return 1.0L + u * u * u * (15.0L * u - 6.0L * u * u - 10.0L); };
    auto dS = [](long double u) { const long double w = u * (1.0L - u); return -30.0L * w * w; };
    switch (c.family) {
    case AMM_NEAR_NONE:
    case AMM_NEAR_SHIFT:
    case AMM_NEAR_FSWITCH: {
        const long double du = r - (long double)c.rs0;
        const long double u = du >= 0 ? du * (long double)c.inv_dr0 : 0.0L;
        if (c.family == AMM_NEAR_FSWITCH) return S(u) * dlj_r;
        long double V = eps4 * (s12 - s6);
        if (c.family == AMM_NEAR_SHIFT) {
            const long double sc2 = sig * sig * (long double)c.inv_rc0_2, sc6 = sc2 * sc2 * sc2, sc12 = sc6 * sc6;
            V = eps4 * (s12 - s6 - (sc12 - sc6));
        }
        return S(u) * dlj_r - dS(u) * (long double)c.inv_dr0 * V * rinv;
    }
    case AMM_DAMPED: {
        const long double V = eps4 * (s12 - s6);
        const int d = c.degree;
        const long double rd1 = powl(r, d - 1);
        const long double du = rd1 * r - (long double)c.rswitch_d;
        const long double u = du >= 0 ? du * (long double)c.inv_sw_den : 0.0L;
        return S(u) * dlj_r - dS(u) * d * rd1 * (long double)c.inv_sw_den * V * rinv;
    }
    case AMM_NONBONDED: {
        long double Sv = 1.0L, dSdr = 0.0L;
        if ((c.flags & AMM_SWITCH) && r > (long double)c.rswitch) {
            const long double t = (r - (long double)c.rswitch) * (long double)c.inv_sw_dr;
            Sv = S(t);
            dSdr = dS(t) * (long double)c.inv_sw_dr;
        }
        return Sv * dlj_r - dSdr * (eps4 * (s12 - s6)) * rinv;
    }
    default: return 0.0L;
    }
}

// The site-site table of a force whose Lennard-Jones sites all carry ONE (sigma, eps, q) -- water: the oxygens.  For a pair
// of two sites the whole force over r is a radial function too,
//     F/r = qq B_C(r^2) + B_LJ(r^2) = qq B'(r^2),   B' = B_C + B_LJ / QQ,   QQ = Kc q_site^2 = that pair's qq,
// so the force-only kernels read B' instead of B_C for such a pair -- same interval, same Horner chain, same multiply by qq --
// and need no Lennard-Jones arithmetic (1/r, two Newton steps, the switch in r ...) at all.  B' is tabulated on the SAME
// intervals as B_C, from the interval that holds 0.7 sigma upwards (4 eps ((1/0.7)^12 - (1/0.7)^6) = 255 eps: sites do not get
// closer; if they do, the pair takes the analytic path like any pair below a table).  Its error is dominated by the r^-14 wall
// (1.0-1.4e-13 relative at 128 intervals per octave, whatever r): bound AMM_TAB_SS_MAX_ERROR.
#define AMM_TAB_SS_MAX_ERROR 3e-13
struct SiteTable {
    bool want = false;                 // in: build it
    double sig = 0, eps4 = 0, QQ = 0;  // in: sigma, 4 eps of a site pair, Kc q_site^2
    std::vector<double> coef;          // out: (nint - first) x 6
    double error = 0;                  // out: largest relative error met
};

// Fills pc.tab and `coef` (nint x 6 doubles) -- and the site-site table if one is wanted.  Returns the largest relative
// interpolation error of the Coulomb table met at the check points.
static inline double amm_build_coulomb_table(PairConsts &pc, std::vector<double> &coef, SiteTable *ss = nullptr) {
    memset(&pc.tab, 0, sizeof(pc.tab));
    coef.clear();
    if (ss) ss->coef.clear();
    if (!amm_family_has_table(pc.family)) return 0.0;
    const bool near_family = pc.family == AMM_NEAR_NONE || pc.family == AMM_NEAR_SHIFT || pc.family == AMM_NEAR_FSWITCH;
    double rkink = pc.rc;
    if (pc.family == AMM_DAMPED && pc.rswitch > 0 && pc.rswitch < pc.rc) rkink = pc.rswitch;
    if (near_family && pc.rs0 > 0 && pc.rs0 < pc.rc) rkink = pc.rs0;
    // (the built-in switch of the NonbondedForce acts on the Lennard-Jones term only: its start is a kink of the site-site table)
    if (pc.family == AMM_NONBONDED && (pc.flags & AMM_SWITCH) && pc.rswitch > 0 && pc.rswitch < pc.rc) rkink = pc.rswitch;
    // zone A reaches down to r <= 0.115 nm (closer pairs take the analytic path): 5 octaves of r^2 for rs0 = 0.5, 6 for 0.9
    const int octaves_below = std::max(3, std::min(10, (int)std::ceil(2.0 * std::log2(rkink / 0.115))));
    auto raw_index = [](double w, int shift) {
        unsigned long long bits;
        memcpy(&bits, &w, 8);
        return (unsigned)(bits >> (32 + shift));
    };
    PairTab T;
    memset(&T, 0, sizeof(T));
    T.scale = 1.0 / (rkink * rkink);
    T.r2min = ldexp(1.0, -octaves_below) / T.scale * (1.0 + 1e-12);
    T.baseA = (int)raw_index(ldexp(1.0, -octaves_below), AMM_TAB_SHIFT);
    T.nA = octaves_below * AMM_TAB_PER_OCTAVE;
    // (a force guarded by step(rc0 - r) is looked up to rc0 only: its table ends there, whatever the nominal cutoff)
    const double reach = ((pc.flags & AMM_GUARD_RC0) && pc.rc0 > 0.0) ? std::min(pc.rc, pc.rc0) : pc.rc;
    const double wtop = reach * reach * T.scale * (1.0 + 1e-12);
    const bool with_ss = ss && ss->want && ss->QQ != 0.0 && ss->eps4 > 0.0 && ss->sig > 0.0;
    if (with_ss) {
        // first interval of the site-site table: the one that holds (0.7 sigma)^2, if zone A reaches that far down
        const double w0 = 0.49 * ss->sig * ss->sig * T.scale;
        const int first = (int)raw_index(w0, AMM_TAB_SHIFT) - T.baseA;
        T.ss_first = std::max(0, std::min(first, T.nA));
        long double lo, width;
        PairTab tmp = T;
        tmp.nint = T.nA + 1;
        amm_tab_interval(tmp, T.ss_first, lo, width);
        T.ss_r2min = (double)lo / T.scale * (1.0 + 1e-12);
        if (!(w0 < 1.0)) T.ss_first = -1;          // sites larger than the kink radius: no table
    } else {
        T.ss_first = -1;
    }
    double worst = 0.0;
    for (int fine = 0; fine <= 6; ++fine) {        // refine the switching zone until the table is as good as the arithmetic
        T.shiftB = AMM_TAB_SHIFT - fine;
        T.halfB = 1 << (T.shiftB - 1);
        const unsigned rawB = raw_index(1.0, T.shiftB);
        T.rawB_minus_nA = (int)rawB - T.nA;
        T.nint = T.nA + (int)(raw_index(wtop, T.shiftB) - rawB) + 2;   // one interval beyond the cutoff (r2 < rc2 may round up)
        std::vector<double> trial, trial_ss;
        const double err = amm_fit_radial_table(pc, T, trial, [&](long double r) { return amm_coul_radial_exact(pc, r); });
        double err_ss = 0.0, err_ss_b = 0.0;
        if (T.ss_first >= 0) {
            const long double sg = ss->sig, e4 = ss->eps4, iqq = 1.0L / (long double)ss->QQ;
            err_ss = amm_fit_radial_table(pc, T, trial_ss,
                                          [&](long double r) { return amm_coul_radial_exact(pc, r) + amm_lj_radial_exact(pc, sg, e4, r) * iqq; },
                                          T.ss_first, &err_ss_b);
        }
        const size_t bytes = ((size_t)T.nint + (T.ss_first >= 0 ? (size_t)(T.nint - T.ss_first) : 0)) * AMM_TAB_STRIDE;
        if (fine > 0 && bytes > (T.ss_first >= 0 ? 112 : 72) * 1024 && !coef.empty()) break;      // LDS budget: keep the previous one
        coef.swap(trial);
        if (ss) {
            ss->coef.swap(trial_ss);
            ss->error = err_ss;
        }
        pc.tab = T;
        worst = err;
        // (the site-site table's error in zone A is that of the r^-14 wall and does not move with the refinement of zone B)
        if (err < 1e-14 && err_ss_b < 1e-13) break;
    }
    return worst;
}

// ------------------------------------------------------------------------------------------------ device
#if defined(__HIPCC__)
#include "pair_math.h"
__device__ __forceinline__ double amm_tab_eval(const char *lds_tab, const PairTab &T, double r2) {
    const double w = r2 * T.scale;
    const unsigned hi = (unsigned)__double2hiint(w);
    const bool above = hi >= 0x3FF00000u;                         // w >= 1: the switching zone (finer intervals)
    const unsigned sh = above ? (unsigned)T.shiftB : (unsigned)AMM_TAB_SHIFT;
    const unsigned raw = hi >> sh;                                // exponent | top mantissa bits = interval
    unsigned idx = raw - (unsigned)(above ? T.rawB_minus_nA : T.baseA);   // below the table: wraps to a huge value -> clamped
    idx = min(idx, (unsigned)(T.nint - 1));                       // (masked, or redone analytically by the caller); beyond rc: clamped, masked
    const double centre = __hiloint2double((int)((raw << sh) | (unsigned)(above ? T.halfB : (1 << (AMM_TAB_SHIFT - 1)))), 0);
    const double t = w - centre;                                  // exact (Sterbenz)
#ifdef AMM_EXP_NOLDS
    const double2 c01 = make_double2(1.0 + idx, 0.5), c23 = make_double2(0.25, 0.125), c45 = make_double2(0.0625, 0.03);
#else
    const double2 *cf = reinterpret_cast<const double2 *>(lds_tab + __umul24(idx, AMM_TAB_STRIDE));
    const double2 c01 = cf[0], c23 = cf[1], c45 = cf[2];
#endif
    double p = fma(c45.y, t, c45.x);
    p = fma(p, t, c23.y);
    p = fma(p, t, c23.x);
    p = fma(p, t, c01.y);
    return fma(p, t, c01.x);
}

// the same look-up in two halves, for kernels that pin the LDS reads of several pairs ahead of the arithmetic (cluster.hip)
struct TabLookup {
    double2 c01, c23, c45;
    double t;
};
// (`extra`: bytes added to the interval's address -- the way from a force's Coulomb table to its site-site table, or 0)
__device__ __forceinline__ TabLookup amm_tab_fetch(const char *lds_tab, const PairTab &T, double r2, unsigned extra = 0u) {
    const double w = r2 * T.scale;
    const unsigned hi = (unsigned)__double2hiint(w);
    const bool above = hi >= 0x3FF00000u;
    const unsigned sh = above ? (unsigned)T.shiftB : (unsigned)AMM_TAB_SHIFT;
    const unsigned raw = hi >> sh;
    unsigned idx = raw - (unsigned)(above ? T.rawB_minus_nA : T.baseA);
    idx = min(idx, (unsigned)(T.nint - 1));
    const double centre = __hiloint2double((int)((raw << sh) | (unsigned)(above ? T.halfB : (1 << (AMM_TAB_SHIFT - 1)))), 0);
    TabLookup L;
    L.t = w - centre;
#if defined(AMM_EXP_TAB_BCAST)          // measurement only: every lane reads interval 0 (no bank conflicts; wrong forces)
    idx &= 0u;
#endif
    const double2 *cf = reinterpret_cast<const double2 *>(lds_tab + (__umul24(idx, AMM_TAB_STRIDE) + extra));
    L.c01 = cf[0];
    L.c23 = cf[1];
#if defined(AMM_EXP_TAB_BYTES) && AMM_EXP_TAB_BYTES == 40          // measurement only (wrong forces): what fewer LDS bytes per pair would buy
    L.c45 = make_double2(*reinterpret_cast<const double *>(cf + 2), 0.0);
#elif defined(AMM_EXP_TAB_BYTES) && AMM_EXP_TAB_BYTES == 32
    L.c45 = make_double2(0.0, 0.0);
#else
    L.c45 = cf[2];
#endif
    return L;
}
__device__ __forceinline__ double amm_tab_horner(const TabLookup &L) {
    double p = fma(L.c45.y, L.t, L.c45.x);
    p = fma(p, L.t, L.c23.y);
    p = fma(p, L.t, L.c23.x);
    p = fma(p, L.t, L.c01.y);
    return fma(p, L.t, L.c01.x);
}

// Lennard-Jones part of (-dE/dr)/r for mixed sig = (sigma_i + sigma_j)/2, eps4 = 4 sqrt(eps_i eps_j): the qq = 0 case of
// amm_pair_math, same expressions and order of operations.  The pieces two forces of a shared list have in common
// (1/r, (sigma/r)^6, (sigma/r)^12, the unswitched force) are formed once.
// a + b rounded on its own, never contracted with a product that made a or b (HIP's __dadd_rn is a plain '+'): the force of a
// pair must come out bit for bit the same from a stand-alone launch and from the guest part of a fused pass
__device__ __forceinline__ double amm_sum_unfused(double a, double b) {
#pragma clang fp contract(off)
    return a + b;
}
struct LJCommon {
    double rinv, r, s6, s12, dlj_r;
};
__device__ __forceinline__ LJCommon amm_lj_common(double r2, double sig, double eps4) {
    LJCommon L;
    L.rinv = amm_rsqrt(r2);
    L.r = r2 * L.rinv;
    const double rinv2 = L.rinv * L.rinv;
    const double s2 = sig * sig * rinv2;
    L.s6 = s2 * s2 * s2;
    L.s12 = L.s6 * L.s6;
    L.dlj_r = eps4 * (12.0 * L.s12 - 6.0 * L.s6) * rinv2;
    return L;
}
template <int FAM, int CMODE>
__device__ __forceinline__ double amm_lj_force(const PairConsts &c, const LJCommon &L, double sig, double eps4) {
    if (FAM == AMM_NEAR_NONE || FAM == AMM_NEAR_SHIFT || FAM == AMM_NEAR_FSWITCH) {
        const double du = L.r - c.rs0;
        const double u = (du >= 0.0) ? du * c.inv_dr0 : 0.0;
        const double S = amm_sw_S(u);
        if (FAM == AMM_NEAR_FSWITCH) return S * L.dlj_r;
        double V;
        if (FAM == AMM_NEAR_SHIFT) {
            const double sc2 = sig * sig * c.inv_rc0_2, sc6 = sc2 * sc2 * sc2, sc12 = sc6 * sc6;
            V = eps4 * (L.s12 - L.s6 - (sc12 - sc6));
        } else {
            V = eps4 * (L.s12 - L.s6);
        }
        return S * L.dlj_r - amm_sw_dS(u) * c.inv_dr0 * V * L.rinv;
    } else if (FAM == AMM_DAMPED) {
        const double V = eps4 * (L.s12 - L.s6);
        const int d = (CMODE == 1) ? 1 : c.degree;
        const double rd1 = (CMODE == 1) ? 1.0 : amm_powi(L.r, d - 1);
        const double du = rd1 * L.r - c.rswitch_d;
        const double u = du >= 0.0 ? du * c.inv_sw_den : 0.0;
        return amm_sw_S(u) * L.dlj_r - amm_sw_dS(u) * d * rd1 * c.inv_sw_den * V * L.rinv;
    } else {   // AMM_NONBONDED: the built-in switch multiplies the LJ term only
        double S = 1.0, dSdr = 0.0;
        if ((c.flags & AMM_SWITCH) && L.r > c.rswitch) {
            const double t = (L.r - c.rswitch) * c.inv_sw_dr;
            S = amm_sw_S(t);
            dSdr = amm_sw_dS(t) * c.inv_sw_dr;
        }
        return S * L.dlj_r - dSdr * (eps4 * (L.s12 - L.s6)) * L.rinv;
    }
}
#endif
