// atomsmm_amd/csrc/pair_math.h -- per-pair energy / force of the AtomsMM pair families (device, fp64).
//
// Every family returns e = sign*E(r) and fr = sign*(-dE/dr)/r for mixed parameters
//   qq  = Kc*q_i*q_j,  sig = (sigma_i+sigma_j)/2,  eps4 = 4*sqrt(eps_i*eps_j)
// following the energy strings of the reference (file:line in /root/reference/src/atomsmm):
//   NEAR_NONE     forces.py:542-543   S*(4 eps ((s/r)^12-(s/r)^6) + Kc qq/r), S = 1+step(r-rs0) u^3 (15u-6u^2-10)
//   NEAR_SHIFT    forces.py:545-548   S*(V(r)-V(rc0)) term-wise
//   NEAR_FSWITCH  forces.py:550-563   force-switched potential; V'(r) = S(u) V'_LJC(r) (forces.py:628)
//   DAMPED        forces.py:448-455   SW*(LJ + erfc(alpha r) Kc qq/r), u = (r^d-rs^d)/(rc^d-rs^d)
//   NONBONDED     forces.py:134-190   S_b*LJ + Coulomb {plain | erfc | reaction field}
//   SOFTCORE      systems.py:266-272  S_b * 4 lambda eps (1-x)/x^2, x = (r/sigma)^6 + (1-lambda)/2, set 1 x set 2 only
//   LJ_VIRIAL     systems.py:894      S_b * 24 eps (2 (s/r)^12 - (s/r)^6)   (ComputingSystem: the virial as an energy)
#pragma once
#include "amm_ctx.h"
#include "erfcx_table.h"

// Contraction is decided per source expression (not by the optimiser across statements), so that the same pair
// evaluated by two different kernels -- alone, or next to the guest force of a shared list -- rounds identically.
#pragma clang fp contract(on)

__device__ __forceinline__ double amm_sw_S(double u) { return 1.0 + u * u * u * (15.0 * u - 6.0 * u * u - 10.0); }
__device__ __forceinline__ double amm_sw_dS(double u) {
    double w = u * (1.0 - u);
    return -30.0 * w * w;
}
__device__ __forceinline__ double amm_powi(double x, int n) {
    double r = 1.0;
    for (int k = 0; k < n; ++k) r *= x;
    return r;
}

// 1/sqrt(x) for x in the pair-distance range (no denormal/overflow scaling needed): hardware estimate
// (v_rsq_f64) + one cubic and one quadratic Newton step -> full fp64 precision in 9 ops instead of the ~30
// of the IEEE sqrt + divide sequence.
__device__ __forceinline__ double amm_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double e = fma(-x * y, y, 1.0);
    y = fma(y, e * fma(0.375, e, 0.5), y);
    e = fma(-x * y, y, 1.0);
    return fma(0.5 * y, e, y);
}

// exp(-y) for y >= 0: n = rint(-y log2 e), two-part ln2 reduction, degree-12 Taylor polynomial, ldexp.
__device__ __forceinline__ double amm_exp_neg(double y) {
    const double n = rint(-y * 1.4426950408889634074);
    double r = fma(n, -6.93147180369123816490e-01, -y);
    r = fma(n, -1.90821492927058770002e-10, r);
    double p = 1.0 / 479001600.0;
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}

// erfc(x) and exp(-x^2) for x >= 0, branch-free: erfc = exp(-x^2) * erfcx(x), erfcx from the piecewise
// degree-11 polynomials of erfcx_table.h staged in LDS (`tab`), max relative error ~2e-15.
__device__ __forceinline__ void amm_erfc_exp(double x, const double *tab, double &ec, double &ex) {
    int k = (int)(x * AMM_ERFCX_INVH);
    k = k > AMM_ERFCX_NI - 1 ? AMM_ERFCX_NI - 1 : k;
    const double t = x - ((double)k + 0.5) * AMM_ERFCX_H;
    const double *cf = tab + __mul24(k, AMM_ERFCX_NC);      // 24-bit multiply: full rate (v_mul_lo_u32 is quarter rate)
    double p = cf[AMM_ERFCX_NC - 1];
#pragma unroll
    for (int m = AMM_ERFCX_NC - 2; m >= 0; --m) p = fma(p, t, cf[m]);
    ex = amm_exp_neg(x * x);
    ec = ex * p;
}

// GROUPED: interaction-group forces (AMM_GROUP_LJ / AMM_GROUP_Q) are separate instantiations, so that the common path
// carries no per-pair flag tests or selects
template <int FAM, int CMODE, bool GUARD, bool EN, bool GROUPED = false>
__device__ __forceinline__ void amm_pair_math(const PairConsts &c, double r2, double qq, double sig, double eps4,
                                              double &e, double &fr, const double *tab = nullptr) {
    const double rinv = amm_rsqrt(r2);
    const double r = r2 * rinv;
    const double rinv2 = rinv * rinv;
    e = 0.0;
    fr = 0.0;
    bool in_group = true;
    if (GROUPED) {
        if (c.flags & AMM_GROUP_LJ) {      // wave-uniform flag: qq carries the product of the atoms' set codes
            in_group = qq == 2.0;
            qq = 0.0;
        }
        if (c.flags & AMM_GROUP_Q) {       // wave-uniform flag: sig carries the sum of the atoms' set codes
            in_group = sig == 3.0;
            sig = 1.0;
            eps4 = 0.0;
        }
    }
    if (GUARD) {
        if (!(c.rc0 - r >= 0.0)) return;   // step(rc0 - r), forces.py:661,714
    }
    const double s2 = sig * sig * rinv2;
    const double s6 = s2 * s2 * s2;
    const double s12 = s6 * s6;
    const double dlj_r = eps4 * (12.0 * s12 - 6.0 * s6) * rinv2;   // (-dV_LJ/dr)/r
    const double coul = qq * rinv;
    const double dcoul_r = coul * rinv2;                            // (-d(qq/r)/dr)/r
    if (FAM == AMM_NEAR_NONE || FAM == AMM_NEAR_SHIFT || FAM == AMM_NEAR_FSWITCH) {
        const double du = r - c.rs0;
        const double u = (du >= 0.0) ? du * c.inv_dr0 : 0.0;
        const double S = amm_sw_S(u);
        if (FAM == AMM_NEAR_FSWITCH) {
            fr = S * (dlj_r + dcoul_r);
            if (EN) {
                double f12 = 1.0, f6 = 1.0, f1 = 1.0;
                if (du >= 0.0) {
                    const double b = c.b, b2 = b * b, b3 = b2 * b;
                    const double R = u / b + 1.0;
                    const double u2 = u * u, u3 = u2 * u, u4 = u3 * u, u5 = u4 * u;
                    const double R2 = R * R, R3 = R2 * R, R6 = R3 * R3, R12 = R6 * R6;
                    f12 = 1.0 + ((6 * b2 - 21 * b + 28) * (b3 * (R12 - 1) - 12 * b2 * u - 66 * b * u2 - 220 * u3) / 462 +
                                 45 * (7 - 2 * b) * u4 / 14 - 72 * u5 / 7);
                    f6 = 1.0 + ((6 * b2 - 3 * b + 1) * (b3 * (R6 - 1) - 6 * b2 * u - 15 * b * u2 - 20 * u3) +
                                45 * (1 - 2 * b) * u4 - 36 * u5);
                    f1 = 1.0 + (5 * (b + 1) * (b + 1) * (6 * b3 * R * log(R) - 6 * b2 * u - 3 * b * u2 + u3) +
                                u4 * (3 * u - 5 * b - 10) / 2);
                }
                const double sc2 = sig * sig * c.inv_rc0_2, sc6 = sc2 * sc2 * sc2, sc12 = sc6 * sc6;
                e = eps4 * (f12 * s12 - f6 * s6) + qq * f1 * rinv;
                if (!(c.flags & AMM_NO_SHIFT)) e -= eps4 * (c.f12c * sc12 - c.f6c * sc6) + qq * c.f1c * c.inv_rc0;
            }
        } else {
            double V;
            if (FAM == AMM_NEAR_SHIFT) {
                const double sc2 = sig * sig * c.inv_rc0_2, sc6 = sc2 * sc2 * sc2, sc12 = sc6 * sc6;
                V = eps4 * (s12 - s6 - (sc12 - sc6)) + qq * (rinv - c.inv_rc0);
            } else {
                V = eps4 * (s12 - s6) + coul;
            }
            const double dSdr = amm_sw_dS(u) * c.inv_dr0;
            fr = S * (dlj_r + dcoul_r) - dSdr * V * rinv;
            if (EN) e = S * V;
        }
    } else if (FAM == AMM_DAMPED) {
        const double ar = c.alpha * r;
        double ec, ex;
        amm_erfc_exp(ar, tab, ec, ex);
        const double V = eps4 * (s12 - s6) + ec * coul;
        const double mdV_r = dlj_r + ec * dcoul_r + coul * c.two_alpha_over_sqrtpi * ex * rinv;
        // u = 0 below rswitch gives S = 1, dS = 0: no branch needed (step(r - rswitch), forces.py:454)
        const int d = (CMODE == 1) ? 1 : c.degree;          // CMODE 1: degree-1 specialisation (same values: x*1 is exact)
        const double rd1 = (CMODE == 1) ? 1.0 : amm_powi(r, d - 1);
        const double du = rd1 * r - c.rswitch_d;
        const double u = du >= 0.0 ? du * c.inv_sw_den : 0.0;
        const double S = amm_sw_S(u);
        const double dSdr = amm_sw_dS(u) * d * rd1 * c.inv_sw_den;
        fr = S * mdV_r - dSdr * V * rinv;
        if (EN) e = S * V;
    } else if (FAM == AMM_SOFTCORE) {
        // qq = code_i*code_j (Kc = 1): 2 for a (set 1, set 2) pair of the interaction group; lambda travels in alpha
        const bool member = (qq == 2.0) && (sig > 0.0);
        const double sg = member ? sig : 1.0;
        const double isg2 = 1.0 / (sg * sg);
        const double t2 = r2 * isg2, t6 = t2 * t2 * t2;                 // (r/sigma)^6
        const double x = t6 + 0.5 * (1.0 - c.alpha);
        const double ix = 1.0 / x, ix2 = ix * ix;
        const double le = c.alpha * eps4;                               // 4 lambda eps
        const double V = le * (1.0 - x) * ix2;
        const double mdV_r = 6.0 * le * (2.0 - x) * ix2 * ix * t6 * rinv2;      // (-dV/dr)/r
        double S = 1.0, dSdr = 0.0;
        if ((c.flags & AMM_SWITCH) && r > c.rswitch) {
            const double t = (r - c.rswitch) * c.inv_sw_dr;
            S = amm_sw_S(t);
            dSdr = amm_sw_dS(t) * c.inv_sw_dr;
        }
        fr = member ? S * mdV_r - dSdr * V * rinv : 0.0;
        if (EN) {
            // AMM_DERIV_LAMBDA: the "energy" output is dE/dlambda of the pair (deriv(energy, lambda), integrators.py:735):
            //   d/dlambda [4 lambda eps (1-x)/x^2] = 4 eps [(1-x)/x^2 - lambda (x-2)/(2 x^3)],  dx/dlambda = -1/2
            const double dVdl = eps4 * ((1.0 - x) * ix2 - 0.5 * c.alpha * (x - 2.0) * ix2 * ix);
            e = member ? S * ((c.flags & AMM_DERIV_LAMBDA) ? dVdl : V) : 0.0;
        }
    } else if (FAM == AMM_LJ_VIRIAL) {
        // W = -r dV_LJ/dr = 24 eps (2 s12 - s6) as an "energy" (systems.py:894); eps4 = 4 eps
        const double W = 6.0 * eps4 * (2.0 * s12 - s6);
        const double mdW_r = 6.0 * eps4 * (24.0 * s12 - 6.0 * s6) * rinv2;
        double S = 1.0, dSdr = 0.0;
        if ((c.flags & AMM_SWITCH) && r > c.rswitch) {
            const double t = (r - c.rswitch) * c.inv_sw_dr;
            S = amm_sw_S(t);
            dSdr = amm_sw_dS(t) * c.inv_sw_dr;
        }
        fr = S * mdW_r - dSdr * W * rinv;
        if (EN) e = S * W;
    } else {   // AMM_NONBONDED
        double S = 1.0, dSdr = 0.0;
        if ((c.flags & AMM_SWITCH) && r > c.rswitch) {
            const double t = (r - c.rswitch) * c.inv_sw_dr;
            S = amm_sw_S(t);
            dSdr = amm_sw_dS(t) * c.inv_sw_dr;
        }
        const double lj = eps4 * (s12 - s6);
        fr = S * dlj_r - dSdr * lj * rinv;
        if (EN) e = S * lj;
        if (CMODE == 1) {
            const double ar = c.alpha * r;
            double ec, ex;
            amm_erfc_exp(ar, tab, ec, ex);
            fr += ec * dcoul_r + coul * c.two_alpha_over_sqrtpi * ex * rinv;
            if (EN) e += ec * coul;
        } else if (CMODE == 2) {
            fr += qq * (rinv2 * rinv - 2.0 * c.krf);
            if (EN) e += qq * (rinv + c.krf * r2 - c.crf);
        } else {
            fr += dcoul_r;
            if (EN) e += coul;
        }
    }
    fr = (!GROUPED || in_group) ? fr * c.sign : 0.0;
    if (EN) e = (!GROUPED || in_group) ? e * c.sign : 0.0;
}

// Runtime-dispatched variant for the bond-list kernels (not hot: O(#exceptions)).
__device__ __forceinline__ void amm_pair_math_rt(const PairConsts &c, double r2, double qq, double sig, double eps4,
                                                 double &e, double &fr) {
    const bool g = (c.flags & AMM_GUARD_RC0) != 0;
    switch (c.family) {
    case AMM_NEAR_NONE:
        if (g) amm_pair_math<AMM_NEAR_NONE, 0, true, true>(c, r2, qq, sig, eps4, e, fr);
        else amm_pair_math<AMM_NEAR_NONE, 0, false, true>(c, r2, qq, sig, eps4, e, fr);
        break;
    case AMM_NEAR_SHIFT:
        if (g) amm_pair_math<AMM_NEAR_SHIFT, 0, true, true>(c, r2, qq, sig, eps4, e, fr);
        else amm_pair_math<AMM_NEAR_SHIFT, 0, false, true>(c, r2, qq, sig, eps4, e, fr);
        break;
    default:
        if (g) amm_pair_math<AMM_NEAR_FSWITCH, 0, true, true>(c, r2, qq, sig, eps4, e, fr);
        else amm_pair_math<AMM_NEAR_FSWITCH, 0, false, true>(c, r2, qq, sig, eps4, e, fr);
        break;
    }
}

__device__ __forceinline__ double amm_min_image(double d, double L, double invL) { return d - L * rint(d * invL); }
