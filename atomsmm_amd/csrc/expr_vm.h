// atomsmm_amd/csrc/expr_vm.h -- the per-DOF expression interpreter (device side), shared by k_expr (expr.hip) and the
// inner-loop kernel (bonded.hip), which runs bath steps of a RESPA loop in place.  See expr.hip for the design notes.
#pragma once
#include "amm_ctx.h"

#define AMM_EXPR_MAXCODE 256
#define AMM_EXPR_MAXCONST 48
#define AMM_EXPR_MAXGLOBAL 48
#define AMM_EXPR_STACK 24
#define AMM_EXPR_LOCALS 16

enum {
    X_CONST = 0, X_GLOBAL = 1, X_BUF = 2, X_MASS = 3, X_GAUSS = 4, X_UNIFORM = 5, X_LOAD = 6, X_STORE = 7, X_DEVG = 8, X_OUT = 9, X_HORNER = 43,
    X_ADD = 10, X_SUB = 11, X_MUL = 12, X_DIV = 13, X_NEG = 14, X_POW = 15, X_POWI = 16,
    X_SQRT = 20, X_EXP = 21, X_LOG = 22, X_SIN = 23, X_COS = 24, X_TAN = 25, X_ASIN = 26, X_ACOS = 27, X_ATAN = 28,
    X_SINH = 29, X_COSH = 30, X_TANH = 31, X_ERF = 32, X_ERFC = 33, X_ABS = 34, X_FLOOR = 35, X_CEIL = 36,
    X_STEP = 37, X_DELTA = 38, X_MIN = 39, X_MAX = 40, X_SELECT = 41, X_ATAN2 = 42
};

struct ExprProg {
    int ncode;
    int code[AMM_EXPR_MAXCODE];           // opcode | arg << 8
    double consts[AMM_EXPR_MAXCONST];
    double globals[AMM_EXPR_MAXGLOBAL];
    const double *bufs[AMM_MAX_SLOTS];     // per-DOF buffers by slot ([n][3])
    const double *mass;                    // [n]
    unsigned long long seed, counter;
};

__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned *out) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// two uniforms in (0,1) with 53 random bits each
__device__ __forceinline__ void amm_uniforms(unsigned long long seed, unsigned long long counter, unsigned dof, unsigned occurrence,
                                             double &u1, double &u2) {
    unsigned r[4];
    philox4x32_10(dof, occurrence, (unsigned)counter, (unsigned)(counter >> 32), (unsigned)seed, (unsigned)(seed >> 32), r);
    const unsigned long long a = ((unsigned long long)r[0] << 21) ^ (r[1] >> 11), b = ((unsigned long long)r[2] << 21) ^ (r[3] >> 11);
    u1 = ((double)(a & 0x1FFFFFFFFFFFFFull) + 0.5) * (1.0 / 9007199254740992.0);
    u2 = ((double)(b & 0x1FFFFFFFFFFFFFull) + 0.5) * (1.0 / 9007199254740992.0);
}
__device__ __forceinline__ void expr_uniforms(const ExprProg &P, unsigned dof, unsigned occurrence, double &u1, double &u2) {
    amm_uniforms(P.seed, P.counter, dof, occurrence, u1, u2);
}
// the `gaussian` of degree of freedom `dof` in launch `counter` (Box-Muller)
__device__ __forceinline__ double amm_gaussian(unsigned long long seed, unsigned long long counter, unsigned dof) {
    double u1, u2;
    amm_uniforms(seed, counter, dof, 0u, u1, u2);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925 * u2);
}
// Ornstein-Uhlenbeck step  v <- z v + sqrt(kT (1 - z z)/m) g  in the operation order of the reference's expression
// `z*v + sqrt(kT*(1 - z*z)/mass)*gaussian` (propagators.py:727), no contraction: the stand-alone kernel and the
// inner-loop kernel round identically
__device__ __forceinline__ double amm_ou_step(double v, double m, double z, double kT, double g) {
#pragma clang fp contract(off)
    const double zz = z * z;
    const double var = kT * (1.0 - zz);
    const double amp = sqrt(var / m);
    const double zv = z * v;
    const double noise = amp * g;
    return zv + noise;
}

// Nose-Hoover-Langevin step of one DOF, in the operation order of the reference's three expressions
// (`v*exp(-(h)*v2)`, `z*v2 + sqrt(kT*(1 - z*z)/mass)*gaussian + force*(1 - z)/(mass*friction)` with force = m*v^2 - kT and
// mass = Q2, then the scaling again): shared by the stand-alone kernel and the inner-loop kernel
__device__ __forceinline__ void amm_nhl_step(double &v, double &w, double m, double h, double z, double kT, double Q, double friction,
                                             double g) {
#pragma clang fp contract(off)
    v = v * exp(-h * w);
    const double force = m * v * v - kT;
    const double amp = sqrt(kT * (1.0 - z * z) / Q);
    w = z * w + amp * g + force * (1.0 - z) / (Q * friction);
    v = v * exp(-h * w);
}

// Stochastic-isokinetic pieces (SIN(R), L = 1), in the operation order of the reference's expressions
//   `v*cosh(z) + sqrt(LkT/m)*sinh(z); z = (c*dt)*(F)/sqrt(m*LkT)` ; `H <- sqrt(LkT/(m*v^2 + 0.5*Q1*(v1_0^2)))` ; `v <- H*v` ;
//   `v1_0 <- H*v1_0` (propagators.py:300-355), shared by the stand-alone kernels and the inner-loop kernel
__device__ __forceinline__ void amm_iso_rescale(double &v, double &v1, double m, double LkT, double Q1) {
#pragma clang fp contract(off)
    const double H = sqrt(LkT / (m * (v * v) + (0.5 * Q1) * (v1 * v1)));
    v = H * v;
    v1 = H * v1;
}
__device__ __forceinline__ void amm_iso_kick(double &v, double &v1, double F, double m, double coef, double LkT, double Q1) {
#pragma clang fp contract(off)
    const double z = (coef * F) / sqrt(m * LkT);
    v = v * cosh(z) + sqrt(LkT / m) * sinh(z);
    amm_iso_rescale(v, v1, m, LkT, Q1);
}
__device__ __forceinline__ void amm_sin_bath_step(double &v, double &v1, double &v2, double m, double h, double z, double kT, double Q2,
                                                  double friction, double Q1, double LkT, double g) {
#pragma clang fp contract(off)
    v1 = v1 * exp(-h * v2);
    amm_iso_rescale(v, v1, m, LkT, Q1);
    const double force = Q1 * (v1 * v1) - kT;
    v2 = z * v2 + sqrt(kT * (1.0 - z * z) / Q2) * g + force * (1.0 - z) / (Q2 * friction);
    v1 = v1 * exp(-h * v2);
    amm_iso_rescale(v, v1, m, LkT, Q1);
}

// Scalar programs (amm_expr_eval_scalar): a sequence of assignments  scalars[dst] <- expression  over constants and other scalars
// (X_DEVG), each closed by X_OUT dst.  One thread runs it; code, constants, stack and locals sit in LDS (a lone thread waits for
// every access: from the launch argument and private memory the same interpreter took ~0.6 us per word).
#define AMM_SCALAR_MAXCODE 640
#define AMM_SCALAR_MAXCONST 96
struct ScalarProg {
    int ncode, nconst;
    int code[AMM_SCALAR_MAXCODE];
    double consts[AMM_SCALAR_MAXCONST];
};

// one copy per kernel, called (not inlined): the interpreter is ~2 k instructions
static __device__ __noinline__ double expr_run(const ExprProg &P, int dof) {
    double st[AMM_EXPR_STACK], loc[AMM_EXPR_LOCALS];
    int sp = 0;
    for (int pc = 0; pc < P.ncode; ++pc) {
        const int word = P.code[pc], op = word & 0xff, arg = word >> 8;
        switch (op) {
        case X_CONST: st[sp++] = P.consts[arg]; break;
        case X_GLOBAL: st[sp++] = P.globals[arg]; break;
        case X_BUF: st[sp++] = P.bufs[arg][dof]; break;
        case X_MASS: st[sp++] = P.mass[dof / 3]; break;
        case X_GAUSS: {
            st[sp++] = amm_gaussian(P.seed, P.counter, (unsigned)dof);      // one draw per evaluation and DOF, as in OpenMM
        } break;
        case X_UNIFORM: {
            double u1, u2;
            expr_uniforms(P, (unsigned)dof, 1u, u1, u2);
            st[sp++] = u1;
        } break;
        case X_LOAD: st[sp++] = loc[arg]; break;
        case X_STORE: loc[arg] = st[--sp]; break;
        case X_ADD: sp--; st[sp - 1] = st[sp - 1] + st[sp]; break;
        case X_SUB: sp--; st[sp - 1] = st[sp - 1] - st[sp]; break;
        case X_MUL: sp--; st[sp - 1] = st[sp - 1] * st[sp]; break;
        case X_DIV: sp--; st[sp - 1] = st[sp - 1] / st[sp]; break;
        case X_NEG: st[sp - 1] = -st[sp - 1]; break;
        case X_POW: sp--; st[sp - 1] = pow(st[sp - 1], st[sp]); break;
        case X_POWI: {
            const double b = st[sp - 1];
            int e = arg < 0 ? -arg : arg;
            double r = 1.0, q = b;
            while (e) {
                if (e & 1) r *= q;
                q *= q;
                e >>= 1;
            }
            st[sp - 1] = arg < 0 ? 1.0 / r : r;
        } break;
        case X_SQRT: st[sp - 1] = sqrt(st[sp - 1]); break;
        case X_EXP: st[sp - 1] = exp(st[sp - 1]); break;
        case X_LOG: st[sp - 1] = log(st[sp - 1]); break;
        case X_SIN: st[sp - 1] = sin(st[sp - 1]); break;
        case X_COS: st[sp - 1] = cos(st[sp - 1]); break;
        case X_TAN: st[sp - 1] = tan(st[sp - 1]); break;
        case X_ASIN: st[sp - 1] = asin(st[sp - 1]); break;
        case X_ACOS: st[sp - 1] = acos(st[sp - 1]); break;
        case X_ATAN: st[sp - 1] = atan(st[sp - 1]); break;
        case X_SINH: st[sp - 1] = sinh(st[sp - 1]); break;
        case X_COSH: st[sp - 1] = cosh(st[sp - 1]); break;
        case X_TANH: st[sp - 1] = tanh(st[sp - 1]); break;
        case X_ERF: st[sp - 1] = erf(st[sp - 1]); break;
        case X_ERFC: st[sp - 1] = erfc(st[sp - 1]); break;
        case X_ABS: st[sp - 1] = fabs(st[sp - 1]); break;
        case X_FLOOR: st[sp - 1] = floor(st[sp - 1]); break;
        case X_CEIL: st[sp - 1] = ceil(st[sp - 1]); break;
        case X_STEP: st[sp - 1] = st[sp - 1] >= 0.0 ? 1.0 : 0.0; break;
        case X_DELTA: st[sp - 1] = st[sp - 1] == 0.0 ? 1.0 : 0.0; break;
        case X_MIN: sp--; st[sp - 1] = fmin(st[sp - 1], st[sp]); break;
        case X_MAX: sp--; st[sp - 1] = fmax(st[sp - 1], st[sp]); break;
        case X_SELECT: sp -= 2; st[sp - 1] = st[sp - 1] != 0.0 ? st[sp] : st[sp + 1]; break;
        case X_ATAN2: sp--; st[sp - 1] = atan2(st[sp - 1], st[sp]); break;
        default: break;
        }
    }
    return st[0];
}

