// atomsmm_amd/csrc/expr.hip -- general per-DOF step-program expressions (gfx950, fp64).
//
// The reference's thermostat / regulated / stochastic propagators (propagators.py:276-827, 1108-2172) emit
// CustomIntegrator steps `addComputePerDof(variable, expression)` and `addComputeSum(variable, expression)` whose
// expressions go beyond the kick / move forms of the RESPA hot path (integrate.hip): products of globals, exp, sqrt,
// `gaussian`, auxiliary definitions after ';' ...  OpenMM compiles such expressions for its VM; here the host
// (atomsmm_amd/expr.py) compiles them to a small POSTFIX program that one kernel interprets per degree of freedom.
// Off the hot path by design: a few launches per step, 3N elements each.
//
// Random numbers: counter-based Philox-4x32-10 (Salmon et al., SC'11), key = seed, counter = (dof, occurrence of
// `gaussian`/`uniform` within the expression, launch counter): reproducible whatever the launch geometry, and the
// same stream on every rank (ranks integrate all atoms redundantly and must stay in lock-step).
#include "amm_ctx.h"

#define AMM_EXPR_MAXCODE 256
#define AMM_EXPR_MAXCONST 48
#define AMM_EXPR_MAXGLOBAL 48
#define AMM_EXPR_STACK 24
#define AMM_EXPR_LOCALS 16

enum {
    X_CONST = 0, X_GLOBAL = 1, X_BUF = 2, X_MASS = 3, X_GAUSS = 4, X_UNIFORM = 5, X_LOAD = 6, X_STORE = 7,
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
__device__ __forceinline__ void expr_uniforms(const ExprProg &P, unsigned dof, unsigned occurrence, double &u1, double &u2) {
    unsigned r[4];
    philox4x32_10(dof, occurrence, (unsigned)P.counter, (unsigned)(P.counter >> 32), (unsigned)P.seed, (unsigned)(P.seed >> 32), r);
    const unsigned long long a = ((unsigned long long)r[0] << 21) ^ (r[1] >> 11), b = ((unsigned long long)r[2] << 21) ^ (r[3] >> 11);
    u1 = ((double)(a & 0x1FFFFFFFFFFFFFull) + 0.5) * (1.0 / 9007199254740992.0);
    u2 = ((double)(b & 0x1FFFFFFFFFFFFFull) + 0.5) * (1.0 / 9007199254740992.0);
}

__device__ double expr_run(const ExprProg &P, int dof) {
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
            double u1, u2;
            expr_uniforms(P, (unsigned)dof, 0u, u1, u2);      // one draw per evaluation and DOF, as in OpenMM
            st[sp++] = sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925 * u2);
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

// dst[dof] <- value (dst may be one of the input buffers: each thread reads only its own element);
// block partial sums of the values -> part (ComputeSum)
__global__ void __launch_bounds__(256) k_expr(int n3, ExprProg P, double *dst, double *part) {
    const int dof = blockIdx.x * blockDim.x + threadIdx.x;
    double val = 0.0;
    if (dof < n3) {
        val = expr_run(P, dof);
        if (dst) dst[dof] = val;
    }
    if (part) {
        __shared__ double red[4];
        for (int off = 32; off > 0; off >>= 1) val += __shfl_xor(val, off);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = val;
        __syncthreads();
        if (threadIdx.x == 0) part[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
    }
}

int amm_reduce_add(amm_ctx *ctx, const double *d_part, int n, double scale, double *d_out);

int amm_expr_eval_impl(amm_ctx *ctx, const int32_t *code, int n_code, const double *consts, int n_consts, const double *globals,
                       int n_globals, unsigned long long seed, unsigned long long counter, double *d_dst, double *d_sum) {
    if (n_code < 1 || n_code > AMM_EXPR_MAXCODE || n_consts > AMM_EXPR_MAXCONST || n_globals > AMM_EXPR_MAXGLOBAL) {
        amm_set_error("amm_expr_eval: program too long (code <= 256, constants <= 48, globals <= 48)");
        return 1;
    }
    ExprProg P;
    P.ncode = n_code;
    int depth = 0, max_depth = 0;
    for (int k = 0; k < n_code; ++k) {
        P.code[k] = code[k];
        const int op = code[k] & 0xff, arg = code[k] >> 8;
        // validate operands and the stack discipline on the host: the kernel indexes without checks
        int push = 0, pop = 0;
        switch (op) {
        case X_CONST: push = 1; if (arg < 0 || arg >= n_consts) depth = -1000; break;
        case X_GLOBAL: push = 1; if (arg < 0 || arg >= n_globals) depth = -1000; break;
        case X_BUF: push = 1; if (arg < 0 || arg >= AMM_MAX_SLOTS || !ctx->slots[arg]) depth = -1000; break;
        case X_MASS: push = 1; if (!ctx->d_mass) depth = -1000; break;
        case X_GAUSS: case X_UNIFORM: push = 1; break;
        case X_LOAD: push = 1; if (arg < 0 || arg >= AMM_EXPR_LOCALS) depth = -1000; break;
        case X_STORE: pop = 1; if (arg < 0 || arg >= AMM_EXPR_LOCALS) depth = -1000; break;
        case X_ADD: case X_SUB: case X_MUL: case X_DIV: case X_POW: case X_MIN: case X_MAX: case X_ATAN2: pop = 2; push = 1; break;
        case X_SELECT: pop = 3; push = 1; break;
        case X_NEG: case X_POWI: case X_SQRT: case X_EXP: case X_LOG: case X_SIN: case X_COS: case X_TAN: case X_ASIN: case X_ACOS:
        case X_ATAN: case X_SINH: case X_COSH: case X_TANH: case X_ERF: case X_ERFC: case X_ABS: case X_FLOOR: case X_CEIL:
        case X_STEP: case X_DELTA: pop = 1; push = 1; break;
        default: depth = -1000; break;
        }
        depth -= pop;
        if (depth < 0) break;
        depth += push;
        max_depth = std::max(max_depth, depth);
    }
    if (depth != 1 || max_depth > AMM_EXPR_STACK) {
        amm_set_error("amm_expr_eval: malformed program (bad operand, unbound buffer, or stack discipline)");
        return 1;
    }
    for (int k = 0; k < n_consts; ++k) P.consts[k] = consts[k];
    for (int k = 0; k < n_globals; ++k) P.globals[k] = globals[k];
    for (int k = 0; k < AMM_MAX_SLOTS; ++k) P.bufs[k] = ctx->slots[k];
    P.mass = ctx->d_mass;
    P.seed = seed;
    P.counter = counter;
    const int n3 = 3 * ctx->n, nblk = (n3 + 255) / 256;
    if (d_sum && !ctx->d_expr_part) AMM_HIP(hipMalloc(&ctx->d_expr_part, sizeof(double) * nblk));
    hipLaunchKernelGGL(k_expr, dim3(nblk), dim3(256), 0, ctx->stream, n3, P, d_dst, d_sum ? ctx->d_expr_part : (double *)nullptr);
    AMM_HIP(hipGetLastError());
    if (d_sum) {
        AMM_HIP(hipMemsetAsync(d_sum, 0, sizeof(double), ctx->stream));
        if (amm_reduce_add(ctx, ctx->d_expr_part, nblk, 1.0, d_sum)) return 1;
    }
    if (d_dst == ctx->d_x) ctx->pos_epoch++;
    return 0;
}
