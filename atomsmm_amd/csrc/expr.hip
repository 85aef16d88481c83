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
#include "device_utils.h"

#include "expr_vm.h"

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

// Scalars of a host-walked program: scalars[dst] <- expression for a sequence of assignments (X_OUT closes each), whose X_DEVG
// operands are other entries of `scalars` (device memory).  CustomIntegrator.addComputeGlobal steps whose operands wait on device
// results -- the extended variable's block of AdiabaticDynamicsIntegrator (integrators.py:701-737): lambda moves, reflects at the
// walls, its Nose-Hoover thermostat acts, all on numbers that depend on deriv(energy, lambda) sums still in flight.
// One wavefront, every lane doing the same: the program word comes by v_readlane from a register that holds 64 words of the code
// (refilled from LDS every 64 words), constants, stack and locals live one per LANE in registers (push = a select on the lane index,
// pop = v_readlane with a scalar index): no memory access on the interpreter's critical path but the scalars themselves.  (Stack and
// code in LDS took 0.25 us per word, in private memory and the launch argument 0.6.)
__global__ void __launch_bounds__(64) k_expr_scalar(ScalarProg P, double *scalars) {
    __shared__ int s_code[AMM_SCALAR_MAXCODE];
    const int lane = threadIdx.x;
    for (int k = lane; k < P.ncode; k += 64) s_code[k] = P.code[k];
    const double c_lo = lane < P.nconst ? P.consts[lane] : 0.0, c_hi = 64 + lane < P.nconst ? P.consts[64 + lane] : 0.0;
    __syncthreads();
    auto lane_of = [&](double v, int l) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
        return __hiloint2double(hi, lo);
    };
    double stk = 0.0, locals = 0.0;       // lane k: stack slot k / local k
    int sp = 0, chunk = 0;
#define X_PUSH(x) do { const double x_ = (x); stk = lane == sp ? x_ : stk; ++sp; } while (0)
#define X_POP() (--sp, lane_of(stk, sp))
#define X_UN(expr) do { const double a = X_POP(); X_PUSH(expr); } while (0)
#define X_BIN(expr) do { const double b = X_POP(); const double a = X_POP(); X_PUSH(expr); } while (0)
    for (int pc = 0; pc < P.ncode; ++pc) {
        if ((pc & 63) == 0) chunk = s_code[min(pc + lane, AMM_SCALAR_MAXCODE - 1)];
        const int word = __builtin_amdgcn_readlane(chunk, pc & 63), op = word & 0xff, arg = word >> 8;
        // the frequent words first, one test each (a lone wavefront pays ~40 cycles per TAKEN branch: the compiler's decision tree
        // over all 40 opcodes cost six of them per word)
        if (op == X_CONST) { X_PUSH(arg < 64 ? lane_of(c_lo, arg) : lane_of(c_hi, arg - 64)); continue; }
        if (op == X_MUL) { X_BIN(a * b); continue; }
        if (op == X_ADD) { X_BIN(a + b); continue; }
        if (op == X_LOAD) { X_PUSH(lane_of(locals, arg)); continue; }
        if (op == X_DEVG) { X_PUSH(__longlong_as_double((long long)amm_ld_l2((const unsigned long long *)&scalars[arg]))); continue; }
        if (op == X_SUB) { X_BIN(a - b); continue; }
        if (op == X_HORNER) {        // top <- top * local 0 + constant: one step of a polynomial in the first auxiliary definition
            const double a = X_POP();
            X_PUSH(fma(a, lane_of(locals, 0), arg < 64 ? lane_of(c_lo, arg) : lane_of(c_hi, arg - 64)));
            continue;
        }
        switch (op) {
        case X_CONST: X_PUSH(arg < 64 ? lane_of(c_lo, arg) : lane_of(c_hi, arg - 64)); break;
        case X_DEVG: X_PUSH(__longlong_as_double((long long)amm_ld_l2((const unsigned long long *)&scalars[arg]))); break;      // (through L2: an X_OUT of this program may have written it)
        case X_OUT: {
            const double a = X_POP();
            if (lane == 0) amm_st_l2((unsigned long long *)&scalars[arg], (unsigned long long)__double_as_longlong(a));
            __builtin_amdgcn_s_waitcnt(0);        // (a later X_DEVG of this program may read it)
        } break;
        case X_LOAD: X_PUSH(lane_of(locals, arg)); break;
        case X_STORE: {
            const double a = X_POP();
            locals = lane == arg ? a : locals;
        } break;
        case X_ADD: X_BIN(a + b); break;
        case X_SUB: X_BIN(a - b); break;
        case X_MUL: X_BIN(a * b); break;
        case X_DIV: X_BIN(a / b); break;
        case X_NEG: X_UN(-a); break;
        case X_POW: X_BIN(pow(a, b)); break;
        case X_POWI: {
            const double base = X_POP();
            int e = arg < 0 ? -arg : arg;
            double r = 1.0, q = base;
            while (e) {
                if (e & 1) r *= q;
                q *= q;
                e >>= 1;
            }
            X_PUSH(arg < 0 ? 1.0 / r : r);
        } break;
        case X_SQRT: X_UN(sqrt(a)); break;
        case X_EXP: X_UN(exp(a)); break;
        case X_LOG: X_UN(log(a)); break;
        case X_SIN: X_UN(sin(a)); break;
        case X_COS: X_UN(cos(a)); break;
        case X_TAN: X_UN(tan(a)); break;
        case X_ASIN: X_UN(asin(a)); break;
        case X_ACOS: X_UN(acos(a)); break;
        case X_ATAN: X_UN(atan(a)); break;
        case X_SINH: X_UN(sinh(a)); break;
        case X_COSH: X_UN(cosh(a)); break;
        case X_TANH: X_UN(tanh(a)); break;
        case X_ERF: X_UN(erf(a)); break;
        case X_ERFC: X_UN(erfc(a)); break;
        case X_ABS: X_UN(fabs(a)); break;
        case X_FLOOR: X_UN(floor(a)); break;
        case X_CEIL: X_UN(ceil(a)); break;
        case X_STEP: X_UN(a >= 0.0 ? 1.0 : 0.0); break;
        case X_DELTA: X_UN(a == 0.0 ? 1.0 : 0.0); break;
        case X_MIN: X_BIN(fmin(a, b)); break;
        case X_MAX: X_BIN(fmax(a, b)); break;
        case X_ATAN2: X_BIN(atan2(a, b)); break;
        case X_SELECT: {
            const double no = X_POP();
            const double yes = X_POP();
            const double cond = X_POP();
            X_PUSH(cond != 0.0 ? yes : no);
        } break;
        default: break;
        }
    }
#undef X_PUSH
#undef X_POP
#undef X_UN
#undef X_BIN
}

static int expr_fill(amm_ctx *ctx, ExprProg &P, const int32_t *code, int n_code, const double *consts, int n_consts, const double *globals,
                     int n_globals, bool per_dof);

int amm_expr_eval_scalar_impl(amm_ctx *ctx, const int32_t *code, int n_code, const double *consts, int n_consts, double *d_scalars,
                              int n_scalars) {
    if (n_code < 1 || n_code > AMM_SCALAR_MAXCODE || n_consts > AMM_SCALAR_MAXCONST) {
        amm_set_error("amm_expr_eval_scalar: program too long (code <= 640, constants <= 96)");
        return 1;
    }
    // validate on the host: the kernel indexes without checks
    ScalarProg P;
    P.ncode = n_code;
    P.nconst = n_consts;
    int depth = 0, max_depth = 0, outs = 0;
    for (int k = 0; k < n_code && depth >= 0; ++k) {
        P.code[k] = code[k];
        const int op = code[k] & 0xff, arg = code[k] >> 8;
        int push = 0, pop = 0;
        switch (op) {
        case X_CONST: push = 1; if (arg < 0 || arg >= n_consts) depth = -1000; break;
        case X_DEVG: push = 1; if (arg < 0 || arg >= n_scalars) depth = -1000; break;
        case X_HORNER: pop = 1; push = 1; if (arg < 0 || arg >= n_consts) depth = -1000; break;
        case X_OUT: pop = 1; outs++; if (arg < 0 || arg >= n_scalars || depth != 1) depth = -1000; break;
        case X_LOAD: push = 1; if (arg < 0 || arg >= AMM_EXPR_LOCALS) depth = -1000; break;
        case X_STORE: pop = 1; if (arg < 0 || arg >= AMM_EXPR_LOCALS) depth = -1000; break;
        case X_ADD: case X_SUB: case X_MUL: case X_DIV: case X_POW: case X_MIN: case X_MAX: case X_ATAN2: pop = 2; push = 1; break;
        case X_SELECT: pop = 3; push = 1; break;
        case X_NEG: case X_POWI: case X_SQRT: case X_EXP: case X_LOG: case X_SIN: case X_COS: case X_TAN: case X_ASIN: case X_ACOS:
        case X_ATAN: case X_SINH: case X_COSH: case X_TANH: case X_ERF: case X_ERFC: case X_ABS: case X_FLOOR: case X_CEIL:
        case X_STEP: case X_DELTA: pop = 1; push = 1; break;
        default: depth = -1000; break;        // (no per-DOF operands, no random draws: the host makes those)
        }
        depth -= pop;
        if (depth < 0) break;
        depth += push;
        max_depth = std::max(max_depth, depth);
    }
    if (depth != 0 || outs == 0 || max_depth > AMM_EXPR_STACK) {
        amm_set_error("amm_expr_eval_scalar: malformed program (bad operand, stack discipline, or an assignment left open)");
        return 1;
    }
    for (int k = 0; k < n_consts; ++k) P.consts[k] = consts[k];
    hipLaunchKernelGGL(k_expr_scalar, dim3(1), dim3(64), 0, ctx->stream, P, d_scalars);
    AMM_HIP(hipGetLastError());
    return 0;
}

int amm_expr_eval_impl(amm_ctx *ctx, const int32_t *code, int n_code, const double *consts, int n_consts, const double *globals,
                       int n_globals, unsigned long long seed, unsigned long long counter, double *d_dst, double *d_sum) {
    ExprProg P;
    if (expr_fill(ctx, P, code, n_code, consts, n_consts, globals, n_globals, true)) return 1;
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

// validate a postfix program on the host (the kernels index without checks) and copy it into the launch argument
static int expr_fill(amm_ctx *ctx, ExprProg &P, const int32_t *code, int n_code, const double *consts, int n_consts, const double *globals,
                     int n_globals, bool per_dof) {
    if (n_code < 1 || n_code > AMM_EXPR_MAXCODE || n_consts > AMM_EXPR_MAXCONST || n_globals > AMM_EXPR_MAXGLOBAL) {
        amm_set_error("amm_expr_eval: program too long (code <= 256, constants <= 48, globals <= 48)");
        return 1;
    }
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
        case X_BUF: push = 1; if (!per_dof || arg < 0 || arg >= AMM_MAX_SLOTS || !ctx->slots[arg]) depth = -1000; break;
        case X_MASS: push = 1; if (!per_dof || !ctx->d_mass) depth = -1000; break;
        case X_GAUSS: case X_UNIFORM: push = 1; if (!per_dof) depth = -1000; break;      // (a scalar's random draws are made by the host)
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
    return 0;
}

// AMM_OP_BATH outside the inner-loop kernel
__global__ void k_bath_ou(int n3, double *v, const double *__restrict__ mass, double z, double kT, unsigned long long seed,
                          unsigned long long counter) {
    const int dof = blockIdx.x * blockDim.x + threadIdx.x;
    if (dof >= n3) return;
    v[dof] = amm_ou_step(v[dof], mass[dof / 3], z, kT, amm_gaussian(seed, counter, (unsigned)dof));
}

__global__ void k_bath_nhl(int n3, double *v, double *w, const double *__restrict__ mass, BathDef b, unsigned long long seed,
                           unsigned long long counter) {
    const int dof = blockIdx.x * blockDim.x + threadIdx.x;
    if (dof >= n3) return;
    double vv = v[dof], ww = w[dof];
    amm_nhl_step(vv, ww, mass[dof / 3], b.h, b.z, b.kT, b.Q, b.friction, amm_gaussian(seed, counter, (unsigned)dof));
    v[dof] = vv;
    w[dof] = ww;
}

__global__ void k_bath_sin(int n3, double *v, double *v1, double *v2, const double *__restrict__ mass, BathDef b, double Q1, double LkT,
                           unsigned long long seed, unsigned long long counter) {
    const int dof = blockIdx.x * blockDim.x + threadIdx.x;
    if (dof >= n3) return;
    double vv = v[dof], a = v1[dof], c = v2[dof];
    amm_sin_bath_step(vv, a, c, mass[dof / 3], b.h, b.z, b.kT, b.Q, b.friction, Q1, LkT, amm_gaussian(seed, counter, (unsigned)dof));
    v[dof] = vv;
    v1[dof] = a;
    v2[dof] = c;
}

// isokinetic kick outside the inner-loop kernel (amm_kick_impl dispatches here when the context is in isokinetic mode)
__global__ void k_isokick(int n3, double *v, double *v1, const double *__restrict__ f, const double *__restrict__ f2, int plus,
                          const double *__restrict__ mass, double coef, double LkT, double Q1) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n3) return;
    double ff = f[t];
    if (f2) ff = plus ? ff + f2[t] : ff - f2[t];
    double vv = v[t], a = v1[t];
    amm_iso_kick(vv, a, ff, mass[t / 3], coef, LkT, Q1);
    v[t] = vv;
    v1[t] = a;
}

int amm_isokick_impl(amm_ctx *ctx, double *d_v, const double *d_f, const double *d_f2, int plus, const double *d_mass, double coef) {
    double *v1 = (ctx->iso.slot >= 0 && ctx->iso.slot < AMM_MAX_SLOTS) ? ctx->slots[ctx->iso.slot] : nullptr;
    if (!v1) {
        amm_set_error("isokinetic kick: the thermostat-velocity buffer is not bound");
        return 1;
    }
    const int n3 = 3 * ctx->n;
    hipLaunchKernelGGL(k_isokick, dim3((n3 + 255) / 256), dim3(256), 0, ctx->stream, n3, d_v, v1, d_f, d_f2, plus, d_mass, coef,
                       ctx->iso.LkT, ctx->iso.Q1);
    AMM_HIP(hipGetLastError());
    return 0;
}

int amm_bath_impl(amm_ctx *ctx, const BathDef &bath, double *d_v, unsigned long long counter) {
    const int n3 = 3 * ctx->n;
    if (bath.kind == 2) {
        double *v2 = (bath.slot >= 0 && bath.slot < AMM_MAX_SLOTS) ? ctx->slots[bath.slot] : nullptr;
        double *v1 = (ctx->iso.on && ctx->iso.slot >= 0 && ctx->iso.slot < AMM_MAX_SLOTS) ? ctx->slots[ctx->iso.slot] : nullptr;
        if (!v1 || !v2) {
            amm_set_error("stochastic-isokinetic bath: isokinetic mode is off or a thermostat-velocity buffer is not bound");
            return 1;
        }
        hipLaunchKernelGGL(k_bath_sin, dim3((n3 + 255) / 256), dim3(256), 0, ctx->stream, n3, d_v, v1, v2, ctx->d_mass, bath, ctx->iso.Q1,
                           ctx->iso.LkT, ctx->expr_seed, counter);
        AMM_HIP(hipGetLastError());
        return 0;
    }
    if (bath.kind == 1) {
        double *w = (bath.slot >= 0 && bath.slot < AMM_MAX_SLOTS) ? ctx->slots[bath.slot] : nullptr;
        if (!w) {
            amm_set_error("Nose-Hoover-Langevin bath: the thermostat-velocity buffer is not bound");
            return 1;
        }
        hipLaunchKernelGGL(k_bath_nhl, dim3((n3 + 255) / 256), dim3(256), 0, ctx->stream, n3, d_v, w, ctx->d_mass, bath, ctx->expr_seed, counter);
        AMM_HIP(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(k_bath_ou, dim3((n3 + 255) / 256), dim3(256), 0, ctx->stream, n3, d_v, ctx->d_mass, bath.z, bath.kT,
                       ctx->expr_seed, counter);
    AMM_HIP(hipGetLastError());
    return 0;
}
