// atomsmm_amd/csrc/integrate.hip -- per-DOF step-program primitives (gfx950, fp64).
//
// The reference's propagators emit CustomIntegrator per-DOF assignments (propagators.py:249 move,
// :271 kick; integrators.py:113 mvv sum); OpenMM's VM evaluates them operation by operation in fp64.
// These kernels do the same arithmetic in the same order, with contraction into FMA disabled
// (#pragma clang fp contract(off) below), so a kick/move here is BIT-IDENTICAL to the oracle's.
// Pure streaming: 3N doubles, 16-byte accesses where the layout allows.
#include "amm_ctx.h"
#include "bonded_terms.h"

// hipcc contracts a*b+c into FMA by default (and __dmul_rn/__dadd_rn are plain * and + in HIP):
// switch contraction off for this file so that mul and add round separately, as in OpenMM's VM.
#pragma clang fp contract(off)

// v <- v + (coef)*(f - fsub)/m      propagators.py:271 with force expression `f`, `(_f2_-f1)`, ...
__global__ void k_kick(int n3, double *__restrict__ v, const double *__restrict__ f, const double *__restrict__ f2,
                       int plus, const double *__restrict__ mass, double coef) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n3) return;
    double ff = f[t];
    if (f2) ff = plus ? ff + f2[t] : ff - f2[t];
    const double num = coef * ff;
    const double dv = num / mass[t / 3];
    v[t] = v[t] + dv;
}

// x <- x + (coef)*v                propagators.py:249
__global__ void k_move(int n3, double *__restrict__ x, const double *__restrict__ v, double coef) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n3) return;
    const double dx = coef * v[t];
    x[t] = x[t] + dx;
}

// up to four kicks and a move in one launch: the same operations per degree of freedom, in the same order, as the separate
// kernels (v <- v + (c*f)/m ... ; x <- x + d*v) -- programs outside the one-launch inner loop (a pair force in the innermost
// group) spend a tenth of their step in these 5 us launches otherwise
__global__ void k_kicks_move(int n3, double *__restrict__ x, double *__restrict__ v, KickList K, const double *__restrict__ mass,
                             int with_move, double dcoef) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n3) return;
    double vt = v[t];
    const double m = mass[t / 3];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k < K.n) {
            double ff = K.f[k][t];
            if (K.f2[k]) ff = K.plus[k] ? ff + K.f2[k][t] : ff - K.f2[k][t];
            const double num = K.coef[k] * ff;
            const double dv = num / m;
            vt = vt + dv;
        }
    }
    v[t] = vt;
    if (with_move) {
        const double dx = dcoef * vt;
        x[t] = x[t] + dx;
    }
}
// the same with a move, one thread per ATOM: the three new coordinates are at hand, so the launch also evaluates the displacement
// triggers of the neighbour lists (WatchArgs) -- the check launch in front of the next pair evaluation is not needed then
__global__ void k_kicks_move_atoms(int n, double *__restrict__ x, double *__restrict__ v, KickList K, const double *__restrict__ mass,
                                   double dcoef, WatchArgs W) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double m = mass[i];
    double xn[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int t = 3 * i + c;
        double vt = v[t];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < K.n) {
                double ff = K.f[k][t];
                if (K.f2[k]) ff = K.plus[k] ? ff + K.f2[k][t] : ff - K.f2[k][t];
                const double num = K.coef[k] * ff;
                const double dv = num / m;
                vt = vt + dv;
            }
        }
        v[t] = vt;
        const double dx = dcoef * vt;
        xn[c] = x[t] + dx;
        x[t] = xn[c];
    }
    amm_watch_atom(W, i, xn);      // (both flags: "rebuild wanted" and AMM_FLAG_FAR -- group.hip's candidate walk trusts the second)
}

int amm_kicks_move_impl(amm_ctx *ctx, const double *const *fa, const double *const *fb, const int *plus, const double *coef, int nk,
                        int with_move, double dcoef) {
    KickList K;
    K.n = nk;
    for (int k = 0; k < 4; ++k) {
        K.f[k] = k < nk ? fa[k] : nullptr;
        K.f2[k] = k < nk ? fb[k] : nullptr;
        K.plus[k] = k < nk ? plus[k] : 0;
        K.coef[k] = k < nk ? coef[k] : 0.0;
    }
    const int n3 = 3 * ctx->n;
    if (with_move) {
        WatchArgs W;
        amm_collect_watches(ctx, W);           // (the caller bumps pos_epoch and calls amm_watch_moved)
        hipLaunchKernelGGL(k_kicks_move_atoms, dim3((ctx->n + 255) / 256), dim3(256), 0, ctx->stream, ctx->n, ctx->d_x, ctx->d_v, K,
                           ctx->d_mass, dcoef, W);
        AMM_HIP(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(k_kicks_move, dim3((n3 + 255) / 256), dim3(256), 0, ctx->stream, n3, ctx->d_x, ctx->d_v, K, ctx->d_mass,
                       with_move, dcoef);
    AMM_HIP(hipGetLastError());
    return 0;
}

// dst <- a + coef*b   (`fm2 <- f2-f1`, propagators.py:951)
__global__ void k_combine(int n3, double *__restrict__ dst, const double *__restrict__ a, const double *__restrict__ b, double coef) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n3) return;
    const double sb = coef * b[t];
    dst[t] = a[t] + sb;
}

__global__ void k_mvv(int n, const double *__restrict__ v, const double *__restrict__ m, double *part) {
    __shared__ double red[4];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    double s = 0.0;
    if (i < n) s = m[i] * (v[3 * i] * v[3 * i] + v[3 * i + 1] * v[3 * i + 1] + v[3 * i + 2] * v[3 * i + 2]);
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

int amm_kick_impl(amm_ctx *ctx, double *d_v, const double *d_f, const double *d_f2, int plus, const double *d_mass, double coef) {
    if (ctx->iso.on) return amm_isokick_impl(ctx, d_v, d_f, d_f2, plus, d_mass, coef);      // SIN(R): every kick is isokinetic
    const int n3 = 3 * ctx->n;
    hipLaunchKernelGGL(k_kick, dim3((n3 + 255) / 256), dim3(256), 0, ctx->stream, n3, d_v, d_f, d_f2, plus, d_mass, coef);
    AMM_HIP(hipGetLastError());
    return 0;
}
int amm_move_impl(amm_ctx *ctx, double *d_x, const double *d_v, double coef) {
    const int n3 = 3 * ctx->n;
    hipLaunchKernelGGL(k_move, dim3((n3 + 255) / 256), dim3(256), 0, ctx->stream, n3, d_x, d_v, coef);
    AMM_HIP(hipGetLastError());
    return 0;
}
int amm_combine_impl(amm_ctx *ctx, double *d_dst, const double *d_a, const double *d_b, double coef) {
    const int n3 = 3 * ctx->n;
    hipLaunchKernelGGL(k_combine, dim3((n3 + 255) / 256), dim3(256), 0, ctx->stream, n3, d_dst, d_a, d_b, coef);
    AMM_HIP(hipGetLastError());
    return 0;
}
int amm_copy_impl(amm_ctx *ctx, double *d_dst, const double *d_src) {
    AMM_HIP(hipMemcpyAsync(d_dst, d_src, sizeof(double) * 3 * (size_t)ctx->n, hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
}
int amm_mvv_impl(amm_ctx *ctx, const double *d_v, const double *d_m, double *d_out) {
    const int nblk = (ctx->n + 255) / 256;
    hipLaunchKernelGGL(k_mvv, dim3(nblk), dim3(256), 0, ctx->stream, ctx->n, d_v, d_m, ctx->d_scratch);
    AMM_HIP(hipGetLastError());
    AMM_HIP(hipMemsetAsync(d_out, 0, sizeof(double), ctx->stream));
    return amm_reduce_add(ctx, ctx->d_scratch, nblk, 1.0, d_out);
}
