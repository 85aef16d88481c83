// fp64 VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction for the operations of the pair kernels.
//   hipcc --offload-arch=gfx950 -O3 -o fp64_rate fp64_rate.hip && ./fp64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64
template <int OP>
__global__ void __launch_bounds__(256) k_rate(double *out, int iters, double seed) {
    double a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = seed + threadIdx.x * 1e-3 + k;
    const double b = 1.0000001, c = 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (OP == 0) a[k] = __builtin_fma(a[k], b, c);
                else if (OP == 1) a[k] = a[k] * b;
                else if (OP == 2) a[k] = a[k] + c;
                else if (OP == 3) a[k] = __builtin_rint(a[k]) + 0.0;   // rndne (+ add folded?)
                else if (OP == 4) a[k] = __builtin_amdgcn_rsq(a[k]);
                else if (OP == 5) a[k] = (a[k] > 1.5) ? b : a[k];
                else if (OP == 6) { float f = (float)a[k]; f = __builtin_fmaf(f, 1.0000001f, 1e-9f); a[k] = f; }
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += a[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(const char *name, int waves_per_simd, int instr_per_elem) {
    const int blocks = 256 * waves_per_simd;        // 256 CUs x (4 waves per block = 1 per SIMD) x waves_per_simd
    double *out;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k_rate<OP><<<blocks, 256>>>(out, 10, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_rate<OP><<<blocks, 256>>>(out, iters, 1.0);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double winst = (double)iters * REP * instr_per_elem * waves_per_simd;     // wave-instructions per SIMD
    printf("%-22s waves/SIMD %d: %.3f ms  -> %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", name, waves_per_simd, ms,
           ms * 1e6 / winst, ms * 1e6 / winst * 2.4);
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 4, 8}) run<0>("v_fma_f64", w, 1);
    for (int w : {1, 4}) run<1>("v_mul_f64", w, 1);
    for (int w : {1, 4}) run<2>("v_add_f64", w, 1);
    for (int w : {1, 4}) run<3>("v_rndne_f64(+add)", w, 2);
    for (int w : {1, 4}) run<4>("v_rsq_f64", w, 1);
    for (int w : {1, 4}) run<5>("cmp_f64+2cndmask", w, 3);
    for (int w : {1, 4}) run<6>("cvt+fma_f32+cvt", w, 3);
    return 0;
}
