// atomsmm_amd/csrc/device_utils.h -- device helpers shared by the kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

// "last block" idiom: every block takes a ticket when its results have reached the device coherence point; the
// block that draws the last one runs the serial tail of the kernel (a scan / a reduction) -- one launch less per
// stage of the chain.  The results handed to the last block are written with device-scope atomics / atomic
// stores only (write-through to the level shared by the 8 XCDs) and read back with device-scope atomic loads:
// waiting for the writes to be acknowledged (s_waitcnt 0) then orders them before the ticket.  A __threadfence()
// here would instead write back each XCD's whole dirty L2 -- including the list being built -- once per block
// (measured: 2.6x on the build kernel).
// `ticket` points at AMM_TICKET_INTS ints.  Big grids draw in two levels -- 32 classes (blockIdx & 31), then the master
// among the last block of each class -- because same-address device-scope atomics serialise at a few ns each: 15 000 blocks
// that have nothing else to do (the culled build of an interaction-group list) spent 0.2 ms queueing for one counter.
#ifndef AMM_TICKET_INTS
#define AMM_TICKET_INTS 64
#endif
__device__ __forceinline__ bool amm_last_block(int *ticket) {
    __shared__ int s_last;
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nb = (int)gridDim.x;
        if (nb < 1024) {
            const int t = atomicAdd(ticket, 1);
            s_last = (t == nb - 1);
            if (s_last) *ticket = 0;        // every other block has already drawn: safe to re-arm
        } else {
            const int k = (int)(blockIdx.x & 31u);
            const int of_class = (nb - k + 31) >> 5;             // blocks b < nb with b & 31 == k
            s_last = 0;
            if (atomicAdd(&ticket[1 + k], 1) == of_class - 1) {
                ticket[1 + k] = 0;
                if (atomicAdd(ticket, 1) == 31) {
                    *ticket = 0;
                    s_last = 1;
                }
            }
        }
    }
    __syncthreads();
    return s_last != 0;
}
// the same with `count` (< 1024, the same in every block that calls) participants out of the grid: ticket[0] alone
__device__ __forceinline__ bool amm_last_of(int *ticket, int count) {
    __shared__ int s_last_of;
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t = atomicAdd(ticket, 1);
        s_last_of = (t == count - 1);
        if (s_last_of) *ticket = 0;
    }
    __syncthreads();
    return s_last_of != 0;
}
__device__ __forceinline__ int amm_ld_l2(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long amm_ld_l2(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void amm_st_l2(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void amm_st_l2(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}


// Two histograms in one word -- count[c] = atoms of cell c (low 16 bits) | those of them with a Lennard-Jones site << 16 -- and
// both exclusive scans in one sweep: start[0..ncell] and start_hi[0..ncell]; count <- 0; returns the largest cell count.
// (One atomic per atom and one pass over the counts instead of two of each.)
__device__ __forceinline__ int amm_block_scan_packed(int ncell, int *count, int *start, int *start_hi) {
    __shared__ int part[256];
    __shared__ int part_hi[256];
    __shared__ int s_most[256];
    const int t = threadIdx.x;
    const int per = (ncell + 255) / 256;
    const int c0 = min(t * per, ncell), c1 = min(c0 + per, ncell);
    constexpr int KEEP = 32;
    int keep[KEEP];
    const bool kept = per <= KEEP;
    int sum = 0, sum_hi = 0, most = 0;
    if (kept) {
#pragma unroll
        for (int j = 0; j < KEEP; ++j) keep[j] = (c0 + j < c1) ? amm_ld_l2(&count[c0 + j]) : 0;
#pragma unroll
        for (int j = 0; j < KEEP; ++j) {
            sum += keep[j] & 0xffff;
            sum_hi += (int)((unsigned)keep[j] >> 16);
            most = max(most, keep[j] & 0xffff);
        }
    } else {
        for (int c = c0; c < c1; ++c) {
            const int v = amm_ld_l2(&count[c]);
            sum += v & 0xffff;
            sum_hi += (int)((unsigned)v >> 16);
            most = max(most, v & 0xffff);
        }
    }
    part[t] = sum;
    part_hi[t] = sum_hi;
    s_most[t] = most;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int add = t >= off ? part[t - off] : 0, add_hi = t >= off ? part_hi[t - off] : 0;
        __syncthreads();
        part[t] += add;
        part_hi[t] += add_hi;
        __syncthreads();
    }
    int run = part[t] - sum, run_hi = part_hi[t] - sum_hi;
    if (kept) {
#pragma unroll
        for (int j = 0; j < KEEP; ++j)
            if (c0 + j < c1) {
                start[c0 + j] = run;
                start_hi[c0 + j] = run_hi;
                count[c0 + j] = 0;
                run += keep[j] & 0xffff;
                run_hi += (int)((unsigned)keep[j] >> 16);
            }
    } else {
        for (int c = c0; c < c1; ++c) {
            const int v = amm_ld_l2(&count[c]);
            start[c] = run;
            start_hi[c] = run_hi;
            count[c] = 0;
            run += v & 0xffff;
            run_hi += (int)((unsigned)v >> 16);
        }
    }
    if (t == 255) {
        start[ncell] = part[255];
        start_hi[ncell] = part_hi[255];
    }
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (t < off) s_most[t] = max(s_most[t], s_most[t + off]);
        __syncthreads();
    }
    return s_most[0];
}

// run by ONE 256-thread block (the last block of the histogram kernel): exclusive scan of count[0..ncell) ->
// start[0..ncell], count <- 0 (fill <- start when given); returns the largest count.  Thread t scans the contiguous segment [t*per, (t+1)*per).
__device__ __forceinline__ int amm_block_scan_counts(int ncell, int *count, int *start, int *fill = nullptr) {
    __shared__ int part[256];
    __shared__ int s_most[256];
    const int t = threadIdx.x;
    const int per = (ncell + 255) / 256;
    const int c0 = min(t * per, ncell), c1 = min(c0 + per, ncell);
    // the counts were written by device-scope atomics and are read past the XCD's L2 (about a microsecond per load):
    // up to 32 per thread are fetched in one go and kept in registers for the second sweep
    constexpr int KEEP = 32;
    int keep[KEEP];
    const bool kept = per <= KEEP;
    int sum = 0, most = 0;
    if (kept) {
#pragma unroll
        for (int j = 0; j < KEEP; ++j) keep[j] = (c0 + j < c1) ? amm_ld_l2(&count[c0 + j]) : 0;
#pragma unroll
        for (int j = 0; j < KEEP; ++j) {
            sum += keep[j];
            most = max(most, keep[j]);
        }
    } else {
        for (int c = c0; c < c1; ++c) {
            const int v = amm_ld_l2(&count[c]);
            sum += v;
            most = max(most, v);
        }
    }
    part[t] = sum;
    s_most[t] = most;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int add = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    int run = part[t] - sum;
    if (kept) {
#pragma unroll
        for (int j = 0; j < KEEP; ++j)
            if (c0 + j < c1) {
                start[c0 + j] = run;
                if (fill) fill[c0 + j] = run;
                count[c0 + j] = 0;
                run += keep[j];
            }
    } else {
        for (int c = c0; c < c1; ++c) {
            const int v = amm_ld_l2(&count[c]);
            start[c] = run;
            if (fill) fill[c] = run;
            count[c] = 0;
            run += v;
        }
    }
    if (t == 255) start[ncell] = part[255];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (t < off) s_most[t] = max(s_most[t], s_most[t + off]);
        __syncthreads();
    }
    return s_most[0];
}
