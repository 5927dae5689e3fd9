// scan.h — deterministic exclusive scan over int32 (three small kernels), shared by coords.hip and kpindex.hip.
#pragma once
#include "agb_common.h"

#ifndef TPB
#define TPB 256
#endif

// ---- exclusive scan over int32 flags (three small kernels, deterministic) ----
#define SCAN_ITEMS 4
#define SCAN_BLOCK (TPB * SCAN_ITEMS)

static __device__ __forceinline__ int block_exclusive_scan(int v, int* total) {
    // 256 threads = 4 waves; wave scan by shuffles, then scan of 4 wave sums
    __shared__ int wsum[TPB / 64];
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int j = 0; j < TPB / 64; ++j) {
        int s = wsum[j];
        if (j < w) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

static __global__ void k_scan_block_sums(const int32_t* __restrict__ in, int n, int32_t* block_sums) {
    int base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_ITEMS;
    int s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j)
        if (base + j < n) s += in[base + j];
    int tot;
    block_exclusive_scan(s, &tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

static __global__ void k_scan_sums(int32_t* block_sums, int nb, int32_t* total_out) {
    // single workgroup; sequential over chunks of 256 block sums
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nb; b0 += TPB) {
        int i = b0 + threadIdx.x;
        int v = i < nb ? block_sums[i] : 0;
        int tot;
        int ex = block_exclusive_scan(v, &tot);
        int c = carry;
        if (i < nb) block_sums[i] = c + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry = c + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = carry;
}

static __global__ void k_scan_apply(const int32_t* __restrict__ in, int n, const int32_t* __restrict__ block_sums,
                             int32_t* out) {
    int base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS];
    int s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        v[j] = (base + j < n) ? in[base + j] : 0;
        s += v[j];
    }
    int tot;
    int ex = block_exclusive_scan(s, &tot) + block_sums[blockIdx.x];
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        if (base + j < n) out[base + j] = ex;
        ex += v[j];
    }
}


// out[i] = sum(in[0..i-1]); *total_out = sum(in); scratch: int32[agb_cdiv(n, SCAN_BLOCK) + 8]
static inline void agb_launch_exclusive_scan(const int32_t* in, int n, int32_t* out, int32_t* scratch,
                                             int32_t* total_out, hipStream_t s) {
    int nb = agb_cdiv(n, SCAN_BLOCK);
    hipLaunchKernelGGL(k_scan_block_sums, dim3(nb), dim3(TPB), 0, s, in, n, scratch);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(TPB), 0, s, scratch, nb, total_out);
    hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(TPB), 0, s, in, n, scratch, out);
}
