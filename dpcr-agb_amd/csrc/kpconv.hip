// kpconv.hip — rigid kernel-point convolution (KPConv.forward, torch_points3d/modules/KPConv/blocks.py:264-400)
// split the MI355X way:
//   (1) gather fwd : wf[n,k,:] = sum_h max(0, 1 - |(s[idx[n,h]] - q[n]) - kp[k]| / extent) * x[idx[n,h],:]
//       one fused pass over the neighbour matrix — the reference materialises [N,H,15,3] differences, [N,H,15]
//       weights and [N,H,Cin] gathered features in HBM; here the neighbour rows are read once as coalesced 64-byte
//       pieces and the walk stops after the last real neighbour (rows are distance-sorted, padding sits at the end).
//   (2) the dense contraction out = wf[N, 15*Cin] @ W[15*Cin, Cout]: the identity-map kernels of spconv.hip.
//   (3) gather bwd : dx[idx[n,h],:] += sum_k infl(n,h,k) * dwf[n,k,:]   (fp32 atomics, 64-byte pieces) — only for
//       layers whose query and support sets differ (strided blocks); a layer on ONE point set with a symmetric
//       neighbour relation runs its backward as pass (1) on dy with mirrored kernel points (kpconv_ops.py).
// plus the max-pooled shortcut of strided blocks (blocks.py:98-114: the zero "shadow" row takes part in the max).
#include "agb_common.h"
#include <float.h>

#define KP_MAX 16

// ------------------------------------------------------------------ the two gather passes on the matrix cores
// Per query row the gather is a small matrix product over the row's neighbours h:
//     wf[n]  (K x Cin) = W^T (K x H) . X (H x Cin)        W[h,k] = influence of kernel point k on neighbour h,
//     g      (H x Cin) = W   (H x K) . dwf[n] (K x Cin)   X[h,:]  = x[idx[n,h],:],  dx[idx[n,h],:] += g[h,:]
// and v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulation) has exactly the operand shape for it with K = 15
// kernel points padded to 16: in the forward pass lane (k, j) of a wave evaluates ONE influence — kernel point k, the
// j-th neighbour of a block of four — which IS its A operand, and loads x[idx[h_j], c] as its B operand (a coalesced
// 64-byte piece per neighbour); nothing is exchanged between lanes.  One wave per (query row, chunk of 16*NC channels).
//
// History (profiles/r02_kpconv_pmc.txt): the first form gave every lane one channel and broadcast the 15 influences of
// a neighbour to the row's lanes — 16 cross-lane moves + 16 multiply-adds + an influence evaluation per neighbour on
// every lane, 14 wave instructions per row-neighbour at Cin = 16; with ds_bpermute the LDS pipe was 57 % busy, with DPP
// row_share moves the VALU was the limit (fwd 5.9 -> 4.5 ms per KPConv step).  This form needs about 5 (-> 2.2 ms).
// Combining the backward scatter in an LDS table (32 consecutive rows share ~75 % of their support rows) was built and
// measured SLOWER: ds_add_f32 retires one 64-lane instruction per ~194 cycles on gfx950 (ds_add_u32: 4.6,
// read+add+write: 14.6 — tools/micro/lds_atomic_bench.hip), so the LDS pipe became the bound.
typedef float kp_f32x4 __attribute__((ext_vector_type(4)));

// max(0, 1 - |rel - kp| / extent) (blocks.py:339-350).  `inv_ext` = 1 / extent: the quotient becomes one multiply-add
// and the root one v_sqrt_f32 (1 ulp) instead of the ~18 instructions of IEEE division + correctly rounded sqrtf — this
// function is evaluated once per (neighbour, kernel point) and was a third of the gather's instructions.  The
// influence moves by < 2e-7 absolute (outputs: 1e-4 bar of the parity tests, floor re-measured in DESIGN.md section 3).
__device__ __forceinline__ float kp_influence(float rx, float ry, float rz, float kx, float ky, float kz, float inv_ext) {
    const float dx = rx - kx, dy = ry - ky, dz = rz - kz;
    const float d = __builtin_amdgcn_sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
    return fmaxf(fmaf(-d, inv_ext, 1.f), 0.f);
}

template <int NC>
__global__ __launch_bounds__(256) void k_kpconv_gather_mm_fwd(const float* __restrict__ q, const float* __restrict__ s,
                                                              const int32_t* __restrict__ idx, int H, int Ns,
                                                              const float* __restrict__ x, int ldx,
                                                              const float* __restrict__ kp, int K, float inv_ext,
                                                              float* __restrict__ wf, int N, int Cin, int chunks,
                                                              const int32_t* __restrict__ row_ptr) {
    const int lane = threadIdx.x & 63;
    const long long item = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= (long long)N * chunks) return;            // (whole waves leave: no barrier in this kernel)
    const int n = (int)(item / chunks);
    const int cbase = (int)(item % chunks) * (16 * NC);
    const int m = lane & 15, g = lane >> 4;               // A: (kernel point m, neighbour g)   B/D: (.., channel m)
    const float kx = m < K ? kp[3 * m] : 0.f, ky = m < K ? kp[3 * m + 1] : 0.f, kz = m < K ? kp[3 * m + 2] : 0.f;
    const float qx = q[3 * (long long)n], qy = q[3 * (long long)n + 1], qz = q[3 * (long long)n + 2];
    kp_f32x4 acc[NC];
#pragma unroll
    for (int t = 0; t < NC; ++t) acc[t] = kp_f32x4{0.f, 0.f, 0.f, 0.f};
    // row_ptr != NULL: ragged rows idx[row_ptr[n] .. row_ptr[n + 1]) cut at H entries (the neighbourhood limit); else
    // the padded matrix idx[n * H + h]
    const int32_t* row = row_ptr ? idx + row_ptr[n] : idx + (long long)n * H;
    const int Hn = row_ptr ? min(H, row_ptr[n + 1] - row_ptr[n]) : H;
    for (int h0 = 0; h0 < Hn; h0 += 64) {
        const int myid = h0 + lane < Hn ? row[h0 + lane] : Ns;
        const unsigned long long valid = __ballot(myid >= 0 && myid < Ns);
        if (valid == 0ull) break;                          // rows are sorted by distance: the rest is shadow padding
        // two blocks of four neighbours per trip: both blocks' loads are requested before either is used
        const int nblk = (64 - __builtin_clzll(valid) + 7) >> 3;
        for (int b = 0; b < nblk; ++b) {
            int id[2];
            bool live[2];
            float rx[2], ry[2], rz[2], xb[2][NC];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                id[u] = __shfl(myid, 8 * b + 4 * u + g, 64);
                live[u] = id[u] >= 0 && id[u] < Ns;
                const long long ic = live[u] ? id[u] : 0;
                rx[u] = s[3 * ic] - qx, ry[u] = s[3 * ic + 1] - qy, rz[u] = s[3 * ic + 2] - qz;
#pragma unroll
                for (int t = 0; t < NC; ++t) {
                    const int c = cbase + 16 * t + m;
                    xb[u][t] = (live[u] && c < Cin) ? x[ic * ldx + c] : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float a = (live[u] && m < K) ? kp_influence(rx[u], ry[u], rz[u], kx, ky, kz, inv_ext) : 0.f;
#pragma unroll
                for (int t = 0; t < NC; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xb[u][t], acc[t], 0, 0, 0);
            }
        }
        if (valid != ~0ull) break;
    }
#pragma unroll
    for (int t = 0; t < NC; ++t) {
        const int c = cbase + 16 * t + m;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = 4 * g + i;
            if (k < K && c < Cin) wf[((long long)n * K + k) * Cin + c] = acc[t][i];
        }
    }
}

// backward: blocks of 16 neighbours; lane (h, j) evaluates the influences of kernel points j, 4+j, 8+j, 12+j on
// neighbour h (its A operands of the four k-steps); dwf[n] stays in registers as the B operands for the whole row.
template <int NC>
__global__ __launch_bounds__(256) void k_kpconv_gather_mm_bwd(const float* __restrict__ q, const float* __restrict__ s,
                                                              const int32_t* __restrict__ idx, int H, int Ns,
                                                              const float* __restrict__ kp, int K, float inv_ext,
                                                              const float* __restrict__ dwf, float* __restrict__ dx,
                                                              int ldx, int N, int Cin, int chunks,
                                                              const int32_t* __restrict__ row_ptr) {
    const int lane = threadIdx.x & 63;
    const long long item = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= (long long)N * chunks) return;
    const int n = (int)(item / chunks);
    const int cbase = (int)(item % chunks) * (16 * NC);
    const int m = lane & 15, g = lane >> 4;               // A: (neighbour m, kernel point 4*step+g)  B/D: (.., channel m)
    float kx[4], ky[4], kz[4];
    float db[4][NC];
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        const int k = 4 * st + g;
        kx[st] = k < K ? kp[3 * k] : 0.f;
        ky[st] = k < K ? kp[3 * k + 1] : 0.f;
        kz[st] = k < K ? kp[3 * k + 2] : 0.f;
#pragma unroll
        for (int t = 0; t < NC; ++t) {
            const int c = cbase + 16 * t + m;
            db[st][t] = (k < K && c < Cin) ? dwf[((long long)n * K + k) * Cin + c] : 0.f;
        }
    }
    const float qx = q[3 * (long long)n], qy = q[3 * (long long)n + 1], qz = q[3 * (long long)n + 2];
    // row_ptr != NULL: ragged rows idx[row_ptr[n] .. row_ptr[n + 1]) cut at H entries (the neighbourhood limit); else
    // the padded matrix idx[n * H + h]
    const int32_t* row = row_ptr ? idx + row_ptr[n] : idx + (long long)n * H;
    const int Hn = row_ptr ? min(H, row_ptr[n + 1] - row_ptr[n]) : H;
    for (int h0 = 0; h0 < Hn; h0 += 64) {
        const int myid = h0 + lane < Hn ? row[h0 + lane] : Ns;
        const unsigned long long valid = __ballot(myid >= 0 && myid < Ns);
        if (valid == 0ull) break;
        const int nblk = (64 - __builtin_clzll(valid) + 15) >> 4;
        for (int b = 0; b < nblk; ++b) {
            const int id = __shfl(myid, 16 * b + m, 64);
            const bool live = id >= 0 && id < Ns;
            const long long ic = live ? id : 0;
            const float rx = s[3 * ic] - qx, ry = s[3 * ic + 1] - qy, rz = s[3 * ic + 2] - qz;
            float a[4];
#pragma unroll
            for (int st = 0; st < 4; ++st)
                a[st] = (live && 4 * st + g < K) ? kp_influence(rx, ry, rz, kx[st], ky[st], kz[st], inv_ext) : 0.f;
            int ido[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) ido[i] = __shfl(myid, 16 * b + 4 * g + i, 64);
#pragma unroll
            for (int t = 0; t < NC; ++t) {
                kp_f32x4 d = kp_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int st = 0; st < 4; ++st) d = __builtin_amdgcn_mfma_f32_16x16x4f32(a[st], db[st][t], d, 0, 0, 0);
                const int c = cbase + 16 * t + m;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (c < Cin && ido[i] >= 0 && ido[i] < Ns) atomicAdd(&dx[(long long)ido[i] * ldx + c], d[i]);
            }
        }
        if (valid != ~0ull) break;
    }
}

// ------------------------------------------------------------------ max-pooled shortcut
// y[n,c] = max_h xpad[idx[n,h], c] with xpad's extra row = 0; arg = winning row or -1 (shadow).
// One thread per (row, 4 channels): 16-B gathers; the walk stops at the first shadow neighbour (rows are sorted by
// distance with the padding at the end — the padded matrix is up to 265 wide for ~20 valid entries, and the plain
// one-thread-per-element walk over all of it was 2.7 ms of a 32 ms training step).
__global__ __launch_bounds__(256) void k_kp_maxpool_fwd4(const float* __restrict__ x, int ldx,
                                                         const int32_t* __restrict__ idx, int H, int Ns,
                                                         float* __restrict__ y, int32_t* __restrict__ arg, int N, int C4,
                                                         const int32_t* __restrict__ row_ptr,
                                                         const int32_t* __restrict__ width_dev) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int n = (int)(t / C4), c = (int)(t % C4) * 4;
    if (n >= N) return;
    float best[4] = {-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
    int bi[4] = {-2, -2, -2, -2};
    // ragged rows: a row shorter than the padded matrix would be (min(H, *width_dev) columns: the batch-wide maximum, cut
    // at the neighbourhood limit) has shadow entries behind it, and the zero shadow row takes part in its max
    const int32_t* row = row_ptr ? idx + row_ptr[n] : idx + (long long)n * H;
    const int Hn = row_ptr ? min(H, row_ptr[n + 1] - row_ptr[n]) : H;
    bool shadow = row_ptr ? Hn < min(H, *width_dev) : false;
    // four neighbours per trip, their rows requested before any is compared (one dependent load per neighbour made this
    // kernel wait 83 % of its wave cycles); the order of the comparisons — and so the winner among equal values — is unchanged
    int h = 0;
    bool stop = false;                                   // a shadow entry was met: every later entry is padding too
    for (; h + 4 <= Hn && !stop; h += 4) {
        int id[4];
        float4 v4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) id[u] = row[h + u];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool ok = id[u] < Ns && id[u] >= 0;
            v4[u] = *reinterpret_cast<const float4*>(x + (long long)(ok ? id[u] : 0) * ldx + c);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!stop) {
                if (id[u] >= Ns || id[u] < 0) {
                    shadow = stop = true;
                } else {
                    const float v[4] = {v4[u].x, v4[u].y, v4[u].z, v4[u].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (v[j] > best[j]) { best[j] = v[j]; bi[j] = id[u]; }
                }
            }
        }
    }
    for (; h < Hn && !stop; ++h) {
        const int id = row[h];
        if (id >= Ns || id < 0) { shadow = stop = true; break; }
        const float4 v4 = *reinterpret_cast<const float4*>(x + (long long)id * ldx + c);
        const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (v[j] > best[j]) { best[j] = v[j]; bi[j] = id; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // the zero shadow row takes part in the max where the row has padding (blocks.py:98-114), and is the only
        // candidate of a row without any neighbour; a real neighbour wins ties (it comes first in the matrix)
        if ((shadow && 0.f > best[j]) || bi[j] == -2) { best[j] = 0.f; bi[j] = -1; }
    }
    *reinterpret_cast<float4*>(y + (long long)n * (C4 * 4) + c) = make_float4(best[0], best[1], best[2], best[3]);
    *reinterpret_cast<int4*>(arg + (long long)n * (C4 * 4) + c) = make_int4(bi[0], bi[1], bi[2], bi[3]);
}

__global__ void k_kp_maxpool_fwd(const float* __restrict__ x, int ldx, const int32_t* __restrict__ idx, int H,
                                 int Ns, float* __restrict__ y, int32_t* __restrict__ arg, int N, int C) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int n = (int)(t / C), c = (int)(t % C);
    if (n >= N) return;
    float best = -FLT_MAX;
    int bi = -2;
    for (int h = 0; h < H; ++h) {
        int id = idx[(long long)n * H + h];
        if (id >= Ns || id < 0) {  // shadow neighbour: the zero row of the padded feature matrix
            if (0.f > best) { best = 0.f; bi = -1; }
            break;                 // sorted rows: the rest is padding
        }
        float v = x[(long long)id * ldx + c];
        if (v > best) { best = v; bi = id; }
    }
    if (bi == -2) { best = 0.f; bi = -1; }
    y[(long long)n * C + c] = best;
    arg[(long long)n * C + c] = bi;
}

__global__ void k_kp_maxpool_bwd(const float* __restrict__ dy, const int32_t* __restrict__ arg, float* dx, int ldx,
                                 int N, int C) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int n = (int)(t / C), c = (int)(t % C);
    if (n >= N) return;
    int a = arg[(long long)n * C + c];
    if (a >= 0) atomicAdd(&dx[(long long)a * ldx + c], dy[(long long)n * C + c]);
}

template <bool BWD>
static int launch_gather(const float* q, const float* s, const int32_t* idx, int H, int Ns, const float* x, int ldx,
                         const float* kp, int K, float ext, float* wf, const float* dwf, float* dx, int N, int Cin,
                         hipStream_t st, const int32_t* row_ptr = nullptr) {
    // one wave per (row, chunk of 16*NC channels)
    const int nc = Cin <= 16 ? 1 : Cin <= 32 ? 2 : 4;
    const int chunks = agb_cdiv(Cin, 16 * nc);
    const float inv_ext = 1.f / ext;
    const dim3 grid(agb_cdiv((long long)N * chunks, 4)), block(256);
#define AGB_KP_MM(NC)                                                                                                 \
    do {                                                                                                              \
        if (BWD) hipLaunchKernelGGL((k_kpconv_gather_mm_bwd<NC>), grid, block, 0, st, q, s, idx, H, Ns, kp, K, inv_ext, \
                                    dwf, dx, ldx, N, Cin, chunks, row_ptr);                                           \
        else hipLaunchKernelGGL((k_kpconv_gather_mm_fwd<NC>), grid, block, 0, st, q, s, idx, H, Ns, x, ldx, kp, K,    \
                                inv_ext, wf, N, Cin, chunks, row_ptr);                                                \
    } while (0)
    if (nc == 1) AGB_KP_MM(1);
    else if (nc == 2) AGB_KP_MM(2);
    else AGB_KP_MM(4);
#undef AGB_KP_MM
    return AGB_OK;
}

// =============================================================== C ABI
extern "C" {

// wf: float[N, K, Cin] out.  idx: int32[N, H], entries >= Ns are shadow neighbours.
int agb_kpconv_gather_fwd(const float* q, const float* s, const int32_t* idx, int H, int Ns, const float* x, int ldx,
                          const float* kp, int K, float extent, float* wf, int N, int Cin, void* stream) {
    AGB_CHECK_ARG(K >= 1 && K <= KP_MAX, "agb_kpconv_gather_fwd: %d kernel points (max %d)", K, KP_MAX);
    if (N == 0) return AGB_OK;
    int rc = launch_gather<false>(q, s, idx, H, Ns, x, ldx, kp, K, extent, wf, nullptr, nullptr, N, Cin,
                                  (hipStream_t)stream);
    if (rc) return rc;
    AGB_CHECK_LAUNCH("agb_kpconv_gather_fwd");
    return AGB_OK;
}

// dx: float[Ns, ldx], zero-filled by the caller (accumulated with fp32 atomics).
int agb_kpconv_gather_bwd(const float* q, const float* s, const int32_t* idx, int H, int Ns, const float* dwf,
                          const float* kp, int K, float extent, float* dx, int ldx, int N, int Cin, void* stream) {
    AGB_CHECK_ARG(K >= 1 && K <= KP_MAX, "agb_kpconv_gather_bwd: %d kernel points (max %d)", K, KP_MAX);
    if (N == 0) return AGB_OK;
    int rc = launch_gather<true>(q, s, idx, H, Ns, nullptr, ldx, kp, K, extent, nullptr, dwf, dx, N, Cin,
                                 (hipStream_t)stream);
    if (rc) return rc;
    AGB_CHECK_LAUNCH("agb_kpconv_gather_bwd");
    return AGB_OK;
}

int agb_kp_maxpool_fwd(const float* x, int ldx, const int32_t* idx, int H, int Ns, float* y, int32_t* argmax, int N,
                       int C, void* stream) {
    if (N == 0) return AGB_OK;
    if (C % 4 == 0 && ldx % 4 == 0)
        hipLaunchKernelGGL(k_kp_maxpool_fwd4, dim3(agb_cdiv((long long)N * (C / 4), 256)), dim3(256), 0,
                           (hipStream_t)stream, x, ldx, idx, H, Ns, y, argmax, N, C / 4, (const int32_t*)nullptr,
                           (const int32_t*)nullptr);
    else
        hipLaunchKernelGGL(k_kp_maxpool_fwd, dim3(agb_cdiv((long long)N * C, 256)), dim3(256), 0, (hipStream_t)stream, x,
                           ldx, idx, H, Ns, y, argmax, N, C);
    AGB_CHECK_LAUNCH("agb_kp_maxpool_fwd");
    return AGB_OK;
}

// The same three passes on RAGGED neighbour rows (row_ptr int32[N + 1], indices int32[total] from agb_ball_query_fill_csr):
// row n = indices[row_ptr[n] .. row_ptr[n + 1]) cut at `limit` entries (the reference's neighborhood_limits crop of the padded
// matrix; INT_MAX: none).  max_count_dev (device int32): width of the padded matrix the reference would have built — a row
// shorter than min(limit, *max_count_dev) has shadow neighbours, whose zero row takes part in the max pool (blocks.py:98-114).
int agb_kpconv_gather_fwd_csr(const float* q, const float* s, const int32_t* row_ptr, const int32_t* indices, int limit,
                              int Ns, const float* x, int ldx, const float* kp, int K, float extent, float* wf, int N,
                              int Cin, void* stream) {
    AGB_CHECK_ARG(K >= 1 && K <= KP_MAX, "agb_kpconv_gather_fwd_csr: %d kernel points (max %d)", K, KP_MAX);
    AGB_CHECK_ARG(limit >= 1, "agb_kpconv_gather_fwd_csr: limit %d", limit);
    if (N == 0) return AGB_OK;
    AGB_CHECK_ARG(row_ptr != nullptr, "agb_kpconv_gather_fwd_csr: row_ptr required");
    int rc = launch_gather<false>(q, s, indices, limit, Ns, x, ldx, kp, K, extent, wf, nullptr, nullptr, N, Cin,
                                  (hipStream_t)stream, row_ptr);
    if (rc) return rc;
    AGB_CHECK_LAUNCH("agb_kpconv_gather_fwd_csr");
    return AGB_OK;
}

int agb_kpconv_gather_bwd_csr(const float* q, const float* s, const int32_t* row_ptr, const int32_t* indices, int limit,
                              int Ns, const float* dwf, const float* kp, int K, float extent, float* dx, int ldx, int N,
                              int Cin, void* stream) {
    AGB_CHECK_ARG(K >= 1 && K <= KP_MAX, "agb_kpconv_gather_bwd_csr: %d kernel points (max %d)", K, KP_MAX);
    AGB_CHECK_ARG(limit >= 1, "agb_kpconv_gather_bwd_csr: limit %d", limit);
    if (N == 0) return AGB_OK;
    AGB_CHECK_ARG(row_ptr != nullptr, "agb_kpconv_gather_bwd_csr: row_ptr required");
    int rc = launch_gather<true>(q, s, indices, limit, Ns, nullptr, ldx, kp, K, extent, nullptr, dwf, dx, N, Cin,
                                 (hipStream_t)stream, row_ptr);
    if (rc) return rc;
    AGB_CHECK_LAUNCH("agb_kpconv_gather_bwd_csr");
    return AGB_OK;
}

int agb_kp_maxpool_fwd_csr(const float* x, int ldx, const int32_t* row_ptr, const int32_t* indices, int limit,
                           const int32_t* max_count_dev, int Ns, float* y, int32_t* argmax, int N, int C, void* stream) {
    AGB_CHECK_ARG(C % 4 == 0 && ldx % 4 == 0 && limit >= 1, "agb_kp_maxpool_fwd_csr: C (%d), ldx multiples of 4", C);
    if (N == 0) return AGB_OK;
    AGB_CHECK_ARG(row_ptr && max_count_dev, "agb_kp_maxpool_fwd_csr: row_ptr and max_count_dev required");
    hipLaunchKernelGGL(k_kp_maxpool_fwd4, dim3(agb_cdiv((long long)N * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, x,
                       ldx, indices, limit, Ns, y, argmax, N, C / 4, row_ptr, max_count_dev);
    AGB_CHECK_LAUNCH("agb_kp_maxpool_fwd_csr");
    return AGB_OK;
}

// dx zero-filled by the caller
int agb_kp_maxpool_bwd(const float* dy, const int32_t* argmax, float* dx, int ldx, int N, int C, void* stream) {
    if (N == 0) return AGB_OK;
    hipLaunchKernelGGL(k_kp_maxpool_bwd, dim3(agb_cdiv((long long)N * C, 256)), dim3(256), 0, (hipStream_t)stream, dy,
                       argmax, dx, ldx, N, C);
    AGB_CHECK_LAUNCH("agb_kp_maxpool_bwd");
    return AGB_OK;
}

}  // extern "C"
