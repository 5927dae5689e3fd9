// kpconv.hip — rigid kernel-point convolution (KPConv.forward, torch_points3d/modules/KPConv/blocks.py:264-400)
// split the MI355X way:
//   (1) k_kpconv_gather_fwd : wf[n,k,:] = sum_h max(0, 1 - |(s[idx[n,h]] - q[n]) - kp[k]| / extent) * x[idx[n,h],:]
//       one fused pass over the neighbour matrix — the reference materialises [N,H,15,3] differences, [N,H,15]
//       weights and [N,H,Cin] gathered features in HBM; here neighbour rows are read once as coalesced
//       Cin-wide pieces, the 15 influences of a neighbour are computed by 15 lanes and shuffled to the row's lanes,
//       and the loop stops at the first shadow neighbour (rows are distance-sorted, padding sits at the end).
//   (2) the dense contraction out = wf[N, 15*Cin] @ W[15*Cin, Cout] is a plain GEMM (MFMA, via rocBLAS).
//   (3) k_kpconv_gather_bwd : dx[idx[n,h],:] += sum_k infl(n,h,k) * dwf[n,k,:]   (fp32 atomics, 4*LPR-byte pieces)
// plus the max-pooled shortcut of strided blocks (blocks.py:98-114: the zero "shadow" row takes part in the max).
#include "agb_common.h"
#include <float.h>

#define KP_MAX 16

template <int LPR, int CPL>
__device__ __forceinline__ void kp_influences(const float rel[3], const float* __restrict__ s_kp, int K, float ext,
                                              int lir, int lane, float w[KP_MAX]) {
    if constexpr (LPR >= 16) {
        // lane (lir % 16) of every 16-lane group evaluates one kernel point, then the values are shuffled around
        int k = lir & 15;
        float mine = 0.f;
        if (k < K) {
            float dx = rel[0] - s_kp[3 * k], dy = rel[1] - s_kp[3 * k + 1], dz = rel[2] - s_kp[3 * k + 2];
            float d = sqrtf((dx * dx + dy * dy) + dz * dz);
            mine = fmaxf(1.f - d / ext, 0.f);
        }
        int base = lane & ~15;
#pragma unroll
        for (int j = 0; j < KP_MAX; ++j) w[j] = __shfl(mine, base + j, 64);
    } else {
#pragma unroll
        for (int j = 0; j < KP_MAX; ++j) {
            w[j] = 0.f;
            if (j < K) {
                float dx = rel[0] - s_kp[3 * j], dy = rel[1] - s_kp[3 * j + 1], dz = rel[2] - s_kp[3 * j + 2];
                float d = sqrtf((dx * dx + dy * dy) + dz * dz);
                w[j] = fmaxf(1.f - d / ext, 0.f);
            }
        }
    }
}

// LPR lanes per query row, each lane owns channels lir, lir+LPR, ... (CPL of them)
template <int LPR, int CPL, bool BWD>
__global__ __launch_bounds__(256) void k_kpconv_gather(const float* __restrict__ q, const float* __restrict__ s,
                                                       const int32_t* __restrict__ idx, int H, int Ns,
                                                       const float* __restrict__ x, int ldx,
                                                       const float* __restrict__ kp, int K, float ext,
                                                       float* __restrict__ wf,         // fwd: out [N,K,Cin]
                                                       const float* __restrict__ dwf,  // bwd: in  [N,K,Cin]
                                                       float* __restrict__ dx,         // bwd: out [Ns,Cin] (atomics)
                                                       int N, int Cin) {
    __shared__ float s_kp[3 * KP_MAX];
    if (threadIdx.x < 3 * K) s_kp[threadIdx.x] = kp[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int lir = threadIdx.x % LPR;
    const int n = (blockIdx.x * 256 + threadIdx.x) / LPR;
    const bool row_ok = n < N;
    const int nn = row_ok ? n : 0;
    const float qx = q[3 * (long long)nn], qy = q[3 * (long long)nn + 1], qz = q[3 * (long long)nn + 2];

    float acc[KP_MAX][CPL];
#pragma unroll
    for (int k = 0; k < KP_MAX; ++k)
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            acc[k][j] = 0.f;
            if (BWD && row_ok && k < K) {
                int c = lir + LPR * j;
                if (c < Cin) acc[k][j] = dwf[((long long)n * K + k) * Cin + c];
            }
        }

    // The walk over a row's neighbours is a chain of dependent loads (index -> support position / feature row): one
    // neighbour per iteration left every wave waiting ~1.5 us per step with nothing else in flight (0.95 ms for the
    // 506 k rows of the first level at ~20 neighbours per row).  Neighbours are therefore taken NB at a time: one lane per
    // neighbour loads its index (a coalesced run of the row), then the NB positions and feature pieces are all requested
    // before the first one is used.  All lanes of a wave walk the same number of blocks (shuffles inside); rows are sorted
    // by distance with the shadow padding at the end, so the walk stops at the first block whose FIRST entry is a shadow
    // neighbour in every row of the wave.
    constexpr int NB = 8;
    const int grp = lane & ~15;
    for (int h0 = 0; h0 < H; h0 += NB) {
        int myid = Ns;
        if (row_ok && (lir & 15) < NB && h0 + (lir & 15) < H) myid = idx[(long long)n * H + h0 + (lir & 15)];
        int ids[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) ids[j] = __shfl(myid, grp + j, 64);
        if (__ballot(ids[0] < Ns && ids[0] >= 0) == 0ull) break;
        float rel[NB][3];
        float xv[NB][CPL];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const bool live = ids[j] < Ns && ids[j] >= 0;
            const long long ic = live ? ids[j] : 0;      // clamped: the loads are unconditional, the use is masked
            rel[j][0] = s[3 * ic] - qx;
            rel[j][1] = s[3 * ic + 1] - qy;
            rel[j][2] = s[3 * ic + 2] - qz;
            if (!BWD) {
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    const int ch = lir + LPR * c;
                    xv[j][c] = ch < Cin ? x[ic * ldx + ch] : 0.f;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const bool live = ids[j] < Ns && ids[j] >= 0;
            if (__ballot(live) == 0ull) break;           // (uniform: sorted rows, nothing live further on in this block)
            float w[KP_MAX];
            kp_influences<LPR, CPL>(rel[j], s_kp, K, ext, lir, lane, w);
            if (!live) continue;
            if (!BWD) {
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
#pragma unroll
                    for (int k = 0; k < KP_MAX; ++k) acc[k][c] += w[k] * xv[j][c];
                }
            } else {
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    const int ch = lir + LPR * c;
                    float g = 0.f;
#pragma unroll
                    for (int k = 0; k < KP_MAX; ++k) g += w[k] * acc[k][c];
                    if (ch < Cin) atomicAdd(&dx[(long long)ids[j] * ldx + ch], g);
                }
            }
        }
    }
    if (!BWD && row_ok) {
#pragma unroll
        for (int k = 0; k < KP_MAX; ++k)
            if (k < K)
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    int c = lir + LPR * j;
                    if (c < Cin) wf[((long long)n * K + k) * Cin + c] = acc[k][j];
                }
    }
}

// ------------------------------------------------------------------ max-pooled shortcut
// y[n,c] = max_h xpad[idx[n,h], c] with xpad's extra row = 0; arg = winning row or -1 (shadow).
// One thread per (row, 4 channels): 16-B gathers; the walk stops at the first shadow neighbour (rows are sorted by
// distance with the padding at the end — the padded matrix is up to 265 wide for ~20 valid entries, and the plain
// one-thread-per-element walk over all of it was 2.7 ms of a 32 ms training step).
__global__ __launch_bounds__(256) void k_kp_maxpool_fwd4(const float* __restrict__ x, int ldx,
                                                         const int32_t* __restrict__ idx, int H, int Ns,
                                                         float* __restrict__ y, int32_t* __restrict__ arg, int N, int C4) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int n = (int)(t / C4), c = (int)(t % C4) * 4;
    if (n >= N) return;
    float best[4] = {-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
    int bi[4] = {-2, -2, -2, -2};
    const int32_t* row = idx + (long long)n * H;
    bool shadow = false;
    for (int h = 0; h < H; ++h) {
        const int id = row[h];
        if (id >= Ns || id < 0) { shadow = true; break; }      // every later entry is padding too
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
                         hipStream_t st) {
#define AGB_KP_LAUNCH(LPR, CPL)                                                                                      \
    hipLaunchKernelGGL((k_kpconv_gather<LPR, CPL, BWD>), dim3(agb_cdiv((long long)N * LPR, 256)), dim3(256), 0, st, \
                       q, s, idx, H, Ns, x, ldx, kp, K, ext, wf, dwf, dx, N, Cin)
    // (Cin <= 4 — the 3-feature input layer — takes the 16-lane form too: one lane per kernel point evaluates its
    // influence once per neighbour; with 4 lanes per row every lane evaluated all 15 square roots itself)
    if (Cin <= 16) AGB_KP_LAUNCH(16, 1);
    else if (Cin <= 32) AGB_KP_LAUNCH(16, 2);
    else if (Cin <= 64) AGB_KP_LAUNCH(64, 1);
    else if (Cin <= 128) AGB_KP_LAUNCH(64, 2);
    else if (Cin <= 256) AGB_KP_LAUNCH(64, 4);
    else {
        agb_set_error("agb_kpconv_gather: Cin %d > 256 is not supported", Cin);
        return AGB_EUNSUPPORTED;
    }
#undef AGB_KP_LAUNCH
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
                           (hipStream_t)stream, x, ldx, idx, H, Ns, y, argmax, N, C / 4);
    else
        hipLaunchKernelGGL(k_kp_maxpool_fwd, dim3(agb_cdiv((long long)N * C, 256)), dim3(256), 0, (hipStream_t)stream, x,
                           ldx, idx, H, Ns, y, argmax, N, C);
    AGB_CHECK_LAUNCH("agb_kp_maxpool_fwd");
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
