// se.hip — the excitation MLP of the squeeze-excite layer as one forward and two backward launches.
//
// Reference: SELayer (torch_points3d/modules/MinkowskiEngine/senet_block.py:33-50):
//   fc = Linear(C, C/r) -> act -> Linear(C/r, C) -> Sigmoid on the per-plot average-pooled features [B, C];
// with ME wrappers + cuBLAS that is ~12 launches forward and ~25 backward per block on [32, C] operands (C = 64..2048,
// hidden H = C/16): pure launch latency.  Here:
//   forward   one workgroup per plot:  h = W1 p + b1;  a = act(h);  s = sigmoid(W2 a + b2)
//   backward  A (one workgroup per plot): dz2 = ds*s*(1-s);  dh = (W2^T dz2) * act'(h);  dp = W1^T dh
//             B (grid over the weights, fixed summation order over the plots: deterministic):
//               dW2 = dz2^T a, db2 = sum_b dz2, dW1 = dh^T p, db1 = sum_b dh
// Weights are nn.Linear layout: W1 [H, C], W2 [C, H].  fp32 throughout; dot products are wave reductions (fixed order).
#include "agb_common.h"

#define SE_ACT_RELU 1
#define SE_ACT_GELU 2

__device__ __forceinline__ float se_act(float z, int act) {
    if (act == SE_ACT_RELU) return z > 0.f ? z : 0.f;
    if (act == SE_ACT_GELU) return 0.5f * z * (1.f + erff(z * 0.70710678118654752440f));
    return z;
}
__device__ __forceinline__ float se_act_grad(float z, int act) {
    if (act == SE_ACT_RELU) return z > 0.f ? 1.f : 0.f;
    if (act == SE_ACT_GELU) {
        float cdf = 0.5f * (1.f + erff(z * 0.70710678118654752440f));
        float pdf = 0.39894228040143267794f * expf(-0.5f * z * z);
        return cdf + z * pdf;
    }
    return 1.f;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

#define SE_MAX_H 256

// grid (B, column slabs), block 256.  h_pre [B,H] and s [B,C] are written (saved for the backward pass).
// Hidden layer: thread (j-group, slice) — 8 slices of 4 consecutive channels each read one contiguous 128-B piece of
// row j of W1 per step, all loads of a thread independent; the 8 slice partials meet by shuffles.  (One wave per hidden
// unit with a shuffle reduction per unit serialised 8 memory round trips: 30 us on the 512-channel level.)
__global__ __launch_bounds__(256) void k_se_fwd(const float* __restrict__ P, const float* __restrict__ W1,
                                                const float* __restrict__ b1, const float* __restrict__ W2,
                                                const float* __restrict__ b2, int C, int H, int act,
                                                float* __restrict__ h_pre, float* __restrict__ S) {
    __shared__ float s_a[SE_MAX_H];
    const int b = blockIdx.x, sl = threadIdx.x & 7, jg = threadIdx.x >> 3;
    const float* p = P + (long long)b * C;
    for (int j0 = 0; j0 < H; j0 += 32) {
        const int j = j0 + jg;
        float acc = 0.f;
        if (j < H) {
            const float* w = W1 + (long long)j * C;
            if ((C & 3) == 0) {
#pragma unroll 4
                for (int c = 4 * sl; c < C; c += 32) {
                    const float4 wv = *reinterpret_cast<const float4*>(w + c);
                    const float4 pv = *reinterpret_cast<const float4*>(p + c);
                    acc += wv.x * pv.x + wv.y * pv.y + wv.z * pv.z + wv.w * pv.w;
                }
            } else {
                for (int c = sl; c < C; c += 8) acc += w[c] * p[c];
            }
        }
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        acc += __shfl_xor(acc, 4);
        if (sl == 0 && j < H) {
            float h = acc + (b1 ? b1[j] : 0.f);
            if (blockIdx.y == 0) h_pre[(long long)b * H + j] = h;
            s_a[j] = se_act(h, act);
        }
    }
    __syncthreads();
    // the columns are divided among gridDim.y workgroups per plot (every one of them recomputes the small hidden layer):
    // wide layers (C = 2048 in SENet50) would otherwise run on 32 CUs only
    for (int c = blockIdx.y * 256 + threadIdx.x; c < C; c += 256 * gridDim.y) {
        const float* w = W2 + (long long)c * H;
        float acc = b2 ? b2[c] : 0.f;
        if ((H & 3) == 0) {
#pragma unroll 4
            for (int j = 0; j < H; j += 4) {
                const float4 wv = *reinterpret_cast<const float4*>(w + j);
                acc += wv.x * s_a[j] + wv.y * s_a[j + 1] + wv.z * s_a[j + 2] + wv.w * s_a[j + 3];
            }
        } else {
            for (int j = 0; j < H; ++j) acc += w[j] * s_a[j];
        }
        S[(long long)b * C + c] = 1.f / (1.f + expf(-acc));
    }
}

// grid (B, column slabs), block 256: per-plot vectors dz2 [B,C], dh [B,H] (scratch for pass B) and dP [B,C]
__global__ __launch_bounds__(256) void k_se_bwd_a(const float* __restrict__ W1, const float* __restrict__ W2, int C,
                                                  int H, int act, const float* __restrict__ h_pre,
                                                  const float* __restrict__ S, const float* __restrict__ dS,
                                                  float* __restrict__ dz2, float* __restrict__ dh,
                                                  float* __restrict__ dP) {
    __shared__ float s_dh[SE_MAX_H];
    __shared__ float s_red[8][32];
    __shared__ __attribute__((aligned(16))) float s_part[1024];   // [256 / (H/4) lane groups][H]
    const int b = blockIdx.x;
    // every workgroup of the plot needs the whole dz2 vector for the hidden-layer gradient: each keeps a private copy in
    // the scratch slice dz2[blockIdx.y][b] (slice 0 is the one pass B reads)
    float* zw = dz2 + ((long long)blockIdx.y * gridDim.x + b) * C;
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = S[(long long)b * C + c];
        zw[c] = dS[(long long)b * C + c] * s * (1.f - s);
    }
    __syncthreads();   // read back below by other threads of the workgroup
    const float* z = zw;
    // dh[j] = act'(h[j]) * sum_c z[c] W2[c][j].  H % 4 == 0: H/4 lanes cover one row of W2 with a float4 each and the
    // 256 / (H/4) lane groups take rows c = group, group + groups, ...: every load instruction of the workgroup covers
    // 4 KB of W2 and the loop is C / groups trips of independent loads (C = 2048, H = 128: 256 trips, eight in flight —
    // the lane-per-column form walked W2 four times with one 4-byte load per trip and ran 50 us); partials meet in LDS.
    if ((H & 3) == 0) {
        const int lj = H >> 2, groups = 256 / lj;          // H <= 128: lj <= 32, a power of two or not — any divisor works
        const int jl = threadIdx.x % lj, grp = threadIdx.x / lj;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (grp < groups) {
#pragma unroll 8
            for (int c = grp; c < C; c += groups) {
                const float4 wv = *reinterpret_cast<const float4*>(W2 + (long long)c * H + 4 * jl);
                const float zc = z[c];
                acc.x += zc * wv.x; acc.y += zc * wv.y; acc.z += zc * wv.z; acc.w += zc * wv.w;
            }
            *reinterpret_cast<float4*>(&s_part[grp * H + 4 * jl]) = acc;
        }
        __syncthreads();
        if (threadIdx.x < H) {
            const int j = threadIdx.x;
            float t = 0.f;
            for (int u = 0; u < groups; ++u) t += s_part[u * H + j];      // fixed order
            float g = t * se_act_grad(h_pre[(long long)b * H + j], act);
            if (blockIdx.y == 0) dh[(long long)b * H + j] = g;
            s_dh[j] = g;
        }
        __syncthreads();
    } else {
    // thread (slice, j) with j fastest — the 32 lanes of a slice read one
    // contiguous piece of row c of W2; 8 slices take rows c = slice, slice + 8, ...; partials meet in LDS
    const int jl = threadIdx.x & 31, sl = threadIdx.x >> 5;
    for (int j0 = 0; j0 < H; j0 += 32) {
        const int j = j0 + jl;
        float acc = 0.f;
        if (j < H) {
#pragma unroll 4
            for (int c = sl; c < C; c += 8) acc += z[c] * W2[(long long)c * H + j];
        }
        s_red[sl][jl] = acc;
        __syncthreads();
        if (threadIdx.x < 32 && j < H) {
            float t = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) t += s_red[u][jl];
            float g = t * se_act_grad(h_pre[(long long)b * H + j], act);
            if (blockIdx.y == 0) dh[(long long)b * H + j] = g;
            s_dh[j] = g;
        }
        __syncthreads();
    }
    }
    for (int c = blockIdx.y * 256 + threadIdx.x; c < C; c += 256 * gridDim.y) {
        float acc = 0.f;
#pragma unroll 4
        for (int j = 0; j < H; ++j) acc += s_dh[j] * W1[(long long)j * C + c];
        dP[(long long)b * C + c] = acc;
    }
}

// eight lanes per weight element (2*C*H of them) or bias (C + H): lane q sums plots q, q+8, ... (independent loads, one
// memory round trip for B <= 32), the eight partial sums meet by shuffles in a fixed order (deterministic)
__global__ __launch_bounds__(256) void k_se_bwd_b(const float* __restrict__ P, int C, int H, int B, int act,
                                                  const float* __restrict__ h_pre, const float* __restrict__ dz2,
                                                  const float* __restrict__ dh, float* __restrict__ dW1,
                                                  float* __restrict__ db1, float* __restrict__ dW2,
                                                  float* __restrict__ db2, int shift) {
    // shift 3: eight lanes per element (small layers: latency-bound); shift 0: one thread per element (C*H >= 64 k:
    // enough threads already, eight times more only add overhead — 33 vs 15 us at C = 2048)
    const long long gt = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int t = (int)(gt >> shift), q = (int)(gt & ((1 << shift) - 1)), st = 1 << shift;
    const int nw = C * H;
    float acc = 0.f;
    if (t < nw) {                       // dW1[j][c] = sum_b dh[b][j] * p[b][c]        (c fastest)
        const int j = t / C, c = t % C;
#pragma unroll 4
        for (int b = q; b < B; b += st) acc += dh[(long long)b * H + j] * P[(long long)b * C + c];
    } else if (t < 2 * nw) {            // dW2[c][j] = sum_b dz2[b][c] * act(h[b][j])
        const int u = t - nw, c = u / H, j = u % H;
#pragma unroll 4
        for (int b = q; b < B; b += st) acc += dz2[(long long)b * C + c] * se_act(h_pre[(long long)b * H + j], act);
    } else if (t < 2 * nw + C) {
        const int c = t - 2 * nw;
#pragma unroll 4
        for (int b = q; b < B; b += st) acc += dz2[(long long)b * C + c];
    } else if (t < 2 * nw + C + H) {
        const int j = t - 2 * nw - C;
#pragma unroll 4
        for (int b = q; b < B; b += st) acc += dh[(long long)b * H + j];
    }
    if (shift == 3) {
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        acc += __shfl_xor(acc, 4);
    }
    if (q != 0) return;
    if (t < nw) dW1[t] = acc;
    else if (t < 2 * nw) dW2[t - nw] = acc;
    else if (t < 2 * nw + C) { if (db2) db2[t - 2 * nw] = acc; }
    else if (t < 2 * nw + C + H) { if (db1) db1[t - 2 * nw - C] = acc; }
}

extern "C" {

// P [B,C] pooled features; W1 [H,C], b1 [H] (or NULL), W2 [C,H], b2 [C] (or NULL); act 0 none / 1 relu / 2 gelu.
// Out: h_pre [B,H] (pre-activation of the hidden layer, kept for the backward pass), S [B,C] = sigmoid(...).
int agb_se_mlp_fwd(const float* P, const float* W1, const float* b1, const float* W2, const float* b2, int B, int C,
                   int H, int act, float* h_pre, float* S, void* stream) {
    AGB_CHECK_ARG(B >= 0 && C >= 1 && H >= 1 && H <= SE_MAX_H, "agb_se_mlp_fwd: B %d, C %d, H %d (H <= %d)", B, C, H,
                  SE_MAX_H);
    AGB_CHECK_ARG(act >= 0 && act <= 2, "agb_se_mlp_fwd: activation %d", act);
    if (B == 0) return AGB_OK;
    hipLaunchKernelGGL(k_se_fwd, dim3(B, agb_cdiv(C, 512)), dim3(256), 0, (hipStream_t)stream, P, W1, b1, W2, b2, C, H, act,
                       h_pre, S);
    AGB_CHECK_LAUNCH("agb_se_mlp_fwd");
    return AGB_OK;
}

// dS [B,C] in; scratch dz2 [ceil(C/512)][B,C] (slice 0 = the values), dh [B,H]; out dP [B,C], dW1 [H,C], db1 [H] (or NULL), dW2 [C,H], db2 [C] (or NULL).
int agb_se_mlp_bwd(const float* P, const float* W1, const float* W2, int B, int C, int H, int act, const float* h_pre,
                   const float* S, const float* dS, float* dz2, float* dh, float* dP, float* dW1, float* db1,
                   float* dW2, float* db2, void* stream) {
    AGB_CHECK_ARG(B >= 0 && C >= 1 && H >= 1 && H <= SE_MAX_H, "agb_se_mlp_bwd: B %d, C %d, H %d (H <= %d)", B, C, H,
                  SE_MAX_H);
    hipStream_t s = (hipStream_t)stream;
    if (B > 0)
        hipLaunchKernelGGL(k_se_bwd_a, dim3(B, agb_cdiv(C, 512)), dim3(256), 0, s, W1, W2, C, H, act, h_pre, S, dS, dz2, dh,
                           dP);
    const int shift = (long long)C * H >= 65536 ? 0 : 3;
    const long long total = (2LL * C * H + C + H) << shift;
    hipLaunchKernelGGL(k_se_bwd_b, dim3(agb_cdiv(total, 256)), dim3(256), 0, s, P, C, H, B, act, h_pre, dz2, dh, dW1,
                       db1, dW2, db2, shift);
    AGB_CHECK_LAUNCH("agb_se_mlp_bwd");
    return AGB_OK;
}

}  // extern "C"
