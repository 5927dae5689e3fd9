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

// grid B, block 256.  h_pre [B,H] and s [B,C] are written (saved for the backward pass).
__global__ __launch_bounds__(256) void k_se_fwd(const float* __restrict__ P, const float* __restrict__ W1,
                                                const float* __restrict__ b1, const float* __restrict__ W2,
                                                const float* __restrict__ b2, int C, int H, int act,
                                                float* __restrict__ h_pre, float* __restrict__ S) {
    __shared__ float s_a[SE_MAX_H];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* p = P + (long long)b * C;
    for (int j = wave; j < H; j += 4) {
        const float* w = W1 + (long long)j * C;
        float acc = 0.f;
        for (int c = lane; c < C; c += 64) acc += w[c] * p[c];
        acc = wave_sum(acc);
        if (lane == 0) {
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
        for (int j = 0; j < H; ++j) acc += w[j] * s_a[j];
        S[(long long)b * C + c] = 1.f / (1.f + expf(-acc));
    }
}

// grid B, block 256: per-plot vectors dz2 [B,C], dh [B,H] (scratch for pass B) and dP [B,C]
__global__ __launch_bounds__(256) void k_se_bwd_a(const float* __restrict__ W1, const float* __restrict__ W2, int C,
                                                  int H, int act, const float* __restrict__ h_pre,
                                                  const float* __restrict__ S, const float* __restrict__ dS,
                                                  float* __restrict__ dz2, float* __restrict__ dh,
                                                  float* __restrict__ dP) {
    __shared__ float s_dh[SE_MAX_H];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // every workgroup of the plot needs the whole dz2 vector for the hidden-layer gradient: each keeps a private copy in
    // the scratch slice dz2[blockIdx.y][b] (slice 0 is the one pass B reads)
    float* zw = dz2 + ((long long)blockIdx.y * gridDim.x + b) * C;
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = S[(long long)b * C + c];
        zw[c] = dS[(long long)b * C + c] * s * (1.f - s);
    }
    __syncthreads();   // read back below by other threads of the workgroup
    const float* z = zw;
    for (int j = wave; j < H; j += 4) {
        float acc = 0.f;
        for (int c = lane; c < C; c += 64) acc += z[c] * W2[(long long)c * H + j];
        acc = wave_sum(acc);
        if (lane == 0) {
            float g = acc * se_act_grad(h_pre[(long long)b * H + j], act);
            if (blockIdx.y == 0) dh[(long long)b * H + j] = g;
            s_dh[j] = g;
        }
    }
    __syncthreads();
    for (int c = blockIdx.y * 256 + threadIdx.x; c < C; c += 256 * gridDim.y) {
        float acc = 0.f;
        for (int j = 0; j < H; ++j) acc += s_dh[j] * W1[(long long)j * C + c];
        dP[(long long)b * C + c] = acc;
    }
}

// one thread per weight element (2*C*H) plus the biases (C + H); sums over the plots in plot order
__global__ void k_se_bwd_b(const float* __restrict__ P, int C, int H, int B, int act,
                           const float* __restrict__ h_pre, const float* __restrict__ dz2,
                           const float* __restrict__ dh, float* __restrict__ dW1, float* __restrict__ db1,
                           float* __restrict__ dW2, float* __restrict__ db2) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int nw = C * H;
    if (t < nw) {                       // dW1[j][c] = sum_b dh[b][j] * p[b][c]        (c fastest: coalesced p reads)
        const int j = t / C, c = t % C;
        float acc = 0.f;
        for (int b = 0; b < B; ++b) acc += dh[(long long)b * H + j] * P[(long long)b * C + c];
        dW1[t] = acc;
    } else if (t < 2 * nw) {            // dW2[c][j] = sum_b dz2[b][c] * act(h[b][j])
        const int u = t - nw, c = u / H, j = u % H;
        float acc = 0.f;
        for (int b = 0; b < B; ++b) acc += dz2[(long long)b * C + c] * se_act(h_pre[(long long)b * H + j], act);
        dW2[u] = acc;
    } else if (t < 2 * nw + C) {
        const int c = t - 2 * nw;
        float acc = 0.f;
        for (int b = 0; b < B; ++b) acc += dz2[(long long)b * C + c];
        if (db2) db2[c] = acc;
    } else if (t < 2 * nw + C + H) {
        const int j = t - 2 * nw - C;
        float acc = 0.f;
        for (int b = 0; b < B; ++b) acc += dh[(long long)b * H + j];
        if (db1) db1[j] = acc;
    }
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
    const long long total = 2LL * C * H + C + H;
    hipLaunchKernelGGL(k_se_bwd_b, dim3(agb_cdiv(total, 256)), dim3(256), 0, s, P, C, H, B, act, h_pre, dz2, dh, dW1,
                       db1, dW2, db2);
    AGB_CHECK_LAUNCH("agb_se_mlp_bwd");
    return AGB_OK;
}

}  // extern "C"
