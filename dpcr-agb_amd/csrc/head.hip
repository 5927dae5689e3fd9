// head.hip — the regression head and its loss as one launch forward and one backward.
//
// Reference: torch_points3d/models/instance/minkowski.py:16-26 (SeparateLinear: one nn.Linear(C, 1) per regression target on
// the globally pooled features), models/instance/base.py:139-146 (output slice / activation), :154-179 (targets standardised
// with the train statistics, smooth-L1 / L2 / L1 with mean reduction, summed over the configured functions, weighted by the
// mean of the task weights).  In the reference that is ~12 small library kernels forward and ~15 backward on [B, T <= 4]
// operands: pure launch latency on the critical path between the forward and the backward pass of the backbone.
//   forward   one workgroup: out[b][t] = <pooled[b], W_t> + bias_t;  d = out - (y - center) / scale;
//             loss_reg = sum_fn mean_{b,t} fn(d);  loss = mean(weights) * loss_reg;  dout[b][t] = dloss / dout (kept)
//   backward  grid over the channels: dW_t[c] = g sum_b dout[b][t] pooled[b][c], dbias_t = g sum_b dout[b][t],
//             dpooled[b][c] = g sum_t dout[b][t] W_t[c]      (g: the gradient arriving at `loss`; plots in order: deterministic)
#include "agb_common.h"

#define HEAD_MAX_T 8
#define HEAD_LOSS_SMOOTHL1 1
#define HEAD_LOSS_L2 2
#define HEAD_LOSS_L1 4

struct HeadParams {
    const float* W[HEAD_MAX_T];      // [C] each (nn.Linear(C, 1).weight)
    const float* bias[HEAD_MAX_T];   // [1] each or NULL
    float* dW[HEAD_MAX_T];
    float* dbias[HEAD_MAX_T];
};

__device__ __forceinline__ float head_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// one workgroup of 1024 threads (16 waves): wave w takes the (plot, target) pairs w, w + 16, ...
__global__ __launch_bounds__(1024) void k_head_fwd(const float* __restrict__ pooled, int ldp, int B, int C, int T, HeadParams P,
                                                   const float* __restrict__ y, const float* __restrict__ center,
                                                   const float* __restrict__ scale, const float* __restrict__ weights,
                                                   int loss_mask, float* __restrict__ out, float* __restrict__ dout,
                                                   float* __restrict__ loss_reg, float* __restrict__ loss) {
    __shared__ float s_val[16], s_w;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float inv_n = 1.f / (float)(B * T);
    if (threadIdx.x == 0) {
        float wm = 0.f;
        for (int t = 0; t < T; ++t) wm += weights[t];
        s_w = wm / (float)T;
    }
    __syncthreads();
    const float wmean = s_w;
    float lsum = 0.f;      // this wave's share of sum fn(d), on lane 0
    for (int e = wave; e < B * T; e += 16) {
        const int b = e / T, t = e - b * T;
        const float* p = pooled + (long long)b * ldp;
        const float* w = P.W[t];
        float acc = 0.f;
        for (int c = lane; c < C; c += 64) acc += p[c] * w[c];
        acc = head_wave_sum(acc);
        if (lane == 0) {
            const float o = acc + (P.bias[t] ? P.bias[t][0] : 0.f);
            out[e] = o;
            const float d = o - (y[e] - center[t]) / scale[t];
            const float ad = fabsf(d), sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            float l = 0.f, g = 0.f;
            if (loss_mask & HEAD_LOSS_SMOOTHL1) { l += ad < 1.f ? 0.5f * d * d : ad - 0.5f; g += ad < 1.f ? d : sg; }
            if (loss_mask & HEAD_LOSS_L2) { l += d * d; g += 2.f * d; }
            if (loss_mask & HEAD_LOSS_L1) { l += ad; g += sg; }
            lsum += l;
            dout[e] = wmean * g * inv_n;
        }
    }
    if (lane == 0) s_val[wave] = lsum;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
        for (int w2 = 0; w2 < 16; ++w2) tot += s_val[w2];      // fixed order
        const float lr = tot * inv_n;
        *loss_reg = lr;
        *loss = wmean * lr;
    }
}

// grid ceil(C / 256): thread c; block 0 also writes the bias gradients
__global__ __launch_bounds__(256) void k_head_bwd(const float* __restrict__ pooled, int ldp, int B, int C, int T, HeadParams P,
                                                  const float* __restrict__ dout, const float* __restrict__ gloss,
                                                  float* __restrict__ dpooled, int lddp) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const float g = gloss ? *gloss : 1.f;
    if (c < C) {
        float dw[HEAD_MAX_T];
#pragma unroll
        for (int t = 0; t < HEAD_MAX_T; ++t) dw[t] = 0.f;
        for (int b = 0; b < B; ++b) {
            const float pv = pooled[(long long)b * ldp + c];
            float dp = 0.f;
#pragma unroll
            for (int t = 0; t < HEAD_MAX_T; ++t) {
                if (t < T) {
                    const float d = dout[b * T + t] * g;
                    dw[t] += d * pv;
                    dp += d * P.W[t][c];
                }
            }
            if (dpooled) dpooled[(long long)b * lddp + c] = dp;
        }
#pragma unroll
        for (int t = 0; t < HEAD_MAX_T; ++t)
            if (t < T && P.dW[t]) P.dW[t][c] = dw[t];
    }
    if (blockIdx.x == 0 && threadIdx.x < T && P.dbias[threadIdx.x]) {
        const int t = threadIdx.x;
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dout[b * T + t] * g;
        P.dbias[t][0] = s;
    }
}

extern "C" {

// pooled [B][ldp >= C] (device); W, bias: HOST arrays of T device pointers (weight [C] / bias [1] of every target's
// nn.Linear(C, 1); a bias pointer may be NULL); y [B][T] raw targets, center / scale / weights [T] (device).
// loss_mask: 1 smooth-L1 (beta 1), 2 L2, 4 L1 (summed).  Out: out [B][T], dout [B][T] (kept for agb_reg_head_bwd),
// loss_reg, loss (device scalars).
int agb_reg_head_fwd(const float* pooled, int ldp, int B, int C, int T, const float* const* W, const float* const* bias,
                     const float* y, const float* center, const float* scale, const float* weights, int loss_mask, float* out,
                     float* dout, float* loss_reg, float* loss, void* stream) {
    AGB_CHECK_ARG(B >= 1 && C >= 1 && T >= 1 && T <= HEAD_MAX_T && ldp >= C, "agb_reg_head_fwd: B %d, C %d, T %d (<= %d)", B, C, T,
                  HEAD_MAX_T);
    AGB_CHECK_ARG(loss_mask >= 1 && loss_mask <= 7, "agb_reg_head_fwd: loss mask %d (1 smooth-L1, 2 L2, 4 L1)", loss_mask);
    AGB_CHECK_ARG(pooled && W && bias && y && center && scale && weights && out && dout && loss_reg && loss,
                  "agb_reg_head_fwd: null argument");
    HeadParams P{};
    for (int t = 0; t < T; ++t) {
        AGB_CHECK_ARG(W[t] != nullptr, "agb_reg_head_fwd: weight %d is NULL", t);
        P.W[t] = W[t];
        P.bias[t] = bias[t];
    }
    hipLaunchKernelGGL(k_head_fwd, dim3(1), dim3(1024), 0, (hipStream_t)stream, pooled, ldp, B, C, T, P, y, center, scale, weights,
                       loss_mask, out, dout, loss_reg, loss);
    AGB_CHECK_LAUNCH("agb_reg_head_fwd");
    return AGB_OK;
}

// gloss: device scalar, the gradient arriving at `loss` (NULL: 1).  dW, dbias: HOST arrays of T device pointers (either
// may hold NULLs); dpooled [B][lddp] or NULL.
int agb_reg_head_bwd(const float* pooled, int ldp, int B, int C, int T, const float* const* W, const float* dout,
                     const float* gloss, float* const* dW, float* const* dbias, float* dpooled, int lddp, void* stream) {
    AGB_CHECK_ARG(B >= 1 && C >= 1 && T >= 1 && T <= HEAD_MAX_T && ldp >= C && (dpooled == nullptr || lddp >= C),
                  "agb_reg_head_bwd: B %d, C %d, T %d (<= %d)", B, C, T, HEAD_MAX_T);
    AGB_CHECK_ARG(pooled && W && dout && dW && dbias, "agb_reg_head_bwd: null argument");
    HeadParams P{};
    for (int t = 0; t < T; ++t) {
        AGB_CHECK_ARG(W[t] != nullptr, "agb_reg_head_bwd: weight %d is NULL", t);
        P.W[t] = W[t];
        P.dW[t] = dW[t];
        P.dbias[t] = dbias[t];
    }
    hipLaunchKernelGGL(k_head_bwd, dim3(agb_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, pooled, ldp, B, C, T, P, dout, gloss,
                       dpooled, lddp);
    AGB_CHECK_LAUNCH("agb_reg_head_bwd");
    return AGB_OK;
}

}  // extern "C"
