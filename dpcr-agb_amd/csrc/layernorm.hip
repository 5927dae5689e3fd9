// layernorm.hip — channel-wise LayerNorm over the rows of a sparse tensor's feature matrix [N, C]
// (the reference's norm_type="ln": modules/MinkowskiEngine/common.py:369-386 MinkowskiLayerNorm = nn.LayerNorm(C, eps=1e-6)
// on .F, chosen in SENet.py:40-41).  HBM-bound: forward reads X once (second pass from L1/L2) and writes Y; backward
// reads X and dY once and writes dX; the parameter gradients are per-workgroup partial column sums folded in a fixed
// order (no atomics).  One 64-lane wave per row, lanes strided over the channels so every load is a coalesced 256-byte run.
#include "agb_common.h"

#define LN_WAVES 4
#define LN_MAX_C 2048

__device__ __forceinline__ float ln_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(64 * LN_WAVES) void k_layernorm_fwd(const float* __restrict__ X, int ldx, int n, int C,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float eps,
                                                                 float* __restrict__ Y, int ldy,
                                                                 float* __restrict__ stats) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float invC = 1.f / (float)C;
    for (long long r = (long long)blockIdx.x * LN_WAVES + w; r < n; r += (long long)gridDim.x * LN_WAVES) {
        const float* x = X + r * ldx;
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += x[c];
        const float mean = ln_wave_sum(s) * invC;
        float q = 0.f;
        for (int c = lane; c < C; c += 64) { const float d = x[c] - mean; q += d * d; }
        const float rstd = rsqrtf(ln_wave_sum(q) * invC + eps);
        float* y = Y + r * ldy;
        for (int c = lane; c < C; c += 64) {
            float v = (x[c] - mean) * rstd;
            if (gamma) v *= gamma[c];
            if (beta) v += beta[c];
            y[c] = v;
        }
        if (lane == 0) { stats[2 * r] = mean; stats[2 * r + 1] = rstd; }
    }
}

// dX = rstd * (g*dY - mean_c(g*dY) - xhat * mean_c(g*dY*xhat));  part[wg][0][c] = sum_r dY*xhat, part[wg][1][c] = sum_r dY
__global__ __launch_bounds__(64 * LN_WAVES) void k_layernorm_bwd(const float* __restrict__ X, int ldx,
                                                                 const float* __restrict__ dY, int ldy, int n, int C,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ stats,
                                                                 float* __restrict__ dX, int lddx,
                                                                 float* __restrict__ part) {
    extern __shared__ float lds[];                 // [LN_WAVES][2][C]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float* dg = lds + (size_t)w * 2 * C;
    float* db = dg + C;
    for (int c = lane; c < C; c += 64) { dg[c] = 0.f; db[c] = 0.f; }
    const float invC = 1.f / (float)C;
    for (long long r = (long long)blockIdx.x * LN_WAVES + w; r < n; r += (long long)gridDim.x * LN_WAVES) {
        const float* x = X + r * ldx;
        const float* dy = dY + r * ldy;
        const float mean = stats[2 * r], rstd = stats[2 * r + 1];
        float s1 = 0.f, s2 = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float xh = (x[c] - mean) * rstd, d = dy[c];
            const float gd = gamma ? gamma[c] * d : d;
            s1 += gd * xh;
            s2 += gd;
            dg[c] += d * xh;          // each lane owns its channels of its wave's row: no conflicts, fixed order
            db[c] += d;
        }
        const float c1 = ln_wave_sum(s1) * invC, c2 = ln_wave_sum(s2) * invC;
        if (dX) {
            float* dx = dX + r * lddx;
            for (int c = lane; c < C; c += 64) {
                const float xh = (x[c] - mean) * rstd, d = dy[c];
                const float gd = gamma ? gamma[c] * d : d;
                dx[c] = rstd * (gd - c2 - xh * c1);
            }
        }
    }
    __syncthreads();
    float* out = part + (size_t)blockIdx.x * 2 * C;
    for (int c = threadIdx.x; c < 2 * C; c += blockDim.x) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < LN_WAVES; ++k) v += lds[(size_t)k * 2 * C + c];
        out[c] = v;
    }
}

__global__ void k_layernorm_fold(const float* __restrict__ part, int chunks, int C, float* __restrict__ dgamma,
                                 float* __restrict__ dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= 2 * C) return;
    float v = 0.f;
    for (int k = 0; k < chunks; ++k) v += part[(size_t)k * 2 * C + c];
    if (c < C) { if (dgamma) dgamma[c] = v; }
    else if (dbeta) dbeta[c - C] = v;
}

extern "C" {

int agb_layernorm_chunks(int n) {
    int wg = agb_cdiv(n > 0 ? n : 1, LN_WAVES * 8);      // >= 8 rows per wave before another workgroup is opened
    return wg < 1 ? 1 : (wg > 2048 ? 2048 : wg);
}

int agb_layernorm_fwd(const float* X, int ldx, int n, int C, const float* gamma, const float* beta, float eps, float* Y,
                      int ldy, float* stats, void* stream) {
    AGB_CHECK_ARG(n >= 0 && C >= 1 && C <= LN_MAX_C && ldx >= C && ldy >= C,
                  "agb_layernorm_fwd: bad shape n=%d C=%d (C <= %d)", n, C, LN_MAX_C);
    if (n == 0) return AGB_OK;
    AGB_CHECK_ARG(X && Y && stats, "agb_layernorm_fwd: null pointer");
    hipLaunchKernelGGL(k_layernorm_fwd, dim3(agb_layernorm_chunks(n)), dim3(64 * LN_WAVES), 0, (hipStream_t)stream, X,
                       ldx, n, C, gamma, beta, eps, Y, ldy, stats);
    AGB_CHECK_LAUNCH("agb_layernorm_fwd");
    return AGB_OK;
}

int agb_layernorm_bwd(const float* X, int ldx, const float* dY, int ldy, int n, int C, const float* gamma,
                      const float* stats, float* dX, int lddx, float* part, float* dgamma, float* dbeta, void* stream) {
    AGB_CHECK_ARG(n >= 0 && C >= 1 && C <= LN_MAX_C && ldx >= C && ldy >= C,
                  "agb_layernorm_bwd: bad shape n=%d C=%d (C <= %d)", n, C, LN_MAX_C);
    AGB_CHECK_ARG(part && (n == 0 || (X && dY && stats)), "agb_layernorm_bwd: null pointer");
    const int chunks = agb_layernorm_chunks(n);
    const size_t lds = (size_t)LN_WAVES * 2 * C * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)k_layernorm_bwd, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) { agb_set_error("agb_layernorm_bwd: %s", hipGetErrorString(e)); return AGB_ELAUNCH; }
    }
    hipLaunchKernelGGL(k_layernorm_bwd, dim3(chunks), dim3(64 * LN_WAVES), lds, st, X, ldx, dY, ldy, n, C, gamma, stats,
                       dX, lddx, part);
    AGB_CHECK_LAUNCH("agb_layernorm_bwd");
    if (dgamma || dbeta) {
        hipLaunchKernelGGL(k_layernorm_fold, dim3(agb_cdiv(2 * C, 256)), dim3(256), 0, st, part, chunks, C, dgamma, dbeta);
        AGB_CHECK_LAUNCH("agb_layernorm_fold");
    }
    return AGB_OK;
}

}  // extern "C"
