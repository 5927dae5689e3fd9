// optim.hip — fused multi-tensor AdaBelief step (+ clip_grad_value_) over every parameter tensor of a group in ONE
// launch.  Same update rule as the reference (torch_points3d/core/optimizer/adabelief.py:89-201, rectified branch,
// decoupled weight decay scaled by lr, in-place eps accumulation into exp_avg_var) and the clip of
// models/base_model.py:241-243.  Elementwise, HBM-bound: 4 reads + 3 writes of 4 B per parameter.
#include "agb_common.h"

struct AdaDesc {
    float* p;
    const float* g;
    float* m;
    float* v;
    long long n;
};

#define ADA_CHUNK 4096

// mode 0: p -= step * m / (sqrt(v) + eps)   (num_sma >= 5)
// mode 1: p -= step * m                      (degenerated to SGD)
// mode 2: no parameter update (step_size <= 0)
// mode 3: non-rectified: p -= step * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void k_adabelief(const AdaDesc* __restrict__ descs,
                                                   const int32_t* __restrict__ chunk_tensor,
                                                   const int32_t* __restrict__ chunk_index, float decay, float beta1,
                                                   float beta2, float omb1, float omb2, float eps, float step,
                                                   float inv_sqrt_bc2, int mode, float clip) {
    const AdaDesc d = descs[chunk_tensor[blockIdx.x]];
    const long long base = (long long)chunk_index[blockIdx.x] * ADA_CHUNK;
    for (int j = 0; j < ADA_CHUNK / 256; ++j) {
        long long i = base + j * 256 + threadIdx.x;
        if (i >= d.n) break;
        float g = d.g[i];
        if (clip > 0.f) g = fminf(fmaxf(g, -clip), clip);
        float p = d.p[i] * decay;
        // omb1 / omb2 = (1 - beta) evaluated in double on the host, as the reference does (1 - 0.999f != 0.001f)
        float m = d.m[i] * beta1 + omb1 * g;
        float r = g - m;
        float v = d.v[i] * beta2 + omb2 * (r * r);
        v += eps;
        if (mode == 0) p -= step * m / (sqrtf(v) + eps);
        else if (mode == 1) p -= step * m;
        else if (mode == 3) p -= step * m / (sqrtf(v) * inv_sqrt_bc2 + eps);
        d.p[i] = p;
        d.m[i] = m;
        d.v[i] = v;
    }
}

extern "C" {

int agb_adabelief_chunk(void) { return ADA_CHUNK; }

// descs: device array of {p, g, m, v (pointers), n (int64)} per tensor (40 bytes each); chunk_tensor / chunk_index:
// device int32[n_chunks] mapping every ADA_CHUNK-element block to its tensor and its chunk number inside the tensor.
int agb_adabelief_step(const void* descs, const int32_t* chunk_tensor, const int32_t* chunk_index, int n_chunks,
                       float decay, float beta1, float beta2, float one_minus_beta1, float one_minus_beta2, float eps,
                       float step, float inv_sqrt_bc2, int mode, float clip, void* stream) {
    AGB_CHECK_ARG(mode >= 0 && mode <= 3, "agb_adabelief_step: mode %d", mode);
    if (n_chunks == 0) return AGB_OK;
    hipLaunchKernelGGL(k_adabelief, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, (const AdaDesc*)descs,
                       chunk_tensor, chunk_index, decay, beta1, beta2, one_minus_beta1, one_minus_beta2, eps, step,
                       inv_sqrt_bc2, mode, clip);
    AGB_CHECK_LAUNCH("agb_adabelief_step");
    return AGB_OK;
}

}  // extern "C"
