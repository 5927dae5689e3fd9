// norm.hip — training-mode BatchNorm over the rows of a [N, C] feature matrix fused with the activation that
// follows it, plus the fused residual tail.  HBM-bound: every kernel streams rows as 16-B pieces.
//
// Replaces, on the reference's hot path,
//   ME.MinkowskiBatchNorm (= nn.BatchNorm1d on .F) + MinkowskiGELU/ReLU
//       torch_points3d/modules/MinkowskiEngine/common.py:215-226 (ConvNormActivation), resnet_block.py:62-72,
//       senet_block.py:83-96, PointNet.py:16-39
//   nn.BatchNorm1d(momentum 0.02) + ReLU of the KPConv blocks: modules/KPConv/blocks.py:460-535
// Statistics are combined with Chan's parallel update (count, mean, M2) so that E[x^2]-E[x]^2 cancellation never
// happens; partials are folded in a fixed order (bitwise reproducible).  The backward pass recomputes the
// pre-activation from the saved conv output instead of storing it.
#include "agb_common.h"

// block = 256 threads = RL row lanes x CGS column groups of 4 channels.  64 channels and more: 16 x 16 (one 64-channel
// slab, 16 rows per pass).  Narrower matrices (the 16/32-channel levels of KPConv and of the sparse stem) would leave 3/4
// or 1/2 of those threads without a column: they take CGS = C/4 groups and 256/CGS row lanes instead (1.2-1.5 TB/s ->
// see DESIGN.md section 5 for the figures).
#define NB_ROWS 16
struct BnLanes {
    int cgs, rl_n;     // column groups per block, row lanes in use
    int cg, rl;        // this thread's; rl >= rl_n: idle (CGS does not divide 256)
};
__device__ __forceinline__ BnLanes bn_lanes(int C) {
    BnLanes L;
    L.cgs = C >= 64 ? 16 : (C >> 2);
    L.rl_n = 256 / L.cgs;
    L.cg = threadIdx.x % L.cgs;
    L.rl = threadIdx.x / L.cgs;
    return L;
}
static inline int bn_slab(int C) { return C >= 64 ? 64 : C; }              // channels per workgroup
static inline int bn_row_lanes(int C) { return C >= 64 ? 16 : 256 / (C >> 2); }

// ---------------------------------------------------------------- statistics
// grid (chunks, ceil(C/64)); part[chunk][3][C] = (count, mean, M2) of the chunk's rows
template <typename T>
__global__ __launch_bounds__(256) void k_bn_stats_partial(const T* __restrict__ X, int ldx, int n, int C,
                                                          int rows_per_chunk, float* __restrict__ part) {
    __shared__ float s_mean[1024];         // [row lane][slab channel]
    __shared__ float s_m2[1024];
    __shared__ float s_cnt[256];
    const BnLanes L = bn_lanes(C);
    const int cg = L.cg, rl = L.rl, slab = L.cgs * 4;
    const int c = blockIdx.y * slab + cg * 4;
    const int r_beg = blockIdx.x * rows_per_chunk;
    const int r_end = min(n, r_beg + rows_per_chunk);
    float mean[4] = {0.f, 0.f, 0.f, 0.f}, m2[4] = {0.f, 0.f, 0.f, 0.f};
    float cnt = 0.f;
    if (rl >= L.rl_n) {
    } else if (c < C) {
        for (int r = r_beg + rl; r < r_end; r += L.rl_n) {
            float4 v = ld4(X + (long long)r * ldx + c);
            cnt += 1.f;
            float inv = 1.f / cnt;
            float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float d = x[j] - mean[j];
                mean[j] += d * inv;
                m2[j] += d * (x[j] - mean[j]);
            }
        }
    } else {
        for (int r = r_beg + rl; r < r_end; r += L.rl_n) cnt += 1.f;
    }
    if (rl < L.rl_n) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s_mean[rl * slab + cg * 4 + j] = mean[j];
            s_m2[rl * slab + cg * 4 + j] = m2[j];
        }
        if (cg == 0) s_cnt[rl] = cnt;
    }
    __syncthreads();
    if (threadIdx.x < slab) {
        int cc = blockIdx.y * slab + threadIdx.x;
        float na = 0.f, ma = 0.f, qa = 0.f;
        for (int j = 0; j < L.rl_n; ++j) {  // fixed order
            float nb = s_cnt[j];
            if (nb == 0.f) continue;
            float mb = s_mean[j * slab + threadIdx.x], qb = s_m2[j * slab + threadIdx.x];
            float nt = na + nb, d = mb - ma;
            ma += d * (nb / nt);
            qa += qb + d * d * (na * nb / nt);
            na = nt;
        }
        if (cc < C) {
            float* p = part + (long long)blockIdx.x * 3 * C;
            p[cc] = na;
            p[C + cc] = ma;
            p[2 * C + cc] = qa;
        }
    }
}

// Chan's parallel combine of (count, mean, M2) partials; b may be empty (count 0)
__device__ __forceinline__ void chan_combine(float& na, float& ma, float& qa, float nb, float mb, float qb) {
    const float nt = na + nb, d = mb - ma;
    const float r = nt > 0.f ? nb / nt : 0.f;
    ma += d * r;
    qa += qb + d * d * (na * r);
    na = nt;
}

// block = 4 channels x 256 chunk lanes (1024 threads).  Lane l owns chunks l, l+256, ... (FOLD_PER = 8 of them with the
// 2048-chunk cap of agb_bn_chunks): all of its partials are loaded up front (one memory round trip; the previous version
// took four, plus a 64-long serial combine by one thread), folded as a binary tree in registers, across the 16 lanes of
// a wave that share the channel with shuffles, and across the 16 waves through LDS.  The tree is fixed: results are
// bitwise reproducible.  Writes mean / rstd, updates the running stats (and the layer's batch counter).  This tiny
// kernel sits on the critical path of every BatchNorm (12 per step) and is latency-bound.
#define FOLD_LANES 256
#define FOLD_CH 4
#define FOLD_PER 8
__global__ __launch_bounds__(1024) void k_bn_stats_fold(const float* __restrict__ part, int chunks, int C, float eps,
                                                        float momentum, float* __restrict__ mean,
                                                        float* __restrict__ rstd, float* running_mean,
                                                        float* running_var, long long* num_batches_tracked) {
    if (num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) *num_batches_tracked += 1;
    __shared__ float s_n[16][FOLD_CH], s_m[16][FOLD_CH], s_q[16][FOLD_CH];
    const int cl = threadIdx.x & (FOLD_CH - 1), lane = threadIdx.x / FOLD_CH;
    const int c = min(blockIdx.x * FOLD_CH + cl, C - 1);   // clamped: every thread takes part in the shuffles
    float nb[FOLD_PER], mb[FOLD_PER], qb[FOLD_PER];
#pragma unroll
    for (int u = 0; u < FOLD_PER; ++u) {
        const int j = lane + u * FOLD_LANES;
        nb[u] = 0.f; mb[u] = 0.f; qb[u] = 0.f;
        if (j < chunks) {
            const float* p = part + (long long)j * 3 * C;
            nb[u] = p[c]; mb[u] = p[C + c]; qb[u] = p[2 * C + c];
        }
    }
    chan_combine(nb[0], mb[0], qb[0], nb[1], mb[1], qb[1]);
    chan_combine(nb[2], mb[2], qb[2], nb[3], mb[3], qb[3]);
    chan_combine(nb[4], mb[4], qb[4], nb[5], mb[5], qb[5]);
    chan_combine(nb[6], mb[6], qb[6], nb[7], mb[7], qb[7]);
    chan_combine(nb[0], mb[0], qb[0], nb[2], mb[2], qb[2]);
    chan_combine(nb[4], mb[4], qb[4], nb[6], mb[6], qb[6]);
    chan_combine(nb[0], mb[0], qb[0], nb[4], mb[4], qb[4]);
    float na = nb[0], ma = mb[0], qa = qb[0];
    // chunks beyond FOLD_PER * FOLD_LANES (not produced by agb_bn_chunks; kept for generality)
    for (int j = lane + FOLD_PER * FOLD_LANES; j < chunks; j += FOLD_LANES) {
        const float* p = part + (long long)j * 3 * C;
        chan_combine(na, ma, qa, p[c], p[C + c], p[2 * C + c]);
    }
    // the 16 chunk lanes of this wave that share the channel sit FOLD_CH threads apart
    for (int o = FOLD_CH; o < 64; o *= 2) {
        const float n2 = __shfl_down(na, o), m2 = __shfl_down(ma, o), q2 = __shfl_down(qa, o);
        chan_combine(na, ma, qa, n2, m2, q2);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) < FOLD_CH) { s_n[wave][cl] = na; s_m[wave][cl] = ma; s_q[wave][cl] = qa; }
    __syncthreads();
    if (threadIdx.x < FOLD_CH && blockIdx.x * FOLD_CH + cl < C) {
        na = s_n[0][cl]; ma = s_m[0][cl]; qa = s_q[0][cl];
        for (int w = 1; w < 16; ++w) chan_combine(na, ma, qa, s_n[w][cl], s_m[w][cl], s_q[w][cl]);
        float var_b = na > 0.f ? qa / na : 0.f;
        mean[c] = ma;
        rstd[c] = rsqrtf(var_b + eps);
        if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * ma;
        if (running_var) {
            float var_u = na > 1.f ? qa / (na - 1.f) : var_b;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * var_u;
        }
    }
}

// eval mode: mean/rstd from the running statistics
__global__ void k_bn_eval_stats(const float* __restrict__ running_mean, const float* __restrict__ running_var, int C,
                                float eps, float* mean, float* rstd) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    mean[c] = running_mean[c];
    rstd[c] = rsqrtf(running_var[c] + eps);
}

// ---------------------------------------------------------------- forward apply: y = act(gamma*(x-mean)*rstd+beta)
// Elementwise kernels: block = 16 column groups (one 64-channel slab) x 16 row lanes, EW_ROWS rows per workgroup; every
// thread keeps the per-channel parameters of its 4 channels in registers and streams EW_ROWS/16 rows (independent
// 16-B loads in flight) — no per-element 64-bit index division, no per-element parameter reloads.
#define EW_ROWS 128
#define EW_PER (EW_ROWS / 16)      // rows per thread of the BatchNorm element-wise kernels
template <typename T>
__global__ __launch_bounds__(256) void k_bn_act_fwd(const T* __restrict__ X, int ldx, int n, int C,
                                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    int act, T* __restrict__ Y, int ldy) {
    const BnLanes L = bn_lanes(C);
    const int cg = L.cg, rl = L.rl;
    const int c = blockIdx.y * (L.cgs * 4) + cg * 4;
    if (c >= C || rl >= L.rl_n) return;
    const float4 m = *reinterpret_cast<const float4*>(mean + c);
    const float4 s = *reinterpret_cast<const float4*>(rstd + c);
    const float4 g = gamma ? *reinterpret_cast<const float4*>(gamma + c) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 b = beta ? *reinterpret_cast<const float4*>(beta + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    const int r0 = blockIdx.x * (EW_PER * L.rl_n) + rl;
    float4 v[EW_PER];
#pragma unroll
    for (int j = 0; j < EW_PER; ++j) {
        const int r = r0 + L.rl_n * j;
        if (r < n) v[j] = ld4(X + (long long)r * ldx + c);
    }
#pragma unroll
    for (int j = 0; j < EW_PER; ++j) {
        const int r = r0 + L.rl_n * j;
        if (r >= n) continue;
        float4 o;
        o.x = act_fwd((v[j].x - m.x) * s.x * g.x + b.x, act);
        o.y = act_fwd((v[j].y - m.y) * s.y * g.y + b.y, act);
        o.z = act_fwd((v[j].z - m.z) * s.z * g.z + b.z, act);
        o.w = act_fwd((v[j].w - m.w) * s.w * g.w + b.w, act);
        st4(Y + (long long)r * ldy + c, o);
    }
}

// ---------------------------------------------------------------- backward
// pass 1: per chunk, sum(dz) and sum(dz * xhat) with dz = dy * act'(z), z recomputed from x
template <typename T>
__global__ __launch_bounds__(256) void k_bn_act_bwd_partial(const T* __restrict__ X, int ldx,
                                                            const T* __restrict__ dY, int ldy, int n, int C,
                                                            int rows_per_chunk, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int act,
                                                            float* __restrict__ part) {
    __shared__ float s_a[1024];
    __shared__ float s_b[1024];
    const BnLanes L = bn_lanes(C);
    const int cg = L.cg, rl = L.rl, slab = L.cgs * 4;
    const int c = blockIdx.y * slab + cg * 4;
    const int r_beg = blockIdx.x * rows_per_chunk;
    const int r_end = min(n, r_beg + rows_per_chunk);
    float sa[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < C && rl < L.rl_n) {
        float m[4], s[4], g[4], b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            m[j] = mean[c + j];
            s[j] = rstd[c + j];
            g[j] = gamma ? gamma[c + j] : 1.f;
            b[j] = beta ? beta[c + j] : 0.f;
        }
        for (int r = r_beg + rl; r < r_end; r += L.rl_n) {
            float4 xv = ld4(X + (long long)r * ldx + c);
            float4 dv = ld4(dY + (long long)r * ldy + c);
            float x[4] = {xv.x, xv.y, xv.z, xv.w}, d[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float xh = (x[j] - m[j]) * s[j];
                float dz = d[j] * act_grad(xh * g[j] + b[j], act);
                sa[j] += dz;
                sb[j] += dz * xh;
            }
        }
    }
    if (rl < L.rl_n) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s_a[rl * slab + cg * 4 + j] = sa[j];
            s_b[rl * slab + cg * 4 + j] = sb[j];
        }
    }
    __syncthreads();
    if (threadIdx.x < slab) {
        int cc = blockIdx.y * slab + threadIdx.x;
        float a = 0.f, b = 0.f;
        for (int j = 0; j < L.rl_n; ++j) {
            a += s_a[j * slab + threadIdx.x];
            b += s_b[j * slab + threadIdx.x];
        }
        if (cc < C) {
            float* p = part + (long long)blockIdx.x * 2 * C;
            p[cc] = a;
            p[C + cc] = b;
        }
    }
}

// block = 4 channels x 256 chunk lanes (1024 threads), like k_bn_stats_fold: every lane loads its (up to 8) chunk
// partials up front, adds them pairwise, the 16 lanes of a wave that share a channel meet by shuffles, the 16 waves in
// LDS.  Fixed tree: deterministic.  (64 lanes walking 32 chunks each + one thread adding 64 partials: 19 us at 2048
// chunks, on the critical path of every BatchNorm backward.)
// colsum (optional): the column sums of dX = the bias gradient of the convolution that feeds this BatchNorm, in closed
// form.  dX = gamma rstd (dz - [training](dbeta + xhat dgamma) / n), dbeta = sum dz, so in training mode
// sum_rows dX = -gamma rstd dgamma sum(xhat) / n = 0 (batch statistics: sum(xhat) = 0; BatchNorm is blind to a constant
// added to its input), and with running statistics it is gamma rstd dbeta.  (Summing dX in the apply pass instead — one
// float atomic per column and workgroup, 3290 atomics per address on the stem level — cost 41 us of a 158 us backward
// there, and produced rounding noise around that zero.)
__global__ __launch_bounds__(1024) void k_bn_bwd_fold(const float* __restrict__ part, int chunks, int C,
                                                      float* dbeta, float* dgamma, float* colsum,
                                                      const float* __restrict__ gamma, const float* __restrict__ rstd,
                                                      int training) {
    __shared__ float s_a[16][FOLD_CH], s_b[16][FOLD_CH];
    const int cl = threadIdx.x & (FOLD_CH - 1), lane = threadIdx.x / FOLD_CH;
    const int c = min(blockIdx.x * FOLD_CH + cl, C - 1);
    float av[FOLD_PER], bv[FOLD_PER];
#pragma unroll
    for (int u = 0; u < FOLD_PER; ++u) {
        const int j = lane + u * FOLD_LANES;
        av[u] = 0.f; bv[u] = 0.f;
        if (j < chunks) {
            const float* p = part + (long long)j * 2 * C;
            av[u] = p[c]; bv[u] = p[C + c];
        }
    }
    float a = ((av[0] + av[1]) + (av[2] + av[3])) + ((av[4] + av[5]) + (av[6] + av[7]));
    float b = ((bv[0] + bv[1]) + (bv[2] + bv[3])) + ((bv[4] + bv[5]) + (bv[6] + bv[7]));
    for (int j = lane + FOLD_PER * FOLD_LANES; j < chunks; j += FOLD_LANES) {
        const float* p = part + (long long)j * 2 * C;
        a += p[c]; b += p[C + c];
    }
    for (int o = FOLD_CH; o < 64; o *= 2) {
        a += __shfl_down(a, o);
        b += __shfl_down(b, o);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) < FOLD_CH) { s_a[wave][cl] = a; s_b[wave][cl] = b; }
    __syncthreads();
    if (threadIdx.x < FOLD_CH && blockIdx.x * FOLD_CH + cl < C) {
        a = s_a[0][cl]; b = s_b[0][cl];
        for (int w = 1; w < 16; ++w) { a += s_a[w][cl]; b += s_b[w][cl]; }
        dbeta[c] = a;
        dgamma[c] = b;
        if (colsum) colsum[c] = training ? 0.f : (gamma ? gamma[c] : 1.f) * rstd[c] * a;
    }
}

// pass 2: dx = gamma*rstd*(dz - [training] (dbeta + xhat*dgamma)/n)
template <typename T>
__global__ __launch_bounds__(256) void k_bn_act_bwd_apply(const T* __restrict__ X, int ldx,
                                                          const T* __restrict__ dY, int ldy, int n, int C,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          int act, const float* __restrict__ dbeta,
                                                          const float* __restrict__ dgamma, int training,
                                                          T* __restrict__ dX, int lddx) {
    const BnLanes L = bn_lanes(C);
    const int cg = L.cg, rl = L.rl;
    if (rl >= L.rl_n) return;
    const int c = min(blockIdx.y * (L.cgs * 4) + cg * 4, C - 4);   // a partial last slab recomputes its last group (C % 4 == 0)
    float m[4], s[4], g[4], b[4], db[4], dg[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        m[j] = mean[c + j];
        s[j] = rstd[c + j];
        g[j] = gamma ? gamma[c + j] : 1.f;
        b[j] = beta ? beta[c + j] : 0.f;
        db[j] = dbeta[c + j];
        dg[j] = dgamma[c + j];
    }
    const float inv_n = training ? 1.f / (float)n : 0.f;
    const int r0 = blockIdx.x * (EW_PER * L.rl_n) + rl;
    constexpr int NR = EW_PER, HALF = NR / 2;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float4 xv[HALF], dv[HALF];
#pragma unroll
        for (int j = 0; j < HALF; ++j) {
            const int r = r0 + L.rl_n * (h * HALF + j);
            if (r < n) {
                xv[j] = ld4(X + (long long)r * ldx + c);
                dv[j] = ld4(dY + (long long)r * ldy + c);
            }
        }
#pragma unroll
        for (int j = 0; j < HALF; ++j) {
            const int r = r0 + L.rl_n * (h * HALF + j);
            if (r >= n) continue;
            const float x[4] = {xv[j].x, xv[j].y, xv[j].z, xv[j].w}, d[4] = {dv[j].x, dv[j].y, dv[j].z, dv[j].w};
            float o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float xh = (x[q] - m[q]) * s[q];
                float dz = d[q] * act_grad(xh * g[q] + b[q], act);
                o[q] = g[q] * s[q] * (dz - (db[q] + xh * dg[q]) * inv_n);
            }
            st4(dX + (long long)r * lddx + c, make_float4(o[0], o[1], o[2], o[3]));
        }
    }
}

// ---------------------------------------------------------------- residual tail: y = act(a * s[batch] + r)
// (the drop-path scale s is optional; reference call sites resnet_block.py:70-73, senet_block.py:92-96)
template <typename T>
__global__ __launch_bounds__(256) void k_add_act_fwd(const T* __restrict__ A, int lda, const T* __restrict__ R,
                                                     int ldr, const float* __restrict__ scale,
                                                     const int32_t* __restrict__ coords, int n, int C, int act,
                                                     T* __restrict__ Y, int ldy) {
    const int cg = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.y * 64 + cg * 4;
    if (c >= C) return;
    const int r0 = blockIdx.x * EW_ROWS + rl;
    constexpr int NR = EW_ROWS / 16;
    float4 av[NR], bv[NR];
    float sv[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int r = r0 + 16 * j;
        if (r < n) {
            av[j] = ld4(A + (long long)r * lda + c);
            bv[j] = ld4(R + (long long)r * ldr + c);
            sv[j] = scale ? scale[coords[4 * (long long)r]] : 1.f;
        }
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int r = r0 + 16 * j;
        if (r >= n) continue;
        float4 o;
        o.x = act_fwd(av[j].x * sv[j] + bv[j].x, act);
        o.y = act_fwd(av[j].y * sv[j] + bv[j].y, act);
        o.z = act_fwd(av[j].z * sv[j] + bv[j].z, act);
        o.w = act_fwd(av[j].w * sv[j] + bv[j].w, act);
        st4(Y + (long long)r * ldy + c, o);
    }
}

// dA = dz * s, dR = dz with dz = dY * act'(a*s + r)
template <typename T>
__global__ __launch_bounds__(256) void k_add_act_bwd(const T* __restrict__ A, int lda, const T* __restrict__ R,
                                                     int ldr, const float* __restrict__ scale,
                                                     const int32_t* __restrict__ coords, const T* __restrict__ dY,
                                                     int ldy, int n, int C, int act, T* __restrict__ dA,
                                                     T* __restrict__ dR) {
    const int cg = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.y * 64 + cg * 4;
    if (c >= C) return;
    const int r0 = blockIdx.x * EW_ROWS + rl;
    constexpr int NR = EW_ROWS / 16, HALF = NR / 2;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float4 av[HALF], bv[HALF], dv[HALF];
        float sv[HALF];
#pragma unroll
        for (int j = 0; j < HALF; ++j) {
            const int r = r0 + 16 * (h * HALF + j);
            if (r < n) {
                av[j] = ld4(A + (long long)r * lda + c);
                bv[j] = ld4(R + (long long)r * ldr + c);
                dv[j] = ld4(dY + (long long)r * ldy + c);
                sv[j] = scale ? scale[coords[4 * (long long)r]] : 1.f;
            }
        }
#pragma unroll
        for (int j = 0; j < HALF; ++j) {
            const int r = r0 + 16 * (h * HALF + j);
            if (r >= n) continue;
            const float s = sv[j];
            float4 dz;
            dz.x = dv[j].x * act_grad(av[j].x * s + bv[j].x, act);
            dz.y = dv[j].y * act_grad(av[j].y * s + bv[j].y, act);
            dz.z = dv[j].z * act_grad(av[j].z * s + bv[j].z, act);
            dz.w = dv[j].w * act_grad(av[j].w * s + bv[j].w, act);
            const long long o = (long long)r * C + c;
            if (dR) st4(dR + o, dz);
            if (dA) st4(dA + o, make_float4(dz.x * s, dz.y * s, dz.z * s, dz.w * s));
        }
    }
}

// ---------------------------------------------------------------- squeeze-excite block tail, fused
// The tail of an SE residual block (senet_block.py:83-96, 126-147; resnet_block.py:70-73)
//     t = BatchNorm(z);  s = sigmoid(W2 act(W1 avgpool_plot(t) + b1) + b2);  y = act(t * s[plot] * keep[plot] + r)
// ran as BatchNorm (statistics + apply), per-plot pooling, excitation MLP, broadcast multiplication and residual kernel:
// 9 passes over [N, C] forward and 14 backward.  Here t and t*s never exist in memory:
//   forward : ONE pass over z for the BatchNorm partials AND the per-plot column sums (the pooled input of the MLP is an
//             affine function of the plot mean of z), then y from (z, r) in one pass                     — 4 passes
//   backward: one pass for the per-plot sums of da = dy act'(.) and da*xhat (they give the gradient of s, and — once the MLP
//             backward has produced the pooled gradient — dbeta / dgamma in closed form per plot), one pass for dz and dr
//                                                                                                           — 8 passes
// Rows of a plot are contiguous.  The row chunks of the reduction kernels are PLOT-ALIGNED — plot b gets
// ceil(rows_b / R) chunks of its own — so every chunk partial belongs to one plot, per-plot quantities are folds of the
// plot's chunk range in a fixed order (no atomics: results are bitwise reproducible, like the BatchNorm statistics), and
// the per-plot mean of z falls out of the BatchNorm partials themselves (Chan's combine over the plot's chunks).
// The grid is sized for the worst case ceil(n / R) + B chunks; the surplus writes empty partials.
struct PlotChunk { int b, r_beg, r_end; };
__device__ __forceinline__ PlotChunk plot_chunk(const int32_t* __restrict__ ptr, int B, int R, int g) {
    if (ptr == nullptr) {                 // no plots: B carries the row count, plain consecutive chunks
        const long long rb = (long long)g * R;
        if (rb >= B) return PlotChunk{-1, 0, 0};
        return PlotChunk{0, (int)rb, (int)min((long long)B, rb + R)};
    }
    int acc = 0;
    for (int b = 0; b < B; ++b) {
        const int lo = ptr[b], hi = ptr[b + 1];
        const int nb = (hi - lo + R - 1) / R;
        if (g < acc + nb) {
            const int rb = lo + (g - acc) * R;
            return PlotChunk{b, rb, min(hi, rb + R)};
        }
        acc += nb;
    }
    return PlotChunk{-1, 0, 0};
}
// first chunk and number of chunks of plot b
__device__ __forceinline__ void plot_chunks_of(const int32_t* __restrict__ ptr, int R, int b, int* first, int* count) {
    int acc = 0;
    for (int i = 0; i < b; ++i) acc += (ptr[i + 1] - ptr[i] + R - 1) / R;
    *first = acc;
    *count = (ptr[b + 1] - ptr[b] + R - 1) / R;
}

// grid (max chunks, slabs): part[chunk][3][C] = (count, mean, M2) of the chunk's rows, as k_bn_stats_partial
template <typename T>
__global__ __launch_bounds__(256) void k_tail_stats(const T* __restrict__ X, int ldx, const int32_t* __restrict__ ptr,
                                                    int B, int C, int R, float* __restrict__ part) {
    __shared__ float s_mean[1024];
    __shared__ float s_m2[1024];
    __shared__ float s_cnt[256];
    const BnLanes L = bn_lanes(C);
    const int cg = L.cg, rl = L.rl, slab = L.cgs * 4;
    const int c = blockIdx.y * slab + cg * 4;
    const PlotChunk pc = plot_chunk(ptr, B, R, blockIdx.x);
    float mean[4] = {0.f, 0.f, 0.f, 0.f}, m2[4] = {0.f, 0.f, 0.f, 0.f};
    float cnt = 0.f;
    if (rl < L.rl_n && c < C) {
        for (int r = pc.r_beg + rl; r < pc.r_end; r += L.rl_n) {
            const float4 v = ld4(X + (long long)r * ldx + c);
            cnt += 1.f;
            const float inv = 1.f / cnt;
            const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = x[j] - mean[j];
                mean[j] += d * inv;
                m2[j] += d * (x[j] - mean[j]);
            }
        }
    } else if (rl < L.rl_n) {
        for (int r = pc.r_beg + rl; r < pc.r_end; r += L.rl_n) cnt += 1.f;
    }
    if (rl < L.rl_n) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s_mean[rl * slab + cg * 4 + j] = mean[j];
            s_m2[rl * slab + cg * 4 + j] = m2[j];
        }
        if (cg == 0) s_cnt[rl] = cnt;
    }
    __syncthreads();
    if (threadIdx.x < slab) {
        const int cc = blockIdx.y * slab + threadIdx.x;
        float na = 0.f, ma = 0.f, qa = 0.f;
        for (int j = 0; j < L.rl_n; ++j) {  // fixed order
            const float nb = s_cnt[j];
            if (nb == 0.f) continue;
            const float mb = s_mean[j * slab + threadIdx.x], qb = s_m2[j * slab + threadIdx.x];
            const float nt = na + nb, d = mb - ma;
            ma += d * (nb / nt);
            qa += qb + d * d * (na * nb / nt);
            na = nt;
        }
        if (cc < C) {
            float* p = part + (long long)blockIdx.x * 3 * C;
            p[cc] = na;
            p[C + cc] = ma;
            p[2 * C + cc] = qa;
        }
    }
}

// zbar[b,c] = plot mean of z (Chan's combine of the plot's chunk partials, fixed order) and
// p[b,c] = BatchNorm(zbar) = (zbar - mean) rstd gamma + beta: the excitation MLP's input
__global__ void k_tail_pool(const float* __restrict__ part, const int32_t* __restrict__ ptr, int B, int C, int R,
                            const float* __restrict__ mean, const float* __restrict__ rstd,
                            const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ zbar,
                            float* __restrict__ p) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * C) return;
    const int b = t / C, c = t - b * C;
    int first, count;
    plot_chunks_of(ptr, R, b, &first, &count);
    float na = 0.f, ma = 0.f;
    for (int j = 0; j < count; ++j) {
        const float* q = part + (long long)(first + j) * 3 * C;
        const float nb = q[c];
        if (nb == 0.f) continue;
        const float nt = na + nb;
        ma += (q[C + c] - ma) * (nb / nt);
        na = nt;
    }
    const float zb = na > 0.f ? ma : mean[c];
    zbar[t] = zb;
    p[t] = (zb - mean[c]) * rstd[c] * (gamma ? gamma[c] : 1.f) + (beta ? beta[c] : 0.f);
}

struct TailParams {
    const float* mean; const float* rstd; const float* gamma; const float* beta;
    const float* s;      // [B, C] excitation
    const float* keep;   // [B] drop-path factor or NULL
};

// y = act(((z - mean) rstd gamma + beta) * s[plot] * keep[plot] + r)
template <typename T>
__global__ __launch_bounds__(256) void k_tail_fwd(const T* __restrict__ Z, int ldz, const T* __restrict__ R, int ldr,
                                                  const int4* __restrict__ coords, TailParams P, int act, int n, int C,
                                                  T* __restrict__ Y, int ldy) {
    const BnLanes L = bn_lanes(C);
    if (L.rl >= L.rl_n) return;
    const int c = min(blockIdx.y * (L.cgs * 4) + L.cg * 4, C - 4);
    float m[4], sc[4], sh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        m[j] = P.mean[c + j];
        sc[j] = P.rstd[c + j] * (P.gamma ? P.gamma[c + j] : 1.f);
        sh[j] = P.beta ? P.beta[c + j] : 0.f;
    }
    const int r0 = blockIdx.x * (EW_PER * L.rl_n) + L.rl;
#pragma unroll
    for (int j = 0; j < EW_PER; ++j) {
        const int r = r0 + L.rl_n * j;
        if (r >= n) continue;
        const int b = coords ? coords[r].x : 0;
        const float kf = P.keep ? P.keep[b] : 1.f;
        const float4 z4 = ld4(Z + (long long)r * ldz + c);
        const float4 r4 = ld4(R + (long long)r * ldr + c);
        const float4 s4 = P.s ? *reinterpret_cast<const float4*>(P.s + (long long)b * C + c) : make_float4(1.f, 1.f, 1.f, 1.f);
        const float z[4] = {z4.x, z4.y, z4.z, z4.w}, rr[4] = {r4.x, r4.y, r4.z, r4.w}, ss[4] = {s4.x, s4.y, s4.z, s4.w};
        float o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = act_fwd(((z[q] - m[q]) * sc[q] + sh[q]) * (ss[q] * kf) + rr[q], act);
        st4(Y + (long long)r * ldy + c, make_float4(o[0], o[1], o[2], o[3]));
    }
}

// chunk partials spart[chunk][2][C] = (sum da, sum da * xhat) over the chunk's rows, da = dy * act'(pre), pre recomputed
template <typename T>
__global__ __launch_bounds__(256) void k_tail_bwd_sums(const T* __restrict__ Z, int ldz, const T* __restrict__ R,
                                                       int ldr, const T* __restrict__ dY, int ldy,
                                                       const int32_t* __restrict__ ptr, int B, TailParams P, int act, int C,
                                                       int RC, float* __restrict__ spart) {
    __shared__ float s_a[1024];
    __shared__ float s_b[1024];
    const BnLanes L = bn_lanes(C);
    const int cg = L.cg, rl = L.rl, slab = L.cgs * 4;
    const int c = blockIdx.y * slab + cg * 4;
    const PlotChunk pc = plot_chunk(ptr, B, RC, blockIdx.x);
    float a2[4] = {0.f, 0.f, 0.f, 0.f}, a3[4] = {0.f, 0.f, 0.f, 0.f};
    if (rl < L.rl_n && c < C && pc.b >= 0) {
        float m[4], rs[4], g[4], be[4], sk[4];
        const float kf = P.keep ? P.keep[pc.b] : 1.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            m[j] = P.mean[c + j];
            rs[j] = P.rstd[c + j];
            g[j] = P.gamma ? P.gamma[c + j] : 1.f;
            be[j] = P.beta ? P.beta[c + j] : 0.f;
            sk[j] = (P.s ? P.s[(long long)pc.b * C + c + j] : 1.f) * kf;
        }
        for (int r = pc.r_beg + rl; r < pc.r_end; r += L.rl_n) {
            const float4 z4 = ld4(Z + (long long)r * ldz + c);
            const float4 r4 = ld4(R + (long long)r * ldr + c);
            const float4 d4 = ld4(dY + (long long)r * ldy + c);
            const float z[4] = {z4.x, z4.y, z4.z, z4.w}, rr[4] = {r4.x, r4.y, r4.z, r4.w}, d[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float xh = (z[q] - m[q]) * rs[q];
                const float da = d[q] * act_grad((xh * g[q] + be[q]) * sk[q] + rr[q], act);
                a2[q] += da;
                a3[q] += da * xh;
            }
        }
    }
    if (rl < L.rl_n) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { s_a[rl * slab + cg * 4 + j] = a2[j]; s_b[rl * slab + cg * 4 + j] = a3[j]; }
    }
    __syncthreads();
    if (threadIdx.x < slab) {
        const int cc = blockIdx.y * slab + threadIdx.x;
        float a = 0.f, b = 0.f;
        for (int j = 0; j < L.rl_n; ++j) { a += s_a[j * slab + threadIdx.x]; b += s_b[j * slab + threadIdx.x]; }
        if (cc < C) {
            float* p = spart + (long long)blockIdx.x * 2 * C;
            p[cc] = a;
            p[C + cc] = b;
        }
    }
}

// S2, S3 [B, C] = the plot's chunk partials summed in order; ds[b,c] = keep[b] * (gamma S3 + beta S2): the gradient of the
// excitation s (sum over the plot's rows of da * t)
__global__ void k_tail_bwd_ds(const float* __restrict__ spart, const int32_t* __restrict__ ptr, int RC,
                              const float* __restrict__ gamma, const float* __restrict__ beta,
                              const float* __restrict__ keep, int B, int C, float* __restrict__ S2, float* __restrict__ S3,
                              float* __restrict__ ds) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * C) return;
    const int b = t / C, c = t - b * C;
    int first, count;
    plot_chunks_of(ptr, RC, b, &first, &count);
    float a = 0.f, q = 0.f;
    for (int j = 0; j < count; ++j) {
        const float* p = spart + (long long)(first + j) * 2 * C;
        a += p[c];
        q += p[C + c];
    }
    S2[t] = a;
    S3[t] = q;
    ds[t] = (keep ? keep[b] : 1.f) * ((gamma ? gamma[c] : 1.f) * q + (beta ? beta[c] : 0.f) * a);
}

// With dp[b,c] = gradient of the pooled MLP input: dte = dp / rows (the per-row share of the pooled gradient, in the space
// of t), and the BatchNorm parameter gradients in closed form over the plots:
//   dbeta  = sum_rows dt        = sum_b (s keep S2 + rows dte)
//   dgamma = sum_rows dt * xhat = sum_b (s keep S3 + dte X1),  X1[b,c] = sum_plot xhat = rows (zbar - mean) rstd
__global__ __launch_bounds__(256) void k_tail_bwd_fold(const float* __restrict__ S2, const float* __restrict__ S3,
                                                       const float* __restrict__ zbar, const int32_t* __restrict__ ptr,
                                                       const float* __restrict__ dp, const float* __restrict__ s,
                                                       const float* __restrict__ keep, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, int B, int C,
                                                       float* __restrict__ dte, float* __restrict__ dbeta,
                                                       float* __restrict__ dgamma) {
    // 64 columns x 4 plot lanes per workgroup (one thread walking all the plots of a column made this tiny kernel 23 us
    // of dependent loads); lane partials meet in LDS in a fixed order
    __shared__ float s_db[4][64], s_dg[4][64];
    const int cl = threadIdx.x & 63, bl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float db = 0.f, dg = 0.f;
    if (c < C) {
        const float mu = mean[c], rs = rstd[c];
        for (int b = bl; b < B; b += 4) {
            const long long t = (long long)b * C + c;
            const float rows = (float)(ptr[b + 1] - ptr[b]);
            const float e = rows > 0.f ? dp[t] / rows : 0.f;
            dte[t] = e;
            const float sk = s[t] * (keep ? keep[b] : 1.f);
            db += sk * S2[t] + rows * e;
            dg += sk * S3[t] + e * rows * (zbar[t] - mu) * rs;
        }
    }
    s_db[bl][cl] = db;
    s_dg[bl][cl] = dg;
    __syncthreads();
    if (bl == 0 && c < C) {
        dbeta[c] = (s_db[0][cl] + s_db[1][cl]) + (s_db[2][cl] + s_db[3][cl]);
        dgamma[c] = (s_dg[0][cl] + s_dg[1][cl]) + (s_dg[2][cl] + s_dg[3][cl]);
    }
}

// dz = gamma rstd (dt - [training](dbeta + xhat dgamma) / n),  dt = da s keep + dte[plot];   dr = da
template <typename T>
__global__ __launch_bounds__(256) void k_tail_bwd_apply(const T* __restrict__ Z, int ldz, const T* __restrict__ R,
                                                        int ldr, const T* __restrict__ dY, int ldy,
                                                        const int4* __restrict__ coords, TailParams P,
                                                        const float* __restrict__ dte, const float* __restrict__ dbeta,
                                                        const float* __restrict__ dgamma, int act, int training, int n,
                                                        int C, T* __restrict__ dZ, int lddz, T* __restrict__ dR,
                                                        int lddr) {
    const BnLanes L = bn_lanes(C);
    if (L.rl >= L.rl_n) return;
    const int c = min(blockIdx.y * (L.cgs * 4) + L.cg * 4, C - 4);
    float m[4], rs[4], g[4], be[4], db[4], dg[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        m[j] = P.mean[c + j];
        rs[j] = P.rstd[c + j];
        g[j] = P.gamma ? P.gamma[c + j] : 1.f;
        be[j] = P.beta ? P.beta[c + j] : 0.f;
        db[j] = dbeta[c + j];
        dg[j] = dgamma[c + j];
    }
    const float inv_n = training ? 1.f / (float)n : 0.f;
    const int r0 = blockIdx.x * (EW_PER * L.rl_n) + L.rl;
#pragma unroll 2
    for (int j = 0; j < EW_PER; ++j) {
        const int r = r0 + L.rl_n * j;
        if (r >= n) continue;
        const int b = coords ? coords[r].x : 0;
        const float kf = P.keep ? P.keep[b] : 1.f;
        const float4 z4 = ld4(Z + (long long)r * ldz + c);
        const float4 r4 = ld4(R + (long long)r * ldr + c);
        const float4 d4 = ld4(dY + (long long)r * ldy + c);
        const float4 s4 = P.s ? *reinterpret_cast<const float4*>(P.s + (long long)b * C + c) : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 e4 = dte ? *reinterpret_cast<const float4*>(dte + (long long)b * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float z[4] = {z4.x, z4.y, z4.z, z4.w}, rr[4] = {r4.x, r4.y, r4.z, r4.w}, d[4] = {d4.x, d4.y, d4.z, d4.w},
                    ss[4] = {s4.x, s4.y, s4.z, s4.w}, ee[4] = {e4.x, e4.y, e4.z, e4.w};
        float oz[4], orr[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float xh = (z[q] - m[q]) * rs[q];
            const float sk = ss[q] * kf;
            const float da = d[q] * act_grad((xh * g[q] + be[q]) * sk + rr[q], act);
            const float dt = da * sk + ee[q];
            oz[q] = g[q] * rs[q] * (dt - (db[q] + xh * dg[q]) * inv_n);
            orr[q] = da;
        }
        if (dZ) st4(dZ + (long long)r * lddz + c, make_float4(oz[0], oz[1], oz[2], oz[3]));
        if (dR) st4(dR + (long long)r * lddr + c, make_float4(orr[0], orr[1], orr[2], orr[3]));
    }
}

// =============================================================== C ABI
extern "C" {

int agb_bn_chunks(int n) {
    // ~128 rows per chunk, at most 2048 chunks (x C/64 column slabs of workgroups): enough workgroups to keep every
    // CU's load queue full on the 64-channel levels (512 chunks left 2 workgroups per CU: latency-bound at ~2.5 TB/s)
    int chunks = agb_cdiv(n > 0 ? n : 1, 128);
    return chunks > 2048 ? 2048 : chunks;
}

static int rows_per_chunk(int n, int chunks) { return agb_cdiv(n > 0 ? n : 1, chunks); }

// chunks actually used (<= agb_bn_chunks(n), which sizes the callers' scratch): every row lane of a workgroup gets at
// least 8 rows, so the narrow layouts (32 / 64 row lanes) do not pay their longer in-block fold for two rows each
static int bn_chunks_for(int n, int C) {
    const int cap = agb_bn_chunks(n), want = agb_cdiv(n > 0 ? n : 1, 8 * bn_row_lanes(C));
    return want < cap ? want : cap;
}

// The second half of agb_bn_stats_tracked on partials produced elsewhere (agb_dense_fwd_bn: the epilogue of the product
// that wrote X): part float[chunks][3][C] = (count, mean, M2) per chunk and column.
int agb_bn_stats_fold(const float* part, int chunks, int C, float eps, float momentum, float* mean, float* rstd,
                      float* running_mean, float* running_var, long long* num_batches_tracked, void* stream) {
    AGB_CHECK_ARG(part != nullptr && chunks >= 1 && C >= 1, "agb_bn_stats_fold: %d chunks, %d columns", chunks, C);
    hipLaunchKernelGGL(k_bn_stats_fold, dim3(agb_cdiv(C, FOLD_CH)), dim3(1024), 0, (hipStream_t)stream, part, chunks, C,
                       eps, momentum, mean, rstd, running_mean, running_var, num_batches_tracked);
    AGB_CHECK_LAUNCH("agb_bn_stats_fold");
    return AGB_OK;
}

// dbeta[c] = sum over chunks of part[chunk][0][c], dgamma[c] = ... part[chunk][1][c] (fixed order): the fold half of
// agb_bn_act_bwd on partials produced by agb_se_tail_bwd_sums without plots
int agb_bn_bwd_fold(const float* part, int chunks, int C, float* dbeta, float* dgamma, void* stream) {
    AGB_CHECK_ARG(part && chunks >= 1 && C >= 1, "agb_bn_bwd_fold: %d chunks, %d columns", chunks, C);
    hipLaunchKernelGGL(k_bn_bwd_fold, dim3(agb_cdiv(C, FOLD_CH)), dim3(1024), 0, (hipStream_t)stream, part, chunks, C, dbeta,
                       dgamma, (float*)nullptr, (const float*)nullptr, (const float*)nullptr, 1);
    AGB_CHECK_LAUNCH("agb_bn_bwd_fold");
    return AGB_OK;
}

// ---- squeeze-excite block tail (see k_tail_*).  coords: int4[n] (batch index in .x), ptr: int32[B+1] row offsets.
static int tail_rows(int n, int C) { return rows_per_chunk(n, bn_chunks_for(n, C)); }

// Plot-aligned row chunks of the tail's reduction kernels (upper bound): sizes `part` (x 3 C floats) and `spart` (x 2 C)
int agb_se_tail_chunks(int n, int C, int B) { return agb_cdiv(n > 0 ? n : 1, tail_rows(n, C)) + (B > 0 ? B : 0); }

// zbar[b, c] = plot mean of z; pooled[b, c] = BatchNorm(zbar): the input of the excitation MLP (agb_se_mlp_fwd)
int agb_se_tail_pool(const float* part, const int32_t* ptr, int n, int B, int C, const float* mean, const float* rstd,
                     const float* gamma, const float* beta, float* zbar, float* pooled, void* stream) {
    hipLaunchKernelGGL(k_tail_pool, dim3(agb_cdiv((long long)B * C, 256)), dim3(256), 0, (hipStream_t)stream, part, ptr, B, C,
                       tail_rows(n, C), mean, rstd, gamma, beta, zbar, pooled);
    AGB_CHECK_LAUNCH("agb_se_tail_pool");
    return AGB_OK;
}

// S2, S3 float[B, C]: per-plot sums (the plot's chunk partials in order); ds float[B, C]: the gradient of the excitation
// (input of agb_se_mlp_bwd)
int agb_se_tail_bwd_ds(const float* spart, const int32_t* ptr, int n, const float* gamma, const float* beta,
                       const float* keep, int B, int C, float* S2, float* S3, float* ds, void* stream) {
    hipLaunchKernelGGL(k_tail_bwd_ds, dim3(agb_cdiv((long long)B * C, 256)), dim3(256), 0, (hipStream_t)stream, spart, ptr,
                       tail_rows(n, C), gamma, beta, keep, B, C, S2, S3, ds);
    AGB_CHECK_LAUNCH("agb_se_tail_bwd_ds");
    return AGB_OK;
}

// dp float[B, C]: gradient of the pooled MLP input (from agb_se_mlp_bwd) -> dte float[B, C], dbeta, dgamma float[C]
int agb_se_tail_bwd_fold(const float* S2, const float* S3, const float* zbar, const int32_t* ptr, const float* dp,
                         const float* s, const float* keep, const float* mean, const float* rstd, int B, int C, float* dte,
                         float* dbeta, float* dgamma, void* stream) {
    hipLaunchKernelGGL(k_tail_bwd_fold, dim3(agb_cdiv(C, 64)), dim3(256), 0, (hipStream_t)stream, S2, S3, zbar, ptr, dp, s,
                       keep, mean, rstd, B, C, dte, dbeta, dgamma);
    AGB_CHECK_LAUNCH("agb_se_tail_bwd_fold");
    return AGB_OK;
}

// the entry points that take row matrices: once per storage type (float: agb_xxx, bf16: agb_xxx_h)
#define AGB_T float
#define AGB_FN(name) name
#include "norm_rows.inc"
#undef AGB_T
#undef AGB_FN
#define AGB_T bf16_t
#define AGB_FN(name) name##_h
#include "norm_rows.inc"
#undef AGB_T
#undef AGB_FN

}  // extern "C"
