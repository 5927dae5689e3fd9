// pointnet.hip — the tail of MinkowskiPointNet's shared per-point MLP (torch_points3d/modules/MinkowskiEngine/
// PointNet.py:16-29, :43-49): BatchNorm + activation of the 1024-wide embedding fused INTO the per-plot pooling, and the
// matching backward.  The reference (and round 1 of this repo) materialises act(bn(z)) [N, 1024] (3.6 GB at B = 64) only to
// reduce it to [B, 1024] right away, and in the backward pass materialises the broadcast pooled gradient [N, 1024] before
// the BatchNorm backward reads it again.  Here
//   forward   pooled[b, c] = reduce_{rows r of plot b} act(gamma_c (z[r,c] - mean_c) rstd_c + beta_c)      (sum/avg/max)
//   backward  dz[r, c]     = gamma_c rstd_c (g - (sum_r g + zhat sum_r g zhat) / n),  g = dpool[b(r), c] act'(.)
//             (max pooling: g is non-zero only in the winning row of its plot and channel)
// read z twice / three times and never write an [N, C] intermediate other than dz (the operand of the two weight /
// data gradient products that follow).  HBM-bound kernels, 16-B row pieces, fixed-order folds (deterministic).
// agb_pointnet_mlp_fwd chains the whole shared MLP (3 x [own MFMA product -> statistics -> BatchNorm + activation]
// -> pooled embedding) for C callers; the Python host composes the same entry points under autograd.
#include "agb_common.h"
#include <float.h>

#define PN_ROWS 16

// grid (B, C / 64, S): plot b, 64-channel slab, row split s of the plot.  part: [B * S][C] (or Y when S == 1)
__global__ __launch_bounds__(256) void k_pn_pool_fwd(const float* __restrict__ Z, int ldz, const int32_t* __restrict__ ptr,
                                                     int C, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     int act, int mode, int S, float* __restrict__ Y,
                                                     int32_t* __restrict__ arg, float* __restrict__ AUX, int B) {
    // AUX (sum / avg pooling, optional): [2][B * S][C] per-plot sums of act'(.) and act'(.) * zhat — everything the
    // BatchNorm parameter gradients need from z, so that the backward pass does not read the [n, C] matrix for them
    __shared__ float s_val[PN_ROWS][64];
    __shared__ int s_arg[PN_ROWS][64];
    __shared__ float s_x1[PN_ROWS][64], s_x2[PN_ROWS][64];
    const int b = blockIdx.x;
    const int cg = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = min(blockIdx.y * 64 + cg * 4, C - 4);
    const int seg_beg = ptr[b], seg_end = ptr[b + 1];
    const int len = seg_end - seg_beg;
    const int chunk = (len + S - 1) / S;
    const int beg = seg_beg + blockIdx.z * chunk;
    const int end = min(seg_end, beg + chunk);
    float m[4], sd[4], g[4], bb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        m[j] = mean[c + j]; sd[j] = rstd[c + j];
        g[j] = gamma ? gamma[c + j] : 1.f; bb[j] = beta ? beta[c + j] : 0.f;
    }
    float acc[4], x1[4] = {0.f, 0.f, 0.f, 0.f}, x2[4] = {0.f, 0.f, 0.f, 0.f};
    int ai[4] = {-1, -1, -1, -1};
    const bool aux = AUX != nullptr && mode != 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (mode == 2) ? -FLT_MAX : 0.f;
    // four rows in flight per thread
    for (int r0 = beg + rl; r0 < end; r0 += 4 * PN_ROWS) {
        float4 v4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = min(r0 + PN_ROWS * u, end - 1);
            v4[u] = *reinterpret_cast<const float4*>(Z + (long long)r * ldz + c);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r0 + PN_ROWS * u;
            if (r >= end) break;
            const float v[4] = {v4[u].x, v4[u].y, v4[u].z, v4[u].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // (z - mean) * rstd * gamma + beta evaluated like the unfused kernel (k_bn_act_fwd): same rounding
                const float zh = (v[j] - m[j]) * sd[j];
                if (aux) {
                    float y, dy;
                    act_fwd_grad(zh * g[j] + bb[j], act, &y, &dy);
                    acc[j] += y;
                    x1[j] += dy;
                    x2[j] += dy * zh;
                    continue;
                }
                const float y = act_fwd(zh * g[j] + bb[j], act);
                if (mode == 2) {
                    if (y > acc[j]) { acc[j] = y; ai[j] = r; }
                } else {
                    acc[j] += y;
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s_val[rl][cg * 4 + j] = acc[j];
        s_arg[rl][cg * 4 + j] = ai[j];
        s_x1[rl][cg * 4 + j] = x1[j];
        s_x2[rl][cg * 4 + j] = x2[j];
    }
    __syncthreads();
    const int l = threadIdx.x;
    const int cc = blockIdx.y * 64 + l;
    if (aux && l < 64 && cc < C) {
        float a1 = 0.f, a2 = 0.f;
        for (int j = 0; j < PN_ROWS; ++j) { a1 += s_x1[j][l]; a2 += s_x2[j][l]; }
        const long long slot = ((long long)b * S + blockIdx.z) * C + cc;
        AUX[slot] = a1;
        AUX[(long long)B * S * C + slot] = a2;
    }
    // (threads of a partial last slab past C recomputed the last channel group; their columns are not written out)
    if (l < 64 && cc < C) {
        const int src = l;
        float a = s_val[0][src];
        int bi = s_arg[0][src];
        if (mode == 2) {
            for (int j = 1; j < PN_ROWS; ++j) {
                const float v = s_val[j][src];
                const int q = s_arg[j][src];
                if (q >= 0 && (bi < 0 || v > a || (v == a && q < bi))) { a = v; bi = q; }
            }
            if (S == 1 && bi < 0) a = 0.f;
            arg[((long long)b * S + blockIdx.z) * C + cc] = bi;
        } else {
            for (int j = 1; j < PN_ROWS; ++j) a += s_val[j][src];
            if (mode == 1 && S == 1) a = len > 0 ? a / (float)len : 0.f;
        }
        Y[((long long)b * S + blockIdx.z) * C + cc] = a;
    }
}

// fold of the row splits of a plot, in split order
__global__ void k_pn_pool_fold(const float* __restrict__ part, const int32_t* __restrict__ part_arg,
                               const int32_t* __restrict__ ptr, int B, int C, int mode, int S, float* __restrict__ Y,
                               int32_t* __restrict__ arg) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * C) return;
    const int b = t / C, c = t % C;
    if (mode == 2) {
        float best = 0.f;
        int bi = -1;
        for (int s = 0; s < S; ++s) {
            const int a = part_arg[((long long)b * S + s) * C + c];
            const float v = part[((long long)b * S + s) * C + c];
            if (a >= 0 && (bi < 0 || v > best || (v == best && a < bi))) { best = v; bi = a; }
        }
        Y[t] = bi >= 0 ? best : 0.f;
        arg[t] = bi;
    } else {
        float a = 0.f;
        for (int s = 0; s < S; ++s) a += part[((long long)b * S + s) * C + c];
        const int len = ptr[b + 1] - ptr[b];
        if (mode == 1) a = len > 0 ? a / (float)len : 0.f;
        Y[t] = a;
    }
}

// aux[2][B][C] = the row splits of aux_part[2][B * S][C] summed in split order
__global__ void k_pn_aux_fold(const float* __restrict__ aux_part, int B, int C, int S, float* __restrict__ aux) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2LL * B * C) return;
    const int which = (int)(t / ((long long)B * C));
    const long long u = t - (long long)which * B * C;
    const int b = (int)(u / C), c = (int)(u % C);
    float a = 0.f;
    for (int s = 0; s < S; ++s) a += aux_part[((long long)which * B * S + (long long)b * S + s) * C + c];
    aux[t] = a;
}

// sum / avg pooling with the forward's per-plot sums: dbeta[c] = sum_b g_b A1[b,c], dgamma[c] = sum_b g_b A2[b,c],
// g_b = dpooled[b,c] (/ rows of the plot for avg) — B terms per channel instead of a pass over [n, C]
__global__ void k_pn_bwd_sums_aux(const float* __restrict__ dP, const float* __restrict__ aux,
                                  const int32_t* __restrict__ ptr, int B, int C, int mode, float* __restrict__ dbeta,
                                  float* __restrict__ dgamma) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float a = 0.f, q = 0.f;
    for (int b = 0; b < B; ++b) {
        float g = dP[(long long)b * C + c];
        if (mode == 1) {
            const int len = ptr[b + 1] - ptr[b];
            g = len > 0 ? g / (float)len : 0.f;
        }
        a += g * aux[(long long)b * C + c];
        q += g * aux[((long long)B + b) * C + c];
    }
    dbeta[c] = a;
    dgamma[c] = q;
}

// upstream gradient of element (r, c .. c+3): the pooled gradient of the row's plot, routed by the pooling mode
__device__ __forceinline__ void pn_upstream(const float* __restrict__ dP, const int32_t* __restrict__ arg,
                                            const int32_t* __restrict__ ptr, int mode, int b, int r, int c, int C,
                                            float d[4]) {
    const float4 p = *reinterpret_cast<const float4*>(dP + (long long)b * C + c);
    d[0] = p.x; d[1] = p.y; d[2] = p.z; d[3] = p.w;
    if (mode == 1) {
        const float inv = 1.f / (float)(ptr[b + 1] - ptr[b]);
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] *= inv;
    } else if (mode == 2) {
        const int4 a = *reinterpret_cast<const int4*>(arg + (long long)b * C + c);
        d[0] = a.x == r ? d[0] : 0.f; d[1] = a.y == r ? d[1] : 0.f;
        d[2] = a.z == r ? d[2] : 0.f; d[3] = a.w == r ? d[3] : 0.f;
    }
}

// pass 1 of the backward: per row chunk, sum(g) and sum(g * zhat);  grid (chunks, C / 64), part[chunk][2][C]
__global__ __launch_bounds__(256) void k_pn_bwd_partial(const float* __restrict__ Z, int ldz, int n, int C,
                                                        int rows_per_chunk, const int4* __restrict__ coords,
                                                        const int32_t* __restrict__ ptr, const float* __restrict__ dP,
                                                        const int32_t* __restrict__ arg, int mode,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int act, float* __restrict__ part) {
    __shared__ float s_a[PN_ROWS][64];
    __shared__ float s_b[PN_ROWS][64];
    const int cg = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.y * 64 + cg * 4;
    const int r_beg = blockIdx.x * rows_per_chunk;
    const int r_end = min(n, r_beg + rows_per_chunk);
    float sa[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        float m[4], s[4], g[4], bb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            m[j] = mean[c + j]; s[j] = rstd[c + j];
            g[j] = gamma ? gamma[c + j] : 1.f; bb[j] = beta ? beta[c + j] : 0.f;
        }
        for (int r = r_beg + rl; r < r_end; r += PN_ROWS) {
            const float4 zv = *reinterpret_cast<const float4*>(Z + (long long)r * ldz + c);
            const float z[4] = {zv.x, zv.y, zv.z, zv.w};
            float d[4];
            pn_upstream(dP, arg, ptr, mode, coords[r].x, r, c, C, d);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float zh = (z[j] - m[j]) * s[j];
                const float gz = d[j] * act_grad(zh * g[j] + bb[j], act);
                sa[j] += gz;
                sb[j] += gz * zh;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s_a[rl][cg * 4 + j] = sa[j];
        s_b[rl][cg * 4 + j] = sb[j];
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int cc = blockIdx.y * 64 + threadIdx.x;
        float a = 0.f, b = 0.f;
        for (int j = 0; j < PN_ROWS; ++j) {
            a += s_a[j][threadIdx.x];
            b += s_b[j][threadIdx.x];
        }
        if (cc < C) {
            float* p = part + (long long)blockIdx.x * 2 * C;
            p[cc] = a;
            p[C + cc] = b;
        }
    }
}

// fold of the chunk partials (fixed order): dbeta = sum g, dgamma = sum g zhat;  one thread per channel, 4 partial sums
__global__ void k_pn_bwd_fold(const float* __restrict__ part, int chunks, int C, float* __restrict__ dbeta,
                              float* __restrict__ dgamma) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
    int j = 0;
    for (; j + 4 <= chunks; j += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float* p = part + (long long)(j + u) * 2 * C;
            a[u] += p[c]; b[u] += p[C + c];
        }
    }
    for (; j < chunks; ++j) {
        const float* p = part + (long long)j * 2 * C;
        a[0] += p[c]; b[0] += p[C + c];
    }
    dbeta[c] = (a[0] + a[1]) + (a[2] + a[3]);
    dgamma[c] = (b[0] + b[1]) + (b[2] + b[3]);
}

// Max pooling: the upstream gradient is non-zero at ONE row per (plot, channel) — the winner — so the two sums are B terms
// per channel gathered through argmax, not a pass over the [n, C] matrix (4.2 GB for the 1024-wide layer of the point MLP):
// dbeta[c] = sum_b g, dgamma[c] = sum_b g zhat with g = dpooled[b,c] act'(.) at row argmax[b,c].  Fixed order over the plots.
__global__ __launch_bounds__(256) void k_pn_bwd_sums_max(const float* __restrict__ Z, int ldz, int C, int B,
                                                         const float* __restrict__ dP, const int32_t* __restrict__ arg,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         int act, float* __restrict__ dbeta, float* __restrict__ dgamma) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float m = mean[c], s = rstd[c], g = gamma ? gamma[c] : 1.f, bb = beta ? beta[c] : 0.f;
    float a = 0.f, q = 0.f;
    for (int b0 = 0; b0 < B; b0 += 8) {
        float z[8], d[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {            // eight independent gathers in flight
            const int b = b0 + u;
            const int r = b < B ? arg[(long long)b * C + c] : -1;
            z[u] = r >= 0 ? Z[(long long)r * ldz + c] : 0.f;
            d[u] = r >= 0 ? dP[(long long)b * C + c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float zh = (z[u] - m) * s;
            const float gz = d[u] * act_grad(zh * g + bb, act);
            a += gz;
            q += gz * zh;
        }
    }
    dbeta[c] = a;
    dgamma[c] = q;
}

// pass 2: dz = gamma rstd (g - [training] (dbeta + zhat dgamma) / n)
__global__ __launch_bounds__(256) void k_pn_bwd_apply(const float* __restrict__ Z, int ldz, int n, int C,
                                                      const int4* __restrict__ coords, const int32_t* __restrict__ ptr,
                                                      const float* __restrict__ dP, const int32_t* __restrict__ arg,
                                                      int mode, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, int act,
                                                      const float* __restrict__ dbeta, const float* __restrict__ dgamma,
                                                      int training, float* __restrict__ dZ, int lddz) {
    const int cg = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = min(blockIdx.y * 64 + cg * 4, C - 4);
    float m[4], s[4], g[4], bb[4], db[4], dg[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        m[j] = mean[c + j]; s[j] = rstd[c + j];
        g[j] = gamma ? gamma[c + j] : 1.f; bb[j] = beta ? beta[c + j] : 0.f;
        db[j] = dbeta[c + j]; dg[j] = dgamma[c + j];
    }
    const float inv_n = training ? 1.f / (float)n : 0.f;
    const int r0 = blockIdx.x * 128 + rl;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float4 zv[4];
        int bs[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = min(r0 + PN_ROWS * (4 * h + u), n - 1);
            zv[u] = *reinterpret_cast<const float4*>(Z + (long long)r * ldz + c);
            bs[u] = coords[r].x;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r0 + PN_ROWS * (4 * h + u);
            if (r >= n) continue;
            const float z[4] = {zv[u].x, zv[u].y, zv[u].z, zv[u].w};
            float d[4], o[4];
            pn_upstream(dP, arg, ptr, mode, bs[u], r, c, C, d);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float zh = (z[q] - m[q]) * s[q];
                const float gz = d[q] * act_grad(zh * g[q] + bb[q], act);
                o[q] = g[q] * s[q] * (gz - (db[q] + zh * dg[q]) * inv_n);
            }
            *reinterpret_cast<float4*>(dZ + (long long)r * lddz + c) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

extern "C" {
int agb_pointnet_pool_fwd_aux(const float* Z, int ldz, int n, int C, const float* mean, const float* rstd,
                              const float* gamma, const float* beta, int act, const int32_t* ptr, int B, int mode,
                              int splits, float* part, int32_t* part_arg, float* pooled, int32_t* argmax, float* aux_part,
                              float* aux, void* stream);
int agb_pointnet_pool_bwd_aux(const float* Z, int ldz, int n, int C, const int32_t* coords, const int32_t* ptr, int B,
                              const float* dpooled, const int32_t* argmax, int mode, const float* mean, const float* rstd,
                              const float* gamma, const float* beta, int act, int training, float* part, const float* aux,
                              float* dZ, int lddz, float* dgamma, float* dbeta, void* stream);

int agb_bn_chunks(int n);
int agb_bn_stats(const float* X, int ldx, int n, int C, float eps, float momentum, int training, float* part,
                 float* mean, float* rstd, float* running_mean, float* running_var, void* stream);
int agb_bn_act_fwd(const float* X, int ldx, int n, int C, const float* mean, const float* rstd, const float* gamma,
                   const float* beta, int act, float* Y, int ldy, void* stream);
int agb_spconv_fwd_ex(const float* X, int ldx, const float* W, const int32_t* nbr, long long nbr_stride, int kflip,
                      const float* bias, float* Y, int ldy, int n_out, int K3, int Cin, int Cout, const int32_t* perm,
                      const int32_t* tile_cls, const int32_t* cls_tab, int n_tiles, int ksplit, float* partial,
                      void* stream);

// Row splits per plot the pooling wants for n rows in B plots (host helper: sizes `part` / `part_arg`).
int agb_pointnet_pool_splits(int n, int B) {
    int s = (n / (B > 0 ? B : 1)) / 512;
    return s < 1 ? 1 : (s > 64 ? 64 : s);
}

// pooled[b, :] = reduce over the rows ptr[b] .. ptr[b+1] of act(batchnorm(Z)), mode 0 sum / 1 average / 2 max.
// mean, rstd: float[C] (agb_bn_stats); gamma, beta: float[C] or NULL; part: float[B * splits * C] and part_arg:
// int32[B * splits * C] scratch when splits > 1 (part_arg only for max); argmax: int32[B * C] out (max mode: winning row).
int agb_pointnet_pool_fwd(const float* Z, int ldz, int n, int C, const float* mean, const float* rstd, const float* gamma,
                          const float* beta, int act, const int32_t* ptr, int B, int mode, int splits, float* part,
                          int32_t* part_arg, float* pooled, int32_t* argmax, void* stream) {
    return agb_pointnet_pool_fwd_aux(Z, ldz, n, C, mean, rstd, gamma, beta, act, ptr, B, mode, splits, part, part_arg,
                                     pooled, argmax, nullptr, nullptr, stream);
}

// The same, and (sum / avg pooling) aux float[2][B][C]: per-plot sums of act'(.) and act'(.) * zhat for
// agb_pointnet_pool_bwd_aux.  aux_part: float[2][B * splits][C] scratch when splits > 1.
int agb_pointnet_pool_fwd_aux(const float* Z, int ldz, int n, int C, const float* mean, const float* rstd,
                              const float* gamma, const float* beta, int act, const int32_t* ptr, int B, int mode,
                              int splits, float* part, int32_t* part_arg, float* pooled, int32_t* argmax, float* aux_part,
                              float* aux, void* stream) {
    AGB_CHECK_ARG(C % 4 == 0 && C >= 4 && ldz % 4 == 0, "agb_pointnet_pool_fwd: C (%d), ldz must be multiples of 4", C);
    AGB_CHECK_ARG(mode >= 0 && mode <= 2 && act >= 0 && act <= 2, "agb_pointnet_pool_fwd: mode %d, act %d", mode, act);
    AGB_CHECK_ARG(mode != 2 || argmax != nullptr, "agb_pointnet_pool_fwd: max pooling needs an argmax buffer");
    AGB_CHECK_ARG(splits >= 1 && splits <= 1024 && (splits == 1 || (part && (mode != 2 || part_arg))),
                  "agb_pointnet_pool_fwd: splits %d needs scratch buffers", splits);
    if (B == 0) return AGB_OK;
    hipStream_t s = (hipStream_t)stream;
    AGB_CHECK_ARG(aux == nullptr || splits == 1 || aux_part != nullptr, "agb_pointnet_pool_fwd_aux: splits need aux_part");
    float* aux_out = (aux && mode != 2) ? (splits == 1 ? aux : aux_part) : nullptr;
    hipLaunchKernelGGL(k_pn_pool_fwd, dim3(B, agb_cdiv(C, 64), splits), dim3(256), 0, s, Z, ldz, ptr, C, mean, rstd, gamma,
                       beta, act, mode, splits, splits == 1 ? pooled : part, splits == 1 ? argmax : part_arg, aux_out, B);
    if (splits > 1) {
        hipLaunchKernelGGL(k_pn_pool_fold, dim3(agb_cdiv((long long)B * C, 256)), dim3(256), 0, s, part, part_arg, ptr, B,
                           C, mode, splits, pooled, argmax);
        if (aux_out)
            hipLaunchKernelGGL(k_pn_aux_fold, dim3(agb_cdiv(2LL * B * C, 256)), dim3(256), 0, s, aux_part, B, C, splits,
                               aux);
    }
    AGB_CHECK_LAUNCH("agb_pointnet_pool_fwd");
    return AGB_OK;
}

// Backward of agb_pointnet_pool_fwd through the pooling, the activation and the BatchNorm: dZ [n, C], dgamma, dbeta
// float[C].  coords: int32[n][4] (the row's plot in column 0); dpooled: float[B, C]; argmax: from the forward (max mode).
// part: float[agb_bn_chunks(n) * 2 * C] scratch.  training != 0: batch statistics were used in the forward pass.
int agb_pointnet_pool_bwd(const float* Z, int ldz, int n, int C, const int32_t* coords, const int32_t* ptr, int B,
                          const float* dpooled, const int32_t* argmax, int mode, const float* mean, const float* rstd,
                          const float* gamma, const float* beta, int act, int training, float* part, float* dZ, int lddz,
                          float* dgamma, float* dbeta, void* stream) {
    return agb_pointnet_pool_bwd_aux(Z, ldz, n, C, coords, ptr, B, dpooled, argmax, mode, mean, rstd, gamma, beta, act,
                                     training, part, nullptr, dZ, lddz, dgamma, dbeta, stream);
}

// aux: the forward's per-plot sums (agb_pointnet_pool_fwd_aux) or NULL; with them (and for max pooling anyway) the
// parameter gradients need no pass over Z and `part` may be NULL.
int agb_pointnet_pool_bwd_aux(const float* Z, int ldz, int n, int C, const int32_t* coords, const int32_t* ptr, int B,
                              const float* dpooled, const int32_t* argmax, int mode, const float* mean, const float* rstd,
                              const float* gamma, const float* beta, int act, int training, float* part, const float* aux,
                              float* dZ, int lddz, float* dgamma, float* dbeta, void* stream) {
    AGB_CHECK_ARG(C % 4 == 0 && C >= 4 && ldz % 4 == 0 && lddz % 4 == 0, "agb_pointnet_pool_bwd: C/ld multiples of 4");
    AGB_CHECK_ARG(mode >= 0 && mode <= 2 && (mode != 2 || argmax), "agb_pointnet_pool_bwd: mode %d", mode);
    hipStream_t s = (hipStream_t)stream;
    const int chunks = agb_bn_chunks(n);
    const int rpc = agb_cdiv(n > 0 ? n : 1, chunks);
    if (mode == 2) {
        hipLaunchKernelGGL(k_pn_bwd_sums_max, dim3(agb_cdiv(C, 256)), dim3(256), 0, s, Z, ldz, C, B, dpooled, argmax, mean,
                           rstd, gamma, beta, act, dbeta, dgamma);
    } else if (aux != nullptr) {
        hipLaunchKernelGGL(k_pn_bwd_sums_aux, dim3(agb_cdiv(C, 256)), dim3(256), 0, s, dpooled, aux, ptr, B, C, mode, dbeta,
                           dgamma);
    } else {
        AGB_CHECK_ARG(part != nullptr, "agb_pointnet_pool_bwd: sum / avg pooling without aux needs the scratch `part`");
        hipLaunchKernelGGL(k_pn_bwd_partial, dim3(chunks, agb_cdiv(C, 64)), dim3(256), 0, s, Z, ldz, n, C, rpc,
                           (const int4*)coords, ptr, dpooled, argmax, mode, mean, rstd, gamma, beta, act, part);
        hipLaunchKernelGGL(k_pn_bwd_fold, dim3(agb_cdiv(C, 256)), dim3(256), 0, s, part, chunks, C, dbeta, dgamma);
    }
    if (n > 0 && dZ)
        hipLaunchKernelGGL(k_pn_bwd_apply, dim3(agb_cdiv(n, 128), agb_cdiv(C, 64)), dim3(256), 0, s, Z, ldz, n, C,
                           (const int4*)coords, ptr, dpooled, argmax, mode, mean, rstd, gamma, beta, act, dbeta, dgamma,
                           training, dZ, lddz);
    AGB_CHECK_LAUNCH("agb_pointnet_pool_bwd");
    return AGB_OK;
}

// ------------------------------------------------------------------------------------------------
// The shared MLP of MinkowskiPointNet as ONE call (inference / C callers; training composes the pieces under autograd):
//   x [n, cin_pad] -> Linear(no bias) -> BatchNorm -> act  (x3: widths c1, c2, c3)  -> per-plot pooling -> pooled [B, c3]
// Weights W_l are [c_{l-1}, c_l] row-major (input-major, i.e. nn.Linear.weight transposed) with c_0 = cin_pad >= 12, all
// widths multiples of 4.  bn_l = {gamma, beta, running_mean, running_var} (float[c_l] each; gamma / beta may be NULL).
// training != 0: batch statistics (running statistics updated with `momentum` when given); else running statistics.
// workspace: agb_pointnet_mlp_workspace_bytes(n, B, c1, c2, c3) bytes, caller-owned.
size_t agb_pointnet_mlp_workspace_bytes(int n, int B, int c1, int c2, int c3) {
    const size_t rows = (size_t)(n > 0 ? n : 1);
    const size_t cmax = (size_t)(c3 > c2 ? (c3 > c1 ? c3 : c1) : (c2 > c1 ? c2 : c1));
    const size_t sp = (size_t)agb_pointnet_pool_splits(n, B);
    return sizeof(float) * (rows * ((size_t)c1 + c2 + c3)                       // pre-activations z1, z2, z3
                            + rows * ((size_t)c1 + c2)                          // activations a1, a2
                            + (size_t)agb_bn_chunks(n) * 3 * cmax + 2 * cmax    // statistics partials, mean, rstd
                            + (size_t)B * sp * c3) + sizeof(int32_t) * (size_t)B * sp * c3 + 256;
}

int agb_pointnet_mlp_fwd(const float* x, int ldx, int n, int cin_pad, const float* W1, const float* const* bn1, int c1,
                         const float* W2, const float* const* bn2, int c2, const float* W3, const float* const* bn3,
                         int c3, int act, float eps, float momentum, int training, const int32_t* ptr, int B, int mode,
                         void* workspace, float* pooled, int32_t* argmax, void* stream) {
    AGB_CHECK_ARG(cin_pad >= 12 && cin_pad % 4 == 0 && c1 % 4 == 0 && c2 % 4 == 0 && c3 % 4 == 0 && c1 >= 12 && c2 >= 12,
                  "agb_pointnet_mlp_fwd: widths %d -> %d -> %d -> %d must be multiples of 4 (>= 12 on the input side)",
                  cin_pad, c1, c2, c3);
    AGB_CHECK_ARG(workspace != nullptr && bn1 && bn2 && bn3, "agb_pointnet_mlp_fwd: workspace and bn tables are required");
    if (n == 0 || B == 0) return AGB_OK;
    const size_t rows = (size_t)n;
    float* z1 = (float*)workspace;
    float* z2 = z1 + rows * c1;
    float* z3 = z2 + rows * c2;
    float* a1 = z3 + rows * c3;
    float* a2 = a1 + rows * c1;
    const int cmax = c3 > c2 ? (c3 > c1 ? c3 : c1) : (c2 > c1 ? c2 : c1);
    float* part = a2 + rows * c2;
    float* mean = part + (size_t)agb_bn_chunks(n) * 3 * cmax;
    float* rstd = mean + cmax;
    const int sp = agb_pointnet_pool_splits(n, B);
    float* ppart = rstd + cmax;
    int32_t* parg = (int32_t*)(ppart + (size_t)B * sp * c3);
    const float* in = x;
    int ld = ldx, cprev = cin_pad;
    const float* Ws[3] = {W1, W2, W3};
    const float* const* bns[3] = {bn1, bn2, bn3};
    const int cs[3] = {c1, c2, c3};
    float* zs[3] = {z1, z2, z3};
    float* as[2] = {a1, a2};
    for (int l = 0; l < 3; ++l) {
        int rc = agb_spconv_fwd_ex(in, ld, Ws[l], nullptr, 0, 0, nullptr, zs[l], cs[l], n, 1, cprev, cs[l], nullptr,
                                   nullptr, nullptr, 0, 1, nullptr, stream);
        if (rc) return rc;
        rc = agb_bn_stats(zs[l], cs[l], n, cs[l], eps, momentum, training, part, mean, rstd, (float*)bns[l][2],
                          (float*)bns[l][3], stream);
        if (rc) return rc;
        if (l < 2) {
            rc = agb_bn_act_fwd(zs[l], cs[l], n, cs[l], mean, rstd, bns[l][0], bns[l][1], act, as[l], cs[l], stream);
            if (rc) return rc;
            in = as[l]; ld = cs[l]; cprev = cs[l];
        }
    }
    return agb_pointnet_pool_fwd(z3, c3, n, c3, mean, rstd, bn3[0], bn3[1], act, ptr, B, mode, sp, ppart, parg, pooled,
                                 argmax, stream);
}

}  // extern "C"
