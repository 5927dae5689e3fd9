// pool.hip — HBM-bound pooling / broadcast kernels of the sparse-voxel path.
//   * max pooling over a kernel map (ME.MinkowskiMaxPooling; reference call site
//     torch_points3d/modules/MinkowskiEngine/SENet.py:53)
//   * per-batch segment reductions = global sum / avg / max pooling (ME.MinkowskiGlobal*Pooling;
//     SENet.py:63, senet_block.py:43, PointNet.py:29) and their gradients
//   * broadcast multiply x.F * y.F[batch] (ME.MinkowskiBroadcastMultiplication; senet_block.py:44-50)
// Rows of a batch element are contiguous (coords.hip keeps that invariant), so reductions are
// deterministic segment sums — no float atomics.
#include "agb_common.h"
#include <float.h>

// ------------------------------------------------------------------ max pool
// one thread per (out row, 4 channels); lanes of a row read one contiguous feature row.  The offsets are taken in batches
// of PB: PB independent index loads, then PB independent row gathers, then the compares (in offset order: ties keep the
// lowest offset, like the serial loop) — the serial index -> gather -> compare chain per offset ran at 1.3 TB/s.
#define PB 9
template <typename T>
__global__ __launch_bounds__(256) void k_maxpool_fwd(const T* __restrict__ X, int ldx, const int32_t* __restrict__ nbr,
                                                     long long nbr_stride, T* __restrict__ Y, int ldy,
                                                     int32_t* __restrict__ arg, int n_out, int K3, int C4) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int r = (int)(t / C4);
    int c = (int)(t % C4) * 4;
    if (r >= n_out) return;
    float4 best = make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX);
    int4 bi = make_int4(-1, -1, -1, -1);
    for (int k0 = 0; k0 < K3; k0 += PB) {
        int idx[PB];
        float4 v[PB];
#pragma unroll
        for (int u = 0; u < PB; ++u) idx[u] = (k0 + u < K3) ? nbr[(long long)(k0 + u) * nbr_stride + r] : -1;
#pragma unroll
        for (int u = 0; u < PB; ++u)
            if (idx[u] >= 0) v[u] = ld4(X + (long long)idx[u] * ldx + c);
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            if (idx[u] < 0) continue;
            if (v[u].x > best.x) { best.x = v[u].x; bi.x = idx[u]; }
            if (v[u].y > best.y) { best.y = v[u].y; bi.y = idx[u]; }
            if (v[u].z > best.z) { best.z = v[u].z; bi.z = idx[u]; }
            if (v[u].w > best.w) { best.w = v[u].w; bi.w = idx[u]; }
        }
    }
    if (bi.x < 0) best.x = 0.f;
    if (bi.y < 0) best.y = 0.f;
    if (bi.z < 0) best.z = 0.f;
    if (bi.w < 0) best.w = 0.f;
    st4(Y + (long long)r * ldy + c, best);
    *reinterpret_cast<int4*>(arg + (long long)r * (C4 * 4) + c) = bi;
}

// input-stationary gradient: each input row collects from the (at most K3) outputs that could have chosen it; same
// batching (index loads, then the argmax / gradient rows of the present outputs, then the adds in offset order)
template <typename T>
__global__ __launch_bounds__(256) void k_maxpool_bwd(const T* __restrict__ dY, int ldy, const int32_t* __restrict__ arg,
                                                     const int32_t* __restrict__ nbrT, long long nbrT_stride,
                                                     T* __restrict__ dX, int ldx, int n_in, int K3, int C4) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int q = (int)(t / C4);
    int c = (int)(t % C4) * 4;
    if (q >= n_in) return;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k0 = 0; k0 < K3; k0 += PB) {
        int o[PB];
        int4 a[PB];
        float4 d[PB];
#pragma unroll
        for (int u = 0; u < PB; ++u) o[u] = (k0 + u < K3) ? nbrT[(long long)(k0 + u) * nbrT_stride + q] : -1;
#pragma unroll
        for (int u = 0; u < PB; ++u)
            if (o[u] >= 0) {
                a[u] = *reinterpret_cast<const int4*>(arg + (long long)o[u] * (C4 * 4) + c);
                d[u] = ld4(dY + (long long)o[u] * ldy + c);
            }
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            if (o[u] < 0) continue;
            if (a[u].x == q) g.x += d[u].x;
            if (a[u].y == q) g.y += d[u].y;
            if (a[u].z == q) g.z += d[u].z;
            if (a[u].w == q) g.w += d[u].w;
        }
    }
    st4(dX + (long long)q * ldx + c, g);
}

// The same pair with the winner kept as its OFFSET index (one byte per output element instead of the 4-byte input row):
// nbrT[k][q] = o  <=>  nbr[k][o] = q, so input row q won channel c of output o iff arg8[o][c] == k.  The gradient pass
// gathers 4 + 16 bytes per (pair, 4 channels) instead of 16 + 16 (it is bound by those gathers: every output row is read
// by ~7 input rows), and the forward pass writes a quarter of the argmax bytes.  K3 <= 255.
template <typename T>
__global__ __launch_bounds__(256) void k_maxpool_fwd8(const T* __restrict__ X, int ldx, const int32_t* __restrict__ nbr,
                                                      long long nbr_stride, T* __restrict__ Y, int ldy,
                                                      uint8_t* __restrict__ arg, int n_out, int K3, int C4) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int r = (int)(t / C4);
    int c = (int)(t % C4) * 4;
    if (r >= n_out) return;
    float4 best = make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX);
    int bx = 255, by = 255, bz = 255, bw = 255;   // 255 = no neighbour at all
    for (int k0 = 0; k0 < K3; k0 += PB) {
        int idx[PB];
        float4 v[PB];
#pragma unroll
        for (int u = 0; u < PB; ++u) idx[u] = (k0 + u < K3) ? nbr[(long long)(k0 + u) * nbr_stride + r] : -1;
#pragma unroll
        for (int u = 0; u < PB; ++u)
            if (idx[u] >= 0) v[u] = ld4(X + (long long)idx[u] * ldx + c);
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            if (idx[u] < 0) continue;
            if (v[u].x > best.x) { best.x = v[u].x; bx = k0 + u; }
            if (v[u].y > best.y) { best.y = v[u].y; by = k0 + u; }
            if (v[u].z > best.z) { best.z = v[u].z; bz = k0 + u; }
            if (v[u].w > best.w) { best.w = v[u].w; bw = k0 + u; }
        }
    }
    if (bx == 255) best.x = 0.f;
    if (by == 255) best.y = 0.f;
    if (bz == 255) best.z = 0.f;
    if (bw == 255) best.w = 0.f;
    st4(Y + (long long)r * ldy + c, best);
    *reinterpret_cast<uchar4*>(arg + (long long)r * (C4 * 4) + c) = make_uchar4(bx, by, bz, bw);
}

template <typename T>
__global__ __launch_bounds__(256) void k_maxpool_bwd8(const T* __restrict__ dY, int ldy, const uint8_t* __restrict__ arg,
                                                      const int32_t* __restrict__ nbrT, long long nbrT_stride,
                                                      T* __restrict__ dX, int ldx, int n_in, int K3, int C4) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int q = (int)(t / C4);
    int c = (int)(t % C4) * 4;
    if (q >= n_in) return;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k0 = 0; k0 < K3; k0 += PB) {
        int o[PB];
        uchar4 a[PB];
        float4 d[PB];
#pragma unroll
        for (int u = 0; u < PB; ++u) o[u] = (k0 + u < K3) ? nbrT[(long long)(k0 + u) * nbrT_stride + q] : -1;
#pragma unroll
        for (int u = 0; u < PB; ++u)
            if (o[u] >= 0) {
                a[u] = *reinterpret_cast<const uchar4*>(arg + (long long)o[u] * (C4 * 4) + c);
                d[u] = ld4(dY + (long long)o[u] * ldy + c);
            }
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            if (o[u] < 0) continue;
            const int k = k0 + u;
            if (a[u].x == k) g.x += d[u].x;
            if (a[u].y == k) g.y += d[u].y;
            if (a[u].z == k) g.z += d[u].z;
            if (a[u].w == k) g.w += d[u].w;
        }
    }
    st4(dX + (long long)q * ldx + c, g);
}

// ------------------------------------------------------------ segment reduce
// grid (B, ceil(C/64), S), block 256 = 4 row lanes x 64 channels. Segment b is cut into S contiguous row
// chunks so that long segments (6.8k rows/plot at 64 channels) still fill the chip; chunk partials go to
// `part` [B*S, C] (and `part_arg`) and a second tiny kernel folds them in a fixed order (deterministic).
// mode 0 sum, 1 average, 2 max (arg receives the winning row); optional second operand: reduce A*Bm.
template <typename T>
__global__ __launch_bounds__(256) void k_segment_reduce(const T* __restrict__ A, int lda,
                                                        const T* __restrict__ Bm, int ldb,
                                                        const int32_t* __restrict__ ptr, int C, int mode, int S,
                                                        float* __restrict__ Y, int32_t* __restrict__ arg) {
    __shared__ float s_val[4][64];
    __shared__ int s_arg[4][64];
    const int b = blockIdx.x;
    const int c = blockIdx.y * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const int seg_beg = ptr[b], seg_end = ptr[b + 1];
    const int len = seg_end - seg_beg;
    const int chunk = (len + S - 1) / S;
    const int beg = seg_beg + blockIdx.z * chunk;
    const int end = min(seg_end, beg + chunk);
    float acc = (mode == 2) ? -FLT_MAX : 0.f;
    int ai = -1;
    if (c < C) {
        for (int r = beg + rl; r < end; r += 4) {
            float v = ld1(A + (long long)r * lda + c);
            if (Bm) v *= ld1(Bm + (long long)r * ldb + c);
            if (mode == 2) {
                if (v > acc) { acc = v; ai = r; }
            } else {
                acc += v;
            }
        }
    }
    s_val[rl][threadIdx.x & 63] = acc;
    s_arg[rl][threadIdx.x & 63] = ai;
    __syncthreads();
    if (rl == 0 && c < C) {
        int l = threadIdx.x & 63;
        if (mode == 2) {
            for (int j = 1; j < 4; ++j) {
                float v = s_val[j][l];
                int a = s_arg[j][l];
                if (a >= 0 && (v > acc || (v == acc && a < ai))) { acc = v; ai = a; }
            }
            if (S == 1 && ai < 0) acc = 0.f;
            arg[((long long)b * S + blockIdx.z) * C + c] = ai;
        } else {
            acc = ((s_val[0][l] + s_val[1][l]) + (s_val[2][l] + s_val[3][l]));
            if (mode == 1 && S == 1) acc = len > 0 ? acc / (float)len : 0.f;
        }
        Y[((long long)b * S + blockIdx.z) * C + c] = acc;
    }
}

// Same reduction with 16-B loads: block 256 = 16 column groups (one 64-channel slab) x 16 row lanes; every thread
// folds rows r, r+16, ... of its 4 channels, the 16 lane partials are combined in lane order (fixed order).
// Used when C, lda (and ldb) are multiples of 4: 4x fewer load instructions than the scalar kernel above
// (the SE squeeze of 206 k x 64 rows ran at 0.9 TB/s with 4-byte loads).
template <typename T>
__global__ __launch_bounds__(256) void k_segment_reduce4(const T* __restrict__ A, int lda,
                                                         const T* __restrict__ Bm, int ldb,
                                                         const int32_t* __restrict__ ptr, int C, int mode, int S,
                                                         float* __restrict__ Y, int32_t* __restrict__ arg) {
    __shared__ float s_val[16][64];
    __shared__ int s_arg[16][64];
    const int b = blockIdx.x;
    const int cg = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.y * 64 + cg * 4;
    const int seg_beg = ptr[b], seg_end = ptr[b + 1];
    const int len = seg_end - seg_beg;
    const int chunk = (len + S - 1) / S;
    const int beg = seg_beg + blockIdx.z * chunk;
    const int end = min(seg_end, beg + chunk);
    float acc[4];
    int ai[4] = {-1, -1, -1, -1};
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (mode == 2) ? -FLT_MAX : 0.f;
    if (c < C) {
        for (int r = beg + rl; r < end; r += 16) {
            float4 v4 = ld4(A + (long long)r * lda + c);
            if (Bm) {
                float4 m4 = ld4(Bm + (long long)r * ldb + c);
                v4.x *= m4.x; v4.y *= m4.y; v4.z *= m4.z; v4.w *= m4.w;
            }
            const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (mode == 2) {
                    if (v[j] > acc[j]) { acc[j] = v[j]; ai[j] = r; }
                } else {
                    acc[j] += v[j];
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s_val[rl][cg * 4 + j] = acc[j];
        s_arg[rl][cg * 4 + j] = ai[j];
    }
    __syncthreads();
    const int l = threadIdx.x;
    const int cc = blockIdx.y * 64 + l;
    if (l < 64 && cc < C) {
        float a = s_val[0][l];
        int bi = s_arg[0][l];
        if (mode == 2) {
            for (int j = 1; j < 16; ++j) {
                float v = s_val[j][l];
                int q = s_arg[j][l];
                if (q >= 0 && (bi < 0 || v > a || (v == a && q < bi))) { a = v; bi = q; }
            }
            if (S == 1 && bi < 0) a = 0.f;
            arg[((long long)b * S + blockIdx.z) * C + cc] = bi;
        } else {
            for (int j = 1; j < 16; ++j) a += s_val[j][l];
            if (mode == 1 && S == 1) a = len > 0 ? a / (float)len : 0.f;
        }
        Y[((long long)b * S + blockIdx.z) * C + cc] = a;
    }
}

__global__ void k_segment_fold(const float* __restrict__ part, const int32_t* __restrict__ part_arg,
                               const int32_t* __restrict__ ptr, int B, int C, int mode, int S,
                               float* __restrict__ Y, int32_t* __restrict__ arg) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * C) return;
    int b = t / C, c = t % C;
    if (mode == 2) {
        float best = -FLT_MAX;
        int bi = -1;
        for (int s = 0; s < S; ++s) {
            int a = part_arg[((long long)b * S + s) * C + c];
            float v = part[((long long)b * S + s) * C + c];
            if (a >= 0 && (v > best || (v == best && a < bi) || bi < 0)) { best = v; bi = a; }
        }
        Y[t] = bi >= 0 ? best : 0.f;
        arg[t] = bi;
    } else {
        float acc = 0.f;
        for (int s = 0; s < S; ++s) acc += part[((long long)b * S + s) * C + c];
        if (mode == 1) {
            int len = ptr[b + 1] - ptr[b];
            acc = len > 0 ? acc / (float)len : 0.f;
        }
        Y[t] = acc;
    }
}

// gradient of sum/avg pooling, and the forward of a broadcast: out[r,c] = S[batch(r),c] * scale(b) [* M[r,c]]
template <typename T>
__global__ void k_segment_broadcast(const float* __restrict__ S, const int32_t* __restrict__ coords,
                                    const int32_t* __restrict__ ptr, const T* __restrict__ M, int ldm,
                                    T* __restrict__ out, int ldo, int n, int C4, int average) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int r = (int)(t / C4);
    int c = (int)(t % C4) * 4;
    if (r >= n) return;
    int b = coords[4 * (long long)r];
    float4 s = *reinterpret_cast<const float4*>(S + (long long)b * (C4 * 4) + c);
    if (average) {
        float inv = 1.f / (float)(ptr[b + 1] - ptr[b]);
        s.x *= inv; s.y *= inv; s.z *= inv; s.w *= inv;
    }
    if (M) {
        float4 m = ld4(M + (long long)r * ldm + c);
        s.x *= m.x; s.y *= m.y; s.z *= m.z; s.w *= m.w;
    }
    st4(out + (long long)r * ldo + c, s);
}

// out[r,c] = M[r,c] * S[batch(r),c] + T[batch(r),c] / rows(batch(r)): the input gradient of the squeeze-excite layer
// (product-rule term of the broadcast multiplication + the gradient that reaches the rows through the average pooling)
// in one pass, instead of two broadcast kernels and an addition
template <typename T>
__global__ void k_segment_scale_add(const float* __restrict__ S, const float* __restrict__ Tb,
                                    const int32_t* __restrict__ coords, const int32_t* __restrict__ ptr,
                                    const T* __restrict__ M, int ldm, T* __restrict__ out, int ldo, int n,
                                    int C4) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int r = (int)(t / C4);
    int c = (int)(t % C4) * 4;
    if (r >= n) return;
    int b = coords[4 * (long long)r];
    const float4 s = *reinterpret_cast<const float4*>(S + (long long)b * (C4 * 4) + c);
    const float4 g = *reinterpret_cast<const float4*>(Tb + (long long)b * (C4 * 4) + c);
    const float inv = 1.f / (float)(ptr[b + 1] - ptr[b]);
    const float4 m = ld4(M + (long long)r * ldm + c);
    float4 o;
    // same operation order as the unfused path: (m * s) + (g * inv)
    o.x = m.x * s.x + g.x * inv; o.y = m.y * s.y + g.y * inv;
    o.z = m.z * s.z + g.z * inv; o.w = m.w * s.w + g.w * inv;
    st4(out + (long long)r * ldo + c, o);
}

// gradient of global max pooling: dX = 0 except dX[arg[b,c], c] = dY[b,c]
template <typename T>
__global__ void k_segment_max_bwd(const float* __restrict__ dY, const int32_t* __restrict__ arg, T* __restrict__ dX, int ldx,
                                  int B, int C) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * C) return;
    int a = arg[t];
    if (a >= 0) st1(dX + (long long)a * ldx + (t % C), dY[t]);
}

// =============================================================== C ABI
extern "C" {

#define AGB_T float
#define AGB_FN(name) name
#include "pool_rows.inc"
#undef AGB_T
#undef AGB_FN
#define AGB_T bf16_t
#define AGB_FN(name) name##_h
#include "pool_rows.inc"
#undef AGB_T
#undef AGB_FN

}  // extern "C"
