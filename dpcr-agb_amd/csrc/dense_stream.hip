// dense_stream.hip — dense products with MANY rows and a SMALL weight matrix: Y[n, Cout] = X[n, Cin] @ W[Cin, Cout] and
// dW[Cin, Cout] = X^T dY, where Cin * Cout <= 16 K elements and n is in the hundreds of thousands.
//
// Where they come from: KPConv's first two levels (kernel-point contraction [N, 15*16] @ [240, 16], [N, 480] @ [480, 32],
// their data gradients [N, 16] @ [16, 240], the unary blocks 32 <-> 128, 16 <-> 64: blocks.py:396-400, 499-535) and the
// narrow front of the shared point MLP (PointNet.py:16-28).  At 8 FLOP per byte and less they are HBM-bound (the fp32
// ridge is 19.7 FLOP/B); the register-accumulator convolution kernels of spconv.hip take them with the identity map but
// stage every X element through LDS in 128-row tiles with two 8 KB chunks in flight per workgroup: 2.6-3.3 TB/s.
//
// Here X never touches LDS.  A wave owns 16 rows; lane (r, q) loads the float4 X[r][16 j + 4 q .. + 3] of chunk j — four
// consecutive reduction indices, which is all v_mfma_f32_16x16x4_f32 asks of an A operand once the weight rows are read
// in the same permuted order — and keeps several chunks in flight.  W (<= 64 KB + padding) sits in LDS for the whole
// workgroup, rows padded by 4 floats so the four k of a step fall on different banks.  fp32 exact products, like every
// fp32 kernel of this library.
#include "agb_common.h"
#include <stdio.h>

typedef float ds_f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------ forward
// NTB: output tiles of 16 columns per pass (all of them for Cout <= 64; wide outputs take passes of 8 tiles and re-read
// their 16 x Cin block of X from L1 — those shapes have Cin <= 64).
template <int NTB, bool NT_LOADS>
__global__ __launch_bounds__(512) void k_dense_stream(const float* __restrict__ X, int ldx, const float* __restrict__ W,
                                                      const float* __restrict__ bias, float* __restrict__ Y, int ldy,
                                                      int n, int Cin, int Cout, int items) {
    extern __shared__ __align__(16) float ds_w[];          // [Cin][ldw], columns >= Cout zero
    const int NT = (Cout + 15) >> 4, ldw = NT * 16 + 4;
    for (int e = threadIdx.x; e < Cin * ldw; e += 512) {
        const int k = e / ldw, c = e - k * ldw;
        ds_w[e] = c < Cout ? W[(long long)k * Cout + c] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, g = lane >> 4;                // A: (row m, k g)   B: (k g, column m)   D: (row 4 g + i, column m)
    const int chunks = Cin >> 4;
    for (int it = blockIdx.x * 8 + wave; it < items; it += gridDim.x * 8) {
        const long long n0 = (long long)it * 16;
        const long long ra = n0 + m < n ? n0 + m : n - 1;  // clamped: rows past the end are loaded twice, never stored
        const float* xr = X + ra * ldx + 4 * g;
        for (int nt0 = 0; nt0 < NT; nt0 += NTB) {
            ds_f32x4 acc[NTB];
#pragma unroll
            for (int q = 0; q < NTB; ++q) acc[q] = ds_f32x4{0.f, 0.f, 0.f, 0.f};
            auto mm = [&](int j, const float4& a4) {
                const float a[4] = {a4.x, a4.y, a4.z, a4.w};
                const float* wb = ds_w + (16 * j + 4 * g) * ldw + 16 * nt0 + m;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int q = 0; q < NTB; ++q)
                        if (nt0 + q < NT)
                            acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], wb[t * ldw + 16 * q], acc[q], 0, 0, 0);
            };
            int j = 0;
            for (; j + 4 <= chunks; j += 4) {              // four chunks (a 256-byte run of the row) requested before use
                float4 a4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (NT_LOADS) {     // rows of 1.5 KB and more: the lines are used once, keep them out of L1
                        const ds_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const ds_f32x4*>(xr + 16 * (j + u)));
                        a4[u] = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
                        a4[u] = *reinterpret_cast<const float4*>(xr + 16 * (j + u));
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) mm(j + u, a4[u]);
            }
            for (; j < chunks; ++j) mm(j, *reinterpret_cast<const float4*>(xr + 16 * j));
#pragma unroll
            for (int q = 0; q < NTB; ++q) {
                const int col = 16 * (nt0 + q) + m;
                if (nt0 + q < NT && col < Cout) {
                    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const long long r = n0 + 4 * g + i;
                        if (r < n) Y[r * ldy + col] = acc[q][i] + bv;
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ weight gradient
// dW[c][o] += sum_rows X[r][c] dY[r][o]: the reduction runs over ROWS, so a lane may load a float4 of its row — lane
// (cg, r) reads X[n0 + 4 s + r][64 cb + 4 cg .. + 3]: four rows x 256 contiguous bytes per load instruction, whole cache
// lines — and use component t as the A operand of tile t, whose 16 "matrix rows" are then the channels 64 cb + 4 cg + t
// (a stride-4 set: undone when the tile is written).  B: dY[n0 + 4 s + r][16 nt + m] (dY is the narrow side).
// The 64-channel column blocks cb are dealt round-robin to the four wave PAIRS of a workgroup (<= DSW_CB blocks x 4
// tiles x NT accumulators each); the two waves of a pair take alternate 16-row blocks and meet in LDS before the one
// atomic add per element and workgroup.
#define DSW_CB 2          // column blocks per wave pair: Cin <= 512
#define DSW_NT 2          // output tiles: Cout <= 32
__global__ __launch_bounds__(512) void k_dense_stream_wgrad(const float* __restrict__ X, int ldx,
                                                            const float* __restrict__ dY, int ldy, float* __restrict__ dW,
                                                            int sa, int sb,      // element (c, o) of the result: dW[c sa + o sb]
                                                            int n, int Cin, int Cout, int blocks16) {
    __shared__ float red[4][DSW_CB * 4 * DSW_NT][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, pair = wave & 3, half = wave >> 2;
    const int m = lane & 15, g = lane >> 4;
    const int NT = (Cout + 15) >> 4;
    ds_f32x4 acc[DSW_CB][4][DSW_NT];
#pragma unroll
    for (int u = 0; u < DSW_CB; ++u)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int q = 0; q < DSW_NT; ++q) acc[u][t][q] = ds_f32x4{0.f, 0.f, 0.f, 0.f};
    for (int rb = blockIdx.x * 2 + half; rb < blocks16; rb += gridDim.x * 2) {
        const long long n0 = (long long)rb * 16;
        float4 a4[4][DSW_CB];
        float bq[4][DSW_NT];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const long long r = n0 + 4 * s + g;
            const bool ok = r < n;
            const float* xr = X + (ok ? r : 0) * ldx + 4 * m;
            const float* dr = dY + (ok ? r : 0) * ldy + m;
#pragma unroll
            for (int u = 0; u < DSW_CB; ++u) {
                const int c = 64 * (pair + 4 * u) + 4 * m;
                a4[s][u] = (ok && c < Cin) ? *reinterpret_cast<const float4*>(xr + 64 * (pair + 4 * u))
                                           : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int q = 0; q < DSW_NT; ++q) bq[s][q] = (ok && q < NT && 16 * q + m < Cout) ? dr[16 * q] : 0.f;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int u = 0; u < DSW_CB; ++u) {
                if (64 * (pair + 4 * u) >= Cin) continue;
                const float a[4] = {a4[s][u].x, a4[s][u].y, a4[s][u].z, a4[s][u].w};
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int q = 0; q < DSW_NT; ++q)
                        if (q < NT) acc[u][t][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], bq[s][q], acc[u][t][q], 0, 0, 0);
            }
    }
    if (half == 1) {
#pragma unroll
        for (int u = 0; u < DSW_CB; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int q = 0; q < DSW_NT; ++q)
#pragma unroll
                    for (int i = 0; i < 4; ++i) red[pair][(u * 4 + t) * DSW_NT + q][i * 64 + lane] = acc[u][t][q][i];
    }
    __syncthreads();
    if (half == 0) {
#pragma unroll
        for (int u = 0; u < DSW_CB; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int q = 0; q < DSW_NT; ++q) {
                    const int col = 16 * q + m;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int c = 64 * (pair + 4 * u) + 4 * (4 * g + i) + t;     // matrix row 4 g + i of tile t
                        if (q < NT && col < Cout && c < Cin)
                            atomicAdd(&dW[(long long)c * sa + (long long)col * sb],
                                      acc[u][t][q][i] + red[pair][(u * 4 + t) * DSW_NT + q][i * 64 + lane]);
                    }
                }
    }
}

// ------------------------------------------------------------------------------------------------ host side
// Shapes these kernels take from the identity-map entry points of spconv.hip (agb_spconv_fwd_opt / _bwd_weight_lp with
// nbr == NULL, fp32 operands): many rows, Cin a multiple of 16, a weight matrix that fits LDS, HBM-bound intensity.
bool agb_dense_stream_ok(int n, int Cin, int Cout) {
    if (n < 16384 || Cin < 16 || Cin % 16 != 0 || Cout < 4 || Cout % 4 != 0) return false;
    const int ldw = ((Cout + 15) / 16) * 16 + 4;
    if ((long long)Cin * ldw * 4 > 96 * 1024) return false;
    // Measured on MI355X against the tiled kernels (GB/s of compulsory traffic): 240 -> 16: 2896 -> 3691, 64 -> 16:
    // 2616 -> 3625, 16 -> 64: 3288 -> 3638, 128 -> 32: 2798 -> 3130, 480 -> 32: 3256 -> 3951; but 32 -> 128: 2662 -> 2002
    // and 32 -> 64: 3382 -> 2751 (wide outputs re-read X per pass of 8 tiles): narrow outputs and 16-channel inputs only.
    if (!(Cout <= 32 || Cin == 16)) return false;
    // FLOP per compulsory byte 2 Cin Cout / (4 (Cin + Cout)) below the fp32 ridge (19.7)
    return (double)Cin * Cout / (2.0 * (Cin + Cout)) < 17.0;
}

bool agb_dense_stream_wgrad_ok(int n, int Cin, int Cout) {
    // X streams as the A side in 64-channel column blocks, one or two per wave pair: at least three of the four pairs
    // busy (Cin >= 192); dY is the narrow B side.  Measured on MI355X against the tiled weight-gradient kernel:
    // [215 k, 480]^T [.., 32] 0.165 -> 0.110 ms, [359 k, 240]^T [.., 16] 0.143 -> 0.084 ms; narrower X (128, 64 channels)
    // and the transposed orientation for wide-output layers were slower and stay with the tiled kernel.
    return n >= 16384 && Cin % 4 == 0 && Cout % 4 == 0 && Cin >= 192 && Cin <= 256 * DSW_CB && Cout <= 16 * DSW_NT;
}

template <int NTB, bool NT_LOADS>
static int launch_stream(const float* X, int ldx, const float* W, const float* bias, float* Y, int ldy, int n, int Cin,
                         int Cout, hipStream_t st) {
    const int ldw = ((Cout + 15) / 16) * 16 + 4, lds = Cin * ldw * 4;
    auto kern = k_dense_stream<NTB, NT_LOADS>;
    if (lds > 48 * 1024) {      // (per device and cheap: set on every such launch rather than cached in a static)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           96 * 1024);
        if (e != hipSuccess) {
            agb_set_error("dense product: %d bytes of LDS refused: %s", lds, hipGetErrorString(e));
            return AGB_ELAUNCH;
        }
    }
    const int items = agb_cdiv(n, 16);
    const int per_cu = lds > 80 * 1024 ? 1 : lds > 40 * 1024 ? 2 : 4;   // 8-wave workgroups
    int grid = 256 * per_cu;
    if (grid > agb_cdiv(items, 8)) grid = agb_cdiv(items, 8);
    char nm[64];
    snprintf(nm, sizeof nm, "k_dense_stream<%d, %s>", NTB, NT_LOADS ? "true" : "false");
    agb_note_kernel(nm);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, X, ldx, W, bias, Y, ldy, n, Cin, Cout, items);
    return AGB_OK;
}

int agb_dense_stream_launch(const float* X, int ldx, const float* W, const float* bias, float* Y, int ldy, int n, int Cin,
                            int Cout, hipStream_t st) {
    const int NT = (Cout + 15) / 16;
    if (Cin >= 384) {
        if (NT == 1) return launch_stream<1, true>(X, ldx, W, bias, Y, ldy, n, Cin, Cout, st);
        return launch_stream<2, true>(X, ldx, W, bias, Y, ldy, n, Cin, Cout, st);
    }
    if (NT == 1) return launch_stream<1, false>(X, ldx, W, bias, Y, ldy, n, Cin, Cout, st);
    if (NT == 2) return launch_stream<2, false>(X, ldx, W, bias, Y, ldy, n, Cin, Cout, st);
    if (NT <= 4) return launch_stream<4, false>(X, ldx, W, bias, Y, ldy, n, Cin, Cout, st);
    return launch_stream<8, false>(X, ldx, W, bias, Y, ldy, n, Cin, Cout, st);
}

// dW accumulated (the caller's buffer is zero-filled, as for the tiled weight-gradient kernels)
int agb_dense_stream_wgrad_launch(const float* X, int ldx, const float* dY, int ldy, float* dW, int n, int Cin, int Cout,
                                  hipStream_t st) {
    const int blocks16 = agb_cdiv(n, 16);
    int grid = 512;
    if (grid > agb_cdiv(blocks16, 2)) grid = agb_cdiv(blocks16, 2);
    AGB_LAUNCH(k_dense_stream_wgrad, dim3(grid), dim3(512), 0, st, X, ldx, dY, ldy, dW, Cout, 1, n, Cin, Cout,
                       blocks16);
    return AGB_OK;
}
