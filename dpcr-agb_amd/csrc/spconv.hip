// spconv.hip — generalized sparse 3-D convolution as an output-stationary implicit GEMM on
// fp32 MFMA (v_mfma_f32_32x32x2_f32: exact fp32 fma chain, 64 FLOP/clk/SIMD).
//
// Replaces MinkowskiEngine's per-offset gather-GEMM-scatter behind ME.MinkowskiConvolution
// (reference call sites: torch_points3d/modules/MinkowskiEngine/common.py:215-226,
// resnet_block.py:48-55,95-107, SENet.py:47-52,93-99).
//
//   forward      Y[r, :]  = bias + sum_k X[nbr[k][r], :] @ W[k]            (nbr = forward kernel map)
//   data grad    dX[q, :] =        sum_k dY[nbrT[k][q], :] @ W[k]^T        (same kernel, W pre-transposed,
//                                                                           nbrT = transposed map or k-flipped map)
//   weight grad  dW[k]    = sum_r X[nbr[k][r], :]^T @ dY[r, :]             (k_spconv_dw, split over row chunks)
//
// No scatter, no float atomics on activations: every workgroup owns a 64x64 output tile and walks the
// kernel offsets, gathering the 64 neighbour rows of each offset into LDS (coalesced 16-B row pieces).
// Results are run-to-run deterministic for fwd/data-grad.
#include "agb_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BM 64
#define BN 64
#define BK 32
#define LDA 36  // A tile row stride (floats): 16-B aligned, conflict-free for ds_read_b128 (36*i mod 64 distinct)
#define LDB 64

// ------------------------------------------------------------------------------------------------
// CPAD == 0 : generic path, Cin % 4 == 0, one K-chunk = 32 input channels of one kernel offset
// CPAD == 4/8: small-Cin path (stem), features/weights padded to CPAD channels; one K-chunk = 32/CPAD offsets
template <int CPAD>
__global__ __launch_bounds__(256) void k_spconv_fwd(const float* __restrict__ X, int ldx,
                                                    const float* __restrict__ W,  // [K3*Cin, Cout]
                                                    const int32_t* __restrict__ nbr, long long nbr_stride, int kflip,
                                                    const float* __restrict__ bias, float* __restrict__ Y, int ldy,
                                                    int n_out, int K3, int Cin, int Cout) {
    __shared__ __attribute__((aligned(16))) float As[BM * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[BK * LDB];
    __shared__ int s_idx[BM];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int row0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int li = lane & 31, lh = lane >> 5;

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    auto mma_chunk = [&]() {
        const float* arow = &As[(wr * 32 + li) * LDA + 4 * lh];
        const float* bcol = &Bs[wc * 32 + li];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float4 a4 = *reinterpret_cast<const float4*>(arow + 8 * t);
            const int kb = 8 * t + 4 * lh;
            float b0 = bcol[(kb + 0) * LDB];
            float b1 = bcol[(kb + 1) * LDB];
            float b2 = bcol[(kb + 2) * LDB];
            float b3 = bcol[(kb + 3) * LDB];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b3, acc, 0, 0, 0);
        }
    };

    if constexpr (CPAD == 0) {
        for (int k = 0; k < K3; ++k) {
            const int kn = kflip ? (K3 - 1 - k) : k;
            int my = -1;
            if (tid < BM) {
                int r = row0 + tid;
                my = (r < n_out) ? nbr[(long long)kn * nbr_stride + r] : -1;
                s_idx[tid] = my;
            }
            if (!__syncthreads_or(my >= 0)) continue;  // whole tile has no neighbour at this offset
            for (int c0 = 0; c0 < Cin; c0 += BK) {
                // gather A: 64 rows x 32 channels (8 float4 per row)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    int r = (tid >> 3) + 32 * j;
                    int c = c0 + (tid & 7) * 4;
                    int idx = s_idx[r];
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (idx >= 0 && c < Cin) v = *reinterpret_cast<const float4*>(X + (long long)idx * ldx + c);
                    *reinterpret_cast<float4*>(&As[r * LDA + (tid & 7) * 4]) = v;
                }
                // load B: 32 channels x 64 outputs
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    int kr = (tid >> 4) + 16 * j;
                    int n = n0 + (tid & 15) * 4;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (c0 + kr < Cin && n < Cout)
                        v = *reinterpret_cast<const float4*>(W + ((long long)k * Cin + c0 + kr) * Cout + n);
                    *reinterpret_cast<float4*>(&Bs[kr * LDB + (tid & 15) * 4]) = v;
                }
                __syncthreads();
                mma_chunk();
                __syncthreads();
            }
        }
    } else {
        constexpr int OPC = BK / CPAD;  // offsets per chunk
        const int nchunks = (K3 + OPC - 1) / OPC;
        constexpr int F4 = CPAD / 4;  // float4 per (row, offset)
        for (int ch = 0; ch < nchunks; ++ch) {
            const int k0 = ch * OPC;
            // gather A: 64 rows x OPC offsets x CPAD channels
#pragma unroll
            for (int j = 0; j < (BM * OPC * F4) / 256; ++j) {
                int e = tid + 256 * j;
                int r = e & (BM - 1);
                int rest = e >> 6;  // 0 .. OPC*F4-1
                int off = rest / F4, f = rest % F4;
                int k = k0 + off;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < K3 && row0 + r < n_out) {
                    int kn = kflip ? (K3 - 1 - k) : k;
                    int idx = nbr[(long long)kn * nbr_stride + row0 + r];
                    if (idx >= 0) v = *reinterpret_cast<const float4*>(X + (long long)idx * ldx + f * 4);
                }
                *reinterpret_cast<float4*>(&As[r * LDA + off * CPAD + f * 4]) = v;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                int kr = (tid >> 4) + 16 * j;
                int n = n0 + (tid & 15) * 4;
                long long wrow = (long long)k0 * CPAD + kr;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (wrow < (long long)K3 * CPAD && n < Cout) v = *reinterpret_cast<const float4*>(W + wrow * Cout + n);
                *reinterpret_cast<float4*>(&Bs[kr * LDB + (tid & 15) * 4]) = v;
            }
            __syncthreads();
            mma_chunk();
            __syncthreads();
        }
    }

    // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const int col = n0 + wc * 32 + li;
    if (col < Cout) {
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            int row = row0 + wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            if (row < n_out) Y[(long long)row * ldy + col] = acc[reg] + bv;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient. grid = (row_chunks, M_tiles, N_tiles); M = K3*Cin rows of the flattened weight.
//   generic: M tile = 64 input channels of one offset      (Cin % 64 handled by bounds)
//   CPAD   : M tile = 64/CPAD offsets x CPAD channels
// Partial tiles are summed into dW with fp32 atomics shaped as 2 x 128-B row segments per instruction.
template <int CPAD>
__global__ __launch_bounds__(256) void k_spconv_dw(const float* __restrict__ X, int ldx,
                                                   const float* __restrict__ dY, int ldy,
                                                   const int32_t* __restrict__ nbr, long long nbr_stride,
                                                   float* __restrict__ dW, int n_out, int K3, int Cin, int Cout,
                                                   int rows_per_wg, int cin_tiles) {
    __shared__ __attribute__((aligned(16))) float As[BK * 64];  // [r][m]
    __shared__ __attribute__((aligned(16))) float Bs[BK * 64];  // [r][n]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int n0 = blockIdx.z * 64;
    const int r_begin = blockIdx.x * rows_per_wg;
    const int r_end = min(n_out, r_begin + rows_per_wg);

    int k = 0, c0 = 0, k0 = 0;
    if constexpr (CPAD == 0) {
        k = blockIdx.y / cin_tiles;
        c0 = (blockIdx.y % cin_tiles) * 64;
    } else {
        k0 = blockIdx.y * (64 / CPAD);
    }

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    for (int rb = r_begin; rb < r_end; rb += BK) {
        float4 av[2];
        int any = 0;
        if constexpr (CPAD == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                int r = rb + (tid >> 4) + 16 * j;
                int c = c0 + (tid & 15) * 4;
                int idx = (r < r_end) ? nbr[(long long)k * nbr_stride + r] : -1;
                av[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (idx >= 0) {
                    any = 1;
                    if (c < Cin) av[j] = *reinterpret_cast<const float4*>(X + (long long)idx * ldx + c);
                }
            }
        } else {
            constexpr int F4 = CPAD / 4;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                int e = tid + 256 * j;  // 0..511 = 32 rows x 16 float4 columns
                int r = rb + (e & 31);
                int m4 = e >> 5;  // float4 column 0..15
                int off = m4 / F4, f = m4 % F4;
                int kk = k0 + off;
                av[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (kk < K3 && r < r_end) {
                    int idx = nbr[(long long)kk * nbr_stride + r];
                    if (idx >= 0) {
                        any = 1;
                        av[j] = *reinterpret_cast<const float4*>(X + (long long)idx * ldx + f * 4);
                    }
                }
            }
        }
        // barrier: previous step's MFMA reads are done; also decides whether this step has any pair
        if (!__syncthreads_or(any)) continue;
        if constexpr (CPAD == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                int r = (tid >> 4) + 16 * j;
                *reinterpret_cast<float4*>(&As[r * 64 + (tid & 15) * 4]) = av[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                int e = tid + 256 * j;
                *reinterpret_cast<float4*>(&As[(e & 31) * 64 + (e >> 5) * 4]) = av[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int rl = (tid >> 4) + 16 * j;
            int r = rb + rl;
            int n = n0 + (tid & 15) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < r_end && n < Cout) v = *reinterpret_cast<const float4*>(dY + (long long)r * ldy + n);
            *reinterpret_cast<float4*>(&Bs[rl * 64 + (tid & 15) * 4]) = v;
        }
        __syncthreads();
        const float* ap = &As[wr * 32 + li];
        const float* bp = &Bs[wc * 32 + li];
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            float a = ap[(2 * s + lh) * 64];
            float b = bp[(2 * s + lh) * 64];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }

    // epilogue
    const int col = n0 + wc * 32 + li;
    if (col < Cout) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            int m = wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            long long wrow;
            bool ok;
            if constexpr (CPAD == 0) {
                ok = (c0 + m) < Cin;
                wrow = (long long)k * Cin + c0 + m;
            } else {
                wrow = (long long)k0 * CPAD + m;
                ok = wrow < (long long)K3 * CPAD;
            }
            if (ok) atomicAdd(&dW[wrow * Cout + col], acc[reg]);
        }
    }
}

// =============================================================== C ABI
extern "C" {

int agb_spconv_fwd(const float* X, int ldx, const float* W, const int32_t* nbr, long long nbr_stride, int kflip,
                   const float* bias, float* Y, int ldy, int n_out, int K3, int Cin, int Cout, void* stream) {
    AGB_CHECK_ARG(n_out >= 0 && K3 >= 1 && Cin >= 1 && Cout >= 1, "agb_spconv_fwd: bad sizes");
    AGB_CHECK_ARG(Cin % 4 == 0 && Cout % 4 == 0 && ldx % 4 == 0 && ldy >= Cout,
                  "agb_spconv_fwd: Cin (%d), Cout (%d), ldx (%d) must be multiples of 4 (pad small inputs)", Cin,
                  Cout, ldx);
    AGB_CHECK_ARG(nbr_stride >= n_out, "agb_spconv_fwd: nbr_stride < n_out");
    if (n_out == 0) return AGB_OK;
    dim3 grid(agb_cdiv(n_out, BM), agb_cdiv(Cout, BN)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (Cin == 4)
        hipLaunchKernelGGL(k_spconv_fwd<4>, grid, block, 0, s, X, ldx, W, nbr, nbr_stride, kflip, bias, Y, ldy, n_out,
                           K3, Cin, Cout);
    else if (Cin == 8)
        hipLaunchKernelGGL(k_spconv_fwd<8>, grid, block, 0, s, X, ldx, W, nbr, nbr_stride, kflip, bias, Y, ldy, n_out,
                           K3, Cin, Cout);
    else
        hipLaunchKernelGGL(k_spconv_fwd<0>, grid, block, 0, s, X, ldx, W, nbr, nbr_stride, kflip, bias, Y, ldy, n_out,
                           K3, Cin, Cout);
    AGB_CHECK_LAUNCH("agb_spconv_fwd");
    return AGB_OK;
}

// dW must be zero-filled by the caller (it is accumulated into).
int agb_spconv_bwd_weight(const float* X, int ldx, const float* dY, int ldy, const int32_t* nbr, long long nbr_stride,
                          float* dW, int n_out, int K3, int Cin, int Cout, void* stream) {
    AGB_CHECK_ARG(n_out >= 0 && K3 >= 1 && Cin >= 1 && Cout >= 1, "agb_spconv_bwd_weight: bad sizes");
    AGB_CHECK_ARG(Cin % 4 == 0 && Cout % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0,
                  "agb_spconv_bwd_weight: Cin (%d), Cout (%d), ldx, ldy must be multiples of 4", Cin, Cout);
    if (n_out == 0) return AGB_OK;
    hipStream_t s = (hipStream_t)stream;
    int m_tiles, cin_tiles = 1;
    if (Cin == 4 || Cin == 8) {
        m_tiles = agb_cdiv(K3, 64 / Cin);
    } else {
        cin_tiles = agb_cdiv(Cin, 64);
        m_tiles = K3 * cin_tiles;
    }
    int n_tiles = agb_cdiv(Cout, 64);
    // aim for ~4096 workgroups; at least 256 rows per workgroup (multiple of 32)
    long long target_chunks = 4096 / ((long long)m_tiles * n_tiles);
    if (target_chunks < 1) target_chunks = 1;
    long long rows = (n_out + target_chunks - 1) / target_chunks;
    if (rows < 256) rows = 256;
    rows = (rows + 31) / 32 * 32;
    int chunks = agb_cdiv(n_out, rows);
    dim3 grid(chunks, m_tiles, n_tiles), block(256);
    if (Cin == 4)
        hipLaunchKernelGGL(k_spconv_dw<4>, grid, block, 0, s, X, ldx, dY, ldy, nbr, nbr_stride, dW, n_out, K3, Cin,
                           Cout, (int)rows, cin_tiles);
    else if (Cin == 8)
        hipLaunchKernelGGL(k_spconv_dw<8>, grid, block, 0, s, X, ldx, dY, ldy, nbr, nbr_stride, dW, n_out, K3, Cin,
                           Cout, (int)rows, cin_tiles);
    else
        hipLaunchKernelGGL(k_spconv_dw<0>, grid, block, 0, s, X, ldx, dY, ldy, nbr, nbr_stride, dW, n_out, K3, Cin,
                           Cout, (int)rows, cin_tiles);
    AGB_CHECK_LAUNCH("agb_spconv_bwd_weight");
    return AGB_OK;
}

}  // extern "C"
