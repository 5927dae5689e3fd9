// dwreg.hip — fp32 weight gradient of the sparse convolution with BOTH MFMA operands gathered straight from global memory
// into registers: no LDS staging, no workgroup barrier, no float atomics.
//
//   dW[k][c][o] = sum over the pairs (in, out) of kernel offset k of  X[in][c] * dY[out][o]
//   (ME.MinkowskiConvolution's kernel gradient; reference call sites modules/MinkowskiEngine/common.py:215-226,
//    resnet_block.py:48-55,95-107)
//
// The reduction index of this product is the PAIR.  v_mfma_f32_16x16x4_f32 takes A[m][p] from lane (m = lane & 15,
// p = lane >> 4) and B[p][n] from lane (n = lane & 15, p = lane >> 4): lane (i, p) loads ONE float4 of X[in_p] (channels
// c0 + 4i .. 4i + 3) and ONE float4 of dY[out_p] (columns n0 + 4i .. 4i + 3); component a of the first and b of the second
// are the operands of MFMA (a, b), whose 16 x 16 result block stands for channels {c0 + 4m + a} x columns {n0 + 4n + b}.
// Two 16-byte gathers per lane feed sixteen MFMAs (a 64 x 64 tile of dW[k], four pairs deep); the sixteen accumulators stay
// in registers for the whole unit and leave as float4 rows.  Gathers run DWR_D steps ahead of their MFMAs.
//
// One workgroup per unit = (row chunk, kernel offset, 64-channel tile, 64-column tile); each of its four WAVES takes a
// quarter of the chunk's rows: it compacts its part of the kernel-map column into a wave-private pair list in LDS (ballot +
// prefix; one wave's LDS operations execute in order: no barrier) and walks it with its own sixteen accumulators; the four
// tiles meet in LDS once, at the end.  k_spconv_dw_cmp — four waves that meet at two barriers per 64-pair step, operands
// staged through LDS — spent 27 % of its wave cycles waiting at those barriers (profiles/r02_mfma_pmc_summary.txt).
// (One wave per unit was measured first: with 4050 units of 40-90 us each on 2560 wave slots the kernel ended on its
// longest units: 375 us against the staged kernel's 282 us on 64 -> 64 channels at 211 k rows.)
//
// Row chunks are summed in two levels: every unit writes its partial tile to part[chunk] and k_dw_fold adds the chunks
// in a fixed order — bitwise reproducible, and shorter fp32 chains than one running sum per element.  Without a
// workspace the units fall back to fp32 atomic adds on dW (the contract of agb_spconv_bwd_weight_lp).
// 1-D grid, XCD-aware: the units of one row chunk run back to back on one XCD (workgroup id % 8), so the chunk's X / dY
// rows are re-read from that XCD's L2.
#include "agb_common.h"
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DWR_CH 2048     // rows per chunk at most (the LDS pair list: 16 KB per wave)
// (template parameter DWR_D: steps of four pairs the gathers run ahead of their MFMAs; 8 measured equal to 4.  A 32 KB
// two-round reduction for three or four workgroups per CU instead of two was measured SLOWER on the deep levels: 256 -> 256
// 335 -> 401 us, 512 -> 512 256 -> 361 us — more resident units thrash the XCD's L2: EXPERIMENTS.md)

struct DwrGeo { int rows, chunks, cin_tiles, cout_tiles; };

static DwrGeo dwr_geometry(int n_out, int K3, int Cin, int Cout) {
    DwrGeo g;
    g.cin_tiles = agb_cdiv(Cin, 64);
    g.cout_tiles = agb_cdiv(Cout, 64);
    const long long M = (long long)K3 * g.cin_tiles * g.cout_tiles;
    long long target = 4096 / M;                  // ~4096 units = eight per resident workgroup slot (2 per CU); more, smaller
                                                  // chunks balance better but every unit writes a 16 KB partial tile
    if (target < 1) target = 1;
    long long rows = (n_out + target - 1) / target;
    rows = (rows + 255) / 256 * 256;              // four waves x a multiple of 64 rows
    if (rows < 256) rows = 256;
    if (rows > DWR_CH) rows = DWR_CH;
    g.rows = (int)rows;
    g.chunks = agb_cdiv(n_out > 0 ? n_out : 1, rows);
    return g;
}

// DENSE: the identity map (nbr == NULL, K3 == 1): pair p of a chunk is (row, row).
// One workgroup per unit; its four waves take a quarter of the chunk's rows each (own pair list, own accumulators, no
// barrier while they multiply) and add their four tiles through LDS at the end, in wave order.
template <bool DENSE, int DWR_D>
__global__ __launch_bounds__(256) void k_spconv_dw_reg(const float* __restrict__ X, int ldx, const float* __restrict__ dY,
                                                       int ldy, const int32_t* __restrict__ nbr, long long nbr_stride,
                                                       float* __restrict__ dW, float* __restrict__ part, int n_out, int K3,
                                                       int Cin, int Cout, int rows_per_chunk, int cin_tiles, int cout_tiles,
                                                       int chunks) {
    // [4][64 x 64] tiles of the final reduction; while the waves multiply, the first 16 KB hold their pair lists
    __shared__ __attribute__((aligned(16))) float red[4 * 4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kk = lane >> 4;
    const int M = K3 * cin_tiles * cout_tiles;
    int chunk, m;
    if (chunks >= 16) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        m = j % M;
        chunk = (j / M) * 8 + xcd;
        if (chunk >= chunks) return;
    } else {
        chunk = blockIdx.x % chunks;
        m = blockIdx.x / chunks;
    }
    const int k = m / (cin_tiles * cout_tiles);
    const int c0 = ((m / cout_tiles) % cin_tiles) * 64, n0 = (m % cout_tiles) * 64;
    const int quarter = rows_per_chunk >> 2;                       // rows_per_chunk is a multiple of 64
    const int r_begin = chunk * rows_per_chunk + wave * quarter;
    const int nrows = max(0, min(quarter, n_out - r_begin));
    int2* plist = reinterpret_cast<int2*>(red) + wave * (DWR_CH / 4);

    // ---- this wave's pairs of offset k, in row order
    int total = DENSE ? nrows : 0;
    if (!DENSE) {
        const int32_t* nrow = nbr + (long long)k * nbr_stride + r_begin;
        const unsigned long long lt = (1ull << lane) - 1ull;
        for (int base = 0; base < nrows; base += 64 * 8) {
            int idx[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {          // eight independent loads in flight
                const int r = base + 64 * u + lane;
                idx[u] = r < nrows ? nrow[r] : -1;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool p = idx[u] >= 0;
                const unsigned long long bal = __ballot(p);
                if (p) plist[total + __popcll(bal & lt)] = make_int2(idx[u], r_begin + base + 64 * u + lane);
                total += __popcll(bal);
            }
        }
        __atomic_signal_fence(__ATOMIC_SEQ_CST);   // wave-private list, in-order LDS: a compiler fence is enough
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // clamped columns: lanes past Cin / Cout read valid memory; their products land in rows / columns that are never stored
    const float* xcol = X + min(c0 + 4 * i, Cin - 4);
    const float* ycol = dY + min(n0 + 4 * i, Cout - 4);
    auto pair_of = [&](int p) -> int2 {
        if (DENSE) return make_int2(r_begin + p, r_begin + p);
        return plist[p];
    };
    auto gather = [&](int step, int last_pair, float4& xa, float4& yb) {
        const int2 e = pair_of(min(4 * step + kk, last_pair));
        xa = *reinterpret_cast<const float4*>(xcol + (long long)e.x * ldx);
        yb = *reinterpret_cast<const float4*>(ycol + (long long)e.y * ldy);
    };
    auto mma = [&](const float4& xa, const float4& yb) {
        const float av[4] = {xa.x, xa.y, xa.z, xa.w}, bv[4] = {yb.x, yb.y, yb.z, yb.w};
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[b], acc[a][b], 0, 0, 0);
    };

    if (total > 0) {
        const int nfull = total >> 2, last_pair = total - 1;
        float4 xa[DWR_D], yb[DWR_D];
#pragma unroll
        for (int d = 0; d < DWR_D; ++d) gather(d, last_pair, xa[d], yb[d]);
        for (int s = 0; s < nfull; s += DWR_D) {
#pragma unroll
            for (int d = 0; d < DWR_D; ++d) {
                if (s + d < nfull) mma(xa[d], yb[d]);
                gather(s + d + DWR_D, last_pair, xa[d], yb[d]);     // (clamped past the end: harmless repeats)
            }
        }
        if (total & 3) {     // the last one to three pairs: lanes past the end contribute zeros
            float4 xt, yt;
            gather(nfull, last_pair, xt, yt);
            if (4 * nfull + kk > last_pair) xt = make_float4(0.f, 0.f, 0.f, 0.f);
            mma(xt, yt);
        }
    }

    // ---- the four waves' tiles -> LDS (after every wave is done with its list), summed in wave order.
    // D layout of a 16x16 block: lane holds rows 4 * (lane >> 4) + v, column lane & 15: tile row (channel) 16 kk + 4 v + a,
    // tile columns 4 i + b (b = 0..3: one float4)
    __syncthreads();
    float* mine = red + wave * 4096;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int v = 0; v < 4; ++v)
            *reinterpret_cast<float4*>(mine + (16 * kk + 4 * v + a) * 64 + 4 * i) =
                make_float4(acc[a][0][v], acc[a][1][v], acc[a][2][v], acc[a][3][v]);
    __syncthreads();
    float* dst = (part ? part + (long long)chunk * K3 * Cin * Cout : dW) + (long long)k * Cin * Cout;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int e = threadIdx.x + 256 * t;          // float4 element of the 64 x 64 tile
        const int row = e >> 4, col = (e & 15) * 4;
        float4 sum = *reinterpret_cast<const float4*>(red + row * 64 + col);
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float4 o = *reinterpret_cast<const float4*>(red + w * 4096 + row * 64 + col);
            sum.x += o.x; sum.y += o.y; sum.z += o.z; sum.w += o.w;
        }
        if (c0 + row < Cin && n0 + col < Cout) {
            float* p = dst + (long long)(c0 + row) * Cout + n0 + col;
            if (part) {
                *reinterpret_cast<float4*>(p) = sum;
            } else {
                atomicAdd(p + 0, sum.x); atomicAdd(p + 1, sum.y); atomicAdd(p + 2, sum.z); atomicAdd(p + 3, sum.w);
            }
        }
    }
}

// dW[e] += sum over chunks (ascending) of part[chunk][e]
__global__ __launch_bounds__(256) void k_dw_fold(const float4* __restrict__ part, int chunks, long long n4,
                                                 float4* __restrict__ dW) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n4) return;
    float4 s = dW[e];
    for (int c = 0; c < chunks; ++c) {
        const float4 v = part[(long long)c * n4 + e];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    dW[e] = s;
}

size_t agb_dwreg_workspace_bytes(int n_out, int K3, int Cin, int Cout) {
    if (n_out <= 0) return 0;
    const DwrGeo g = dwr_geometry(n_out, K3, Cin, Cout);
    return (size_t)g.chunks * K3 * Cin * Cout * sizeof(float);
}

// dW += X_gathered^T dY.  workspace: agb_dwreg_workspace_bytes() bytes, or NULL (atomic accumulation).
int agb_dwreg_launch(const float* X, int ldx, const float* dY, int ldy, const int32_t* nbr, long long nbr_stride, float* dW,
                     int n_out, int K3, int Cin, int Cout, void* workspace, size_t workspace_bytes, hipStream_t s) {
    const DwrGeo g = dwr_geometry(n_out, K3, Cin, Cout);
    float* part = (float*)workspace;
    if (part && workspace_bytes < (size_t)g.chunks * K3 * Cin * Cout * sizeof(float)) {
        agb_set_error("weight gradient: workspace of %zu bytes, %zu needed", workspace_bytes,
                      (size_t)g.chunks * K3 * Cin * Cout * sizeof(float));
        return AGB_EINVAL;
    }
    if (g.chunks == 1) part = nullptr;     // one chunk: every element has one writer — plain accumulation is already exact
    const long long M = (long long)K3 * g.cin_tiles * g.cout_tiles;
    const long long units = (g.chunks >= 16 ? 8LL * agb_cdiv(g.chunks, 8) : g.chunks) * M;
    if (units > 0x7fffffffLL) { agb_set_error("weight gradient: too many units"); return AGB_ERANGE; }
#define DWR_LAUNCH(DENSE)                                                                                                  \
    AGB_LAUNCH((k_spconv_dw_reg<DENSE, 4>), dim3((unsigned)units), dim3(256), 0, s, X, ldx, dY, ldy, nbr, nbr_stride, \
                       dW, part, n_out, K3, Cin, Cout, g.rows, g.cin_tiles, g.cout_tiles, g.chunks)
    if (nbr) DWR_LAUNCH(false); else DWR_LAUNCH(true);
#undef DWR_LAUNCH
    if (part) {
        const long long n4 = (long long)K3 * Cin * Cout / 4;
        hipLaunchKernelGGL(k_dw_fold, dim3((unsigned)agb_cdiv(n4, 256)), dim3(256), 0, s, (const float4*)part, g.chunks, n4,
                           (float4*)dW);
    }
    return AGB_OK;
}
