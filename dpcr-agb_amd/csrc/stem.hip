// stem.hip — the 3 -> 64 channel stem convolution (7^3 offsets on the raw voxels) as PAIR-SPARSE vector kernels.
//
// Reference: the first layer of every MinkowskiEngine ResNet/SENet of the reference
// (torch_points3d/modules/MinkowskiEngine/resnet.py: conv1 = MinkowskiConvolution(in_channels, 64, kernel_size=7...),
// executed by ME's gather-GEMM-scatter, MinkowskiEngine/src/convolution_kernel.cu).
//
// With 3 input channels a (row, offset) pair carries 3 x 64 multiply-adds, and only ~17 % of the 343 offsets of a row
// have a neighbour (58 pairs per row on the NFI plots).  The MFMA kernels (k_spconv_fwd3 / k_spconv_dw_small_cmp) run
// the dense 343 x 3 reduction and spend five sixths of their MFMA issue slots on zero rows; a K = 3 pair cannot be
// compacted into an MFMA operand.  Here one lane owns one OUTPUT CHANNEL and a wave walks the present pairs of a
// 64-row tile, offset by offset:
//   forward   acc[row][lane] (LDS, 16 KB per wave) += x[nbr][0..2] (broadcast by v_readlane) * W[k][0..2][lane]
//   weight    acc[k][0..2] (registers, lane = channel) += x[nbr][0..2] * dY[row][lane] (tile staged in LDS)
// Work is proportional to the pairs (9 GFLOP instead of 68 GFLOP dense on 410 k rows); the bounds are LDS bandwidth
// (one 256-B read-modify-write per pair forward, one 256-B read backward) and scalar issue of the pair loop.
// Summation order is fixed (offsets ascending, rows ascending, row groups folded in order): results are deterministic.
#include "agb_common.h"

#define ST_ROWS 64
#ifndef ST_KB
#define ST_KB 4    // offset PAIRS per block of the forward pipeline (two register sets of ST_KB x 9 floats)
#endif
#define ST_D 4     // gathers in flight in the weight-gradient kernel

__device__ __forceinline__ float st_bcast(float v, int r) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), r));
}

// ---------------------------------------------------------------------------------------------------------------
// forward: Y[row, :] = bias + sum_k X[nbr[k][row], 0..2] . W[k][0..2][:]        (Cout == 64, X rows 4 floats wide)
// grid: 8 * ceil(ntiles / 8) single-wave workgroups, XCD-aware (consecutive tiles share an L2: their neighbour rows
// overlap).
// Latency: the offsets are taken in blocks of ST_KB.  While block b is processed, the gathers and weights of block b+1
// and the neighbour indices of block b+2 are in flight; every block starts by waiting for all of them (they had a
// whole block of pair work, ~1.5 k cycles per wave, to arrive).  Two register sets (A/B) alternate statically (the loop
// is unrolled by two blocks): a rotating queue indexed k % depth made the compiler copy registers at the loop back
// edge, and the copy of a just-issued load drains the pipeline.
struct StemSet {
    float x[ST_KB][3];
    float w[ST_KB][2][3];          // weights of the two offsets a lane pair (row, row + 32 lanes) covers
    unsigned long long m[ST_KB];   // bit l: lane l has a neighbour (bits 0-31: even offset, 32-63: odd offset)
};

// Clears bit r of m (s_bitset0_b64: one scalar instruction instead of the three of m &= m - 1).
__device__ __forceinline__ void st_clear(unsigned long long& m, int r) {
    asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(r));
}

// A wave owns FW_ROWS = 32 rows: lanes 0-31 hold the neighbour of offset 2j of those rows, lanes 32-63 the neighbour of
// offset 2j+1 (8 KB of LDS accumulators per wave: ~19 waves per CU, enough to overlap scalar, vector and LDS issue —
// a wave issues one instruction per 4 cycles whatever its type).
#define FW_ROWS 32
__global__ __launch_bounds__(64) void k_stem_fwd_sparse(const float* __restrict__ X, const float* __restrict__ W,
                                                        const int32_t* __restrict__ nbr, long long nbr_stride,
                                                        int kflip, const float* __restrict__ bias,
                                                        float* __restrict__ Y, int ldy, int n_out, int K3,
                                                        int ntiles) {
    __shared__ __attribute__((aligned(16))) float acc[FW_ROWS * 64];
    const int lane = threadIdx.x;
    const int xcd = blockIdx.x & 7, per = (ntiles + 7) >> 3, j = blockIdx.x >> 3;
    const int tile = xcd * per + j;
    if (j >= per || tile >= ntiles) return;
    const int row0 = tile * FW_ROWS;
    const int half = lane >> 5;
    const int row = row0 + (lane & 31);
    const bool rv = row < n_out;
    const int rowc = rv ? row : n_out - 1;
    const float4* X4 = reinterpret_cast<const float4*>(X);

    {
        const float b = bias ? bias[lane] : 0.f;
#pragma unroll 8
        for (int r = 0; r < FW_ROWS; ++r) acc[r * 64 + lane] = b;
    }

    const int nk2 = (K3 + 1) >> 1;   // offset pairs
    int nn[ST_KB];   // neighbour indices of the block after the one whose gathers are in flight
    // address = wave-uniform part (offset pair) + per-lane part (row, and one map row further for the odd half)
    const int32_t* nrow = nbr + rowc;
    const long long hs = half ? (kflip ? -nbr_stride : nbr_stride) : 0;
    auto load_idx = [&](int blk) {
#pragma unroll
        for (int i = 0; i < ST_KB; ++i) {
            int jc = blk * ST_KB + i;
            jc = jc < nk2 ? jc : nk2 - 1;
            const int k0 = 2 * jc;
            const long long base = (long long)(kflip ? (K3 - 1 - k0) : k0) * nbr_stride;
            const long long lo = (k0 + 1 < K3) ? hs : 0;   // the odd offset of the last pair may not exist: reload the even
            nn[i] = nrow[base + lo];
        }
    };
    // gathers + weights of block blk from the indices in nn (which must be those of blk)
    auto load_set = [&](StemSet& s, int blk) {
#pragma unroll
        for (int i = 0; i < ST_KB; ++i) {
            const int k0 = 2 * (blk * ST_KB + i);
            const int nv = nn[i];
            s.m[i] = __ballot(nv >= 0 && rv && (k0 + half) < K3);
            const float4 v = X4[nv >= 0 ? nv : 0];
            s.x[i][0] = v.x; s.x[i][1] = v.y; s.x[i][2] = v.z;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int kc = (k0 + h) < K3 ? (k0 + h) : K3 - 1;
#pragma unroll
                for (int c = 0; c < 3; ++c) s.w[i][h][c] = W[(kc * 3 + c) * 64 + lane];
            }
        }
    };
    // adds the pairs of one offset: mask bits -> lanes holding x; the row is lane & 31
    auto run = [&](unsigned long long m, float x0, float x1, float x2, float w0, float w1, float w2) {
        while (__builtin_popcountll(m) >= 4) {
            const int r0 = __builtin_ctzll(m); st_clear(m, r0);
            const int r1 = __builtin_ctzll(m); st_clear(m, r1);
            const int r2 = __builtin_ctzll(m); st_clear(m, r2);
            const int r3 = __builtin_ctzll(m); st_clear(m, r3);
            float* p0 = &acc[(r0 & 31) * 64 + lane]; float* p1 = &acc[(r1 & 31) * 64 + lane];
            float* p2 = &acc[(r2 & 31) * 64 + lane]; float* p3 = &acc[(r3 & 31) * 64 + lane];
            float a0 = *p0, a1 = *p1, a2 = *p2, a3 = *p3;
            a0 = fmaf(st_bcast(x0, r0), w0, a0); a1 = fmaf(st_bcast(x0, r1), w0, a1);
            a2 = fmaf(st_bcast(x0, r2), w0, a2); a3 = fmaf(st_bcast(x0, r3), w0, a3);
            a0 = fmaf(st_bcast(x1, r0), w1, a0); a1 = fmaf(st_bcast(x1, r1), w1, a1);
            a2 = fmaf(st_bcast(x1, r2), w1, a2); a3 = fmaf(st_bcast(x1, r3), w1, a3);
            a0 = fmaf(st_bcast(x2, r0), w2, a0); a1 = fmaf(st_bcast(x2, r1), w2, a1);
            a2 = fmaf(st_bcast(x2, r2), w2, a2); a3 = fmaf(st_bcast(x2, r3), w2, a3);
            *p0 = a0; *p1 = a1; *p2 = a2; *p3 = a3;
        }
        while (m) {
            const int r0 = __builtin_ctzll(m); st_clear(m, r0);
            float* p0 = &acc[(r0 & 31) * 64 + lane];
            float a0 = *p0;
            a0 = fmaf(st_bcast(x0, r0), w0, a0);
            a0 = fmaf(st_bcast(x1, r0), w1, a0);
            a0 = fmaf(st_bcast(x2, r0), w2, a0);
            *p0 = a0;
        }
    };
    auto process = [&](const StemSet& s) {
#pragma unroll
        for (int i = 0; i < ST_KB; ++i) {
            const unsigned long long m = s.m[i];
            // even offset first, then the odd one (ascending offsets: fixed summation order)
            run(m & 0xffffffffull, s.x[i][0], s.x[i][1], s.x[i][2], s.w[i][0][0], s.w[i][0][1], s.w[i][0][2]);
            run(m & 0xffffffff00000000ull, s.x[i][0], s.x[i][1], s.x[i][2], s.w[i][1][0], s.w[i][1][1], s.w[i][1][2]);
        }
    };

    StemSet A, B;
    const int nblk = (nk2 + ST_KB - 1) / ST_KB;
    load_idx(0);
    load_set(A, 0);
    load_idx(1);
    for (int b = 0; b < nblk; b += 2) {
        load_set(B, b + 1);          // waits for the indices of block b+1 (and with them for set A)
        load_idx(b + 2);
        asm volatile("" ::: "memory");
        process(A);
        load_set(A, b + 2);
        load_idx(b + 3);
        asm volatile("" ::: "memory");
        process(B);
    }
    __syncthreads();
    // tile out: 4 rows of 16 float4 per pass
    const int c4 = lane & 15, rl = lane >> 4;
#pragma unroll 4
    for (int i = 0; i < FW_ROWS / 4; ++i) {
        const int r = rl + 4 * i;
        if (row0 + r < n_out)
            *reinterpret_cast<float4*>(Y + (long long)(row0 + r) * ldy + 4 * c4) =
                *reinterpret_cast<const float4*>(&acc[r * 64 + 4 * c4]);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// weight gradient: dW[k][c][co] = sum_rows X[nbr[k][row], c] * dY[row, co]     (Cout == 64)
// grid (G row groups, ceil(K3 / KC) offset chunks), one wave each; every wave keeps the KC x 3 sums of its offsets in
// registers (lane = co) over the tiles of its group and writes them to partial[g][K3][3][64]; k_stem_wgrad_fold adds
// the groups in order.
template <int KC>
__global__ __launch_bounds__(64) void k_stem_wgrad_sparse(const float* __restrict__ X, const float* __restrict__ dY,
                                                          int ldy, const int32_t* __restrict__ nbr,
                                                          long long nbr_stride, float* __restrict__ partial,
                                                          int n_out, int K3, int tiles_per_group) {
    __shared__ __attribute__((aligned(16))) float dys[ST_ROWS * 64];
    const int lane = threadIdx.x;
    const int g = blockIdx.x, k0 = blockIdx.y * KC;
    const float4* X4 = reinterpret_cast<const float4*>(X);
    const int c4 = lane & 15, rl = lane >> 4;

    float acc[KC][3];
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) { acc[kk][0] = 0.f; acc[kk][1] = 0.f; acc[kk][2] = 0.f; }

    for (int t = 0; t < tiles_per_group; ++t) {
        const int row0 = (g * tiles_per_group + t) * ST_ROWS;
        if (row0 >= n_out) break;
        const int row = row0 + lane;
        const bool rv = row < n_out;
        // neighbour indices of all KC offsets of this tile: KC independent loads in flight
        int nv[KC];
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) {
            const int k = k0 + kk;
            const int kc = k < K3 ? k : K3 - 1;
            const int v = rv ? nbr[(long long)kc * nbr_stride + row] : -1;
            nv[kk] = k < K3 ? v : -1;
        }
        __syncthreads();   // the previous tile's reads of dys are done
#pragma unroll 4
        for (int i = 0; i < ST_ROWS / 4; ++i) {
            const int r = rl + 4 * i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row0 + r < n_out) v = *reinterpret_cast<const float4*>(dY + (long long)(row0 + r) * ldy + 4 * c4);
            *reinterpret_cast<float4*>(&dys[r * 64 + 4 * c4]) = v;
        }
        float4 xq[ST_D];
#pragma unroll
        for (int d = 0; d < ST_D; ++d) xq[d] = X4[nv[d] >= 0 ? nv[d] : 0];
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) {
            const float4 xv = xq[kk % ST_D];
            if (kk + ST_D < KC) xq[kk % ST_D] = X4[nv[kk + ST_D] >= 0 ? nv[kk + ST_D] : 0];
            unsigned long long m = __ballot(nv[kk] >= 0);
            float s0 = acc[kk][0], s1 = acc[kk][1], s2 = acc[kk][2];
            while (__builtin_popcountll(m) >= 4) {
                const int r0 = __builtin_ctzll(m); m &= m - 1;
                const int r1 = __builtin_ctzll(m); m &= m - 1;
                const int r2 = __builtin_ctzll(m); m &= m - 1;
                const int r3 = __builtin_ctzll(m); m &= m - 1;
                const float d0 = dys[r0 * 64 + lane], d1 = dys[r1 * 64 + lane];
                const float d2 = dys[r2 * 64 + lane], d3 = dys[r3 * 64 + lane];
                s0 = fmaf(st_bcast(xv.x, r0), d0, s0); s1 = fmaf(st_bcast(xv.y, r0), d0, s1); s2 = fmaf(st_bcast(xv.z, r0), d0, s2);
                s0 = fmaf(st_bcast(xv.x, r1), d1, s0); s1 = fmaf(st_bcast(xv.y, r1), d1, s1); s2 = fmaf(st_bcast(xv.z, r1), d1, s2);
                s0 = fmaf(st_bcast(xv.x, r2), d2, s0); s1 = fmaf(st_bcast(xv.y, r2), d2, s1); s2 = fmaf(st_bcast(xv.z, r2), d2, s2);
                s0 = fmaf(st_bcast(xv.x, r3), d3, s0); s1 = fmaf(st_bcast(xv.y, r3), d3, s1); s2 = fmaf(st_bcast(xv.z, r3), d3, s2);
            }
            while (m) {
                const int r0 = __builtin_ctzll(m); m &= m - 1;
                const float d0 = dys[r0 * 64 + lane];
                s0 = fmaf(st_bcast(xv.x, r0), d0, s0); s1 = fmaf(st_bcast(xv.y, r0), d0, s1); s2 = fmaf(st_bcast(xv.z, r0), d0, s2);
            }
            acc[kk][0] = s0; acc[kk][1] = s1; acc[kk][2] = s2;
        }
    }
    float* p = partial + (long long)g * K3 * 192;
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) {
        const int k = k0 + kk;
        if (k < K3) {
            p[(k * 3 + 0) * 64 + lane] = acc[kk][0];
            p[(k * 3 + 1) * 64 + lane] = acc[kk][1];
            p[(k * 3 + 2) * 64 + lane] = acc[kk][2];
        }
    }
}

// dW[i] = sum_g partial[g][i]   (groups in order; total = K3 * 192 floats, a multiple of 4)
__global__ __launch_bounds__(256) void k_stem_wgrad_fold(const float* __restrict__ partial, int G, int total4,
                                                         float* __restrict__ dW) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const float4* p = reinterpret_cast<const float4*>(partial) + i;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int g = 0;
    for (; g + 8 <= G; g += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(long long)(g + u) * total4];
#pragma unroll
        for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; g < G; ++g) {
        const float4 v = p[(long long)g * total4];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    reinterpret_cast<float4*>(dW)[i] = s;
}

#define ST_KC 32
static int g_stem_mode = -1;   // 1: pair-sparse kernels, 0: the dense MFMA kernels of spconv.hip (default)

extern "C" int agb_stem_sparse_enabled() {
    if (g_stem_mode < 0) {
        const char* e = getenv("AGB_STEM_SPARSE");
        g_stem_mode = e ? (atoi(e) != 0) : 0;
    }
    return g_stem_mode;
}

extern "C" int agb_stem_fwd_sparse_launch(const float* X, const float* W, const int32_t* nbr, long long nbr_stride, int kflip,
                               const float* bias, float* Y, int ldy, int n_out, int K3, hipStream_t s) {
    const int ntiles = agb_cdiv(n_out, FW_ROWS);
    hipLaunchKernelGGL(k_stem_fwd_sparse, dim3(8 * agb_cdiv(ntiles, 8)), dim3(64), 0, s, X, W, nbr, nbr_stride, kflip,
                       bias, Y, ldy, n_out, K3, ntiles);
    return AGB_OK;
}

static void stem_wgrad_geometry(int n_out, int K3, int* G, int* tpg, int* nchunks) {
    const int tiles = agb_cdiv(n_out > 0 ? n_out : 1, ST_ROWS);
    *nchunks = agb_cdiv(K3, ST_KC);
    // one resident round: ~3 waves per SIMD x 1024 SIMDs, equal tile counts per group
    int g = 3072 / *nchunks;
    if (g < 1) g = 1;
    if (g > tiles) g = tiles;
    *tpg = agb_cdiv(tiles, g);
    *G = agb_cdiv(tiles, *tpg);
}

extern "C" {

int agb_spconv_set_stem_mode(int mode) {
    AGB_CHECK_ARG(mode == 0 || mode == 1, "agb_spconv_set_stem_mode: mode %d", mode);
    g_stem_mode = mode;
    return AGB_OK;
}

// floats of scratch agb_spconv_bwd_weight3 needs for n_out rows and K3 offsets
int agb_spconv_bwd_weight3_scratch(int n_out, int K3) {
    int G, tpg, nchunks;
    stem_wgrad_geometry(n_out, K3, &G, &tpg, &nchunks);
    return G * K3 * 192;   // G * ceil(K3 / 32) <= 3072: at most 19 M floats
}

// Weight gradient of a 3 -> 64 channel convolution.  X [n_in, 4] (rows 4 floats wide, 3 used), dY [n_out, ldy >= 64],
// nbr [K3][n_out]; dW [K3, 3, 64] is WRITTEN (not accumulated).  scratch: agb_spconv_bwd_weight3_scratch() floats.
int agb_spconv_bwd_weight3(const float* X, const float* dY, int ldy, const int32_t* nbr, long long nbr_stride,
                           float* dW, float* scratch, int n_out, int K3, int Cout, void* stream) {
    AGB_CHECK_ARG(n_out >= 0 && K3 >= 1, "agb_spconv_bwd_weight3: bad sizes");
    AGB_CHECK_ARG(Cout == 64 && ldy % 4 == 0 && ldy >= 64, "agb_spconv_bwd_weight3: Cout (%d) must be 64, ldy (%d) a "
                  "multiple of 4", Cout, ldy);
    hipStream_t s = (hipStream_t)stream;
    int G, tpg, nchunks;
    stem_wgrad_geometry(n_out, K3, &G, &tpg, &nchunks);
    if (n_out == 0) G = 0;
    if (G > 0)
        hipLaunchKernelGGL(k_stem_wgrad_sparse<ST_KC>, dim3(G, nchunks), dim3(64), 0, s, X, dY, ldy, nbr, nbr_stride,
                           scratch, n_out, K3, tpg);
    const int total4 = K3 * 192 / 4;
    hipLaunchKernelGGL(k_stem_wgrad_fold, dim3(agb_cdiv(total4, 256)), dim3(256), 0, s, scratch, G, total4, dW);
    AGB_CHECK_LAUNCH("agb_spconv_bwd_weight3");
    return AGB_OK;
}

}  // extern "C"
