// stem.hip — PAIR-SPARSE kernels of the 7^3 stem convolution (3 -> 64 channels; ME.MinkowskiConvolution of
// modules/MinkowskiEngine/SENet.py:47-53, kernel_size 7, on the first coordinate level).
//
// The stem's kernel map is 84-87 % empty (56 of 343 neighbours per voxel on the synthetic NFI plots) and the emptiness has no
// tile structure: a 64-row tile sees 81 % of the 343 offsets in its natural (z-major) row order and 88 % when the rows are
// made spatially coherent (Morton order) — profiles/r04_stem_coherence.txt — so a dense-over-offsets MFMA form (csrc/spconv.hip
// k_spconv_dw_small_cmp: 48 GFLOP issued for 9.4 GFLOP useful) cannot skip its way out.  The kernels here touch PAIRS.
//
// Weight gradient  dW[k][c][o] = sum over the pairs (n, m = nbr[k][n]) of offset k of  x[m][c] * dy[n][o]:
//   one v_mfma_f32_4x4x1_16B_f32 PER PAIR.  The instruction computes 16 independent 4 x 4 outer products (block b = lane >> 2):
//   D_b[i][j] += A_b[i] * B_b[j].  With B = dy[n][lane] (block b = output channels 4b .. 4b+3, a coalesced row read) and
//   A_b[i] = x[m][i] for every block, register i of lane `o` accumulates dW[k][i][o]: three useful accumulator registers per
//   lane, the whole 3 x 64 outer product of a pair in one 8-cycle instruction — 24.6 M pairs x 8 cycles / 1024 SIMDs = 0.08 ms
//   of matrix-pipe time at B = 32 against 0.31 ms for the dense form at the MFMA peak.
//   The A operand of SIXTEEN pairs comes from ONE gather instruction: lane l loads x[m_(l >> 2)][l & 3] (16 cache lines), and
//   pair p's MFMA takes block p of that register for all blocks (the CBSZ / ABID broadcast of the MFMA encoding: cbsz = 4,
//   abid = p).  dy rows of a 256-row chunk are staged in LDS once and read by every offset (a row of dy is used by ~56 pairs).
//   A 1024-thread workgroup (16 waves) owns a class q of the offsets (k % nq == q) and walks row chunks; wave w owns the
//   offsets k = (16 j + w) nq + q, j < 11, and keeps their accumulators in registers over all of its chunks: no atomics, one
//   partial tile per (row partition, offset), folded in ascending order (k_stem_dw_fold) — bitwise reproducible.
#include "agb_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SDW_R 256        // rows per chunk (dy staged in LDS: 64 KB, double-buffered)
#define SDW_OPW 11       // offsets a wave owns (accumulators: 4 VGPRs each)
#define SDW_WAVES 16

struct StemDwGeo { int nq, nchunks, nparts; };

static StemDwGeo stem_dw_geometry(int n_out, int K3) {
    StemDwGeo g;
    g.nq = agb_cdiv(K3, SDW_WAVES * SDW_OPW);                 // 343 offsets -> 2 classes of 176 slots
    g.nchunks = agb_cdiv(n_out > 0 ? n_out : 1, SDW_R);
    int parts = 256 / g.nq;                                   // one workgroup per CU
    if (parts < 1) parts = 1;
    g.nparts = g.nchunks < parts ? g.nchunks : parts;
    return g;
}

bool agb_stem_dw_ok(int n_out, int K3, int Cin, int Cout, int ldx, int ldy) {
    return Cin == 4 && Cout == 64 && ldx == 4 && ldy % 4 == 0 && n_out > 0 && n_out < (1 << 24) && K3 >= 1 &&
           agb_cdiv(K3, SDW_WAVES * SDW_OPW) <= 16;
}

size_t agb_stem_dw_workspace_bytes(int n_out, int K3) {
    const StemDwGeo g = stem_dw_geometry(n_out, K3);
    return (size_t)g.nparts * K3 * 4 * 64 * sizeof(float);
}

__global__ __launch_bounds__(1024) void k_stem_dw_pairs(const float* __restrict__ X, const float* __restrict__ dY, int ldy,
                                                        const int32_t* __restrict__ nbr, long long nbr_stride,
                                                        float* __restrict__ part, int n_out, int K3, int nq, int nchunks,
                                                        int nparts) {
    __shared__ __attribute__((aligned(16))) float s_dy[2][SDW_R * 64];
    __shared__ unsigned s_list[SDW_WAVES][SDW_R];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int q = blockIdx.x % nq, rp = blockIdx.x / nq;
    unsigned* list = s_list[w];

    f32x4 acc[SDW_OPW];
#pragma unroll
    for (int j = 0; j < SDW_OPW; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // dy chunk staging: thread t moves four float4 (row t >> 2 of the chunk, columns 16 (t & 3) .. + 15)
    // (four named registers and macros, not an array written inside a lambda: those stay in scratch memory)
    const int st_row = tid >> 2, st_col = (tid & 3) * 16;
    float4 stg0, stg1, stg2, stg3;
#define SDW_LOAD_CHUNK(CH)                                                                         \
    do {                                                                                           \
        const float* src_ = dY + (long long)min((CH) * SDW_R + st_row, n_out - 1) * ldy + st_col;  \
        stg0 = *reinterpret_cast<const float4*>(src_);                                             \
        stg1 = *reinterpret_cast<const float4*>(src_ + 4);                                         \
        stg2 = *reinterpret_cast<const float4*>(src_ + 8);                                         \
        stg3 = *reinterpret_cast<const float4*>(src_ + 12);                                        \
    } while (0)
#define SDW_STORE_CHUNK(BUF)                                                                       \
    do {                                                                                           \
        float* dst_ = &s_dy[BUF][st_row * 64 + st_col];                                            \
        *reinterpret_cast<float4*>(dst_) = stg0;                                                   \
        *reinterpret_cast<float4*>(dst_ + 4) = stg1;                                               \
        *reinterpret_cast<float4*>(dst_ + 8) = stg2;                                               \
        *reinterpret_cast<float4*>(dst_ + 12) = stg3;                                              \
    } while (0)

    int chunk = rp;
    if (chunk < nchunks) {
        SDW_LOAD_CHUNK(chunk);
        SDW_STORE_CHUNK(0);
    }
    __syncthreads();
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int it = 0; chunk < nchunks; ++it, chunk += nparts) {
        const int buf = it & 1;
        const bool more = chunk + nparts < nchunks;
        if (more) SDW_LOAD_CHUNK(chunk + nparts);                  // in flight while this chunk multiplies
        const float* dyb = s_dy[buf];
        const int r0 = chunk * SDW_R;
#pragma unroll
        for (int j = 0; j < SDW_OPW; ++j) {
            const int k = (j * SDW_WAVES + w) * nq + q;
            if (k >= K3) continue;                                 // (wave-uniform)
            // ---- the pairs of offset k in this chunk, in row order: list entry = (input row << 8) | local output row
            const int32_t* col = nbr + (long long)k * nbr_stride + r0;
            const int i0 = (r0 + lane < n_out) ? col[lane] : -1;
            const int i1 = (r0 + 64 + lane < n_out) ? col[64 + lane] : -1;
            const int i2 = (r0 + 128 + lane < n_out) ? col[128 + lane] : -1;
            const int i3 = (r0 + 192 + lane < n_out) ? col[192 + lane] : -1;
            int total = 0;
#define SDW_COMPACT(IDX, S)                                                                                  \
    do {                                                                                                     \
        const bool p_ = (IDX) >= 0;                                                                          \
        const unsigned long long bal_ = __ballot(p_);                                                        \
        if (p_) list[total + __popcll(bal_ & lt)] = ((unsigned)(IDX) << 8) | (unsigned)(64 * (S) + lane);    \
        total += __popcll(bal_);                                                                             \
    } while (0)
            SDW_COMPACT(i0, 0); SDW_COMPACT(i1, 1); SDW_COMPACT(i2, 2); SDW_COMPACT(i3, 3);
#undef SDW_COMPACT
            if (total == 0) continue;
            // pad the last group of 16 with entries of pair 0 (their A operand is zeroed below)
            if (lane < 16 && total + lane < SDW_R) list[total + lane] = 0u;
            __atomic_signal_fence(__ATOMIC_SEQ_CST);               // wave-private list, in-order LDS: a compiler fence is enough
            const int ngroups = (total + 15) >> 4;
            const int g4 = lane >> 2, c4 = lane & 3;
            unsigned e = list[g4];
            float a = X[(long long)(e >> 8) * 4 + c4];
            for (int g = 0; g < ngroups; ++g) {
                const unsigned e_cur = e;
                float a_cur = (16 * g + g4 < total) ? a : 0.f;
                if (g + 1 < ngroups) {                              // next group's entries and gathered x in flight
                    e = list[16 * (g + 1) + g4];
                    a = X[(long long)(e >> 8) * 4 + c4];
                }
                f32x4 d = acc[j];
#pragma unroll
                for (int p = 0; p < 16; ++p) {
                    const int n = __builtin_amdgcn_readlane((int)e_cur, 4 * p) & 255;
                    const float b = dyb[n * 64 + lane];
                    // block p of `a_cur` (lanes 4p .. 4p+3: x[m_p][0..3]) is the A operand of all 16 blocks
                    switch (p) {
#define SDW_CASE(P) case P: d = __builtin_amdgcn_mfma_f32_4x4x1f32(a_cur, b, d, 4, P, 0); break;
                        SDW_CASE(0) SDW_CASE(1) SDW_CASE(2) SDW_CASE(3) SDW_CASE(4) SDW_CASE(5) SDW_CASE(6) SDW_CASE(7)
                        SDW_CASE(8) SDW_CASE(9) SDW_CASE(10) SDW_CASE(11) SDW_CASE(12) SDW_CASE(13) SDW_CASE(14) SDW_CASE(15)
#undef SDW_CASE
                    }
                }
                acc[j] = d;
            }
        }
        if (more) SDW_STORE_CHUNK(buf ^ 1);
        __syncthreads();
    }
    // ---- partial tiles: part[rp][k][c][o], register i of lane o = dW[k][c = i][o]
    float* dst = part + (long long)rp * K3 * 256;
#pragma unroll
    for (int j = 0; j < SDW_OPW; ++j) {
        const int k = (j * SDW_WAVES + w) * nq + q;
        if (k >= K3) continue;
#pragma unroll
        for (int c = 0; c < 4; ++c) dst[(long long)k * 256 + c * 64 + lane] = acc[j][c];
    }
}

// dW[e] += sum over the row partitions (ascending) of part[rp][e]
__global__ __launch_bounds__(256) void k_stem_dw_fold(const float4* __restrict__ part, int nparts, long long n4,
                                                      float4* __restrict__ dW) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n4) return;
    float4 s = dW[e];
    for (int c = 0; c < nparts; ++c) {
        const float4 v = part[(long long)c * n4 + e];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    dW[e] = s;
}

// dW [K3][4][64] += gathered(X)^T dY through `workspace` (agb_stem_dw_workspace_bytes): X rows 4 floats wide (channel 3 = 0)
int agb_stem_dw_launch(const float* X, const float* dY, int ldy, const int32_t* nbr, long long nbr_stride, float* dW, int n_out,
                       int K3, void* workspace, size_t workspace_bytes, hipStream_t s) {
    const StemDwGeo g = stem_dw_geometry(n_out, K3);
    const size_t need = (size_t)g.nparts * K3 * 256 * sizeof(float);
    if (workspace == nullptr || workspace_bytes < need) {
        agb_set_error("stem weight gradient: workspace of %zu bytes, %zu needed", workspace_bytes, need);
        return AGB_EINVAL;
    }
    AGB_LAUNCH(k_stem_dw_pairs, dim3(g.nparts * g.nq), dim3(1024), 0, s, X, dY, ldy, nbr, nbr_stride, (float*)workspace, n_out,
               K3, g.nq, g.nchunks, g.nparts);
    const long long n4 = (long long)K3 * 64;
    hipLaunchKernelGGL(k_stem_dw_fold, dim3((unsigned)agb_cdiv(n4, 256)), dim3(256), 0, s, (const float4*)workspace, g.nparts,
                       n4, (float4*)dW);
    return AGB_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// Forward  y[n][o] = bias[o] + sum over the present neighbours m = grid[cell(n) + delta_k] of  sum_c x[m][c] * W[k][c][o]
//
// One wave owns 64 consecutive output rows: LANE l = ROW l.  Per offset k: every lane probes the level's dense grid for its
// row's neighbour (the cells of consecutive rows of the z-major level order are mostly adjacent: a handful of cache lines),
// gathers the neighbour's 16-byte feature row, and — optionally — writes the kernel-map column (coalesced).  The rows form
// sixteen groups of four; v_mfma_f32_4x4x1_16B_f32 with the A-block broadcast (cbsz = 4, abid = p) multiplies group p's four
// feature values of channel c (lanes 4p .. 4p+3 of the gathered register) with W[k][c][lane] into acc[p]: register i of lane
// o = y[row 4p + i][o].  A group none of whose four rows has offset k is skipped (wave-uniform test of four ballot bits):
// 42 % of the (group, offset) steps remain at map density 0.13, three 8-cycle MFMAs each — 23 GFLOP issued for 9.4 useful
// where the dense-over-offsets kernel (k_spconv_fwd3) issues 57.  The accumulators of all 64 rows stay in registers for the
// whole kernel (64 VGPRs): no LDS, no barrier, no scatter.  W[k] (768 bytes) is read by every wave as three coalesced
// dword loads one offset ahead; probes run two offsets ahead, gathers one.
#define SFW_DELTA_MAX 736

__device__ __forceinline__ int stem_probe_base(const int4 c, int ox, int oy, int oz, int X, int Y, int Z, int ts) {
    return ((c.x * Z + (c.w - oz) / ts) * Y + (c.z - oy) / ts) * X + (c.y - ox) / ts;
}

template <bool WRITE_MAP>
__global__ __launch_bounds__(256) void k_stem_fwd_pairs(const float* __restrict__ X, const float* __restrict__ W,
                                                        const float* __restrict__ bias, float* __restrict__ Yo, int ldy,
                                                        int n_out, int K3, const int4* __restrict__ coords,
                                                        const int32_t* __restrict__ grid, int ox, int oy, int oz, int GX, int GY,
                                                        int GZ, int ts, int K, int32_t* __restrict__ nbr_out,
                                                        long long nbr_out_stride) {
    __shared__ int s_delta[SFW_DELTA_MAX];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int k = tid; k < K3; k += 256) {
        const int h = K >> 1, ix = k % K, iy = (k / K) % K, iz = k / (K * K);
        s_delta[k] = ((iz - h) * GY + (iy - h)) * GX + (ix - h);
    }
    __syncthreads();
    const int r0 = (blockIdx.x * 4 + w) * 64;
    if (r0 >= n_out) return;
    const int row = r0 + lane;
    const bool row_ok = row < n_out;
    const int base = stem_probe_base(coords[min(row, n_out - 1)], ox, oy, oz, GX, GY, GZ, ts);

    f32x4 acc[16];
    const float bv = bias ? bias[lane] : 0.f;
#pragma unroll
    for (int p = 0; p < 16; ++p) acc[p] = (f32x4){bv, bv, bv, bv};

    // software pipeline over the offsets: cell value of k + 2, feature row of k + 1, weights of k + 1
    int cell_n2, cell_n1;
    float4 x_n1;
    float w0_n1, w1_n1, w2_n1;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    cell_n1 = grid[base + s_delta[0]];
    cell_n2 = grid[base + s_delta[min(1, K3 - 1)]];
    {
        const bool pr = row_ok && cell_n1 != INT_MAX;
        x_n1 = *reinterpret_cast<const float4*>(X + (long long)(pr ? cell_n1 : 0) * 4);
        w0_n1 = W[lane]; w1_n1 = W[64 + lane]; w2_n1 = W[128 + lane];
    }
    for (int k = 0; k < K3; ++k) {
        const int cell = cell_n1;
        const bool present = row_ok && cell != INT_MAX;
        float4 x = x_n1;
        if (!present) x = zero4;
        const float w0 = w0_n1, w1 = w1_n1, w2 = w2_n1;
        // next offset's gather and weights, the probe after it
        cell_n1 = cell_n2;
        {
            const int kn = min(k + 1, K3 - 1);
            const bool pn = row_ok && cell_n1 != INT_MAX;
            x_n1 = *reinterpret_cast<const float4*>(X + (long long)(pn ? cell_n1 : 0) * 4);
            const float* wn = W + (long long)kn * 192;
            w0_n1 = wn[lane]; w1_n1 = wn[64 + lane]; w2_n1 = wn[128 + lane];
            cell_n2 = grid[base + s_delta[min(k + 2, K3 - 1)]];
        }
        if (WRITE_MAP) {
            if (row_ok) nbr_out[(long long)k * nbr_out_stride + row] = present ? cell : -1;
        }
        const unsigned long long mask = __ballot(present);
        if (mask == 0ull) continue;
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            if ((mask >> (4 * p)) & 0xFull) {
                switch (p) {
#define SFW_CASE(P)                                                              \
    case P:                                                                      \
        acc[P] = __builtin_amdgcn_mfma_f32_4x4x1f32(x.x, w0, acc[P], 4, P, 0);  \
        acc[P] = __builtin_amdgcn_mfma_f32_4x4x1f32(x.y, w1, acc[P], 4, P, 0);  \
        acc[P] = __builtin_amdgcn_mfma_f32_4x4x1f32(x.z, w2, acc[P], 4, P, 0);  \
        break;
                    SFW_CASE(0) SFW_CASE(1) SFW_CASE(2) SFW_CASE(3) SFW_CASE(4) SFW_CASE(5) SFW_CASE(6) SFW_CASE(7)
                    SFW_CASE(8) SFW_CASE(9) SFW_CASE(10) SFW_CASE(11) SFW_CASE(12) SFW_CASE(13) SFW_CASE(14) SFW_CASE(15)
#undef SFW_CASE
                }
            }
        }
    }
    // register i of lane o in acc[p] = y[r0 + 4p + i][o]
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + 4 * p + i;
            if (r < n_out) Yo[(long long)r * ldy + lane] = acc[p][i];
        }
}

bool agb_stem_fwd_ok(int n_out, int K, int Cout, int ldx) {
    return Cout == 64 && ldx == 4 && n_out > 0 && K * K * K <= SFW_DELTA_MAX;
}

// Y [n_out][ldy] = bias + conv; desc = the level's lookup grid ({ox, oy, oz, X, Y, Z, ts, B | halo << 16}); nbr_out optional
int agb_stem_fwd_launch(const float* X, const float* W, const float* bias, float* Y, int ldy, int n_out, int K,
                        const int32_t* coords, const int32_t* grid, const int32_t* desc, int32_t* nbr_out,
                        long long nbr_out_stride, hipStream_t s) {
    const int K3 = K * K * K;
    const dim3 g(agb_cdiv(n_out, 256)), b(256);
    if (nbr_out)
        AGB_LAUNCH((k_stem_fwd_pairs<true>), g, b, 0, s, X, W, bias, Y, ldy, n_out, K3, (const int4*)coords, grid, desc[0], desc[1],
                   desc[2], desc[3], desc[4], desc[5], desc[6], K, nbr_out, nbr_out_stride);
    else
        AGB_LAUNCH((k_stem_fwd_pairs<false>), g, b, 0, s, X, W, bias, Y, ldy, n_out, K3, (const int4*)coords, grid, desc[0],
                   desc[1], desc[2], desc[3], desc[4], desc[5], desc[6], K, nbr_out, nbr_out_stride);
    return AGB_OK;
}

extern "C" {
// The pair-sparse stem forward by itself (same arguments as agb_spconv_fwd3_grid; Cout == 64, X rows 4 floats wide):
// what agb_spconv_fwd3_grid* dispatch to for fp32 operands; exported for A/B measurements (tools/bench_stem.py).
int agb_stem_fwd_pairs(const float* X, int ldx, const float* W, const int32_t* coords, const int32_t* grid, const int32_t* desc,
                       int K, const float* bias, float* Y, int ldy, int n_out, int Cout, int32_t* nbr_out,
                       long long nbr_out_stride, void* stream) {
    AGB_CHECK_ARG(coords && grid && desc, "agb_stem_fwd_pairs: coords, grid and desc are required");
    AGB_CHECK_ARG(K >= 1 && K <= 9 && (K & 1), "agb_stem_fwd_pairs: kernel size %d (odd, <= 9)", K);
    AGB_CHECK_ARG(agb_stem_fwd_ok(n_out, K, Cout, ldx) && ldy >= Cout, "agb_stem_fwd_pairs: takes 64 output channels and 4-float "
                  "input rows (Cout %d, ldx %d)", Cout, ldx);
    AGB_CHECK_ARG(((desc[7] >> 16) & 0xff) >= K / 2, "agb_stem_fwd_pairs: the grid's halo is narrower than K/2");
    AGB_CHECK_ARG(nbr_out == nullptr || nbr_out_stride >= n_out, "agb_stem_fwd_pairs: nbr_out_stride < n_out");
    int rc = agb_stem_fwd_launch(X, W, bias, Y, ldy, n_out, K, coords, grid, desc, nbr_out, nbr_out_stride, (hipStream_t)stream);
    if (rc) return rc;
    AGB_CHECK_LAUNCH("agb_stem_fwd_pairs");
    return AGB_OK;
}
}  // extern "C"
