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
#include <math.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SDW_R 256        // rows per chunk (dy staged in LDS: 64 KB, double-buffered)
#define SDW_OPW 11       // offsets a wave owns (accumulators: 4 VGPRs each)
#define SDW_WAVES 16

// cell of a row in the level's dense lookup grid int32[B][Z][Y][X] (origin (ox, oy, oz), tensor stride ts), as csrc/spconv.hip
// GridProbe: the neighbour of row r at kernel offset (ix, iy, iz) is grid[cell(r) + ((iz - h) Y + (iy - h)) X + (ix - h)],
// INT_MAX = none (the grid carries a halo of >= K/2 empty cells)
struct StemGrid {
    const int4* coords;
    const int32_t* grid;
    int ox, oy, oz, X, Y, Z, ts, K;
};
__device__ __forceinline__ int stem_cell(const StemGrid& g, int row) {
    const int4 c = g.coords[row];
    return ((c.x * g.Z + (c.w - g.oz) / g.ts) * g.Y + (c.z - g.oy) / g.ts) * g.X + (c.y - g.ox) / g.ts;
}
__device__ __forceinline__ int stem_delta(const StemGrid& g, int k) {
    const int h = g.K >> 1, ix = k % g.K, iy = (k / g.K) % g.K, iz = k / (g.K * g.K);
    return ((iz - h) * g.Y + (iy - h)) * g.X + (ix - h);
}

struct StemDwGeo { int nq, nchunks, nparts; };

// Which (wave, class) column owns which kernel offset, and as which of its SDW_OPW accumulators: e[k] = (column << 4) | j.
// The pair count of an offset falls with its distance from the centre (measured on the synthetic NFI plots: the centre
// offset is present for EVERY row = 6.9 x the mean, its face neighbours 1.7 x, the corners of the 7^3 cube 0.55 x), and a
// row chunk ends at a workgroup barrier: with k = (16 j + w) nq + q the wave that owned the centre carried 48 % more pairs
// than the mean.  The host deals the offsets longest-first to the least loaded column (LPT) under the isotropic model
// weight = 1 / (1 + 0.9 d)^1.6, d = distance from the centre in cells: max / mean load 1.12 on the measured counts.
// Passed to the kernel by value (1.5 KB of kernel arguments: nothing is allocated or copied).
#define SDW_MAX_K3 729      // 9^3: the tables below hold 736 entries; agb_stem_dw_ok sends larger maps to the generic kernels
struct StemOwners { unsigned short e[736]; };
static_assert(sizeof(StemOwners::e) / sizeof(unsigned short) >= SDW_MAX_K3, "owner table smaller than the largest map");

static void stem_dw_owners_build(int K3, int nq, StemOwners* own) {
    int K = 1;
    while (K * K * K < K3) ++K;
    const bool cube = K * K * K == K3;
    const int ncol = SDW_WAVES * nq, h = K >> 1;
    float wgt[736];
    int order[736];
    static_assert(sizeof(wgt) / sizeof(float) >= SDW_MAX_K3 && sizeof(order) / sizeof(int) >= SDW_MAX_K3, "K3 bound");
    for (int k = 0; k < K3; ++k) {
        float d = 0.f;
        if (cube) {
            const int dx = k % K - h, dy = (k / K) % K - h, dz = k / (K * K) - h;
            d = sqrtf((float)(dx * dx + dy * dy + dz * dz));
        }
        wgt[k] = d == 0.f ? 1.f : 1.f / powf(1.f + 0.9f * d, 1.6f);
        order[k] = k;
    }
    for (int i = 1; i < K3; ++i) {                    // stable insertion sort, weight descending (K3 <= 729)
        const int k = order[i];
        int p = i;
        while (p > 0 && wgt[order[p - 1]] < wgt[k]) { order[p] = order[p - 1]; --p; }
        order[p] = k;
    }
    float load[256];
    int cnt[256];
    for (int c = 0; c < ncol; ++c) { load[c] = 0.f; cnt[c] = 0; }
    for (int i = 0; i < K3; ++i) {
        const int k = order[i];
        int best = -1;
        for (int c = 0; c < ncol; ++c)
            if (cnt[c] < SDW_OPW && (best < 0 || load[c] < load[best])) best = c;
        own->e[k] = (unsigned short)((best << 4) | cnt[best]);
        load[best] += wgt[k];
        ++cnt[best];
    }
}

// The table depends on (K3, nq) only: built once per thread and shape, not at every launch (powf per offset, an O(K3^2) sort
// and the LPT deal are tens of microseconds of host time in a step whose enqueue is the scarce resource).
static const StemOwners& stem_dw_owners(int K3, int nq) {
    thread_local StemOwners memo;
    thread_local int memo_K3 = -1, memo_nq = -1;
    if (memo_K3 != K3 || memo_nq != nq) {
        stem_dw_owners_build(K3, nq, &memo);
        memo_K3 = K3; memo_nq = nq;
    }
    return memo;
}

static StemDwGeo stem_dw_geometry(int n_out, int K3) {
    StemDwGeo g;
    g.nq = agb_cdiv(K3, SDW_WAVES * SDW_OPW);                 // 343 offsets -> 2 classes of 176 slots
    g.nchunks = agb_cdiv(n_out > 0 ? n_out : 1, SDW_R);
    int parts = 256 / g.nq;                                   // one workgroup per CU
    if (parts < 1) parts = 1;
    g.nparts = g.nchunks < parts ? g.nchunks : parts;
    return g;
}

// (A pair-list entry packs (input row << 8) | local output row: input rows below 2^24.  The grid path reads the level it
// writes — n_in == n_out, checked here; a kernel map handed to the generic entry points must index rows below 2^24 as well:
// true for every stride-1 map of a level this check accepts, stated as a requirement in include/agb_hip.h for the rest.)
bool agb_stem_dw_ok(int n_out, int K3, int Cin, int Cout, int ldx, int ldy) {
    return Cin == 4 && Cout == 64 && ldx == 4 && ldy % 4 == 0 && n_out > 0 && n_out < (1 << 24) && K3 >= 1 &&
           K3 <= SDW_MAX_K3 && agb_cdiv(K3, SDW_WAVES * SDW_OPW) <= 16;
}

size_t agb_stem_dw_workspace_bytes(int n_out, int K3) {
    const StemDwGeo g = stem_dw_geometry(n_out, K3);
    return (size_t)g.nparts * K3 * 4 * 64 * sizeof(float);
}

// PROBE: the neighbours come from the level's dense grid (the probes of the forward kernel) instead of a kernel map: the
// 343 x N int32 map of the stem (578 MB at B = 32) is then neither written by the forward pass nor read here.
template <bool PROBE>
__global__ __launch_bounds__(1024) void k_stem_dw_pairs(const float* __restrict__ X, const float* __restrict__ dY, int ldy,
                                                        const int32_t* __restrict__ nbr, long long nbr_stride,
                                                        float* __restrict__ part, int n_out, int K3, int nq, int nchunks,
                                                        int nparts, StemGrid sg, StemOwners own) {
    __shared__ __attribute__((aligned(16))) float s_dy[2][SDW_R * 64];
    __shared__ unsigned s_list[SDW_WAVES][2][SDW_R];     // two pair lists per wave: the current step's and the next one's
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int q = blockIdx.x % nq, rp = blockIdx.x / nq;

    f32x4 acc[SDW_OPW];
#pragma unroll
    for (int j = 0; j < SDW_OPW; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- which offsets this wave owns: own.e[k] = (column << 4) | j, column = wave * nq + class (host: stem_dw_owners)
    // lane j of a wave keeps the wave's j-th offset (-1: none).  Scratch: the second dy buffer, not yet in use.
    int own_v = -1;
    {
        int* s_own = reinterpret_cast<int*>(s_dy[1]);
        if (tid < SDW_WAVES * SDW_OPW) s_own[tid] = -1;
        __syncthreads();
        if (tid < K3) {
            const int e = own.e[tid], col = e >> 4, j = e & 15;
            if (col % nq == q) s_own[(col / nq) * SDW_OPW + j] = tid;
        }
        __syncthreads();
        if (lane < SDW_OPW) own_v = s_own[w * SDW_OPW + lane];
        __syncthreads();
    }
#define SDW_OWN(J) __builtin_amdgcn_readlane(own_v, (J))

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

    // ---- software pipeline over the STEPS (chunk, owned offset j) of this wave: while step s multiplies, the neighbour
    // indices of step s + 2 are in flight, step s + 1 is compacted into the other pair list and its first gather is issued.
    // (A step's own chain index load -> compaction -> list read -> gather is two memory latencies; without the pipeline
    // every step of ~40 pairs exposed them: 562 us for the B = 32 stem, tools/bench_stem.py)
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int g4 = lane >> 2, c4 = lane & 3;
    // step descriptor: (chunk, j) -> neighbour indices of this lane's four rows of the chunk
#define SDW_CELLS(CH, C0, C1, C2, C3)                                                                               \
    do {                                                                                                            \
        C0 = C1 = C2 = C3 = 0;                                                                                       \
        if (PROBE && (CH) < nchunks) {                                                                               \
            C0 = stem_cell(sg, min((CH) * SDW_R + lane, n_out - 1));                                                 \
            C1 = stem_cell(sg, min((CH) * SDW_R + 64 + lane, n_out - 1));                                            \
            C2 = stem_cell(sg, min((CH) * SDW_R + 128 + lane, n_out - 1));                                           \
            C3 = stem_cell(sg, min((CH) * SDW_R + 192 + lane, n_out - 1));                                           \
        }                                                                                                            \
    } while (0)
#define SDW_LOAD_IDX(CH, J, C0, C1, C2, C3, I0, I1, I2, I3)                                                         \
    do {                                                                                                            \
        const int k_ = SDW_OWN(J);                                                                                   \
        const int r0_ = (CH) * SDW_R;                                                                                \
        I0 = I1 = I2 = I3 = -1;                                                                                      \
        if (k_ >= 0 && (CH) < nchunks) {                                                                             \
            if (PROBE) {                                                                                             \
                const int dk_ = stem_delta(sg, k_);                                                                  \
                const int a0_ = sg.grid[(C0) + dk_];                                                                 \
                const int a1_ = sg.grid[(C1) + dk_];                                                                 \
                const int a2_ = sg.grid[(C2) + dk_];                                                                 \
                const int a3_ = sg.grid[(C3) + dk_];                                                                 \
                I0 = (a0_ == INT_MAX || r0_ + lane >= n_out) ? -1 : a0_;                                             \
                I1 = (a1_ == INT_MAX || r0_ + 64 + lane >= n_out) ? -1 : a1_;                                        \
                I2 = (a2_ == INT_MAX || r0_ + 128 + lane >= n_out) ? -1 : a2_;                                       \
                I3 = (a3_ == INT_MAX || r0_ + 192 + lane >= n_out) ? -1 : a3_;                                       \
            } else {                                                                                                 \
                const int32_t* col_ = nbr + (long long)k_ * nbr_stride + r0_;                                        \
                if (r0_ + lane < n_out) I0 = col_[lane];                                                             \
                if (r0_ + 64 + lane < n_out) I1 = col_[64 + lane];                                                   \
                if (r0_ + 128 + lane < n_out) I2 = col_[128 + lane];                                                 \
                if (r0_ + 192 + lane < n_out) I3 = col_[192 + lane];                                                 \
            }                                                                                                        \
        }                                                                                                            \
    } while (0)
    // ordered compaction of a step's present pairs into LIST: entry = (input row << 8) | local output row; the slots up to
    // the next multiple of 16 are padded with entries of pair 0 (their A operand is zeroed when they are multiplied)
#define SDW_COMPACT1(LIST, TOTAL, IDX, S)                                                                           \
    do {                                                                                                            \
        const bool p_ = (IDX) >= 0;                                                                                  \
        const unsigned long long bal_ = __ballot(p_);                                                                \
        if (p_) (LIST)[(TOTAL) + __popcll(bal_ & lt)] = ((unsigned)(IDX) << 8) | (unsigned)(64 * (S) + lane);        \
        (TOTAL) += __popcll(bal_);                                                                                   \
    } while (0)
#define SDW_COMPACT(LIST, TOTAL, I0, I1, I2, I3)                                                                    \
    do {                                                                                                            \
        (TOTAL) = 0;                                                                                                 \
        SDW_COMPACT1(LIST, TOTAL, I0, 0); SDW_COMPACT1(LIST, TOTAL, I1, 1);                                          \
        SDW_COMPACT1(LIST, TOTAL, I2, 2); SDW_COMPACT1(LIST, TOTAL, I3, 3);                                          \
        if (lane < 16 && (TOTAL) + lane < SDW_R) (LIST)[(TOTAL) + lane] = 0u;                                        \
        __atomic_signal_fence(__ATOMIC_SEQ_CST); /* wave-private list, in-order LDS: a compiler fence is enough */  \
    } while (0)

    int chunk = rp;
    if (chunk < nchunks) {
        SDW_LOAD_CHUNK(chunk);
        SDW_STORE_CHUNK(0);
    }
    __syncthreads();
    // prologue: step 0 compacted and its first gather issued, indices of step 1 in flight
    unsigned* lcur = s_list[w][0];
    unsigned* lnxt = s_list[w][1];
    int n0, n1, n2, n3;               // neighbour indices of the step after the current one
    int total_cur;
    unsigned e_first;
    float a_first;
    // PROBE: the grid cells of this lane's four rows, of the current chunk (cc*) and of the wave's next chunk (cn*)
    int cc0, cc1, cc2, cc3, cn0, cn1, cn2, cn3;
    SDW_CELLS(chunk, cc0, cc1, cc2, cc3);
    SDW_CELLS(chunk + nparts, cn0, cn1, cn2, cn3);
    {
        int t0, t1, t2, t3;
        SDW_LOAD_IDX(chunk, 0, cc0, cc1, cc2, cc3, t0, t1, t2, t3);
        static_assert(SDW_OPW >= 3, "the step pipeline looks two offsets ahead inside one chunk");
        SDW_LOAD_IDX(chunk, 1, cc0, cc1, cc2, cc3, n0, n1, n2, n3);
        SDW_COMPACT(lcur, total_cur, t0, t1, t2, t3);
        e_first = lcur[g4];
        a_first = X[(e_first >> 8) * 4u + (unsigned)c4];
    }
    for (int it = 0; chunk < nchunks; ++it, chunk += nparts) {
        const int buf = it & 1;
        const bool more = chunk + nparts < nchunks;
        if (more) SDW_LOAD_CHUNK(chunk + nparts);                  // in flight while this chunk multiplies
        const float* dyb = s_dy[buf];
#pragma unroll
        for (int j = 0; j < SDW_OPW; ++j) {
            // (1) indices of step s + 2
            int f0, f1, f2, f3;
            if (j + 2 >= SDW_OPW) SDW_LOAD_IDX(chunk + nparts, (j + 2) % SDW_OPW, cn0, cn1, cn2, cn3, f0, f1, f2, f3);
            else SDW_LOAD_IDX(chunk, j + 2, cc0, cc1, cc2, cc3, f0, f1, f2, f3);
            // (2) step s + 1: compaction into the other list, first group's entries and gather
            int total_nxt;
            SDW_COMPACT(lnxt, total_nxt, n0, n1, n2, n3);
            const unsigned e_nf = lnxt[g4];
            const float a_nf = X[(e_nf >> 8) * 4u + (unsigned)c4];
            // (3) step s: one 4x4x1 MFMA per pair
            if (total_cur > 0) {
                const int ngroups = (total_cur + 15) >> 4;
                unsigned e = e_first;
                float a = a_first;
                f32x4 d = acc[j];
                for (int g = 0; g < ngroups; ++g) {
                    // byte offset of this lane group's dy row in the chunk buffer plus this lane's column: per pair only a
                    // readlane and an add are left (the shift and mask of the packed entry are done here, once per 16 pairs)
                    const int noff = (int)((e & 255u) << 8);
                    const float a_cur = (16 * g + g4 < total_cur) ? a : 0.f;
                    if (g + 1 < ngroups) {                          // next group's entries and gathered x in flight
                        e = lcur[16 * (g + 1) + g4];
                        a = X[(e >> 8) * 4u + (unsigned)c4];
                    }
#pragma unroll
                    for (int p = 0; p < 16; ++p) {
                        const int nb = __builtin_amdgcn_readlane(noff, 4 * p);
                        const float b = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(dyb) + nb + 4 * lane);
                        // block p of `a_cur` (lanes 4p .. 4p+3: x[m_p][0..3]) is the A operand of all 16 blocks
                        switch (p) {
#define SDW_CASE(P) case P: d = __builtin_amdgcn_mfma_f32_4x4x1f32(a_cur, b, d, 4, P, 0); break;
                            SDW_CASE(0) SDW_CASE(1) SDW_CASE(2) SDW_CASE(3) SDW_CASE(4) SDW_CASE(5) SDW_CASE(6) SDW_CASE(7)
                            SDW_CASE(8) SDW_CASE(9) SDW_CASE(10) SDW_CASE(11) SDW_CASE(12) SDW_CASE(13) SDW_CASE(14)
                            SDW_CASE(15)
#undef SDW_CASE
                        }
                    }
                }
                acc[j] = d;
            }
            // (4) rotate: s + 1 becomes the current step
            {
                unsigned* t = lcur; lcur = lnxt; lnxt = t;
            }
            total_cur = total_nxt; e_first = e_nf; a_first = a_nf;
            n0 = f0; n1 = f1; n2 = f2; n3 = f3;
        }
        if (more) SDW_STORE_CHUNK(buf ^ 1);
        cc0 = cn0; cc1 = cn1; cc2 = cn2; cc3 = cn3;
        SDW_CELLS(chunk + 2 * nparts, cn0, cn1, cn2, cn3);
        __syncthreads();
    }
#undef SDW_CELLS
#undef SDW_LOAD_IDX
#undef SDW_COMPACT
#undef SDW_COMPACT1
    // ---- partial tiles: part[rp][k][c][o], register i of lane o = dW[k][c = i][o]
    float* dst = part + (long long)rp * K3 * 256;
#pragma unroll
    for (int j = 0; j < SDW_OPW; ++j) {
        const int k = SDW_OWN(j);
        if (k < 0) continue;
#pragma unroll
        for (int c = 0; c < 4; ++c) dst[(long long)k * 256 + c * 64 + lane] = acc[j][c];
    }
#undef SDW_OWN
}

// dW[e] += sum over the row partitions of part[rp][e], in a FIXED order: the partitions are cut into four consecutive
// ranges, one per thread group, each range summed in ascending order (loads eight deep), the four range sums added in
// ascending order.  (One thread per element walking all 128 partitions: 86 workgroups and 40 us for the 45 MB; this form
// 10 us.  Round 4 had taken it back because the R2 table's bare-median outcome moved with the summation order; that outcome
// is a printed line since round 5)
__global__ __launch_bounds__(256) void k_stem_dw_fold(const float4* __restrict__ part, int nparts, long long n4,
                                                      float4* __restrict__ dW) {
    __shared__ float4 s_sum[4][64];
    const int el = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const long long e = (long long)blockIdx.x * 64 + el;
    const int per = (nparts + 3) >> 2;
    const int c0 = grp * per, c1 = min(nparts, c0 + per);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < n4) {
        int c = c0;
        for (; c + 8 <= c1; c += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(long long)(c + u) * n4 + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        for (; c < c1; ++c) {
            const float4 v = part[(long long)c * n4 + e];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    s_sum[grp][el] = s;
    __syncthreads();
    if (grp == 0 && e < n4) {
        float4 t = dW[e];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 v = s_sum[g][el];
            t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
        }
        dW[e] = t;
    }
}

// dW [K3][4][64] += gathered(X)^T dY through `workspace` (agb_stem_dw_workspace_bytes): X rows 4 floats wide (channel 3 = 0).
// Neighbours from the kernel map `nbr`, or (nbr == NULL) probed in the level's dense grid (coords, grid, desc, K).
int agb_stem_dw_launch(const float* X, const float* dY, int ldy, const int32_t* nbr, long long nbr_stride, float* dW, int n_out,
                       int K3, void* workspace, size_t workspace_bytes, hipStream_t s, const int32_t* coords,
                       const int32_t* grid, const int32_t* desc, int K) {
    const StemDwGeo g = stem_dw_geometry(n_out, K3);
    const size_t need = (size_t)g.nparts * K3 * 256 * sizeof(float);
    if (workspace == nullptr || workspace_bytes < need) {
        agb_set_error("stem weight gradient: workspace of %zu bytes, %zu needed", workspace_bytes, need);
        return AGB_EINVAL;
    }
    StemGrid sg{};
    const StemOwners& own = stem_dw_owners(K3, g.nq);
    if (nbr == nullptr) {
        sg.coords = (const int4*)coords; sg.grid = grid;
        sg.ox = desc[0]; sg.oy = desc[1]; sg.oz = desc[2]; sg.X = desc[3]; sg.Y = desc[4]; sg.Z = desc[5]; sg.ts = desc[6];
        sg.K = K;
        AGB_LAUNCH((k_stem_dw_pairs<true>), dim3(g.nparts * g.nq), dim3(1024), 0, s, X, dY, ldy, nbr, nbr_stride,
                   (float*)workspace, n_out, K3, g.nq, g.nchunks, g.nparts, sg, own);
    } else {
        AGB_LAUNCH((k_stem_dw_pairs<false>), dim3(g.nparts * g.nq), dim3(1024), 0, s, X, dY, ldy, nbr, nbr_stride,
                   (float*)workspace, n_out, K3, g.nq, g.nchunks, g.nparts, sg, own);
    }
    const long long n4 = (long long)K3 * 64;
    hipLaunchKernelGGL(k_stem_dw_fold, dim3((unsigned)agb_cdiv(n4, 64)), dim3(256), 0, s, (const float4*)workspace, g.nparts,
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

// W is staged through LDS in blocks of SFW_WB offsets (12 KB, double-buffered; the four waves of a workgroup share it, one
// barrier per block): three ds_read_b32 per offset and wave instead of three global loads.  Probes run four offsets ahead,
// gathers two (rotating registers, the offset loop unrolled by four): with one offset of look-ahead an iteration took one
// memory latency (1640 cycles against ~300 of instruction issue, 548 us for the B = 32 stem).
#define SFW_WB 16
template <bool WRITE_MAP>
__global__ __launch_bounds__(256) void k_stem_fwd_pairs(const float* __restrict__ X, const float* __restrict__ W,
                                                        const float* __restrict__ bias, float* __restrict__ Yo, int ldy,
                                                        int n_out, int K3, const int4* __restrict__ coords,
                                                        const int32_t* __restrict__ grid, int ox, int oy, int oz, int GX, int GY,
                                                        int GZ, int ts, int K, int32_t* __restrict__ nbr_out,
                                                        long long nbr_out_stride) {
    __shared__ int s_delta[SFW_DELTA_MAX + 16];
    __shared__ __attribute__((aligned(16))) float s_w[2][SFW_WB * 192];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int k = tid; k < SFW_DELTA_MAX + 16; k += 256) {
        const int kc = min(k, K3 - 1);
        const int h = K >> 1, ix = kc % K, iy = (kc / K) % K, iz = kc / (K * K);
        s_delta[k] = ((iz - h) * GY + (iy - h)) * GX + (ix - h);
    }
    // W block staging: thread t moves three float4 of the block's 3072 floats
    const long long wtotal = (long long)K3 * 192;
    float4 wst0, wst1, wst2;
#define SFW_LOAD_W(B)                                                                              \
    do {                                                                                           \
        const long long o_ = (long long)(B) * (SFW_WB * 192) + 4 * tid;                            \
        wst0 = *reinterpret_cast<const float4*>(W + min(o_, wtotal - 4));                          \
        wst1 = *reinterpret_cast<const float4*>(W + min(o_ + 1024, wtotal - 4));                   \
        wst2 = *reinterpret_cast<const float4*>(W + min(o_ + 2048, wtotal - 4));                   \
    } while (0)
#define SFW_STORE_W(BUF)                                                                           \
    do {                                                                                           \
        *reinterpret_cast<float4*>(&s_w[BUF][4 * tid]) = wst0;                                     \
        *reinterpret_cast<float4*>(&s_w[BUF][4 * tid + 1024]) = wst1;                              \
        *reinterpret_cast<float4*>(&s_w[BUF][4 * tid + 2048]) = wst2;                              \
    } while (0)
    SFW_LOAD_W(0);
    SFW_STORE_W(0);
    __syncthreads();
    const int r0 = (blockIdx.x * 4 + w) * 64;
    const bool wave_ok = r0 < n_out;                 // (idle waves of the last workgroup still stage W and meet the barriers)
    const int row = r0 + lane;
    const bool row_ok = row < n_out;
    const int base = stem_probe_base(coords[min(row, n_out - 1)], ox, oy, oz, GX, GY, GZ, ts);

    f32x4 acc[16];
    const float bv = bias ? bias[lane] : 0.f;
#pragma unroll
    for (int p = 0; p < 16; ++p) acc[p] = (f32x4){bv, bv, bv, bv};

    // rotating registers: cells of offsets k .. k + 7 (c0 = offset k when k % 8 == 0), feature rows of k and k + 1
    int c0 = grid[base + s_delta[0]], c1 = grid[base + s_delta[1]], c2 = grid[base + s_delta[2]],
        c3 = grid[base + s_delta[3]], c4 = grid[base + s_delta[4]], c5 = grid[base + s_delta[5]],
        c6 = grid[base + s_delta[6]], c7 = grid[base + s_delta[7]];
    float xa0, xa1, xa2, xb0, xb1, xb2;
#define SFW_GATHER(CELL, X0, X1, X2)                                                               \
    do {                                                                                           \
        const float* xr_ = X + (unsigned)((row_ok && (CELL) != INT_MAX) ? (CELL) : 0) * 4u;          \
        X0 = xr_[0]; X1 = xr_[1]; X2 = xr_[2];                                                      \
    } while (0)
    SFW_GATHER(c0, xa0, xa1, xa2);
    SFW_GATHER(c1, xb0, xb1, xb2);
    const int nblocks = (K3 + SFW_WB - 1) / SFW_WB;
    // one offset: CELL = its cell value; (X0, X1, X2) = its gathered feature row; afterwards CELL receives the probe of
    // offset k + 8 (a probe that misses the L2 comes from the Infinity Cache or HBM: ~1500 cycles; six offsets pass before
    // its gather is issued) and (X0, X1, X2) the gather of offset k + 2, whose cell value is CELL2
#define SFW_STEP(KK, CELL, CELL2, X0, X1, X2)                                                       \
    do {                                                                                           \
        const int k_ = kb * SFW_WB + (KK);                                                          \
        const int cell_ = CELL;                                                                     \
        const bool present_ = row_ok && cell_ != INT_MAX && k_ < K3;                                \
        const float w0_ = wb[(KK) * 192 + lane], w1_ = wb[(KK) * 192 + 64 + lane],                  \
                    w2_ = wb[(KK) * 192 + 128 + lane];                                              \
        CELL = grid[(unsigned)(base + s_delta[k_ + 8])];                                            \
        if (WRITE_MAP) {                                                                            \
            if (row_ok && k_ < K3) nbr_out[(long long)k_ * nbr_out_stride + row] = present_ ? cell_ : -1; \
        }                                                                                           \
        const float x0_ = present_ ? X0 : 0.f, x1_ = present_ ? X1 : 0.f, x2_ = present_ ? X2 : 0.f; \
        const unsigned long long mask_ = __ballot(present_);                                        \
        if (mask_ != 0ull) {                                                                        \
            SFW_GROUPS(mask_, x0_, x1_, x2_, w0_, w1_, w2_);                                        \
        }                                                                                           \
        SFW_GATHER(CELL2, X0, X1, X2);                                                              \
    } while (0)
#define SFW_GROUP(P, M, A0, A1, A2, B0, B1, B2)                                                     \
    if (((M) >> (4 * (P))) & 0xFull) {                                                              \
        acc[P] = __builtin_amdgcn_mfma_f32_4x4x1f32(A0, B0, acc[P], 4, P, 0);                       \
        acc[P] = __builtin_amdgcn_mfma_f32_4x4x1f32(A1, B1, acc[P], 4, P, 0);                       \
        acc[P] = __builtin_amdgcn_mfma_f32_4x4x1f32(A2, B2, acc[P], 4, P, 0);                       \
    }
#define SFW_GROUPS(M, A0, A1, A2, B0, B1, B2)                                                       \
    SFW_GROUP(0, M, A0, A1, A2, B0, B1, B2) SFW_GROUP(1, M, A0, A1, A2, B0, B1, B2)                 \
    SFW_GROUP(2, M, A0, A1, A2, B0, B1, B2) SFW_GROUP(3, M, A0, A1, A2, B0, B1, B2)                 \
    SFW_GROUP(4, M, A0, A1, A2, B0, B1, B2) SFW_GROUP(5, M, A0, A1, A2, B0, B1, B2)                 \
    SFW_GROUP(6, M, A0, A1, A2, B0, B1, B2) SFW_GROUP(7, M, A0, A1, A2, B0, B1, B2)                 \
    SFW_GROUP(8, M, A0, A1, A2, B0, B1, B2) SFW_GROUP(9, M, A0, A1, A2, B0, B1, B2)                 \
    SFW_GROUP(10, M, A0, A1, A2, B0, B1, B2) SFW_GROUP(11, M, A0, A1, A2, B0, B1, B2)               \
    SFW_GROUP(12, M, A0, A1, A2, B0, B1, B2) SFW_GROUP(13, M, A0, A1, A2, B0, B1, B2)               \
    SFW_GROUP(14, M, A0, A1, A2, B0, B1, B2) SFW_GROUP(15, M, A0, A1, A2, B0, B1, B2)
    for (int kb = 0; kb < nblocks; ++kb) {
        const bool more = kb + 1 < nblocks;
        if (more) SFW_LOAD_W(kb + 1);
        const float* wb = s_w[kb & 1];
        if (wave_ok) {
#pragma unroll 1
            for (int k8 = 0; k8 < SFW_WB; k8 += 8) {
                SFW_STEP(k8 + 0, c0, c2, xa0, xa1, xa2);
                SFW_STEP(k8 + 1, c1, c3, xb0, xb1, xb2);
                SFW_STEP(k8 + 2, c2, c4, xa0, xa1, xa2);
                SFW_STEP(k8 + 3, c3, c5, xb0, xb1, xb2);
                SFW_STEP(k8 + 4, c4, c6, xa0, xa1, xa2);
                SFW_STEP(k8 + 5, c5, c7, xb0, xb1, xb2);
                SFW_STEP(k8 + 6, c6, c0, xa0, xa1, xa2);
                SFW_STEP(k8 + 7, c7, c1, xb0, xb1, xb2);
            }
        }
        if (more) SFW_STORE_W((kb + 1) & 1);
        __syncthreads();
    }
#undef SFW_STEP
#undef SFW_GROUPS
#undef SFW_GROUP
#undef SFW_GATHER
#undef SFW_LOAD_W
#undef SFW_STORE_W
    if (!wave_ok) return;
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

// Weight gradient of the same stem with the neighbours probed in the level's dense grid (no kernel map): dW [K^3][4][Cout]
// fp32, accumulated into; workspace of agb_stem_bwd_weight_grid_workspace_bytes(n_out, K) bytes (row partitions folded in a
// fixed order: bitwise reproducible).
size_t agb_stem_bwd_weight_grid_workspace_bytes(int n_out, int K) { return agb_stem_dw_workspace_bytes(n_out, K * K * K); }
int agb_stem_bwd_weight_grid(const float* X, int ldx, const float* dY, int ldy, const int32_t* coords, const int32_t* grid,
                             const int32_t* desc, int K, float* dW, int n_out, int Cout, void* workspace,
                             size_t workspace_bytes, void* stream) {
    AGB_CHECK_ARG(coords && grid && desc, "agb_stem_bwd_weight_grid: coords, grid and desc are required");
    AGB_CHECK_ARG(K >= 1 && K <= 9 && (K & 1), "agb_stem_bwd_weight_grid: kernel size %d (odd, <= 9)", K);
    AGB_CHECK_ARG(n_out >= 0, "agb_stem_bwd_weight_grid: n_out %d", n_out);
    if (n_out == 0) return AGB_OK;
    AGB_CHECK_ARG(agb_stem_dw_ok(n_out, K * K * K, 4, Cout, ldx, ldy), "agb_stem_bwd_weight_grid: takes 64 output channels and "
                  "4-float input rows (Cout %d, ldx %d, ldy %d)", Cout, ldx, ldy);
    AGB_CHECK_ARG(((desc[7] >> 16) & 0xff) >= K / 2, "agb_stem_bwd_weight_grid: the grid's halo is narrower than K/2");
    int rc = agb_stem_dw_launch(X, dY, ldy, nullptr, 0, dW, n_out, K * K * K, workspace, workspace_bytes, (hipStream_t)stream,
                                coords, grid, desc, K);
    if (rc) return rc;
    AGB_CHECK_LAUNCH("agb_stem_bwd_weight_grid");
    return AGB_OK;
}
}  // extern "C"
