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
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

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
// Direct probes of the level's dense lookup grid instead of a pre-built kernel map (stride-1 odd kernels on a grid-mode
// level whose grid carries a halo of >= K/2 empty cells on every side, so a neighbour cell is always inside the grid and
// inside the plot's own block of it): neighbour of row r at offset (dx, dy, dz) = grid[cell(r) + (dz * Y + dy) * X + dx],
// INT_MAX = none.  The stem forward kernel probes while it convolves and writes the [K^3][N] map out as a by-product for
// the weight gradient: the map kernel (0.5 ms on the side stream, the same probes plus 578 MB of stores, contending
// with the compute stream) disappears.
struct GridProbe {
    const int4* coords;      // [n] (batch, x, y, z) of the level's rows
    const int32_t* grid;     // int32[B][Z][Y][X], INT_MAX = empty
    int ox, oy, oz, X, Y, Z, ts, K;
    int32_t* nbr_out;        // optional: the kernel map [K^3][nbr_out_stride] to write (rows or -1)
    long long nbr_out_stride;
};
__device__ __forceinline__ int probe_base(const GridProbe& g, int row) {
    const int4 c = g.coords[row];
    return ((c.x * g.Z + (c.w - g.oz) / g.ts) * g.Y + (c.z - g.oy) / g.ts) * g.X + (c.y - g.ox) / g.ts;
}
__device__ __forceinline__ int probe_delta(const GridProbe& g, int k) {
    const int h = g.K >> 1;
    const int ix = k % g.K, iy = (k / g.K) % g.K, iz = k / (g.K * g.K);
    return ((iz - h) * g.Y + (iy - h)) * g.X + (ix - h);
}

// ------------------------------------------------------------------------------------------------
// Three input channels (the 7^3 x 3 -> 64 stem of the NFI models), PACKED: a K-chunk of 32 holds 10 offsets x 3 channels
// (+ 2 zero rows) instead of 8 offsets x 4 padded channels: 35 chunks instead of 43 for the 343 offsets, a fifth less
// MFMA work in an MFMA-bound kernel.  X rows are still 4 floats wide (one aligned 16-B gather per pair), W is the layer's
// own [K3*3, Cout] matrix.
template <bool GRID>
__global__ __launch_bounds__(256) void k_spconv_fwd3(const float* __restrict__ X, int ldx,
                                                     const float* __restrict__ W,  // [K3*3, Cout]
                                                     const int32_t* __restrict__ nbr, long long nbr_stride, int kflip,
                                                     const float* __restrict__ bias, float* __restrict__ Y, int ldy,
                                                     int n_out, int K3, int Cout, GridProbe gp) {
    constexpr int OPC = 10;
    __shared__ __attribute__((aligned(16))) float As[BM * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[BK * LDB];
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
    // K rows 30 and 31 of the A tile stay zero for the whole kernel
    if (tid < BM) { As[tid * LDA + 30] = 0.f; As[tid * LDA + 31] = 0.f; }

    const int nchunks = (K3 + OPC - 1) / OPC;
    // Software pipeline (registers): while chunk ch multiplies, the input rows and weights of chunk ch+1 are in flight
    // and the neighbour indices of chunk ch+2 are being fetched — the dependent index -> gather chain (two memory
    // latencies per 0.43 us of MFMA work per chunk) was hidden by occupancy alone and left the MFMA pipe 54 % busy.
    // All loads are unconditional (clamped addresses) and validity is applied when a value is stored to LDS: a load
    // under a branch makes the compiler wait for it at the join.
    int idxn[3];         // neighbour index of this thread's (row, offset) pairs of the chunk to gather next
    int idxc[3];         // ... of the chunk whose rows are in xv
    f32x4 xv[3];         // gathered rows of the chunk to multiply next (kept as whole 4-register tuples, see below)
    float4 wv[2];        // weight rows of the chunk to multiply next
    const int rowc[3] = {min(row0 + ((tid + 0) & (BM - 1)), n_out - 1), min(row0 + ((tid + 256) & (BM - 1)), n_out - 1),
                         min(row0 + ((tid + 512) & (BM - 1)), n_out - 1)};
    const int wcol = min(n0 + (tid & 15) * 4, Cout - 4);
    // GRID: the three (row, offset) slots of a thread share the row (256 % 64 == 0): one cell index per thread; the cell
    // offsets of the K^3 kernel offsets are tabulated once per workgroup (three runtime divisions per probe otherwise)
    const int base = GRID ? probe_base(gp, rowc[0]) : 0;
    const bool row_ok = row0 + (tid & (BM - 1)) < n_out;
    __shared__ int s_delta[GRID ? 736 : 1];   // K <= 9: 729 offsets
    if (GRID) {
        for (int k = tid; k < K3; k += 256) s_delta[k] = probe_delta(gp, k);
        __syncthreads();
    }
    auto load_idx = [&](int ch) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int k = min(ch * OPC + ((tid + 256 * j) >> 6), K3 - 1);
            if (GRID) {
                idxn[j] = gp.grid[base + s_delta[k]];   // raw cell value (INT_MAX = none): decoded at use
            } else {
                const int kn = kflip ? (K3 - 1 - k) : k;
                idxn[j] = nbr[(long long)kn * nbr_stride + rowc[j]];
            }
        }
    };
    // rows of chunk ch (whose indices are in idxn)
    auto gather = [&](int ch) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            idxc[j] = (GRID && idxn[j] == INT_MAX) ? -1 : idxn[j];
            xv[j] = *reinterpret_cast<const f32x4*>(X + (long long)max(idxc[j], 0) * ldx);
            if (GRID && gp.nbr_out && blockIdx.y == 0) {
                // the kernel map as a by-product (64 consecutive rows per offset: coalesced 256-B stores)
                const int e = tid + 256 * j, k = ch * OPC + (e >> 6);
                if (e < BM * OPC && k < K3 && row_ok)
                    gp.nbr_out[(long long)k * gp.nbr_out_stride + row0 + (tid & (BM - 1))] = idxc[j];
            }
        }
    };
    auto load_w = [&](int ch) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long long wrow = min((long long)ch * OPC * 3 + (tid >> 4) + 16 * j, (long long)K3 * 3 - 1);
            wv[j] = *reinterpret_cast<const float4*>(W + wrow * Cout + wcol);
        }
    };
    load_idx(0);
    gather(0);
    load_w(0);
    load_idx(nchunks > 1 ? 1 : 0);
    for (int ch = 0; ch < nchunks; ++ch) {
        const int k0 = ch * OPC;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int e = tid + 256 * j;          // 64 rows x 10 offsets = 640 (row, offset) pairs per chunk
            if (e < BM * OPC) {
                const int r = e & (BM - 1), off = e >> 6;
                const bool ok = idxc[j] >= 0 && k0 + off < K3 && row0 + r < n_out;
                float* dst = &As[r * LDA + off * 3];
                // the row stays one 128-bit register tuple until here: with a 3-dword load the compiler moved the
                // components out of the tuple right behind the load (and waited for it there)
                asm volatile("" : "+v"(xv[j]));
                dst[0] = ok ? xv[j][0] : 0.f; dst[1] = ok ? xv[j][1] : 0.f; dst[2] = ok ? xv[j][2] : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int kr = (tid >> 4) + 16 * j;
            const bool ok = kr < 3 * OPC && (long long)k0 * 3 + kr < (long long)K3 * 3 && n0 + (tid & 15) * 4 < Cout;
            *reinterpret_cast<float4*>(&Bs[kr * LDB + (tid & 15) * 4]) = ok ? wv[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        // rows / weights of chunk ch+1 (its indices arrived during the previous chunk), indices of chunk ch+2; past
        // the last chunk the (clamped) loads are harmless repeats
        gather(min(ch + 1, nchunks - 1));
        load_w(min(ch + 1, nchunks - 1));
        load_idx(min(ch + 2, nchunks - 1));
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
        __syncthreads();
    }
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
// Generic path, software-pipelined: while the MFMAs of K-chunk i run, the gathers of chunk i+1 are in
// flight into registers and the neighbour indices of chunk i+2 are being fetched (the index -> gather
// dependency would otherwise expose two L2 latencies per chunk).  TM x 64 output tile:
//   TM = 64 : 2x2 waves, each 32x32            (few-row layers: keeps enough workgroups to fill 256 CUs)
//   TM = 128: 4x1 waves, each 32 rows x 64 cols (W tile reused by 128 rows: 21.8 FLOP per staged byte)
// PERM: rows come from a class-partitioned permutation (agb_parity_partition): every tile holds rows of ONE
//   lattice-parity class and visits only the kernel offsets that class can reach — the data gradient of a
//   stride-2 convolution touches 1/2/4/8 of the 27 offsets per row instead of walking a map of density 0.11.
// ksplit > 1 (gridDim.z): the offset/channel chunks are divided among gridDim.z workgroups per tile, partial
//   tiles go to `partial` [ksplit][n_out][Cout] and k_split_reduce folds them in order (deterministic); used
//   when a layer has too few rows to fill 256 CUs (2.8 k rows x 512 channels at tensor stride 16).
struct ConvArgs {
    const float* X; int ldx;
    const float* W;
    const int32_t* nbr; long long nbr_stride; int kflip;
    const float* bias;
    float* Y; int ldy;
    int n_out, K3, Cin, Cout;
    const int32_t* perm; const int32_t* tile_cls; const int32_t* cls_tab;
    int ksplit; float* partial;
    int cmp_mode;   // pair-compacted kernel: 1 = automatic, 0 = never, 64 / 128 = always with that tile height
    int cmp_il;     // log2 of the interleave block of its tiles (0 = contiguous, -1 = by level size)
    float* bn_part; // optional [row tiles][3][Cout]: (count, mean, M2) of every output column over the tile's rows —
                    // the partial statistics of the BatchNorm that follows (register-accumulator kernels, no split)
    const int32_t* tile_blocks;   // optional [tiles][rows per tile >> il]: the row blocks of every tile of the pair-compacted
                                  // kernel (work-balanced tiles, agb_spconv_balance_tiles) instead of the fixed interleave
    const float* addend; int ld_add;   // optional [n_out][ld_add]: Y = addend + (bias + sum): the gradient sum at a residual join
                                       // rides in the data-gradient kernel's final store (fp32 kernels; may alias Y)
};

// Which (row tile, column tile) a workgroup of a 2-D tile grid takes.  The hardware hands out workgroups in linear order
// (x fastest) and deals them round-robin to the eight XCDs; with the plain (blockIdx.x, blockIdx.y) reading the column
// tiles of one row tile are gridDim.x workgroups apart in time and land on any XCD: each gathers the row tile's A operand
// again from beyond its L2.  Here eight consecutive row tiles form a group whose workgroups are consecutive in dispatch
// order: linear id L -> XCD L % 8 -> row tile 8 g + L % 8, column tile (L / 8) % ny: the ny workgroups of a row tile run
// back to back on ONE XCD, whose L2 serves the gathered rows ny - 1 times.  A permutation of the tiles: results unchanged.
// (Measured: headline +0.25 % through the class-partitioned strided data gradients; the DENSE products were faster alone —
// [870 k, 128] x [128, 1024]: 2754 -> 2580 us — but slower inside the PointNet step, 11.45 -> 11.55 ms: they keep the plain order.)
__device__ __forceinline__ void xcd_tile_of(int& bx, int& by) {
    if (gridDim.y <= 1) return;
    const int nx = gridDim.x, ny = gridDim.y;
    const int L = bx + nx * by, per = 8 * ny;
    const int g = L / per, w = L - g * per;
    const int rows_here = min(8, nx - 8 * g);           // (the last group has nx % 8 row tiles)
    bx = 8 * g + w % rows_here;
    by = w / rows_here;
}

// CW: output columns per workgroup (64, or 128 for wide dense layers: the staged A tile serves twice the columns)
template <int TM, bool PERM, int CW = 64>
__global__ __launch_bounds__(256) void k_spconv_pipe(ConvArgs a) {
    constexpr int WAVES_M = TM / 32;       // 2 or 4
    constexpr int WAVES_N = 4 / WAVES_M;   // 2 or 1
    constexpr int NT = (CW / 32) / WAVES_N;   // 32-col accumulators per wave: 1, 2 or 4
    constexpr int BJ = CW / 32;               // float4 weight loads per thread per chunk
    constexpr int BT = CW / 4;                // threads per weight row
    constexpr int AJ = TM / 32;            // float4 A gathers per thread per chunk
    __shared__ __attribute__((aligned(16))) float As[TM * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[BK * CW];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    // class-partitioned tiles are taken LAST class first: the classes are ordered by parity bits, the all-odd class
    // (8 of the 27 offsets of a stride-2 3^3 kernel, 30 % of the work in an eighth of the tiles) comes last, and
    // dispatched last it ran on a mostly empty chip; heavy tiles first, the one-offset tiles fill the tail
    // (row tile, column tile) of this workgroup: XCD-aware where the A operand is gathered (xcd_tile_of)
    int bx = blockIdx.x, by = blockIdx.y;
    if (a.nbr != nullptr) xcd_tile_of(bx, by);      // (gathered A; the dense products measured better in plain order)
    const int tile = PERM ? (int)gridDim.x - 1 - bx : bx;
    const int row0 = tile * TM;
    const int n0 = by * CW;
    const int li = lane & 31, lh = lane >> 5;
    const int a_r = tid >> 3, a_c = (tid & 7) * 4;     // A: row (+32j), channel offset in chunk
    const int b_r = tid / BT, b_c = (tid % BT) * 4;    // B: k row (+ (256 / BT) j), output offset
    const int K3 = a.K3, Cin = a.Cin, Cout = a.Cout;

    int noff = K3;
    const int32_t* offs = nullptr;
    if (PERM) {
        int cls = a.tile_cls[tile];
        if (cls < 0) return;  // padding tile (uniform for the workgroup)
        noff = a.cls_tab[cls * (1 + K3)];
        offs = a.cls_tab + cls * (1 + K3) + 1;
    }

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    const int cpk = (Cin + BK - 1) / BK;
    const int nchunks = noff * cpk;
    const int ch_beg = (int)((long long)nchunks * blockIdx.z / a.ksplit);
    const int ch_end = (int)((long long)nchunks * (blockIdx.z + 1) / a.ksplit);

    int grow[AJ];  // global output row of the A rows this thread gathers for
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        int r = row0 + a_r + 32 * j;
        if (PERM) grow[j] = a.perm[r];
        else grow[j] = r < a.n_out ? r : -1;
    }

    int idx_cur[AJ], idx_nxt[AJ];
    float4 a_reg[AJ], b_reg[BJ];

    auto offset_of = [&](int ch) {
        int ko = ch / cpk;
        return PERM ? offs[ko] : ko;
    };
    auto load_idx = [&](int ch, int* dst) {
        int k = ch < ch_end ? offset_of(ch) : 0;
        int kn = a.kflip ? (K3 - 1 - k) : k;
        // nbr == nullptr: the identity map (K3 == 1): a dense [n, Cin] x [Cin, Cout] product (1x1 stride-1 convolution)
#pragma unroll
        for (int j = 0; j < AJ; ++j)
            dst[j] = (ch < ch_end && grow[j] >= 0) ? (a.nbr ? a.nbr[(long long)kn * a.nbr_stride + grow[j]] : grow[j]) : -1;
    };
    auto load_data = [&](int ch, const int* idx) {
        int k = offset_of(ch);
        int c0 = (ch % cpk) * BK;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            a_reg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx[j] >= 0 && c0 + a_c < Cin)
                a_reg[j] = *reinterpret_cast<const float4*>(a.X + (long long)idx[j] * a.ldx + c0 + a_c);
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            int kr = b_r + (256 / BT) * j;
            b_reg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c0 + kr < Cin && n0 + b_c < Cout)
                b_reg[j] = *reinterpret_cast<const float4*>(a.W + ((long long)k * Cin + c0 + kr) * Cout + n0 + b_c);
        }
    };

    if (ch_beg < ch_end) {
        load_idx(ch_beg, idx_cur);
        load_idx(ch_beg + 1, idx_nxt);
        load_data(ch_beg, idx_cur);
    }

    for (int ch = ch_beg; ch < ch_end; ++ch) {
#pragma unroll
        for (int j = 0; j < AJ; ++j)
            *reinterpret_cast<float4*>(&As[(a_r + 32 * j) * LDA + a_c]) = a_reg[j];
#pragma unroll
        for (int j = 0; j < BJ; ++j) *reinterpret_cast<float4*>(&Bs[(b_r + (256 / BT) * j) * CW + b_c]) = b_reg[j];
        __syncthreads();
        if (ch + 1 < ch_end) {
#pragma unroll
            for (int j = 0; j < AJ; ++j) idx_cur[j] = idx_nxt[j];
            load_data(ch + 1, idx_cur);   // in flight during the MFMAs below
            load_idx(ch + 2, idx_nxt);
        }
        const float* arow = &As[(wm * 32 + li) * LDA + 4 * lh];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float4 a4 = *reinterpret_cast<const float4*>(arow + 8 * t);
            const int kb = 8 * t + 4 * lh;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float* bcol = &Bs[(wn * NT + nt) * 32 + li];
                float b0 = bcol[(kb + 0) * CW];
                float b1 = bcol[(kb + 1) * CW];
                float b2 = bcol[(kb + 2) * CW];
                float b3 = bcol[(kb + 3) * CW];
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b0, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b1, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b2, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b3, acc[nt], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    float* out = a.ksplit > 1 ? a.partial + (long long)blockIdx.z * a.n_out * Cout : a.Y;
    const int ldo = a.ksplit > 1 ? Cout : a.ldy;
    const float* __restrict__ addp = a.ksplit > 1 ? nullptr : a.addend;
    // the 16 permutation entries of this lane are loaded together (a load + branch per register serialised 16 memory
    // latencies at the end of every workgroup)
    int rows[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int rt = row0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
        rows[reg] = PERM ? a.perm[rt] : (rt < a.n_out ? rt : -1);
    }
    float bvs[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int col = n0 + (wn * NT + nt) * 32 + li;
        bvs[nt] = (a.bias && a.ksplit == 1 && col < Cout) ? a.bias[col] : 0.f;
    }
    if (addp) {
        // the addend of a residual join: all of this lane's values requested before the first is used (a load, an addition
        // and a store per element — the addend may alias the output, so every load waited for the store before it —
        // cost an HBM-bound dense data gradient a third of its time: 61 -> 81 us on KPConv's 128-column layers)
        float ad[NT][16];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = n0 + (wn * NT + nt) * 32 + li;
                ad[nt][reg] = (rows[reg] >= 0 && col < Cout) ? addp[(long long)rows[reg] * a.ld_add + col] : 0.f;
            }
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = rows[reg];
            if (row < 0) continue;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = n0 + (wn * NT + nt) * 32 + li;
                if (col < Cout) out[(long long)row * ldo + col] = (acc[nt][reg] + bvs[nt]) + ad[nt][reg];
            }
        }
    } else {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = rows[reg];
            if (row < 0) continue;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = n0 + (wn * NT + nt) * 32 + li;
                if (col < Cout) out[(long long)row * ldo + col] = acc[nt][reg] + bvs[nt];
            }
        }
    }
    // Partial BatchNorm statistics of the tile while it is still in registers: the statistics pass of the BatchNorm
    // that follows (one more read of the [N, Cout] output: 4.2 GB for the 1024-wide layer of the point MLP) becomes a
    // fold of these partials.  Two-pass inside the tile (sum -> mean -> squared deviations), Chan's combine across tiles.
    if (!PERM && a.bn_part != nullptr && a.ksplit == 1) {
        float* red = As;                               // [WAVES_M][CW] (the K loop is over: the staging tiles are free)
        const int cnt_rows = min(TM, a.n_out - row0);
        float s[NT], q[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            s[nt] = 0.f;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                if (rows[reg] >= 0) s[nt] += acc[nt][reg] + bvs[nt];
            s[nt] += __shfl_xor(s[nt], 32, 64);
            if (lh == 0) red[wm * CW + (wn * NT + nt) * 32 + li] = s[nt];
        }
        __syncthreads();
        float mean[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES_M; ++w) t += red[w * CW + (wn * NT + nt) * 32 + li];
            mean[nt] = t / (float)cnt_rows;
        }
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            q[nt] = 0.f;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                if (rows[reg] >= 0) {
                    const float d = acc[nt][reg] + bvs[nt] - mean[nt];
                    q[nt] += d * d;
                }
            q[nt] += __shfl_xor(q[nt], 32, 64);
            if (lh == 0) red[wm * CW + (wn * NT + nt) * 32 + li] = q[nt];
        }
        __syncthreads();
        if (wm == 0 && lh == 0) {
            float* p = a.bn_part + (long long)tile * 3 * Cout;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = n0 + (wn * NT + nt) * 32 + li;
                if (col < Cout) {
                    float m2 = 0.f;
#pragma unroll
                    for (int w = 0; w < WAVES_M; ++w) m2 += red[w * CW + (wn * NT + nt) * 32 + li];
                    p[col] = (float)cnt_rows;
                    p[Cout + col] = mean[nt];
                    p[2 * Cout + col] = m2;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Low-precision-operand variant of k_spconv_pipe: fp32 features/weights in HBM are rounded to bf16 while being staged
// into LDS and multiplied with v_mfma_f32_32x32x16_bf16 (fp32 accumulate, 16x the fp32 MFMA rate).
//   X3 = false : plain bf16 operands (BASELINE.json config 5: "bf16 with fp32 index kernels")
//   X3 = true  : split-bf16: a = a_hi + a_lo, acc += a_hi*b_hi + a_hi*b_lo + a_lo*b_hi — relative error ~2^-16 per
//                product, i.e. fp32-level accuracy for this network (1e-4 bar) at 3/16 of the fp32 MFMA cost
// W is taken K-major: Wt[K3][Cout][Cin] (the k index contiguous) so that both MFMA operands are staged as plain 16-B
// row pieces: forward passes the transposed kernel, the data gradient passes the kernel itself.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define LBK 64        // K elements per chunk
#define LLD 72        // LDS row stride in bf16 (144 B = 36 dwords: conflict-free ds_read_b128, as LDA)

// fp32 -> bf16, round to nearest even: a plain cast compiles to v_cvt_pk_bf16_f32 on gfx950 (one instruction per PAIR of
// values instead of ~5 integer operations per value: the staging of the bf16 kernels is VALU-heavy)
__device__ __forceinline__ unsigned short f2bf(float x) {
    const __bf16 h = (__bf16)x;
    return __builtin_bit_cast(unsigned short, h);
}
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2_bf16(float a, float b) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {a, b};
    const bf16x2_t h = __builtin_convertvector(v, bf16x2_t);
    return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

template <bool X3>
__device__ __forceinline__ void stage_bf16(unsigned short* hi, unsigned short* lo, int off, float4 v) {
    const unsigned p0 = pack2_bf16(v.x, v.y), p1 = pack2_bf16(v.z, v.w);
    *reinterpret_cast<uint2*>(hi + off) = make_uint2(p0, p1);
    if (X3) {
        const float r0 = v.x - __uint_as_float(p0 << 16), r1 = v.y - __uint_as_float(p0 & 0xFFFF0000u);
        const float r2 = v.z - __uint_as_float(p1 << 16), r3 = v.w - __uint_as_float(p1 & 0xFFFF0000u);
        *reinterpret_cast<uint2*>(lo + off) = make_uint2(pack2_bf16(r0, r1), pack2_bf16(r2, r3));
    }
}

// CW: output columns per workgroup (64 or 128): a gathered / converted A tile serves CW columns
template <int TM, bool PERM, bool X3, int CW = 64>
__global__ __launch_bounds__(256) void k_spconv_pipe_bf16(ConvArgs a) {
    constexpr int WAVES_M = TM / 32;
    constexpr int WAVES_N = 4 / WAVES_M;
    constexpr int NT = (CW / 32) / WAVES_N;
    constexpr int BJ = CW / 16;            // float4 weight loads per thread per chunk
    constexpr int AJ = TM / 16;            // float4 A gathers per thread per chunk (16 rows per pass)
    constexpr int NP = X3 ? 2 : 1;
    __shared__ __attribute__((aligned(16))) unsigned short As[NP][TM * LLD];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[NP][CW * LLD];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int tile = blockIdx.x;
    const int row0 = tile * TM;
    const int n0 = blockIdx.y * CW;
    const int li = lane & 31, lh = lane >> 5;
    const int t_r = tid >> 4, t_c = (tid & 15) * 4;   // staging: row (+16j), k offset inside the chunk
    const int K3 = a.K3, Cin = a.Cin, Cout = a.Cout;

    int noff = K3;
    const int32_t* offs = nullptr;
    if (PERM) {
        int cls = a.tile_cls[tile];
        if (cls < 0) return;
        noff = a.cls_tab[cls * (1 + K3)];
        offs = a.cls_tab + cls * (1 + K3) + 1;
    }
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    const int cpk = (Cin + LBK - 1) / LBK;
    const int nchunks = noff * cpk;
    const int ch_beg = (int)((long long)nchunks * blockIdx.z / a.ksplit);
    const int ch_end = (int)((long long)nchunks * (blockIdx.z + 1) / a.ksplit);

    int grow[AJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        int r = row0 + t_r + 16 * j;
        if (PERM) grow[j] = a.perm[r];
        else grow[j] = r < a.n_out ? r : -1;
    }
    int idx_cur[AJ], idx_nxt[AJ];
    float4 a_reg[AJ], b_reg[BJ];

    auto offset_of = [&](int ch) {
        int ko = ch / cpk;
        return PERM ? offs[ko] : ko;
    };
    auto load_idx = [&](int ch, int* dst) {
        int k = ch < ch_end ? offset_of(ch) : 0;
        int kn = a.kflip ? (K3 - 1 - k) : k;
        // nbr == nullptr: the identity map (K3 == 1): a dense [n, Cin] x [Cin, Cout] product (1x1 stride-1 convolution)
#pragma unroll
        for (int j = 0; j < AJ; ++j)
            dst[j] = (ch < ch_end && grow[j] >= 0) ? (a.nbr ? a.nbr[(long long)kn * a.nbr_stride + grow[j]] : grow[j]) : -1;
    };
    auto load_data = [&](int ch, const int* idx) {
        int k = offset_of(ch);
        int c0 = (ch % cpk) * LBK;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            a_reg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx[j] >= 0 && c0 + t_c < Cin)
                a_reg[j] = *reinterpret_cast<const float4*>(a.X + (long long)idx[j] * a.ldx + c0 + t_c);
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            int n = n0 + t_r + 16 * j;
            b_reg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n < Cout && c0 + t_c < Cin)
                b_reg[j] = *reinterpret_cast<const float4*>(a.W + ((long long)k * Cout + n) * Cin + c0 + t_c);
        }
    };

    if (ch_beg < ch_end) {
        load_idx(ch_beg, idx_cur);
        load_idx(ch_beg + 1, idx_nxt);
        load_data(ch_beg, idx_cur);
    }
    for (int ch = ch_beg; ch < ch_end; ++ch) {
#pragma unroll
        for (int j = 0; j < AJ; ++j) stage_bf16<X3>(As[0], As[NP - 1], (t_r + 16 * j) * LLD + t_c, a_reg[j]);
#pragma unroll
        for (int j = 0; j < BJ; ++j) stage_bf16<X3>(Bs[0], Bs[NP - 1], (t_r + 16 * j) * LLD + t_c, b_reg[j]);
        __syncthreads();
        if (ch + 1 < ch_end) {
#pragma unroll
            for (int j = 0; j < AJ; ++j) idx_cur[j] = idx_nxt[j];
            load_data(ch + 1, idx_cur);
            load_idx(ch + 2, idx_nxt);
        }
        // fragments: lane (r = lane&31, h = lane>>5) holds A[row r][k = 16 s + 8 h + j] and B[same k][col r], j = 0..7
        const int aoff = (wm * 32 + li) * LLD + 8 * lh;
#pragma unroll
        for (int s2 = 0; s2 < LBK / 16; ++s2) {
            bf16x8 ah = *reinterpret_cast<const bf16x8*>(&As[0][aoff + 16 * s2]);
            bf16x8 al;
            if (X3) al = *reinterpret_cast<const bf16x8*>(&As[NP - 1][aoff + 16 * s2]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int boff = ((wn * NT + nt) * 32 + li) * LLD + 8 * lh + 16 * s2;
                bf16x8 bh = *reinterpret_cast<const bf16x8*>(&Bs[0][boff]);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[nt], 0, 0, 0);
                if (X3) {
                    bf16x8 bl = *reinterpret_cast<const bf16x8*>(&Bs[NP - 1][boff]);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[nt], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    float* out = a.ksplit > 1 ? a.partial + (long long)blockIdx.z * a.n_out * Cout : a.Y;
    const int ldo = a.ksplit > 1 ? Cout : a.ldy;
    // the 16 permutation entries of this lane are loaded together (a load + branch per register serialised 16 memory
    // latencies at the end of every workgroup)
    int rows[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int rt = row0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
        rows[reg] = PERM ? a.perm[rt] : (rt < a.n_out ? rt : -1);
    }
    float bvs[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int col = n0 + (wn * NT + nt) * 32 + li;
        bvs[nt] = (a.bias && a.ksplit == 1 && col < Cout) ? a.bias[col] : 0.f;
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int row = rows[reg];
        if (row < 0) continue;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = n0 + (wn * NT + nt) * 32 + li;
            if (col < Cout) out[(long long)row * ldo + col] = acc[nt][reg] + bvs[nt];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// bf16 operands FROM bf16 STORAGE: the rows of a bf16 twin of the activations (k_to_bf16 below; a.X reinterpreted,
// a.ldx in bf16 elements) and K-major bf16 weights Wt16[K3][Cout][Cin] go straight from global memory to LDS — 16-byte
// pieces of 8 channels, no conversion, half the loads and half the LDS writes of k_spconv_pipe_bf16, which gathers fp32
// rows and rounds them while staging: that kernel is bound by exactly those staging instructions and by the bytes its
// gathers pull through the CU's vector-memory path (profiles/r02_v9_senet50_bf16_bench.json: 0.06 of the bf16 MFMA peak;
// tools/gather_locality_probe.py: not by the gathers' locality).  Cin % 8 == 0.  Same tiling, fp32 accumulate and output.
// Y16: the output rows are bf16 as well (the bf16-activation mode: a.Y reinterpreted, a.ldy in bf16 elements); split
// partials stay fp32 (k_split_reduce rounds once, at the end).
template <int TM, bool PERM, int CW = 64, bool Y16 = false>
__global__ __launch_bounds__(256) void k_spconv_pipe_b16(ConvArgs a) {
    constexpr int WAVES_M = TM / 32;
    constexpr int WAVES_N = 4 / WAVES_M;
    constexpr int NT = (CW / 32) / WAVES_N;
    constexpr int BJ = CW / 32;            // 16-byte weight loads per thread per chunk (32 rows per pass)
    constexpr int AJ = TM / 32;            // 16-byte A gathers per thread per chunk
    __shared__ __attribute__((aligned(16))) unsigned short As[TM * LLD];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[CW * LLD];
    const unsigned short* __restrict__ X16 = reinterpret_cast<const unsigned short*>(a.X);
    const unsigned short* __restrict__ W16 = reinterpret_cast<const unsigned short*>(a.W);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    int bx = blockIdx.x, by = blockIdx.y;
    if (a.nbr != nullptr) xcd_tile_of(bx, by);
    const int tile = PERM ? (int)gridDim.x - 1 - bx : bx;
    const int row0 = tile * TM;
    const int n0 = by * CW;
    const int li = lane & 31, lh = lane >> 5;
    const int t_r = tid >> 3, t_c = (tid & 7) * 8;    // staging: row (+32j), k offset inside the chunk (8 channels)
    const int K3 = a.K3, Cin = a.Cin, Cout = a.Cout;

    int noff = K3;
    const int32_t* offs = nullptr;
    if (PERM) {
        int cls = a.tile_cls[tile];
        if (cls < 0) return;
        noff = a.cls_tab[cls * (1 + K3)];
        offs = a.cls_tab + cls * (1 + K3) + 1;
    }
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    const int cpk = (Cin + LBK - 1) / LBK;
    const int nchunks = noff * cpk;
    const int ch_beg = (int)((long long)nchunks * blockIdx.z / a.ksplit);
    const int ch_end = (int)((long long)nchunks * (blockIdx.z + 1) / a.ksplit);

    int grow[AJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        int r = row0 + t_r + 32 * j;
        if (PERM) grow[j] = a.perm[r];
        else grow[j] = r < a.n_out ? r : -1;
    }
    int idx_cur[AJ], idx_nxt[AJ];
    uint4 a_reg[AJ], b_reg[BJ];

    auto offset_of = [&](int ch) {
        int ko = ch / cpk;
        return PERM ? offs[ko] : ko;
    };
    auto load_idx = [&](int ch, int* dst) {
        int k = ch < ch_end ? offset_of(ch) : 0;
        int kn = a.kflip ? (K3 - 1 - k) : k;
#pragma unroll
        for (int j = 0; j < AJ; ++j)
            dst[j] = (ch < ch_end && grow[j] >= 0) ? (a.nbr ? a.nbr[(long long)kn * a.nbr_stride + grow[j]] : grow[j]) : -1;
    };
    auto load_data = [&](int ch, const int* idx) {
        int k = offset_of(ch);
        int c0 = (ch % cpk) * LBK;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            a_reg[j] = make_uint4(0u, 0u, 0u, 0u);
            if (idx[j] >= 0 && c0 + t_c < Cin)
                a_reg[j] = *reinterpret_cast<const uint4*>(X16 + (long long)idx[j] * a.ldx + c0 + t_c);
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            int n = n0 + t_r + 32 * j;
            b_reg[j] = make_uint4(0u, 0u, 0u, 0u);
            if (n < Cout && c0 + t_c < Cin)
                b_reg[j] = *reinterpret_cast<const uint4*>(W16 + ((long long)k * Cout + n) * Cin + c0 + t_c);
        }
    };

    if (ch_beg < ch_end) {
        load_idx(ch_beg, idx_cur);
        load_idx(ch_beg + 1, idx_nxt);
        load_data(ch_beg, idx_cur);
    }
    for (int ch = ch_beg; ch < ch_end; ++ch) {
#pragma unroll
        for (int j = 0; j < AJ; ++j) *reinterpret_cast<uint4*>(&As[(t_r + 32 * j) * LLD + t_c]) = a_reg[j];
#pragma unroll
        for (int j = 0; j < BJ; ++j) *reinterpret_cast<uint4*>(&Bs[(t_r + 32 * j) * LLD + t_c]) = b_reg[j];
        __syncthreads();
        if (ch + 1 < ch_end) {
#pragma unroll
            for (int j = 0; j < AJ; ++j) idx_cur[j] = idx_nxt[j];
            load_data(ch + 1, idx_cur);
            load_idx(ch + 2, idx_nxt);
        }
        const int aoff = (wm * 32 + li) * LLD + 8 * lh;
#pragma unroll
        for (int s2 = 0; s2 < LBK / 16; ++s2) {
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&As[aoff + 16 * s2]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int boff = ((wn * NT + nt) * 32 + li) * LLD + 8 * lh + 16 * s2;
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&Bs[boff]);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[nt], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    float* out = a.ksplit > 1 ? a.partial + (long long)blockIdx.z * a.n_out * Cout : a.Y;
    const int ldo = a.ksplit > 1 ? Cout : a.ldy;
    int rows[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int rt = row0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
        rows[reg] = PERM ? a.perm[rt] : (rt < a.n_out ? rt : -1);
    }
    float bvs[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int col = n0 + (wn * NT + nt) * 32 + li;
        bvs[nt] = (a.bias && a.ksplit == 1 && col < Cout) ? a.bias[col] : 0.f;
    }
    if (Y16 && a.ksplit == 1) {
        // two rows per step: lanes 2j / 2j+1 swap one value each (DPP quad permutation), the even lane then holds columns
        // (li, li+1) of the first row, the odd lane columns (li-1, li) of the second: one 4-byte store of a bf16 pair per
        // lane instead of two 2-byte stores (Cout % 4 == 0 and n0, 32 nt even: a pair never straddles the matrix edge)
        bf16_t* __restrict__ Y16p = reinterpret_cast<bf16_t*>(a.Y);
        const bool odd = li & 1;
        int prow[8];
#pragma unroll
        for (int h2 = 0; h2 < 8; ++h2) {
            const int reg = 2 * h2;
            const int rt = row0 + wm * 32 + (reg & 3) + (int)odd + 8 * (reg >> 2) + 4 * lh;
            prow[h2] = PERM ? a.perm[rt] : (rt < a.n_out ? rt : -1);
        }
        // addend of a residual join (bf16 rows [n_out][ld_add]; data gradients: no bias): the pair at every store position of
        // this lane, all requested before the first is used, added in fp32 before the one rounding
        const bf16_t* __restrict__ ad16 = reinterpret_cast<const bf16_t*>(a.addend);
        unsigned adw[8][NT];
        if (ad16) {
#pragma unroll
            for (int h2 = 0; h2 < 8; ++h2) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int col = n0 + (wn * NT + nt) * 32 + (li & ~1);
                    adw[h2][nt] = (prow[h2] >= 0 && col < Cout)
                        ? *reinterpret_cast<const unsigned*>(ad16 + (long long)prow[h2] * a.ld_add + col) : 0u;
                }
            }
        }
#pragma unroll
        for (int h2 = 0; h2 < 8; ++h2) {
            const int row = prow[h2];
            unsigned w[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                w[nt] = ad16 ? agb_bf16_pair_rows_add(acc[nt][2 * h2] + bvs[nt], acc[nt][2 * h2 + 1] + bvs[nt], odd, adw[h2][nt])
                             : agb_bf16_pair_rows(acc[nt][2 * h2] + bvs[nt], acc[nt][2 * h2 + 1] + bvs[nt], odd);
            if (row >= 0) {
                bf16_t* yrow = Y16p + (long long)row * a.ldy + n0 + wn * NT * 32 + (li & ~1);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)     // (Cout % 4 == 0, the pair's first column even: both inside or both outside)
                    if (n0 + (wn * NT + nt) * 32 + (li & ~1) < Cout) *reinterpret_cast<unsigned*>(yrow + nt * 32) = w[nt];
            }
        }
        return;
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int row = rows[reg];
        if (row < 0) continue;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = n0 + (wn * NT + nt) * 32 + li;
            if (col < Cout) out[(long long)row * ldo + col] = acc[nt][reg] + bvs[nt];
        }
    }
}

// fp32 [n, C] (row stride ldx) -> bf16 [n, C] (row stride ldy), round to nearest even: the bf16 twin of an activation /
// gradient / weight matrix.  C % 4 == 0.
__global__ __launch_bounds__(256) void k_to_bf16(const float* __restrict__ X, long long ldx, long long n, int C4,
                                                 unsigned short* __restrict__ Y, long long ldy) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n * C4) return;
    const long long r = t / C4;
    const int c = (int)(t % C4) * 4;
    const float4 v = *reinterpret_cast<const float4*>(X + r * ldx + c);
    *reinterpret_cast<uint2*>(Y + r * ldy + c) = make_uint2(pack2_bf16(v.x, v.y), pack2_bf16(v.z, v.w));
}

// ------------------------------------------------------------------------------------------------
// PAIR-COMPACTED output-stationary kernel (Cin a multiple of 64).
// The register-accumulator kernels above multiply zero rows wherever a neighbour is absent (3^3 maps are 50-70 %
// dense).  Here every WAVE (= workgroup) owns up to R consecutive output rows x 64 output channels whose running sums
// live in LDS (wave-private: no barriers, no atomics, fixed offset / group order -> deterministic).  Per kernel
// offset the wave compacts its neighbour indices (ballot + prefix) into a pair list and multiplies only the present
// pairs, 16 at a time, on v_mfma_f32_16x16x4_f32 (exact fp32, same peak as 32x32x2):
//   A fragment  lane (m, q) = row m of the group, channels 4q..4q+3 of each 16-channel block: one 16-B gather
//               straight from global memory into registers, prefetched one group ahead (across offsets too);
//   B fragment  MFMA column j of accumulator ct stands for output column 4j+ct, so lane m reads W[k][c][4m..4m+3]
//               as ONE float4 and the 16x64 product block leaves as float4 read-add-writes on the LDS rows;
//               the fragments of a 64-channel step stay in registers for all groups, those of the next step are
//               loaded while the current one multiplies (two register sets, loop unrolled by two).
// MFMA work = pairs rounded up to 16 per (wave, offset) instead of the tile height per offset.
// 1-D grid, XCD-aware: every XCD (workgroup id % 8) walks one contiguous range of row tiles, so the gathers of
// neighbouring tiles share that XCD's L2.  rows_per_tile <= R is chosen by the host so that the number of workgroups is
// a multiple of the resident-wave capacity (tiles of equal cost: no half-empty last round).
// -DAGB_TIMELINE (make TIMELINE=1; tools/wave_timeline.py): every wave of k_spconv_cmp leaves its start / end time
// (s_memrealtime, 100 MHz), its s_memtime ticks (shader clocks) and its 16-pair group count in g_cmp_timeline — the
// measurement behind DESIGN.md section 5's clock x slots x in-wave decomposition.  Not part of the product build.
#ifdef AGB_TIMELINE
#define AGB_TIMELINE_SLOTS 8192
__device__ unsigned long long g_cmp_timeline[4 * AGB_TIMELINE_SLOTS];
#endif
#define CMP_YS 68   // LDS row stride of the running sums (floats): 16-B aligned rows, 4-bank skew per row
#define CMP_CB 4    // 16-channel blocks per step (64 input channels)

// il_shift > 0: INTERLEAVED tiles — a tile is rows_per_tile / 2^il_shift blocks of 2^il_shift consecutive rows, block j of
// tile t being global block j * ntiles + t.  Pair density varies by region (tile work: std 19 % of the mean on the NFI
// plots, and with ~2 tiles per resident wave slot the slowest slots set the kernel time: 796 of 1024 slots busy on
// average); a tile that samples several regions has a third of that spread.  Row blocks keep the 32/64-byte coalescing
// of the map reads and the x+-1 neighbour reuse; consecutive tiles still walk every region consecutively (L2).
template <int R>
__global__ __launch_bounds__(64) void k_spconv_cmp(ConvArgs a, int ntiles, int nct, int rows_per_tile, int csplit,
                                                   int il_shift) {
    constexpr int NJ = R / 64;
    constexpr int CB = CMP_CB;
    // row R of Ys is a sink: list padding (up to 15 entries per offset) multiplies input row 0 into it
    __shared__ __attribute__((aligned(16))) float Ys[(R + 1) * CMP_YS];
    __shared__ __attribute__((aligned(16))) int pl_in[2][R + 16];
    __shared__ __attribute__((aligned(16))) int pl_out[2][R + 16];
    __shared__ int s_row[R];      // row of the level behind every local row of the tile (-1: none)
    const int lane = threadIdx.x;
    const int m = lane & 15, q = lane >> 4;
    // workgroup id -> (XCD, row tile, input-channel split, column tile); csplit > 1: the 64-channel steps of every
    // offset are divided among csplit workgroups per tile, partial tiles go to a.partial [csplit][n_out][Cout] and
    // k_split_reduce folds them in order (few-row, wide layers: 2.9 k rows x 512 channels would otherwise give 184 waves)
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int per_xcd = (ntiles + 7) >> 3;
    const int ct0 = jx % nct;
    const int sp = (jx / nct) % csplit;
    const int tile = xcd * per_xcd + jx / (nct * csplit);
    if (tile >= ntiles || jx / (nct * csplit) >= per_xcd) return;
    const int row0 = tile * rows_per_tile;
    const int row_end = min(a.n_out, row0 + rows_per_tile);
    // local row of the tile -> row of the level (-1: past the end)
    auto grow = [&](int rl) -> int {
        if (il_shift == 0) return row0 + rl < row_end ? row0 + rl : -1;
        if (rl >= rows_per_tile) return -1;
        const int blk = a.tile_blocks ? a.tile_blocks[tile * (rows_per_tile >> il_shift) + (rl >> il_shift)]
                                      : (rl >> il_shift) * ntiles + tile;
        const int r = (blk << il_shift) | (rl & ((1 << il_shift) - 1));
        return (blk >= 0 && r < a.n_out) ? r : -1;
    };
    int myrow[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        myrow[j] = grow(64 * j + lane);
        s_row[64 * j + lane] = myrow[j];
    }
    const int n0 = ct0 * 64;
    const int K3 = a.K3, Cin = a.Cin, Cout = a.Cout;
    const int NSB = Cin / (CB * 16) / csplit;            // 64-channel steps per offset handled here
    const int c_first = sp * NSB * (CB * 16);            // first input channel of this split
    const int nsteps = K3 * NSB;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const bool colok = (n0 + 4 * m) < Cout;
    // clamped column: the weight loads are unconditional; columns past Cout hold sums that the epilogue never stores
    const int coff = colok ? 4 * m : 0;

#ifdef AGB_TIMELINE
    const unsigned long long tl_t0 = wall_clock64(), tl_c0 = __builtin_readcyclecounter();
    unsigned long long tl_groups = 0;
#endif
    for (int i = lane; i < (R + 1) * CMP_YS / 4; i += 64)
        reinterpret_cast<float4*>(Ys)[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    int nv[NJ];
    auto load_nbr = [&](int k) {
        const int kn = a.kflip ? (K3 - 1 - k) : k;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int r = myrow[j];
            nv[j] = r >= 0 ? a.nbr[(long long)kn * a.nbr_stride + r] : -1;
        }
    };
    // pairs of offset k (whose neighbour indices are in nv) -> list k&1, padded to a multiple of 16; returns groups
    auto compact = [&](int k) {
        int* li = pl_in[k & 1];
        int* lo = pl_out[k & 1];
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const bool p = nv[j] >= 0;
            const unsigned long long bal = __ballot(p);
            if (p) {
                int pos = cnt + __popcll(bal & lt_mask);
                li[pos] = nv[j];
                lo[pos] = 64 * j + lane;
            }
            cnt += __popcll(bal);
        }
        const int cpad = (cnt + 15) & ~15;
        if (lane < cpad - cnt) {
            li[cnt + lane] = 0;
            lo[cnt + lane] = R;
        }
        // The lists are wave-private and one wave's LDS operations execute in order: a compiler barrier is all that is
        // needed.  (A __threadfence_block() here also waits for every outstanding global load — the refills in
        // flight — once per step.)
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        return cpad >> 4;
    };
    auto load_b = [&](int step, float4 (&b)[CB][4]) {
        const int k = step / NSB, cbase = c_first + (step % NSB) * (CB * 16);
        const float* Wk = a.W + ((long long)k * Cin + cbase + 4 * q) * Cout + n0 + coff;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                b[cb][s2] = *reinterpret_cast<const float4*>(Wk + (cb * 16 + s2) * Cout);
            }
    };
    float4 an[CB];
    // A fragment of one 16-channel block of group g of `step` (list padding points at input row 0)
    auto gather_cb = [&](int step, int g, int cb) {
        const int k = step / NSB, cbase = c_first + (step % NSB) * (CB * 16);
        const float* xr = a.X + (long long)pl_in[k & 1][16 * g + m] * a.ldx + cbase + 4 * q;
        an[cb] = *reinterpret_cast<const float4*>(xr + cb * 16);
    };
    auto gather = [&](int step, int g) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) gather_cb(step, g, cb);
    };

    int ng_cur, ng_nxt = 0;
    // one step = 64 input channels of one offset.  bcur: this step's weights; bnxt: filled for the next step
    auto body = [&](int step, float4 (&bcur)[CB][4], float4 (&bnxt)[CB][4]) {
        const int k = step / NSB;
        const int nstep = step + 1;
        if (nstep < nsteps) {
            const int nk = nstep / NSB;
            if (nk != k) {
                ng_nxt = compact(nk);
                if (nk + 1 < K3) load_nbr(nk + 1);   // in flight during this offset's MFMAs
            } else {
                ng_nxt = ng_cur;
            }
            load_b(nstep, bnxt);
        } else {
            ng_nxt = 0;
        }
        const int* lo = pl_out[k & 1];
        if (ng_cur == 0 && ng_nxt > 0) gather(nstep, 0);
        // list / channel offset of this step and of the next one (for the refills)
        const int* li_cur = pl_in[k & 1];
        const int* li_nxt = pl_in[(nstep / NSB) & 1];
        const int xoff_cur = c_first + (step % NSB) * (CB * 16) + 4 * q;
        const int xoff_nxt = c_first + (nstep % NSB) * (CB * 16) + 4 * q;

#ifdef AGB_TIMELINE
        tl_groups += ng_cur;
#endif
        for (int g = 0; g < ng_cur; ++g) {
            // D layout: MFMA col = lane & 15 (-> output column 4m + ct), row = 4 * (lane >> 4) + v.  The four rows of a
            // lane group are distinct (a row occurs once per offset) or the sink row.
            const int4 o4 = *reinterpret_cast<const int4*>(&lo[16 * g + 4 * q]);
            const int ov[4] = {o4.x, o4.y, o4.z, o4.w};
            float4 yv[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) yv[v] = *reinterpret_cast<const float4*>(&Ys[ov[v] * CMP_YS + 4 * m]);
            // where the registers of a 16-channel block are refilled from once its MFMAs are issued: the next group of
            // this step, or the first group of the next step (three quarters of a group of MFMAs ahead of use); with
            // no next group at all, group 0 of this step is re-read: the loads stay unconditional so that the
            // compiler's wait counts are exact instead of "everything outstanding"
            const bool last = g + 1 == ng_cur;
            const bool tonext = last && ng_nxt > 0;
            const int pg = last ? 0 : g + 1;
            const float* xnext = a.X + (long long)(tonext ? li_nxt : li_cur)[16 * pg + m] * a.ldx +
                                 (tonext ? xoff_nxt : xoff_cur);
            f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const float av[4] = {an[cb].x, an[cb].y, an[cb].z, an[cb].w};
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s2], bcur[cb][s2].x, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s2], bcur[cb][s2].y, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s2], bcur[cb][s2].z, c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s2], bcur[cb][s2].w, c3, 0, 0, 0);
                }
                // Pin the refill of this block right behind its MFMAs: left alone, instruction selection and the
                // scheduler sink all four loads to the end of the group, where their latency is exposed.  The empty
                // asm ties the accumulators (the MFMAs stay above it) and clobbers memory (the load stays between).
                asm volatile("" : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3) : : "memory");
                const float4 ld = *reinterpret_cast<const float4*>(xnext + cb * 16);
                asm volatile("" : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3) : : "memory");
                an[cb] = ld;
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                float4 y = yv[v];
                y.x += c0[v]; y.y += c1[v]; y.z += c2[v]; y.w += c3[v];
                *reinterpret_cast<float4*>(&Ys[ov[v] * CMP_YS + 4 * m]) = y;
            }
        }
        ng_cur = ng_nxt;
    };

    float4 bA[CB][4], bB[CB][4];
    load_nbr(0);
    ng_cur = compact(0);
    if (K3 > 1) load_nbr(1);
    load_b(0, bA);
    if (ng_cur > 0) gather(0, 0);
    for (int step = 0; step < nsteps; step += 2) {
        body(step, bA, bB);
        if (step + 1 < nsteps) body(step + 1, bB, bA);
    }
    __threadfence_block();
    // ---- epilogue: tile -> global (4 rows per instruction, float4 per lane)
    const int c4 = (lane & 15) * 4;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.bias && csplit == 1 && n0 + c4 < Cout) bv = *reinterpret_cast<const float4*>(a.bias + n0 + c4);
    float* out = csplit > 1 ? a.partial + (long long)sp * a.n_out * Cout : a.Y;
    const int ldo = csplit > 1 ? Cout : a.ldy;
    // eight row quads per trip: the LDS reads of a trip are issued together (one exposed LDS latency per 32 rows instead
    // of one per 4: the rolled loop was ~3 % of a wave's life)
    const bool col_ok = n0 + c4 < Cout;
    for (int r0 = lane >> 4; r0 < rows_per_tile; r0 += 32) {
        int rows[8];
        float4 ys[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = r0 + 4 * u;
            rows[u] = r < rows_per_tile ? s_row[r] : -1;
            ys[u] = *reinterpret_cast<const float4*>(&Ys[min(r, R) * CMP_YS + c4]);
        }
        // the addend's eight row pieces are requested together, before the first store (the addend may alias the output:
        // a load placed after a store has to wait for it)
        float4 ads[8];
        const bool with_add = a.addend && csplit == 1;
        if (with_add) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                ads[u] = (rows[u] >= 0 && col_ok)
                    ? *reinterpret_cast<const float4*>(a.addend + (long long)rows[u] * a.ld_add + n0 + c4)
                    : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (rows[u] >= 0 && col_ok) {
                float4 y = ys[u];
                y.x += bv.x; y.y += bv.y; y.z += bv.z; y.w += bv.w;
                if (with_add) { y.x += ads[u].x; y.y += ads[u].y; y.z += ads[u].z; y.w += ads[u].w; }
                *reinterpret_cast<float4*>(out + (long long)rows[u] * ldo + n0 + c4) = y;
            }
        }
    }
#ifdef AGB_TIMELINE
    if (lane == 0 && blockIdx.x < AGB_TIMELINE_SLOTS) {
        unsigned long long* e = g_cmp_timeline + 4 * blockIdx.x;
        e[0] = tl_t0; e[1] = wall_clock64(); e[2] = __builtin_readcyclecounter() - tl_c0; e[3] = tl_groups;
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// k_spconv_cmpt: the pair-compacted kernel with the MFMA OPERAND ROLES SWAPPED (round 5).
// k_spconv_cmp multiplies (gathered rows) x (weights): a lane then holds 4 PAIRS x 1 column per accumulator register
// quad, the running sums of its four rows are four separate LDS reads, and the 16 x 64 product block has to be moved out of
// the accumulators (16 v_accvgpr_read), added (8 v_pk_add_f32) and written back after the LAST MFMA of the group — a chain of
// ~250 clocks with an empty matrix pipe (one wave per SIMD: nothing else issues), plus an LDS round trip in front of it.
// Here the weights are the A operand and the gathered rows the B operand: D = W[k]^T X^T, so lane (j, q) holds, for
// pair j of the group, accumulator ct = output columns 16q + 4v + ct (v = 0..3) of that pair's row.  The tile in LDS keeps
// those four sums adjacent (position 16q + 4ct + v inside a row: a 4 x 4 transposition of every 16-column block, undone
// by the tile's final store), so
//   * ONE ds_read_b128 per accumulator fetches the running sums straight into the MFMA's C operand — the products are
//     accumulated ON TOP of the sums: no zero-init, no accumulator reads, no adds;
//   * ONE ds_write_b128 per accumulator writes them back;
//   * the sums of group g + 1 are fetched while group g multiplies (same offset: no row occurs twice, and one wave's LDS
//     operations execute in program order), so a group neither starts nor ends on an LDS round trip.
// Same pair lists, same gathers, same weight loads as k_spconv_cmp.  A row's sum is accumulated in the order
// (offset, 64-channel step, 4-channel MFMA), each MFMA adding its 4 products to the running sum: deterministic, but not the
// rounding order of k_spconv_cmp (which sums a 64-channel step from zero first).
#ifdef AGB_TIMELINE
__device__ unsigned long long g_cmpt_stamp[16];   // stamp sums of the diagnostic build of k_spconv_cma
#endif
template <int R>
__global__ __launch_bounds__(64) void k_spconv_cmpt(ConvArgs a, int ntiles, int nct, int rows_per_tile, int csplit,
                                                    int il_shift) {
    constexpr int NJ = R / 64;
    constexpr int CB = CMP_CB;
    __shared__ __attribute__((aligned(16))) float Ys[(R + 1) * CMP_YS];   // row R: sink of the list padding
    __shared__ __attribute__((aligned(16))) int pl_in[2][R + 16];
    __shared__ __attribute__((aligned(16))) int pl_out[2][R + 16];
    __shared__ int s_row[R];
    const int lane = threadIdx.x;
    const int m = lane & 15, q = lane >> 4;
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int per_xcd = (ntiles + 7) >> 3;
    const int ct0 = jx % nct;
    const int sp = (jx / nct) % csplit;
    const int tile = xcd * per_xcd + jx / (nct * csplit);
    if (tile >= ntiles || jx / (nct * csplit) >= per_xcd) return;
    const int row0 = tile * rows_per_tile;
    const int row_end = min(a.n_out, row0 + rows_per_tile);
    auto grow = [&](int rl) -> int {
        if (il_shift == 0) return row0 + rl < row_end ? row0 + rl : -1;
        if (rl >= rows_per_tile) return -1;
        const int blk = a.tile_blocks ? a.tile_blocks[tile * (rows_per_tile >> il_shift) + (rl >> il_shift)]
                                      : (rl >> il_shift) * ntiles + tile;
        const int r = (blk << il_shift) | (rl & ((1 << il_shift) - 1));
        return (blk >= 0 && r < a.n_out) ? r : -1;
    };
    int myrow[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        myrow[j] = grow(64 * j + lane);
        s_row[64 * j + lane] = myrow[j];
    }
    const int n0 = ct0 * 64;
    const int K3 = a.K3, Cin = a.Cin, Cout = a.Cout;
    const int NSB = Cin / (CB * 16) / csplit;
    const int c_first = sp * NSB * (CB * 16);
    const int nsteps = K3 * NSB;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const bool colok = (n0 + 4 * m) < Cout;
    const int coff = colok ? 4 * m : 0;

#ifdef AGB_TIMELINE
    const unsigned long long tl_t0 = wall_clock64(), tl_c0 = __builtin_readcyclecounter();
    unsigned long long tl_groups = 0;
#endif
    for (int i = lane; i < (R + 1) * CMP_YS / 4; i += 64)
        reinterpret_cast<float4*>(Ys)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    // every list entry is a valid (input row, local row) at any time: the sums of a step's first group are fetched
    // unconditionally, also when the step has no group
    for (int i = lane; i < 2 * (R + 16); i += 64) {
        (&pl_in[0][0])[i] = 0;
        (&pl_out[0][0])[i] = R;
    }

    int nv[NJ];
    auto load_nbr = [&](int k) {
        const int kn = a.kflip ? (K3 - 1 - k) : k;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int r = myrow[j];
            nv[j] = r >= 0 ? a.nbr[(long long)kn * a.nbr_stride + r] : -1;
        }
    };
    auto compact = [&](int k) {
        int* li = pl_in[k & 1];
        int* lo = pl_out[k & 1];
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const bool p = nv[j] >= 0;
            const unsigned long long bal = __ballot(p);
            if (p) {
                int pos = cnt + __popcll(bal & lt_mask);
                li[pos] = nv[j];
                lo[pos] = 64 * j + lane;
            }
            cnt += __popcll(bal);
        }
        const int cpad = (cnt + 15) & ~15;
        if (lane < cpad - cnt) {
            li[cnt + lane] = 0;
            lo[cnt + lane] = R;
        }
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        return cpad >> 4;
    };
    // weights of one step = the MFMA's A operand: lane (m, q) holds W[k][c0 + 16 cb + 4 q + s2][n0 + 4 m + ct]
    auto load_b = [&](int step, float4 (&b)[CB][4]) {
        const int k = step / NSB, cbase = c_first + (step % NSB) * (CB * 16);
        const float* Wk = a.W + ((long long)k * Cin + cbase + 4 * q) * Cout + n0 + coff;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) b[cb][s2] = *reinterpret_cast<const float4*>(Wk + (cb * 16 + s2) * Cout);
    };
    float4 an[CB];   // gathered row pieces of the current group = the B operand: lane (j, q): X[in_j][c0 + 16 cb + 4 q ..]
    auto gather = [&](int step, int g) {
        const int k = step / NSB, cbase = c_first + (step % NSB) * (CB * 16);
        const float* xr = a.X + (long long)pl_in[k & 1][16 * g + m] * a.ldx + cbase + 4 * q;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) an[cb] = *reinterpret_cast<const float4*>(xr + cb * 16);
    };

    f32x4* const Ys4 = reinterpret_cast<f32x4*>(Ys);
    f32x4 cin[4];   // running sums of the group about to be multiplied (accumulator ct: 4 adjacent floats of its row)
    int ng_cur, ng_nxt = 0;
    auto body = [&](int step, float4 (&bcur)[CB][4], float4 (&bnxt)[CB][4]) {
        const int k = step / NSB;
        const int nstep = step + 1;
        const int* lo = pl_out[k & 1];
        // sums of the step's first group: behind every earlier write in program order (the previous step may have
        // written the same rows); their latency is covered by the list work and the weight loads below
        int addr_cur = lo[m] * (CMP_YS / 4) + 4 * q;   // in 16-byte units
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) cin[ct] = Ys4[addr_cur + ct];
        if (nstep < nsteps) {
            const int nk = nstep / NSB;
            if (nk != k) {
                ng_nxt = compact(nk);
                if (nk + 1 < K3) load_nbr(nk + 1);
            } else {
                ng_nxt = ng_cur;
            }
            load_b(nstep, bnxt);
        } else {
            ng_nxt = 0;
        }
        if (ng_cur == 0 && ng_nxt > 0) gather(nstep, 0);
        const int* li_cur = pl_in[k & 1];
        const int* li_nxt = pl_in[(nstep / NSB) & 1];
        const int xoff_cur = c_first + (step % NSB) * (CB * 16) + 4 * q;
        const int xoff_nxt = c_first + (nstep % NSB) * (CB * 16) + 4 * q;
#ifdef AGB_TIMELINE
        tl_groups += ng_cur;
#endif
        for (int g = 0; g < ng_cur; ++g) {
            const bool last = g + 1 == ng_cur;
            const bool tonext = last && ng_nxt > 0;
            const int pg = last ? 0 : g + 1;
            // the next group of this step (the last group re-reads group 0: unconditional loads, exact wait counts; the
            // next step fetches its own first sums behind this group's write)
            const int addr_nxt = lo[16 * pg + m] * (CMP_YS / 4) + 4 * q;
            const float* xnext = a.X + (long long)(tonext ? li_nxt : li_cur)[16 * pg + m] * a.ldx +
                                 (tonext ? xoff_nxt : xoff_cur);
            f32x4 c0 = cin[0], c1 = cin[1], c2 = cin[2], c3 = cin[3];
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const float av[4] = {an[cb].x, an[cb].y, an[cb].z, an[cb].w};
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(bcur[cb][s2].x, av[s2], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(bcur[cb][s2].y, av[s2], c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(bcur[cb][s2].z, av[s2], c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(bcur[cb][s2].w, av[s2], c3, 0, 0, 0);
                }
                asm volatile("" : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3) : : "memory");
                const float4 ld = *reinterpret_cast<const float4*>(xnext + cb * 16);
                if (cb == 1) {
                    // sums of the next group, two thirds of a group ahead of their use
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) cin[ct] = Ys4[addr_nxt + ct];
                }
                asm volatile("" : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3) : : "memory");
                an[cb] = ld;
            }
            Ys4[addr_cur + 0] = c0;
            Ys4[addr_cur + 1] = c1;
            Ys4[addr_cur + 2] = c2;
            Ys4[addr_cur + 3] = c3;
            addr_cur = addr_nxt;
        }
        ng_cur = ng_nxt;
    };

    float4 bA[CB][4], bB[CB][4];
    load_nbr(0);
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    ng_cur = compact(0);
    if (K3 > 1) load_nbr(1);
    load_b(0, bA);
    if (ng_cur > 0) gather(0, 0);
    for (int step = 0; step < nsteps; step += 2) {
        body(step, bA, bB);
        if (step + 1 < nsteps) body(step + 1, bB, bA);
    }
    __threadfence_block();
    // ---- epilogue: tile -> global.  Column 16 a + 4 b + c of a row sits at position 16 a + 4 c + b: the four columns
    // 4 mm .. 4 mm + 3 of a lane are four floats 16 bytes apart.
    const int mm = lane & 15;
    const int c4 = mm * 4;
    const int pos0 = 16 * (mm >> 2) + (mm & 3);
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.bias && csplit == 1 && n0 + c4 < Cout) bv = *reinterpret_cast<const float4*>(a.bias + n0 + c4);
    float* out = csplit > 1 ? a.partial + (long long)sp * a.n_out * Cout : a.Y;
    const int ldo = csplit > 1 ? Cout : a.ldy;
    const bool col_ok = n0 + c4 < Cout;
    for (int r0 = lane >> 4; r0 < rows_per_tile; r0 += 32) {
        int rows[8];
        float4 ys[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = r0 + 4 * u;
            rows[u] = r < rows_per_tile ? s_row[r] : -1;
            const float* yr = &Ys[min(r, R) * CMP_YS + pos0];
            ys[u] = make_float4(yr[0], yr[4], yr[8], yr[12]);
        }
        // the addend's eight row pieces are requested together, before the first store (the addend may alias the output:
        // a load placed after a store has to wait for it)
        float4 ads[8];
        const bool with_add = a.addend && csplit == 1;
        if (with_add) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                ads[u] = (rows[u] >= 0 && col_ok)
                    ? *reinterpret_cast<const float4*>(a.addend + (long long)rows[u] * a.ld_add + n0 + c4)
                    : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (rows[u] >= 0 && col_ok) {
                float4 y = ys[u];
                y.x += bv.x; y.y += bv.y; y.z += bv.z; y.w += bv.w;
                if (with_add) { y.x += ads[u].x; y.y += ads[u].y; y.z += ads[u].z; y.w += ads[u].w; }
                *reinterpret_cast<float4*>(out + (long long)rows[u] * ldo + n0 + c4) = y;
            }
        }
    }
#ifdef AGB_TIMELINE
    if (lane == 0 && blockIdx.x < AGB_TIMELINE_SLOTS) {
        unsigned long long* e = g_cmp_timeline + 4 * blockIdx.x;
        e[0] = tl_t0; e[1] = wall_clock64(); e[2] = __builtin_readcyclecounter() - tl_c0; e[3] = tl_groups;
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// k_spconv_cma: k_spconv_cmpt's arithmetic with the whole main loop HAND-SCHEDULED (cmp_asm.inc, generated by
// gen_cmp_asm.py: register map, program and wait accounting are documented there).  C++ keeps the tile mapping, the LDS
// initialisation and the tile's final store; everything between is one asm statement.  128-row tiles, K3 >= 3.
#ifdef AGB_TIMELINE
#include "cmp_asm_stamps.inc"
#else
#include "cmp_asm.inc"
#endif
__global__ __launch_bounds__(64) void k_spconv_cma(ConvArgs a, int ntiles, int nct, int rows_per_tile, int csplit,
                                                   int il_shift) {
    constexpr int R = 128, CB = CMP_CB;
    __shared__ __attribute__((aligned(16))) char lds[CMA_LDS_BYTES];
    float* const Ys = reinterpret_cast<float*>(lds);
    int2* const lists = reinterpret_cast<int2*>(lds + CMA_LDS_LISTS);
    int* const s_row = reinterpret_cast<int*>(lds + CMA_LDS_SROW);
    const int lane = threadIdx.x;
    const int m = lane & 15, q = lane >> 4;
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int per_xcd = (ntiles + 7) >> 3;
    const int ct0 = jx % nct;
    const int sp = (jx / nct) % csplit;
    const int tile = xcd * per_xcd + jx / (nct * csplit);
    if (tile >= ntiles || jx / (nct * csplit) >= per_xcd) return;
    const int row0 = tile * rows_per_tile;
    const int row_end = min(a.n_out, row0 + rows_per_tile);
    auto grow = [&](int rl) -> int {
        if (il_shift == 0) return row0 + rl < row_end ? row0 + rl : -1;
        if (rl >= rows_per_tile) return -1;
        const int blk = a.tile_blocks ? a.tile_blocks[tile * (rows_per_tile >> il_shift) + (rl >> il_shift)]
                                      : (rl >> il_shift) * ntiles + tile;
        const int r = (blk << il_shift) | (rl & ((1 << il_shift) - 1));
        return (blk >= 0 && r < a.n_out) ? r : -1;
    };
    const int myrow0 = grow(lane), myrow1 = grow(64 + lane);
    s_row[lane] = myrow0;
    s_row[64 + lane] = myrow1;
    const int n0 = ct0 * 64;
    const int K3 = a.K3, Cin = a.Cin, Cout = a.Cout;
    const int NSB = Cin / (CB * 16) / csplit;
    const int c_first = sp * NSB * (CB * 16);
    const bool colok = (n0 + 4 * m) < Cout;
    const int coff = colok ? 4 * m : 0;
    for (int i = lane; i < (R + 1) * CMP_YS / 4; i += 64)
        reinterpret_cast<float4*>(Ys)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = lane; i < 3 * CMA_LP; i += 64) lists[i] = make_int2(0, R);
    {
        const float* xb = a.X + c_first;
        const float* wb = a.W + (long long)c_first * Cout + n0;
        const int32_t* nb = a.nbr + (a.kflip ? (long long)(K3 - 1) * a.nbr_stride : 0ll);
        const long long nstep = (a.kflip ? -a.nbr_stride : a.nbr_stride) * 4ll;
        const unsigned long long valid0 = __ballot(myrow0 >= 0), valid1 = __ballot(myrow1 >= 0);
        const int rowc0 = max(myrow0, 0) * 4, rowc1 = max(myrow1, 0) * 4;
        const int woff = (4 * q * Cout + coff) * 4;
        const unsigned lds_base = (unsigned)(uintptr_t)lds;
#ifdef AGB_TIMELINE
        unsigned o0, o1, o2, o3, o4, o5, o6, o7, o8, o9;
        asm volatile(CMA_ASM_TEXT
                     : [o0] "=v"(o0), [o1] "=v"(o1), [o2] "=v"(o2), [o3] "=v"(o3), [o4] "=v"(o4), [o5] "=v"(o5), [o6] "=v"(o6),
                       [o7] "=v"(o7), [o8] "=v"(o8), [o9] "=v"(o9)
                     : [x] "s"(xb), [w] "s"(wb), [nb] "s"(nb), [nstep] "s"(nstep), [ldx4] "s"(a.ldx * 4), [cout4] "s"(Cout * 4),
                       [wk4] "s"(Cin * Cout * 4), [k3] "s"(K3), [nsb] "s"(NSB), [valid0] "s"(valid0), [valid1] "s"(valid1),
                       [lds] "s"(lds_base), [lane] "v"(lane), [woff] "v"(woff), [rowc0] "v"(rowc0), [rowc1] "v"(rowc1)
                     : CMA_ASM_CLOBBERS);
        if (lane == 0) {   // top / F / P clocks (64 bit each), F groups, P groups
            atomicAdd(&g_cmpt_stamp[0], ((unsigned long long)o1 << 32) | o0);
            atomicAdd(&g_cmpt_stamp[1], ((unsigned long long)o3 << 32) | o2);
            atomicAdd(&g_cmpt_stamp[2], ((unsigned long long)o5 << 32) | o4);
            atomicAdd(&g_cmpt_stamp[3], (unsigned long long)o6);
            atomicAdd(&g_cmpt_stamp[4], (unsigned long long)o7);
            atomicAdd(&g_cmpt_stamp[5], 1ull);
            atomicAdd(&g_cmpt_stamp[6], (unsigned long long)o8);     // clocks in the five vmcnt waits of all groups
            atomicAdd(&g_cmpt_stamp[7], (unsigned long long)o9);     // clocks in the list-entry waits
        }
#else
        asm volatile(CMA_ASM_TEXT
                     :
                     : [x] "s"(xb), [w] "s"(wb), [nb] "s"(nb), [nstep] "s"(nstep), [ldx4] "s"(a.ldx * 4), [cout4] "s"(Cout * 4),
                       [wk4] "s"(Cin * Cout * 4), [k3] "s"(K3), [nsb] "s"(NSB), [valid0] "s"(valid0), [valid1] "s"(valid1),
                       [lds] "s"(lds_base), [lane] "v"(lane), [woff] "v"(woff), [rowc0] "v"(rowc0), [rowc1] "v"(rowc1)
                     : CMA_ASM_CLOBBERS);
#endif
    }
    // ---- epilogue: tile -> global.  Column 16 a + 4 b + c of a row sits at position 16 a + 4 c + b.
    const int mm = lane & 15;
    const int c4 = mm * 4;
    const int pos0 = 16 * (mm >> 2) + (mm & 3);
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.bias && csplit == 1 && n0 + c4 < Cout) bv = *reinterpret_cast<const float4*>(a.bias + n0 + c4);
    float* out = csplit > 1 ? a.partial + (long long)sp * a.n_out * Cout : a.Y;
    const int ldo = csplit > 1 ? Cout : a.ldy;
    const bool col_ok = n0 + c4 < Cout;
    for (int r0 = lane >> 4; r0 < rows_per_tile; r0 += 32) {
        int rows[8];
        float4 ys[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = r0 + 4 * u;
            rows[u] = r < rows_per_tile ? s_row[r] : -1;
            const float* yr = &Ys[min(r, R) * CMP_YS + pos0];
            ys[u] = make_float4(yr[0], yr[4], yr[8], yr[12]);
        }
        // the addend's eight row pieces are requested together, before the first store (the addend may alias the output:
        // a load placed after a store has to wait for it)
        float4 ads[8];
        const bool with_add = a.addend && csplit == 1;
        if (with_add) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                ads[u] = (rows[u] >= 0 && col_ok)
                    ? *reinterpret_cast<const float4*>(a.addend + (long long)rows[u] * a.ld_add + n0 + c4)
                    : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (rows[u] >= 0 && col_ok) {
                float4 y = ys[u];
                y.x += bv.x; y.y += bv.y; y.z += bv.z; y.w += bv.w;
                if (with_add) { y.x += ads[u].x; y.y += ads[u].y; y.z += ads[u].z; y.w += ads[u].w; }
                *reinterpret_cast<float4*>(out + (long long)rows[u] * ldo + n0 + c4) = y;
            }
        }
    }
}

// Y[r, :] = bias + sum_s partial[s][r, :]   (fixed order)
template <typename T = float>
__global__ void k_split_reduce(const float* __restrict__ partial, int S, int n_out, int C4,
                               const float* __restrict__ bias, T* __restrict__ Y, int ldy,
                               const float* addend = nullptr, int ld_add = 0) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int r = (int)(t / C4), c = (int)(t % C4) * 4;
    if (r >= n_out) return;
    float4 acc = bias ? *reinterpret_cast<const float4*>(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < S; ++s) {
        float4 v = *reinterpret_cast<const float4*>(partial + ((long long)s * n_out + r) * (C4 * 4) + c);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (addend) {      // (the sum first, then the addend: the value an addition after the convolution would give)
        const float4 ad = *reinterpret_cast<const float4*>(addend + (long long)r * ld_add + c);
        acc.x += ad.x; acc.y += ad.y; acc.z += ad.z; acc.w += ad.w;
    }
    st4(Y + (long long)r * ldy + c, acc);
}

// WT[k][c][r] = W[k][r][c]: the data gradient multiplies by W[k]^T, and the weights change every step, so this runs
// once per layer per step.  64x64 tiles through LDS, float4 on both sides (R, C multiples of 4).
// zero (optional, same element count): the layer's weight-gradient buffer, cleared here instead of by a separate fill.
__global__ __launch_bounds__(256) void k_weight_transpose(const float* __restrict__ W, float* __restrict__ WT, int R,
                                                          int C, float* __restrict__ zero) {
    __shared__ float tile[64][65];
    const float* w = W + (long long)blockIdx.z * R * C;
    float* wt = WT + (long long)blockIdx.z * R * C;
    float* zp = zero ? zero + (long long)blockIdx.z * R * C : nullptr;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int g = threadIdx.x & 15, h = threadIdx.x >> 4;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = h + 16 * p;
        if (r0 + r < R && c0 + 4 * g < C) {
            float4 v = *reinterpret_cast<const float4*>(w + (long long)(r0 + r) * C + c0 + 4 * g);
            tile[r][4 * g + 0] = v.x; tile[r][4 * g + 1] = v.y; tile[r][4 * g + 2] = v.z; tile[r][4 * g + 3] = v.w;
        }
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int c = h + 16 * p;
        if (c0 + c < C && r0 + 4 * g < R) {
            float4 v = make_float4(tile[4 * g + 0][c], tile[4 * g + 1][c], tile[4 * g + 2][c], tile[4 * g + 3][c]);
            *reinterpret_cast<float4*>(wt + (long long)(c0 + c) * R + r0 + 4 * g) = v;
            if (zp) *reinterpret_cast<float4*>(zp + (long long)(c0 + c) * R + r0 + 4 * g) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}

// The same for EVERY convolution of a model in one launch (the transposes live in a buffer the caller keeps until the next
// optimiser step): tab int64 [n][6] = (W, WT, K3, R, C, first tile) per layer, tiles = 64 x 64 pieces of one offset.
__global__ __launch_bounds__(256) void k_weight_transpose_batched(const long long* __restrict__ tab, int n) {
    __shared__ float tile[64][65];
    __shared__ long long s_e[6];
    if (threadIdx.x == 0) {
        int i = 0;
        while (i + 1 < n && (long long)blockIdx.x >= tab[(i + 1) * 6 + 5]) ++i;
        for (int j = 0; j < 6; ++j) s_e[j] = tab[i * 6 + j];
    }
    __syncthreads();
    const int R = (int)s_e[3], C = (int)s_e[4];
    const int tx = (C + 63) >> 6, ty = (R + 63) >> 6;
    const int t = (int)((long long)blockIdx.x - s_e[5]);
    const int k = t / (tx * ty), rem = t - k * tx * ty;
    const float* w = reinterpret_cast<const float*>(s_e[0]) + (long long)k * R * C;
    float* wt = reinterpret_cast<float*>(s_e[1]) + (long long)k * R * C;
    const int r0 = (rem / tx) * 64, c0 = (rem % tx) * 64;
    const int g = threadIdx.x & 15, h = threadIdx.x >> 4;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = h + 16 * p;
        if (r0 + r < R && c0 + 4 * g < C) {
            float4 v = *reinterpret_cast<const float4*>(w + (long long)(r0 + r) * C + c0 + 4 * g);
            tile[r][4 * g + 0] = v.x; tile[r][4 * g + 1] = v.y; tile[r][4 * g + 2] = v.z; tile[r][4 * g + 3] = v.w;
        }
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int c = h + 16 * p;
        if (c0 + c < C && r0 + 4 * g < R)
            *reinterpret_cast<float4*>(wt + (long long)(c0 + c) * R + r0 + 4 * g) =
                make_float4(tile[4 * g + 0][c], tile[4 * g + 1][c], tile[4 * g + 2][c], tile[4 * g + 3][c]);
    }
}

// Both bf16 operand forms of a kernel W [K3][R][C] (fp32) in one launch: W16 [K3][R][C] (K-major for the data gradient)
// and Wt16 [K3][C][R] (K-major for the forward pass), round to nearest even.  The bf16-row mode converted every layer's
// weights with three launches per step (transpose, two conversions: 160 launches, 1.0 ms of MSENet50's 19.4 ms step).
__global__ __launch_bounds__(256) void k_weight_twins(const float* __restrict__ W, bf16_t* __restrict__ W16,
                                                      bf16_t* __restrict__ Wt16, int R, int C) {
    __shared__ float tile[64][65];
    const long long base = (long long)blockIdx.z * R * C;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int g = threadIdx.x & 15, h = threadIdx.x >> 4;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = h + 16 * p;
        if (r0 + r < R && c0 + 4 * g < C) {
            const long long o = base + (long long)(r0 + r) * C + c0 + 4 * g;
            const float4 v = *reinterpret_cast<const float4*>(W + o);
            tile[r][4 * g + 0] = v.x; tile[r][4 * g + 1] = v.y; tile[r][4 * g + 2] = v.z; tile[r][4 * g + 3] = v.w;
            st4(W16 + o, v);
        }
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int c = h + 16 * p;
        if (c0 + c < C && r0 + 4 * g < R)
            st4(Wt16 + base + (long long)(c0 + c) * R + r0 + 4 * g,
                make_float4(tile[4 * g + 0][c], tile[4 * g + 1][c], tile[4 * g + 2][c], tile[4 * g + 3][c]));
    }
}

// The same for EVERY layer of a model in one launch: tab int64 [n][7] = (W, W16, Wt16, K3, R, C, first tile).  A linear scan
// of the table (n <= a few hundred): the binary-search-free form keeps the kernel trivial; 52 layers of SENet50.
__global__ __launch_bounds__(256) void k_weight_twins_batched(const long long* __restrict__ tab, int n) {
    __shared__ float tile[64][65];
    __shared__ long long s_e[7];
    if (threadIdx.x == 0) {
        int lo = 0, hi = n - 1;           // last layer whose first tile is <= blockIdx.x
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (tab[mid * 7 + 6] <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
        }
        for (int j = 0; j < 7; ++j) s_e[j] = tab[lo * 7 + j];
    }
    __syncthreads();
    const int R = (int)s_e[4], C = (int)s_e[5];
    const int tx = (C + 63) >> 6, ty = (R + 63) >> 6;
    const int t = (int)((long long)blockIdx.x - s_e[6]);
    const int k = t / (tx * ty), rem = t - k * tx * ty;
    const float* W = reinterpret_cast<const float*>(s_e[0]);
    bf16_t* W16 = reinterpret_cast<bf16_t*>(s_e[1]);
    bf16_t* Wt16 = reinterpret_cast<bf16_t*>(s_e[2]);
    const long long base = (long long)k * R * C;
    const int r0 = (rem / tx) * 64, c0 = (rem % tx) * 64;
    const int g = threadIdx.x & 15, h = threadIdx.x >> 4;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = h + 16 * p;
        if (r0 + r < R && c0 + 4 * g < C) {
            const long long o = base + (long long)(r0 + r) * C + c0 + 4 * g;
            const float4 v = *reinterpret_cast<const float4*>(W + o);
            tile[r][4 * g + 0] = v.x; tile[r][4 * g + 1] = v.y; tile[r][4 * g + 2] = v.z; tile[r][4 * g + 3] = v.w;
            st4(W16 + o, v);
        }
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int c = h + 16 * p;
        if (c0 + c < C && r0 + 4 * g < R)
            st4(Wt16 + base + (long long)(c0 + c) * R + r0 + 4 * g,
                make_float4(tile[4 * g + 0][c], tile[4 * g + 1][c], tile[4 * g + 2][c], tile[4 * g + 3][c]));
    }
}

// ------------------------------------------------------------------------------------------------
// Lattice-parity partition of the rows of a level for a stride-s operator: class = (c/ts mod s) per axis.
// perm [n + ncls*TM] receives the rows grouped by class, every class padded with -1 to a multiple of TM;
// tile_cls[t] = class of tile t (or -1).  Order inside a class is arbitrary (results do not depend on it).
__device__ __forceinline__ int parity_class(int4 c, int ts, int s) {
    int px = ((c.y / ts) % s + s) % s, py = ((c.z / ts) % s + s) % s, pz = ((c.w / ts) % s + s) % s;
    return px + s * (py + s * pz);
}

__global__ void k_parity_count(const int4* __restrict__ coords, int n, int ts, int s, int32_t* counts) {
    __shared__ int h[64];
    if (threadIdx.x < 64) h[threadIdx.x] = 0;
    __syncthreads();
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(&h[parity_class(coords[i], ts, s)], 1);
    __syncthreads();
    if (threadIdx.x < s * s * s && h[threadIdx.x]) atomicAdd(&counts[threadIdx.x], h[threadIdx.x]);
}

// counts[0..ncls) -> start[0..ncls] (padded to TM), cursor[c] = start[c]; tile classes
__global__ void k_parity_layout(int32_t* counts, int ncls, int TM, int32_t* start, int32_t* cursor, int32_t* tile_cls,
                                int max_tiles) {
    __shared__ int s_start[65];
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int c = 0; c < ncls; ++c) {
            s_start[c] = acc;
            acc += (counts[c] + TM - 1) / TM * TM;
        }
        s_start[ncls] = acc;
    }
    __syncthreads();
    if (threadIdx.x <= ncls) start[threadIdx.x] = s_start[threadIdx.x];
    if (threadIdx.x < ncls) cursor[threadIdx.x] = s_start[threadIdx.x];
    for (int t = threadIdx.x; t < max_tiles; t += blockDim.x) {
        int r = t * TM, cls = -1;
        for (int c = 0; c < ncls; ++c)
            if (r >= s_start[c] && r < s_start[c + 1]) cls = c;
        tile_cls[t] = cls;
    }
}

__global__ void k_parity_scatter(const int4* __restrict__ coords, int n, int ts, int s, int32_t* cursor,
                                 int32_t* perm) {
    __shared__ int h[64], base[64];
    if (threadIdx.x < 64) h[threadIdx.x] = 0;
    __syncthreads();
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int cls = -1, rank = 0;
    if (i < n) {
        cls = parity_class(coords[i], ts, s);
        rank = atomicAdd(&h[cls], 1);
    }
    __syncthreads();
    if (threadIdx.x < s * s * s && h[threadIdx.x]) base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], h[threadIdx.x]);
    __syncthreads();
    if (i < n) perm[base[cls] + rank] = i;
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
                                                   int rows_per_wg, int cin_tiles, int chunks, int m_tiles) {
    __shared__ __attribute__((aligned(16))) float As[BK * 64];  // [r][m]
    __shared__ __attribute__((aligned(16))) float Bs[BK * 64];  // [r][n]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int n0 = blockIdx.y * 64;
    // 1-D grid, XCD-aware (as k_spconv_dw_cmp): the m-tiles of one row chunk run back to back on one XCD, so the
    // chunk's dY rows (573 KB for the stem) are read from HBM once instead of once per m-tile (PMC: 2.4 GB -> per
    // launch for the 22 m-tiles of the 7^3 stem before this mapping)
    int mt, chunk;
    if (chunks >= 16) {
        const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
        mt = jx % m_tiles;
        chunk = (jx / m_tiles) * 8 + xcd;
        if (chunk >= chunks) return;
    } else {
        chunk = blockIdx.x % chunks;
        mt = blockIdx.x / chunks;
    }
    const int r_begin = chunk * rows_per_wg;
    const int r_end = min(n_out, r_begin + rows_per_wg);

    int k = 0, c0 = 0, k0 = 0;
    if constexpr (CPAD == 0) {
        k = mt / cin_tiles;
        c0 = (mt % cin_tiles) * 64;
    } else {
        k0 = mt * (64 / CPAD);
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

// Weight gradient, generic path, PAIR-COMPACTED: the reduction runs over rows, so rows whose neighbour at offset k is
// absent need not be multiplied at all.  Every workgroup first compacts its row chunk of nbr[k] into an (in, out) pair
// list in LDS (row order kept: deterministic inside the workgroup), then walks that list 64 pairs at a time with the
// same register-prefetch pipeline — MFMA work falls by the map density (0.5-0.8 on the 3^3 maps).
// 1-D grid, XCD-aware: consecutive workgroups of one XCD (id % 8) walk the m-tiles (offset x cin tile) of ONE row
// chunk, so the chunk's dY / X rows are re-read from that XCD's L2 instead of crossing to HBM/MALL 27 times.
#define DW_KS 64
#define DW_MAXROWS 2048
#define DW_CMP_ROWS_F32 1280     // fp32 operands: 16 + 16 KB of staged rows + 7.5 KB of pair list = 40 KB -> four workgroups per CU
static_assert(DW_CMP_ROWS_F32 == 1280 && DW_MAXROWS == 2048, "the launches below spell the literals");
// PREC = 0: exact fp32 MFMA (v_mfma_f32_32x32x2_f32).
// PREC = 1 / 2: bf16 / split-bf16x3 operands on v_mfma_f32_32x32x16_bf16 (fp32 accumulate).  The reduction index of
//   this product is the PAIR, which is the slow index of both operands in memory, while the bf16 MFMA wants 8 consecutive
//   reduction elements per lane: the gathered rows are transposed while they are staged — a thread loads the same four
//   channels of two consecutive pairs, packs them pairwise into bf16x2 dwords and writes T[channel][pair / 2].  Rows of
//   48 dwords with the 4-dword chunk index XORed with (channel >> 2) & 7 keep the ds_write_b32 2-way (free) and the
//   fragment ds_read_b128 conflict-free (searched by brute force over the bank rules of MI355X_MICROARCH.md §LDS).
// nbr == nullptr: identity map (dense X^T dY of a 1x1 stride-1 convolution).
#define DWT_LD 48     // dwords per transposed row (32 used: 64 pairs)
__device__ __forceinline__ unsigned pack_bf16(float a, float b) { return pack2_bf16(a, b); }
__device__ __forceinline__ unsigned pack_bf16_lo(float a, float b) {
    const unsigned h = pack2_bf16(a, b);
    return pack2_bf16(a - __uint_as_float(h << 16), b - __uint_as_float(h & 0xFFFF0000u));
}
// IN16 (PREC 1 only): X and dY are bf16 twins (uint16 rows, ldx / ldy in bf16 elements): 8-byte loads of four channels,
// no conversion — the two pairs of a dword are interleaved with two bit operations per channel.
// TR16 (with IN16): the bf16 rows are staged AS THEY ARE — image [pair][64 channels] of 128-byte rows, one ds_write_b128 per
// 16-byte piece, 16-byte chunk index XORed with 4 * ((pair >> 1) & 1) — and the transposition the MFMA wants (8 consecutive
// pairs per lane) is done by the LDS hardware: two ds_read_b64_tr_b16 per operand and MFMA (each 16-lane group reads a block
// of 4 pairs x 16 channels and receives it column-major; conflict-free with that XOR by the bank rule of
// MI355X_MICROARCH.md section LDS).  Four 16-byte loads and four LDS writes per thread and 64-pair step instead of eight
// 8-byte loads, sixteen 4-byte writes and the bit interleave.
typedef short s16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 lds_read_tr16(const unsigned short* p) {
    const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
    return __builtin_bit_cast(uint2, v);
}
template <int PREC, bool IN16 = false, bool TR16 = false, int MAXR = DW_MAXROWS, int KS = DW_KS>
__global__ __launch_bounds__(256) void k_spconv_dw_cmp(const float* __restrict__ X, int ldx,
                                                       const float* __restrict__ dY, int ldy,
                                                       const int32_t* __restrict__ nbr, long long nbr_stride,
                                                       float* __restrict__ dW, int n_out, int K3, int Cin, int Cout,
                                                       int rows_per_wg, int cin_tiles, int chunks, int m_tiles,
                                                       int il_shift, float* __restrict__ part) {
    // il_shift > 0: a row chunk is made of 2^il_shift-row blocks taken `chunks` blocks apart (see k_spconv_cmp: evens
    // out the pair count per workgroup where the density varies by region)
    // part != nullptr (the REPRODUCIBLE form, any operand precision): the workgroup's tile goes to part[chunk] with plain
    // stores — one writer per element — and k_dw_fold_small adds the chunks in ascending order; nullptr: fp32 atomic adds
    // on dW, whose arrival order differs from run to run.
    constexpr int NP = PREC == 2 ? 2 : 1;
    // fp32: As [pair][m], Bs [pair][n];  bf16: At [m][pair/2] / Bt [n][pair/2] (transposed, swizzled), hi (and lo) planes
    __shared__ __attribute__((aligned(16))) float As[PREC == 0 ? KS * 64 : NP * 64 * DWT_LD];
    __shared__ __attribute__((aligned(16))) float Bs[PREC == 0 ? KS * 64 : NP * 64 * DWT_LD];
    // pair list of the chunk: input row (global) and output row as its LOCAL index in the chunk (16 bits: with MAXR = 1280 the
    // workgroup's LDS is 40 KB — four workgroups per CU instead of three)
    __shared__ int p_in[MAXR];
    __shared__ unsigned short p_out[MAXR];
    __shared__ int s_wcnt[2][4];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    int mt, chunk, ny0 = blockIdx.y;
    if (chunks >= 16) {  // XCD-aware: the m-tiles of one row chunk run back to back on one XCD
        // ... and, innermost, the OUTPUT column tiles of one m-tile (gridDim.y of them: they share the m-tile's gathered X rows;
        // in plain order they are gridDim.x workgroups apart).  gridDim.x is a multiple of 8 here: the linear workgroup id
        // L = x + gridDim.x y has L % 8 = x % 8 = its XCD.
        const int L = blockIdx.x + gridDim.x * blockIdx.y;
        const int xcd = L & 7, j = L >> 3, ny = gridDim.y;
        const int jx = j / ny;
        ny0 = j - jx * ny;
        mt = jx % m_tiles;
        chunk = (jx / m_tiles) * 8 + xcd;
        if (chunk >= chunks) return;
    } else {             // few row chunks: spread the m-tiles over the XCDs
        chunk = blockIdx.x % chunks;
        mt = blockIdx.x / chunks;
    }
    const int n0 = ny0 * 64;
    const int r_begin = chunk * rows_per_wg;
    const int r_end = il_shift ? r_begin + rows_per_wg : min(n_out, r_begin + rows_per_wg);
    const int k = mt / cin_tiles;
    const int c0 = (mt % cin_tiles) * 64;
    const int t_r = tid >> 4, t_c = (tid & 15) * 4;
    const int32_t* nrow = nbr ? nbr + (long long)k * nbr_stride : nullptr;
    // local row of the chunk -> row of the level (-1: past the end)
    auto grow = [&](int i) -> int {
        if (il_shift == 0) return r_begin + i;
        const int r = ((((i >> il_shift) * chunks + chunk) << il_shift) | (i & ((1 << il_shift) - 1)));
        return r < n_out ? r : -1;
    };

    // ---- phase 1: ordered compaction of the present pairs.  All index loads are issued up front (one exposed memory
    // latency per workgroup instead of one per 256 rows: the prologue was ~30 % of a workgroup's time)
    int total = 0;
    const int nrows = r_end - r_begin;
    constexpr int NIT = MAXR / 256;
    static_assert(MAXR % 256 == 0 && MAXR <= 65536, "chunk rows: whole passes of the workgroup, 16-bit local indices");
    int idxs[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = it * 256 + tid;
        const int gr = i < nrows ? grow(i) : -1;
        idxs[it] = gr >= 0 ? (nrow ? nrow[gr] : gr) : -1;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int base = it * 256;
        if (base >= nrows) break;
        const int i = base + tid;
        const int idx = idxs[it];
        const unsigned long long bal = __ballot(idx >= 0);
        if (lane == 0) s_wcnt[it & 1][wave] = __popcll(bal);
        __syncthreads();
        int off = total, round = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            int c = s_wcnt[it & 1][w];
            if (w < wave) off += c;
            round += c;
        }
        if (idx >= 0) {
            int p = off + __popcll(bal & ((1ull << lane) - 1ull));
            p_in[p] = idx;
            p_out[p] = (unsigned short)i;
        }
        total += round;
    }
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    if (total == 0) {
        if (part == nullptr) return;
        // (the fold reads every chunk's tile: an empty chunk still writes its zeros)
        const int colz = n0 + wc * 32 + li;
        if (colz < Cout) {
            float* dst = part + (long long)chunk * K3 * Cin * Cout;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int m = wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                if (c0 + m < Cin) dst[((long long)k * Cin + c0 + m) * Cout + colz] = 0.f;
            }
        }
        return;
    }

    static_assert(KS == DW_KS || PREC == 0, "only the fp32 form takes another step length");
    constexpr int LJ = KS / 16;
    float4 a_reg[LJ], b_reg[LJ];
    uint2 a16[IN16 ? LJ : 1], b16[IN16 ? LJ : 1];
    uint4 a8_0, a8_1, b8_0, b8_1;                             // TR16: pair (tid >> 3) + 32 j, channels 8 (tid & 7) ..
    // (four named registers, not arrays: as arrays captured by the load lambda they were left in scratch memory)
    const int t8_p = tid >> 3, t8_c = (tid & 7) * 8;
    const int a8_col = min(c0 + t8_c, Cin - 8), b8_col = min(n0 + t8_c, Cout - 8);
    const bool a8_ok = c0 + t8_c < Cin, b8_ok = n0 + t8_c < Cout;
    const unsigned short* X16 = reinterpret_cast<const unsigned short*>(X);
    const unsigned short* dY16 = reinterpret_cast<const unsigned short*>(dY);
    // loads are unconditional (clamped pair / column) and masked when stored to LDS: under branches every pair's
    // "LDS list read -> global load" chain sat in its own block, four of them back to back before the first MFMA
    const int a_col = min(c0 + t_c, Cin - 4), b_col = min(n0 + t_c, Cout - 4);
    const bool a_ok = c0 + t_c < Cin, b_ok = n0 + t_c < Cout;
    // pair (inside the 64-pair step) of register j: fp32 rows t_r + 16 j; bf16: the two pairs of pair-pair t_r + 16 (j >> 1)
    auto pair_of = [&](int j) { return PREC == 0 ? t_r + 16 * j : 2 * (t_r + 16 * (j >> 1)) + (j & 1); };
    auto load_data = [&](int pb) {
        if constexpr (TR16) {
            const int p0 = min(pb + t8_p, total - 1), p1 = min(pb + t8_p + 32, total - 1);
            const int i0 = p_in[p0], o0 = grow(p_out[p0]), i1 = p_in[p1], o1 = grow(p_out[p1]);
            a8_0 = *reinterpret_cast<const uint4*>(X16 + (long long)i0 * ldx + a8_col);
            b8_0 = *reinterpret_cast<const uint4*>(dY16 + (long long)o0 * ldy + b8_col);
            a8_1 = *reinterpret_cast<const uint4*>(X16 + (long long)i1 * ldx + a8_col);
            b8_1 = *reinterpret_cast<const uint4*>(dY16 + (long long)o1 * ldy + b8_col);
            return;
        }
#pragma unroll
        for (int j = 0; j < LJ; ++j) {
            const int p = min(pb + pair_of(j), total - 1);
            const int in = p_in[p], out = grow(p_out[p]);
            if constexpr (IN16) {
                a16[j] = *reinterpret_cast<const uint2*>(X16 + (long long)in * ldx + a_col);
                b16[j] = *reinterpret_cast<const uint2*>(dY16 + (long long)out * ldy + b_col);
            } else {
                a_reg[j] = *reinterpret_cast<const float4*>(X + (long long)in * ldx + a_col);
                b_reg[j] = *reinterpret_cast<const float4*>(dY + (long long)out * ldy + b_col);
            }
        }
    };
    load_data(0);
    for (int pb = 0; pb < total; pb += KS) {
        if constexpr (PREC == 0) {
#pragma unroll
            for (int j = 0; j < LJ; ++j) {
                const bool live = pb + t_r + 16 * j < total;
                float4 av = a_reg[j], bv = b_reg[j];
                if (!(live && a_ok)) av = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!(live && b_ok)) bv = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(&As[(t_r + 16 * j) * 64 + t_c]) = av;
                *reinterpret_cast<float4*>(&Bs[(t_r + 16 * j) * 64 + t_c]) = bv;
            }
        } else if constexpr (TR16) {
            unsigned short* A16 = reinterpret_cast<unsigned short*>(As);
            unsigned short* B16 = reinterpret_cast<unsigned short*>(Bs);
            // (component-wise masks: a ternary between two uint4 STRUCTS is compiled as a select between two stack slots)
            auto masked = [](uint4 v, bool keep) {
                const unsigned m = keep ? 0xFFFFFFFFu : 0u;
                return make_uint4(v.x & m, v.y & m, v.z & m, v.w & m);
            };
            const bool live0 = pb + t8_p < total, live1 = pb + t8_p + 32 < total;
            // (pairs p and p + 32 have the same swizzle: ((p >> 1) & 1) is bit 1 of the pair)
            const int off = t8_p * 64 + (((t8_c >> 3) ^ (((t8_p >> 1) & 1) << 2)) << 3);
            *reinterpret_cast<uint4*>(&A16[off]) = masked(a8_0, live0 && a8_ok);
            *reinterpret_cast<uint4*>(&B16[off]) = masked(b8_0, live0 && b8_ok);
            *reinterpret_cast<uint4*>(&A16[off + 32 * 64]) = masked(a8_1, live1 && a8_ok);
            *reinterpret_cast<uint4*>(&B16[off + 32 * 64]) = masked(b8_1, live1 && b8_ok);
        } else {
            unsigned* At = reinterpret_cast<unsigned*>(As);
            unsigned* Bt = reinterpret_cast<unsigned*>(Bs);
#pragma unroll
            for (int jj = 0; jj < LJ / 2; ++jj) {
                if constexpr (IN16) {
                    const bool l0 = pb + pair_of(2 * jj) < total, l1 = pb + pair_of(2 * jj + 1) < total;
                    const uint2 z = make_uint2(0u, 0u);
                    const uint2 a0 = (l0 && a_ok) ? a16[2 * jj] : z, a1 = (l1 && a_ok) ? a16[2 * jj + 1] : z;
                    const uint2 b0 = (l0 && b_ok) ? b16[2 * jj] : z, b1 = (l1 && b_ok) ? b16[2 * jj + 1] : z;
                    const int pp = t_r + 16 * jj;
                    const int col = (((pp >> 2) ^ ((t_c >> 2) & 7)) << 2) | (pp & 3);
                    At[(t_c + 0) * DWT_LD + col] = (a0.x & 0xffffu) | (a1.x << 16);
                    At[(t_c + 1) * DWT_LD + col] = (a0.x >> 16) | (a1.x & 0xffff0000u);
                    At[(t_c + 2) * DWT_LD + col] = (a0.y & 0xffffu) | (a1.y << 16);
                    At[(t_c + 3) * DWT_LD + col] = (a0.y >> 16) | (a1.y & 0xffff0000u);
                    Bt[(t_c + 0) * DWT_LD + col] = (b0.x & 0xffffu) | (b1.x << 16);
                    Bt[(t_c + 1) * DWT_LD + col] = (b0.x >> 16) | (b1.x & 0xffff0000u);
                    Bt[(t_c + 2) * DWT_LD + col] = (b0.y & 0xffffu) | (b1.y << 16);
                    Bt[(t_c + 3) * DWT_LD + col] = (b0.y >> 16) | (b1.y & 0xffff0000u);
                    continue;
                }
                float av[2][4], bv[2][4];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const bool live = pb + pair_of(2 * jj + e) < total;
                    const float4 a4 = a_reg[2 * jj + e], b4 = b_reg[2 * jj + e];
                    const bool la = live && a_ok, lb = live && b_ok;
                    av[e][0] = la ? a4.x : 0.f; av[e][1] = la ? a4.y : 0.f; av[e][2] = la ? a4.z : 0.f; av[e][3] = la ? a4.w : 0.f;
                    bv[e][0] = lb ? b4.x : 0.f; bv[e][1] = lb ? b4.y : 0.f; bv[e][2] = lb ? b4.z : 0.f; bv[e][3] = lb ? b4.w : 0.f;
                }
                const int pp = t_r + 16 * jj;                          // pair-pair (dword column) 0..31
                const int col = (((pp >> 2) ^ ((t_c >> 2) & 7)) << 2) | (pp & 3);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    At[(t_c + i) * DWT_LD + col] = pack_bf16(av[0][i], av[1][i]);
                    Bt[(t_c + i) * DWT_LD + col] = pack_bf16(bv[0][i], bv[1][i]);
                    if constexpr (PREC == 2) {
                        At[(64 + t_c + i) * DWT_LD + col] = pack_bf16_lo(av[0][i], av[1][i]);
                        Bt[(64 + t_c + i) * DWT_LD + col] = pack_bf16_lo(bv[0][i], bv[1][i]);
                    }
                }
            }
        }
        __syncthreads();
        if (pb + KS < total) load_data(pb + KS);   // in flight during the MFMAs below
        if constexpr (PREC == 0) {
            const float* ap = &As[wr * 32 + li];
            const float* bp = &Bs[wc * 32 + li];
#pragma unroll
            for (int s2 = 0; s2 < KS / 2; ++s2) {
                float av = ap[(2 * s2 + lh) * 64];
                float bv = bp[(2 * s2 + lh) * 64];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
            }
        } else if constexpr (TR16) {
            // 16-lane group g = lane >> 4 reads the block of pairs 16 s + 8 (g >> 1) + {0..3} (second read: + 4) x channels
            // 32 w + 16 (g & 1) + {0..15}: lane 4 q + p of the group supplies the address of pair q, channels 4 p .. 4 p + 3, and
            // receives channel (lane & 15) of the four pairs = A[m = lane & 31][8 h + 0..3] of the MFMA operand
            const unsigned short* A16 = reinterpret_cast<const unsigned short*>(As);
            const unsigned short* B16 = reinterpret_cast<const unsigned short*>(Bs);
            const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
            const int prow = 8 * (g >> 1) + q;                       // (+ 16 s, + 4 for the second read)
            const int acol = wr * 32 + 16 * (g & 1) + 4 * pp, bcol = wc * 32 + 16 * (g & 1) + 4 * pp;
#pragma unroll
            for (int s2 = 0; s2 < DW_KS / 16; ++s2) {
                const int r0 = 16 * s2 + prow, r1 = r0 + 4;
                const int x0 = ((r0 >> 1) & 1) << 2, x1 = ((r1 >> 1) & 1) << 2;
                const uint2 a0 = lds_read_tr16(&A16[r0 * 64 + ((((acol >> 3) ^ x0) << 3) | (acol & 7))]);
                const uint2 a1 = lds_read_tr16(&A16[r1 * 64 + ((((acol >> 3) ^ x1) << 3) | (acol & 7))]);
                const uint2 b0 = lds_read_tr16(&B16[r0 * 64 + ((((bcol >> 3) ^ x0) << 3) | (bcol & 7))]);
                const uint2 b1 = lds_read_tr16(&B16[r1 * 64 + ((((bcol >> 3) ^ x1) << 3) | (bcol & 7))]);
                const bf16x8 ah = __builtin_bit_cast(bf16x8, make_uint4(a0.x, a0.y, a1.x, a1.y));
                const bf16x8 bh = __builtin_bit_cast(bf16x8, make_uint4(b0.x, b0.y, b1.x, b1.y));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
            }
        } else {
            // lane (r = lane & 31, h = lane >> 5) holds A[m = r][pair 16 s + 8 h + j] and B[same pairs][n = r], j = 0..7:
            // chunk 2 s + h of row r of the transposed tiles
            const unsigned* At = reinterpret_cast<const unsigned*>(As);
            const unsigned* Bt = reinterpret_cast<const unsigned*>(Bs);
            const int ar = wr * 32 + li, br = wc * 32 + li;
            const int ag = (ar >> 2) & 7, bg = (br >> 2) & 7;
#pragma unroll
            for (int s2 = 0; s2 < DW_KS / 16; ++s2) {
                const int q = 2 * s2 + lh;
                bf16x8 ah = *reinterpret_cast<const bf16x8*>(&At[ar * DWT_LD + ((q ^ ag) << 2)]);
                bf16x8 bh = *reinterpret_cast<const bf16x8*>(&Bt[br * DWT_LD + ((q ^ bg) << 2)]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
                if constexpr (PREC == 2) {
                    bf16x8 al = *reinterpret_cast<const bf16x8*>(&At[(64 + ar) * DWT_LD + ((q ^ ag) << 2)]);
                    bf16x8 bl = *reinterpret_cast<const bf16x8*>(&Bt[(64 + br) * DWT_LD + ((q ^ bg) << 2)]);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    const int col = n0 + wc * 32 + li;
    if (col < Cout) {
        float* dst = part ? part + (long long)chunk * K3 * Cin * Cout : dW;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            int m = wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            if (c0 + m < Cin) {
                float* e = &dst[((long long)k * Cin + c0 + m) * Cout + col];
                if (part) *e = acc[reg]; else atomicAdd(e, acc[reg]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The 3-channel grid-probing stem forward with bf16 / split-bf16x3 operands (operand modes "bf16" / "bf16x3" only: the
// fp32 mode keeps k_spconv_fwd3).  Five sixths of the dense-over-offsets products multiply absent neighbours; on
// v_mfma_f32_32x32x16_bf16 those zeros cost 1/16 (bf16) or 3/16 (bf16x3) of what they cost the fp32 MFMA.  With the MFMA
// cheap the staging is what counts: 16 offsets x 4 (padded) channels per 64-deep K-chunk, so that a gathered row goes to LDS
// as ONE 8-byte write of four bf16 (21 x 3 packed channels needed three 2-byte writes per pair: 550 us; this form: see
// DESIGN.md section 5).  Same probing, same by-product map, same software pipeline (indices two chunks ahead, rows and
// weights one chunk ahead) as k_spconv_fwd3<true>.
template <bool X3, bool Y16 = false>
__global__ __launch_bounds__(256) void k_spconv_fwd3_lp(const float* __restrict__ X, int ldx,
                                                        const float* __restrict__ W,  // [K3*3, Cout]
                                                        const float* __restrict__ bias, float* __restrict__ Y, int ldy,
                                                        int n_out, int K3, int Cout, GridProbe gp) {
    constexpr int OPC = 16, NP = X3 ? 2 : 1, NJ = 4, NW = 4;
    __shared__ __attribute__((aligned(16))) unsigned short As[NP][BM * LLD];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[NP][BN * LLD];
    __shared__ int s_delta[736];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int row0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int li = lane & 31, lh = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int k = tid; k < K3; k += 256) s_delta[k] = probe_delta(gp, k);
    const int myrow = row0 + (tid & (BM - 1));
    const bool row_ok = myrow < n_out;
    const int base = probe_base(gp, min(myrow, n_out - 1));
    __syncthreads();
    const int nchunks = (K3 + OPC - 1) / OPC;
    int idxn[NJ], idxc[NJ];
    f32x4 xv[NJ];
    float4 wv[NW];
    auto load_idx = [&](int ch) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int off = (tid + 256 * j) >> 6;                       // 0 .. 15
            const int k = min(ch * OPC + off, K3 - 1);
            idxn[j] = gp.grid[base + s_delta[k]];
        }
    };
    auto gather = [&](int ch) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            idxc[j] = idxn[j] == INT_MAX ? -1 : idxn[j];
            xv[j] = *reinterpret_cast<const f32x4*>(X + (long long)max(idxc[j], 0) * ldx);
            if (gp.nbr_out && blockIdx.y == 0) {
                const int off = (tid + 256 * j) >> 6, k = ch * OPC + off;
                if (off < OPC && k < K3 && row_ok) gp.nbr_out[(long long)k * gp.nbr_out_stride + myrow] = idxc[j];
            }
        }
    };
    // weights of a chunk: K row kr = 4 * offset + channel (channel 3: zero), float4 pieces along the 64 columns
    auto load_w = [&](int ch) {
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int e = tid + 256 * j, kr = e >> 4, c4 = (e & 15) * 4;     // kr 0 .. 63
            const long long wrow = min(((long long)ch * OPC + (kr >> 2)) * 3 + min(kr & 3, 2), (long long)K3 * 3 - 1);
            wv[j] = *reinterpret_cast<const float4*>(W + wrow * Cout + min(n0 + c4, Cout - 4));
        }
    };
    auto put = [&](unsigned short (*T)[BM * LLD], int at, float v) {
        const unsigned short h = f2bf(v);
        T[0][at] = h;
        if (X3) T[NP - 1][at] = f2bf(v - bf2f(h));
    };
    load_idx(0);
    gather(0);
    load_w(0);
    load_idx(nchunks > 1 ? 1 : 0);
    for (int ch = 0; ch < nchunks; ++ch) {
        const int k0 = ch * OPC;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int e = tid + 256 * j, off = e >> 6, r = e & (BM - 1);
            const bool ok = idxc[j] >= 0 && k0 + off < K3 && row0 + r < n_out;
            asm volatile("" : "+v"(xv[j]));
            // (x rows are 4 floats wide with a zero in the fourth: the padded channel multiplies a zero weight row anyway)
            const float4 v = ok ? make_float4(xv[j][0], xv[j][1], xv[j][2], 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
            stage_bf16<X3>(As[0], As[NP - 1], r * LLD + off * 4, v);
        }
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int e = tid + 256 * j, kr = e >> 4, c4 = (e & 15) * 4;
            const bool ok = (kr & 3) < 3 && k0 + (kr >> 2) < K3 && n0 + c4 < Cout;
            const float w4[4] = {wv[j].x, wv[j].y, wv[j].z, wv[j].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) put(Bs, (c4 + c) * LLD + kr, ok ? w4[c] : 0.f);
        }
        __syncthreads();
        gather(min(ch + 1, nchunks - 1));
        load_w(min(ch + 1, nchunks - 1));
        load_idx(min(ch + 2, nchunks - 1));
        const int aoff = (wr * 32 + li) * LLD + 8 * lh, boff = (wc * 32 + li) * LLD + 8 * lh;
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&As[0][aoff + 16 * s2]);
            const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&Bs[0][boff + 16 * s2]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
            if (X3) {
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(&As[NP - 1][aoff + 16 * s2]);
                const bf16x8 bl = *reinterpret_cast<const bf16x8*>(&Bs[NP - 1][boff + 16 * s2]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
            }
        }
        __syncthreads();
    }
    const int col = n0 + wc * 32 + li;
    if (col < Cout) {
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = row0 + wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            if (row >= n_out) continue;
            if (Y16) st1(reinterpret_cast<bf16_t*>(Y) + (long long)row * ldy + col, acc[reg] + bv);   // (bf16 rows, ldy in bf16)
            else Y[(long long)row * ldy + col] = acc[reg] + bv;
        }
    }
}

// Weight gradient of the small-Cin (stem) path, ROW-COMPACTED: an M tile covers 64/CPAD kernel offsets; a row whose
// neighbours at ALL of those offsets are absent contributes nothing (7^3 stem map: 35 % of the rows of a 16-offset
// tile).  The workgroup compacts its row chunk to the rows with at least one present neighbour, then walks them 32 at a
// time.  Same XCD-aware 1-D grid as k_spconv_dw_cmp.
template <int CPAD>
__global__ __launch_bounds__(256) void k_spconv_dw_small_cmp(const float* __restrict__ X, int ldx,
                                                             const float* __restrict__ dY, int ldy,
                                                             const int32_t* __restrict__ nbr, long long nbr_stride,
                                                             float* __restrict__ dW, int n_out, int K3, int Cout,
                                                             int rows_per_wg, int chunks, int m_tiles, int nsub,
                                                             float* __restrict__ part) {
    // nsub > 1: the workgroup walks nsub consecutive sub-chunks of rows_per_wg rows with ONE set of accumulators (`chunks`
    // counts the groups of nsub sub-chunks).  part != NULL: it leaves its 64 x 64 tile in part[chunk] instead of adding it
    // to dW with fp32 atomics; k_dw_fold_small adds the chunks in ascending order (bitwise reproducible, and a two-level
    // sum: the 343 x 3 x 64 stem gradient sums 22.7 M cancelling terms behind a BatchNorm).
    constexpr int OPT = 64 / CPAD;   // offsets per M tile
    constexpr int F4 = CPAD / 4;     // float4 per (row, offset)
    __shared__ __attribute__((aligned(16))) float As[BK * 64];  // [r][m]
    __shared__ __attribute__((aligned(16))) float Bs[BK * 64];  // [r][n]
    __shared__ int p_row[DW_MAXROWS];
    __shared__ int s_wcnt[2][4];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int n0 = blockIdx.y * 64;
    int mt, chunk;
    if (chunks >= 16) {
        const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
        mt = jx % m_tiles;
        chunk = (jx / m_tiles) * 8 + xcd;
        if (chunk >= chunks) return;
    } else {
        chunk = blockIdx.x % chunks;
        mt = blockIdx.x / chunks;
    }
    const int k0 = mt * OPT;
    const int nk = min(OPT, K3 - k0);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int sub = 0; sub < nsub; ++sub) {
    const int r_begin = (chunk * nsub + sub) * rows_per_wg;
    const int r_end = min(n_out, r_begin + rows_per_wg);
    if (r_begin >= n_out) break;          // (uniform)
    if (sub > 0) __syncthreads();         // the previous sub-chunk's row list and staging tiles are free

    // ---- phase 1: rows with at least one present neighbour among the tile's offsets (order kept).  The neighbour
    // indices of the whole chunk are loaded up front (unconditional, clamped: one exposed latency per workgroup instead
    // of one per 256 rows, as in k_spconv_dw_cmp)
    int total = 0;
    const int nrows = r_end - r_begin;
    constexpr int NIT = 4;                 // rows_per_wg <= 1024 on this path
    bool anyv[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int rc = min(r_begin + it * 256 + tid, n_out - 1);
        int mx = -1;                        // >= 0 iff a neighbour exists at one of the offsets (absent = -1)
#pragma unroll
        for (int o = 0; o < OPT; ++o) mx = max(mx, nbr[(long long)min(k0 + o, K3 - 1) * nbr_stride + rc]);
        anyv[it] = mx >= 0 && it * 256 + tid < nrows;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        if (it * 256 >= nrows) break;
        const int i = it * 256 + tid;
        const bool any = anyv[it];
        const unsigned long long bal = __ballot(any);
        if (lane == 0) s_wcnt[it & 1][wave] = __popcll(bal);
        __syncthreads();
        int off = total, round = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            int c = s_wcnt[it & 1][w];
            if (w < wave) off += c;
            round += c;
        }
        if (any) p_row[off + __popcll(bal & ((1ull << lane) - 1ull))] = r_begin + i;
        total += round;
    }
    __syncthreads();
    if (total == 0) continue;

    // software pipeline: the neighbour indices of step i+2 and the gathers of step i+1 are in flight during the MFMAs
    // of step i (index -> gather is a dependent pair of L2 latencies; one step of MFMAs is only ~0.4 us)
    float4 a_reg[2], b_reg[2];
    int idx_cur[2], idx_nxt[2];
    auto load_idx = [&](int pb, int* idx) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int e = tid + 256 * j;          // A: 32 rows x 16 float4 columns (offset, channel quad)
            int p = pb + (e & 31);
            int o = (e >> 5) / F4;
            idx[j] = (p < total && o < nk) ? nbr[(long long)(k0 + o) * nbr_stride + p_row[p]] : -1;
        }
    };
    auto load_data = [&](int pb, const int* idx) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int f = ((tid + 256 * j) >> 5) % F4;
            a_reg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx[j] >= 0) a_reg[j] = *reinterpret_cast<const float4*>(X + (long long)idx[j] * ldx + f * 4);
            // B: 32 rows x 64 output channels
            int pr = pb + (tid >> 4) + 16 * j;
            int n = n0 + (tid & 15) * 4;
            b_reg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pr < total && n < Cout) b_reg[j] = *reinterpret_cast<const float4*>(dY + (long long)p_row[pr] * ldy + n);
        }
    };
    load_idx(0, idx_cur);
    load_idx(BK, idx_nxt);
    load_data(0, idx_cur);
    for (int pb = 0; pb < total; pb += BK) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int e = tid + 256 * j;
            *reinterpret_cast<float4*>(&As[(e & 31) * 64 + (e >> 5) * 4]) = a_reg[j];
            *reinterpret_cast<float4*>(&Bs[((tid >> 4) + 16 * j) * 64 + (tid & 15) * 4]) = b_reg[j];
        }
        __syncthreads();
        if (pb + BK < total) {
            idx_cur[0] = idx_nxt[0];
            idx_cur[1] = idx_nxt[1];
            load_data(pb + BK, idx_cur);
            load_idx(pb + 2 * BK, idx_nxt);
        }
        const float* ap = &As[wr * 32 + li];
        const float* bp = &Bs[wc * 32 + li];
#pragma unroll
        for (int s2 = 0; s2 < BK / 2; ++s2) {
            float av = ap[(2 * s2 + lh) * 64];
            float bv = bp[(2 * s2 + lh) * 64];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    }   // sub-chunks
    const int col = n0 + wc * 32 + li;
    if (col < Cout) {
        float* dst = part ? part + (long long)chunk * K3 * CPAD * Cout : dW;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            int m = wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            long long wrow = (long long)k0 * CPAD + m;
            if (wrow < (long long)K3 * CPAD) {
                if (part) dst[wrow * Cout + col] = acc[reg];
                else atomicAdd(&dst[wrow * Cout + col], acc[reg]);
            }
        }
    }
}

// dW[e] += sum over chunks (ascending) of part[chunk][e]
__global__ __launch_bounds__(256) void k_dw_fold_small(const float4* __restrict__ part, int chunks, long long n4,
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

// =============================================================== C ABI
extern "C" {
int agb_spconv_bwd_weight_ws(const float* X, int ldx, const float* dY, int ldy, const int32_t* nbr, long long nbr_stride,
                             float* dW, int n_out, int K3, int Cin, int Cout, int precision, int variant, void* workspace,
                             size_t workspace_bytes, void* stream);
int agb_spconv_fwd3_grid_lp(const float* X, int ldx, const float* W, const int32_t* coords, const int32_t* grid,
                            const int32_t* desc, int K, const float* bias, float* Y, int ldy, int n_out, int Cout,
                            int32_t* nbr_out, long long nbr_out_stride, int precision, void* stream);

// tile height of the generic kernel: 128 rows when there are plenty of tiles anyway, or when the layer is W-heavy
// (Cin >= 256: every workgroup streams Cin x 64 weights per offset, so halving the number of row tiles halves the
// dominant L2 traffic); else 64
static int conv_tile_rows(int n_out, int Cin, int Cout) {
    long long t128 = (long long)agb_cdiv(n_out, 128) * agb_cdiv(Cout, BN);
    if (t128 >= 1024) return 128;
    if (Cin >= 256 && n_out >= 1024) return 128;
    return 64;
}

// Input-channel split the pair-compacted kernel wants for this shape (1, 2, 4, ... dividing Cin/64), 0 when the layer
// should take the register-accumulator kernels: enough waves to fill the 1024 resident-wave slots.
static int cmp_want_split(int n_out, int Cin, int Cout) {
    if (Cin % 64 != 0 || Cout % 4 != 0 || n_out <= 0) return 0;
    const long long waves128 = (long long)agb_cdiv(n_out, 128) * agb_cdiv(Cout, 64);
    const int nsb = Cin / 64;
    int sp = 1;
    while (waves128 * sp < 768 && sp * 2 <= nsb && nsb % (sp * 2) == 0) sp *= 2;
    return waves128 * sp >= 384 ? sp : 0;
}

// Rows per wave of the pair-compacted kernel, or 0 when the layer should take the register-accumulator kernels
// (a.cmp_mode: 1 = automatic, 0 = never, 64 / 128 = always with that tile height — tests, tuning; a per-call argument,
// the library keeps no state).
static int cmp_rows(const ConvArgs& a) {
    const int mode = a.cmp_mode;
    if (mode == 0 || a.perm || a.Cin % 64 != 0 || a.ldx % 4 != 0 || a.ldy % 4 != 0 || a.Cout % 4 != 0) return 0;
    if (a.ksplit < 1 || (a.Cin / 64) % a.ksplit != 0 || (a.ksplit > 1 && a.partial == nullptr)) return 0;
    if (mode == 64 || mode == 128) return mode;
    if (mode == 129) return 128;
    // measured on the SENet14 pyramid (tools/bench_conv.py): 128-row tiles win from ~400 waves up (64->64 at 210 k rows
    // 351 vs 497 us, 128->128 at 61 k rows 438 vs 566 us, 256->256 at 14 k rows 477 vs 519 us); few-row wide layers
    // reach that by splitting the input channels (the caller sizes `partial` from agb_spconv_split_hint)
    const long long waves = (long long)agb_cdiv(a.n_out, 128) * agb_cdiv(a.Cout, 64) * a.ksplit;
    return waves >= 384 ? 128 : 0;
}

// Tile geometry of the pair-compacted kernel: equal-cost tiles, workgroup count a multiple of the resident-wave
// capacity (LDS: 4 / 8 waves per CU) so that the last round of workgroups is not half empty.
// Interleave block (log2 rows) of a level with n rows; request: -1 = by level size, else the requested shift.
// Measured (tools/bench_conv.py --il 0,2,3,4, us per launch, contiguous / 4-row / 8-row / 16-row blocks): 64->64 at 211 k
// rows 377 / 326 / 308 / 320; 128->128 at 61 k rows 432 / 387 / 396 / 424; 256->256 at 14 k rows 416 / 390 / 401 / 452;
// 512->512 at 2.9 k rows 289 / - / 311 / 302 (few-row levels: contiguous).
static int cmp_interleave(int n, int request) {
    if (n < 8192) return 0;
    if (request >= 0 && request <= 5) return request;
    return n >= 100000 ? 3 : 2;
}

static void cmp_geometry(const ConvArgs& a, int* R_out, int* rpt_out, int* ntiles_out, int* nct_out, int* il_out) {
    const int R = cmp_rows(a), nct = agb_cdiv(a.Cout, 64) * 1;
    const long long per_tile = (long long)nct * a.ksplit;
    const long long slots = (R == 128 ? 4 : 8) * 256;
    const long long rounds = ((long long)agb_cdiv(a.n_out, R) * per_tile + slots - 1) / slots;
    long long want_tiles = rounds * slots / per_tile;
    if (want_tiles < 1) want_tiles = 1;
    const int il = cmp_interleave(a.n_out, a.cmp_il);
    *R_out = R; *nct_out = nct; *il_out = 0;
    if (il > 0) {
        // tiles of `bpt` row blocks taken ntiles blocks apart
        const int bs = 1 << il, nblk = agb_cdiv(a.n_out, bs);
        int bpt = agb_cdiv(nblk, want_tiles);
        if (bpt > R / bs) bpt = R / bs;
        if (bpt >= 2) {
            *rpt_out = bpt * bs; *ntiles_out = agb_cdiv(nblk, bpt); *il_out = il;
            return;
        }
    }
    int rpt = agb_cdiv(a.n_out, want_tiles);
    if (rpt > R) rpt = R;
    if (rpt < 16) rpt = 16;
    *rpt_out = rpt; *ntiles_out = agb_cdiv(a.n_out, rpt);
}

static int launch_conv(const ConvArgs& a, int n_tiles_perm, hipStream_t s) {
    const bool small = (a.Cin == 4 || a.Cin == 8);
    dim3 block(256);
    if (a.Cin == 3) {
        if (a.perm || a.ksplit > 1) {
            agb_set_error("agb_spconv_fwd_ex: the packed 3-channel path takes neither a class partition nor a split");
            return AGB_EUNSUPPORTED;
        }
        AGB_LAUNCH(k_spconv_fwd3<false>, dim3(agb_cdiv(a.n_out, BM), agb_cdiv(a.Cout, BN)), block, 0, s, a.X, a.ldx,
                           a.W, a.nbr, a.nbr_stride, a.kflip, a.bias, a.Y, a.ldy, a.n_out, a.K3, a.Cout, GridProbe{});
        return AGB_OK;
    }
    if (a.perm) {
        if (small) {
            agb_set_error("agb_spconv_fwd_ex: the class-partitioned path needs Cin >= 12");
            return AGB_EUNSUPPORTED;
        }
        dim3 grid(n_tiles_perm, agb_cdiv(a.Cout, BN), a.ksplit);
        AGB_LAUNCH((k_spconv_pipe<64, true, 64>), grid, block, 0, s, a);
    } else if (small) {
        dim3 grid(agb_cdiv(a.n_out, BM), agb_cdiv(a.Cout, BN));
        if (a.Cin == 4)
            AGB_LAUNCH(k_spconv_fwd<4>, grid, block, 0, s, a.X, a.ldx, a.W, a.nbr, a.nbr_stride, a.kflip, a.bias,
                               a.Y, a.ldy, a.n_out, a.K3, a.Cin, a.Cout);
        else
            AGB_LAUNCH(k_spconv_fwd<8>, grid, block, 0, s, a.X, a.ldx, a.W, a.nbr, a.nbr_stride, a.kflip, a.bias,
                               a.Y, a.ldy, a.n_out, a.K3, a.Cin, a.Cout);
    } else if (a.nbr != nullptr && cmp_rows(a) > 0) {   // (a dense product has no absent pairs to skip)
        int R, rpt, ntiles, nct, il;
        cmp_geometry(a, &R, &rpt, &ntiles, &nct, &il);
        dim3 grid(8 * agb_cdiv(ntiles, 8) * nct * a.ksplit), blk(64);
        // 128-row tiles: the hand-scheduled kernel (k_spconv_cma); cmp_mode 129 forces its C++ twin k_spconv_cmpt (same
        // sums bit for bit: tests), maps of fewer than three offsets keep the first-generation kernel
        if (R == 128 && a.cmp_mode == 129) AGB_LAUNCH((k_spconv_cmpt<128>), grid, blk, 0, s, a, ntiles, nct, rpt, a.ksplit, il);
        else if (R == 128 && a.K3 >= 3) AGB_LAUNCH(k_spconv_cma, grid, blk, 0, s, a, ntiles, nct, rpt, a.ksplit, il);
        else if (R == 128) AGB_LAUNCH((k_spconv_cmp<128>), grid, blk, 0, s, a, ntiles, nct, rpt, a.ksplit, il);
        else AGB_LAUNCH((k_spconv_cmp<64>), grid, blk, 0, s, a, ntiles, nct, rpt, a.ksplit, il);
    } else if (conv_tile_rows(a.n_out, a.Cin, a.Cout) == 128) {
        // 128-row tiles: the W tile is reused by twice as many rows (layers with many rows, or W-heavy layers);
        // 128-column tiles on top where the layer is wide and still fills the chip (the A tile — re-read from L2 once
        // per column tile — serves twice the columns: the dense products are bound by that traffic)
        const bool wide = a.Cout >= 128 && (long long)agb_cdiv(a.n_out, 128) * agb_cdiv(a.Cout, 128) * a.ksplit >= 512;
        if (wide)
            AGB_LAUNCH((k_spconv_pipe<128, false, 128>),
                               dim3(agb_cdiv(a.n_out, 128), agb_cdiv(a.Cout, 128), a.ksplit), block, 0, s, a);
        else
            AGB_LAUNCH((k_spconv_pipe<128, false, 64>), dim3(agb_cdiv(a.n_out, 128), agb_cdiv(a.Cout, BN), a.ksplit),
                               block, 0, s, a);
    } else {
        AGB_LAUNCH((k_spconv_pipe<64, false, 64>), dim3(agb_cdiv(a.n_out, 64), agb_cdiv(a.Cout, BN), a.ksplit),
                           block, 0, s, a);
    }
    if (a.ksplit > 1) {
        long long total = (long long)a.n_out * (a.Cout / 4);
        hipLaunchKernelGGL(k_split_reduce<float>, dim3(agb_cdiv(total, 256)), dim3(256), 0, s, a.partial, a.ksplit, a.n_out,
                           a.Cout / 4, a.bias, a.Y, a.ldy, a.addend, a.ld_add);
    }
    return AGB_OK;
}

#ifdef AGB_TIMELINE
// out: unsigned long long[4 * 8192] = (start, end [10 ns], shader clocks, groups) per workgroup of the last k_spconv_cmp
// launches (slot = blockIdx.x); reset != 0 clears the table afterwards.
int agb_debug_cmp_timeline(unsigned long long* out, int reset) {
    if (out) (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cmp_timeline), sizeof(unsigned long long) * 4 * AGB_TIMELINE_SLOTS);
    if (reset) {
        void* p = nullptr;
        (void)hipGetSymbolAddress(&p, HIP_SYMBOL(g_cmp_timeline));
        (void)hipMemset(p, 0, sizeof(unsigned long long) * 4 * AGB_TIMELINE_SLOTS);
    }
    return AGB_OK;
}
#endif

#ifdef AGB_TIMELINE
// out: unsigned long long[16] = stamp sums of the diagnostic build of k_spconv_cma (tools/cma_stamps.py)
int agb_debug_cmpt_stamps(unsigned long long* out, int reset) {
    if (out) (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cmpt_stamp), sizeof(unsigned long long) * 16);
    if (reset) {
        void* p = nullptr;
        (void)hipGetSymbolAddress(&p, HIP_SYMBOL(g_cmpt_stamp));
        (void)hipMemset(p, 0, sizeof(unsigned long long) * 16);
    }
    return AGB_OK;
}
#endif

// Resident workgroups per CU of the pair-compacted kernel (R = 64 / 128), as the runtime computes it (tuning aid).
int agb_spconv_cmp_occupancy(int R) {
    int n = -1;
    hipError_t e = R == 128
        ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)k_spconv_cmp<128>, 64, 0)
        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)k_spconv_cmp<64>, 64, 0);
    return e == hipSuccess ? n : -1;
}

// WT [K3][C][R] = transpose of W [K3][R][C] per offset (the weights the data gradient multiplies by); R, C % 4 == 0.
int agb_spconv_weight_transpose_z(const float* W, float* WT, float* zero, int K3, int R, int C, void* stream) {
    AGB_CHECK_ARG(K3 >= 1 && R >= 4 && C >= 4 && R % 4 == 0 && C % 4 == 0 && K3 <= 65535,
                  "agb_spconv_weight_transpose: K3 %d, R %d, C %d (multiples of 4)", K3, R, C);
    hipLaunchKernelGGL(k_weight_transpose, dim3(agb_cdiv(C, 64), agb_cdiv(R, 64), K3), dim3(256), 0,
                       (hipStream_t)stream, W, WT, R, C, zero);
    AGB_CHECK_LAUNCH("agb_spconv_weight_transpose");
    return AGB_OK;
}

// W fp32 [K3][R][C] -> W16 uint16 [K3][R][C] and Wt16 uint16 [K3][C][R] (bf16, round to nearest even): the K-major operand
// forms of the data gradient and of the forward pass of agb_spconv_fwd_b16 / _h, in one launch
int agb_weight_twins_bf16(const float* W, int K3, int R, int C, uint16_t* W16, uint16_t* Wt16, void* stream) {
    AGB_CHECK_ARG(K3 >= 1 && R >= 4 && C >= 4 && R % 4 == 0 && C % 4 == 0 && K3 <= 65535,
                  "agb_weight_twins_bf16: K3 %d, R %d, C %d (multiples of 4)", K3, R, C);
    AGB_CHECK_ARG(W && W16 && Wt16, "agb_weight_twins_bf16: W, W16 and Wt16 are required");
    hipLaunchKernelGGL(k_weight_twins, dim3(agb_cdiv(C, 64), agb_cdiv(R, 64), K3), dim3(256), 0, (hipStream_t)stream, W, W16,
                       Wt16, R, C);
    AGB_CHECK_LAUNCH("agb_weight_twins_bf16");
    return AGB_OK;
}

// Both bf16 operand forms of EVERY layer in one launch: tab (DEVICE) int64 [n][7] = (W, W16, Wt16, K3, R, C, first tile), first
// tile as in agb_spconv_weight_transpose_batched.  Values identical to agb_weight_twins_bf16 layer by layer.
int agb_weight_twins_batched(const long long* tab, int n, long long total_tiles, void* stream) {
    AGB_CHECK_ARG(tab != nullptr && n >= 1 && total_tiles >= 1 && total_tiles <= 0x7fffffffLL,
                  "agb_weight_twins_batched: %d layers, %lld tiles", n, total_tiles);
    hipLaunchKernelGGL(k_weight_twins_batched, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, tab, n);
    AGB_CHECK_LAUNCH("agb_weight_twins_batched");
    return AGB_OK;
}

// Every layer of a model in ONE launch: tab (DEVICE) int64 [n][6] = (W, WT, K3, R, C, first tile) per layer with
// first tile = the running sum of K3 * ceil(R / 64) * ceil(C / 64) over the layers before it; total_tiles = that sum over
// all layers.  R, C multiples of 4.  Values identical to agb_spconv_weight_transpose layer by layer.
int agb_spconv_weight_transpose_batched(const long long* tab, int n, long long total_tiles, void* stream) {
    AGB_CHECK_ARG(tab != nullptr && n >= 1 && total_tiles >= 1 && total_tiles <= 0x7fffffffLL,
                  "agb_spconv_weight_transpose_batched: %d layers, %lld tiles", n, total_tiles);
    hipLaunchKernelGGL(k_weight_transpose_batched, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, tab, n);
    AGB_CHECK_LAUNCH("agb_spconv_weight_transpose_batched");
    return AGB_OK;
}

int agb_spconv_weight_transpose(const float* W, float* WT, int K3, int R, int C, void* stream) {
    return agb_spconv_weight_transpose_z(W, WT, nullptr, K3, R, C, stream);
}

// How many offset splits a layer of n_out rows wants (1 = none): host helper for sizing `partial`.
int agb_spconv_split_hint_opt(int n_out, int K3, int Cin, int Cout, int cmp_mode) {
    if (Cin == 3 || Cin == 4 || Cin == 8 || K3 < 8 || n_out <= 0) return 1;
    const int sp = cmp_mode == 0 ? 0 : cmp_want_split(n_out, Cin, Cout);
    if (sp > 0) return sp;   // the pair-compacted kernel takes the layer, input channels split sp ways
    if (cmp_mode == 64 || cmp_mode == 128 || cmp_mode == 129) return 1;
    long long tiles = (long long)agb_cdiv(n_out, conv_tile_rows(n_out, Cin, Cout)) * agb_cdiv(Cout, BN);
    if (tiles >= 768) return 1;
    long long s = (1024 + tiles - 1) / tiles;
    if (s > 8) s = 8;
    return (int)(s < 1 ? 1 : s);
}

// The same for the identity map (nbr == NULL: a dense [n, Cin] x [Cin, Cout] product): the reduction dimension is split
// over workgroups when the output tiles alone leave most of the chip idle and Cin is long enough to pay for the partial
// buffer (KPConv's deepest level: 2860 rows x 3840 -> 256 is 92 tiles of a 256-CU chip; 0.19 -> 0.07 ms with 8 splits).
int agb_dense_split_hint(int n_out, int Cin, int Cout) {
    if (Cin < 512 || n_out <= 0) return 1;
    long long tiles = (long long)agb_cdiv(n_out, conv_tile_rows(n_out, Cin, Cout)) * agb_cdiv(Cout, BN);
    if (tiles >= 512) return 1;
    long long s = (1024 + tiles - 1) / tiles, cap = Cin / 128;      // at least 128 input channels per split
    if (s > 8) s = 8;
    if (s > cap) s = cap;
    return (int)(s < 1 ? 1 : s);
}

// Row tiles of a dense fp32 product that can deliver partial BatchNorm statistics from its epilogue
// (agb_dense_fwd_bn), 0 when this shape takes a path that cannot (streaming kernels, split reduction).
int agb_dense_bn_chunks(int n_out, int Cin, int Cout) {
    if (n_out <= 0 || Cin < 12 || Cin % 4 != 0 || Cout % 4 != 0) return 0;
    if (agb_dense_stream_ok(n_out, Cin, Cout) || agb_dense_split_hint(n_out, Cin, Cout) > 1) return 0;
    return agb_cdiv(n_out, conv_tile_rows(n_out, Cin, Cout));
}

// Y = X W + bias (identity map, fp32) and bn_part float[agb_dense_bn_chunks][3][Cout] = (count, mean, M2) of every
// column of Y over each row tile: the input of agb_bn_stats_fold.
int agb_dense_fwd_bn(const float* X, int ldx, const float* W, const float* bias, float* Y, int ldy, int n_out, int Cin,
                     int Cout, float* bn_part, void* stream) {
    AGB_CHECK_ARG(bn_part != nullptr && agb_dense_bn_chunks(n_out, Cin, Cout) > 0,
                  "agb_dense_fwd_bn: no statistics from this shape (n %d, Cin %d, Cout %d): see agb_dense_bn_chunks", n_out,
                  Cin, Cout);
    AGB_CHECK_ARG(ldx % 4 == 0 && ldy >= Cout, "agb_dense_fwd_bn: ldx %d (a multiple of 4), ldy %d", ldx, ldy);
    ConvArgs a{X, ldx, W, nullptr, 0, 0, bias, Y, ldy, n_out, 1, Cin, Cout, nullptr, nullptr, nullptr, 1, nullptr, 0, -1,
               bn_part};
    int rc = launch_conv(a, 0, (hipStream_t)stream);
    if (rc) return rc;
    AGB_CHECK_LAUNCH("agb_dense_fwd_bn");
    return AGB_OK;
}

int agb_spconv_split_hint(int n_out, int K3, int Cin, int Cout) {
    return agb_spconv_split_hint_opt(n_out, K3, Cin, Cout, 1);
}

// agb_spconv_fwd_opt with an ADDEND: Y = addend + (bias + sum) — the second gradient of a residual join added in the
// final store of the data-gradient kernel instead of by a separate pass (addend float [n_out][ld_add], may alias Y; fp32
// kernels; with a split reduction the addend joins in k_split_reduce).  Internal: behind agb_spconv_bwd_data and csrc/net.hip.
AGB_INTERNAL int agb_spconv_fwd_opt_add(const float* X, int ldx, const float* W, const int32_t* nbr, long long nbr_stride,
                                        int kflip, const float* bias, float* Y, int ldy, int n_out, int K3, int Cin, int Cout,
                                        const int32_t* perm, const int32_t* tile_cls, const int32_t* cls_tab, int n_tiles,
                                        int ksplit, float* partial, int cmp_mode, int cmp_interleave_shift,
                                        const float* addend, int ld_add, void* stream);

int agb_spconv_fwd_opt(const float* X, int ldx, const float* W, const int32_t* nbr, long long nbr_stride, int kflip,
                       const float* bias, float* Y, int ldy, int n_out, int K3, int Cin, int Cout,
                       const int32_t* perm, const int32_t* tile_cls, const int32_t* cls_tab, int n_tiles, int ksplit,
                       float* partial, int cmp_mode, int cmp_interleave_shift, void* stream) {
    return agb_spconv_fwd_opt_add(X, ldx, W, nbr, nbr_stride, kflip, bias, Y, ldy, n_out, K3, Cin, Cout, perm, tile_cls,
                                  cls_tab, n_tiles, ksplit, partial, cmp_mode, cmp_interleave_shift, nullptr, 0, stream);
}

int agb_spconv_fwd_opt_add(const float* X, int ldx, const float* W, const int32_t* nbr, long long nbr_stride, int kflip,
                           const float* bias, float* Y, int ldy, int n_out, int K3, int Cin, int Cout,
                           const int32_t* perm, const int32_t* tile_cls, const int32_t* cls_tab, int n_tiles, int ksplit,
                           float* partial, int cmp_mode, int cmp_interleave_shift, const float* addend, int ld_add,
                           void* stream) {
    AGB_CHECK_ARG(cmp_mode == 0 || cmp_mode == 1 || cmp_mode == 64 || cmp_mode == 128 || cmp_mode == 129,
                  "agb_spconv_fwd_opt: cmp_mode %d (1 automatic, 0 never, 64 / 128 forced, 129: 128 with the C++ twin of the "
                  "hand-scheduled kernel)", cmp_mode);
    AGB_CHECK_ARG(cmp_interleave_shift >= -1 && cmp_interleave_shift <= 5, "agb_spconv_fwd_opt: interleave shift %d "
                  "(-1: by level size)", cmp_interleave_shift);
    AGB_CHECK_ARG(n_out >= 0 && K3 >= 1 && Cin >= 1 && Cout >= 1, "agb_spconv_fwd: bad sizes");
    AGB_CHECK_ARG((Cin % 4 == 0 || Cin == 3) && Cout % 4 == 0 && ldx % 4 == 0 && ldy >= Cout,
                  "agb_spconv_fwd: Cin (%d: a multiple of 4, or 3 with 4-float rows), Cout (%d), ldx (%d) must be "
                  "multiples of 4 (pad small inputs)", Cin, Cout, ldx);
    AGB_CHECK_ARG(nbr == nullptr || nbr_stride >= n_out, "agb_spconv_fwd: nbr_stride < n_out");
    AGB_CHECK_ARG(nbr != nullptr || (K3 == 1 && perm == nullptr && Cin >= 12),
                  "agb_spconv_fwd: the identity map (nbr == NULL: dense 1x1 stride-1 product) needs K3 == 1, Cin >= 12");
    AGB_CHECK_ARG(ksplit >= 1 && (ksplit == 1 || partial != nullptr), "agb_spconv_fwd_ex: ksplit needs `partial`");
    AGB_CHECK_ARG(perm == nullptr || (tile_cls != nullptr && cls_tab != nullptr && n_tiles > 0),
                  "agb_spconv_fwd_ex: perm needs tile_cls, cls_tab and n_tiles");
    AGB_CHECK_ARG(addend == nullptr || (ld_add >= Cout && ld_add % 4 == 0 && Cin >= 12),
                  "agb_spconv_fwd: an addend needs ld_add >= Cout, a multiple of 4, and Cin >= 12 (ld_add %d)", ld_add);
    if (n_out == 0) return AGB_OK;
    // HBM-bound dense products (many rows, a weight matrix that fits LDS): the streaming kernels of dense_stream.hip
    if (nbr == nullptr && ksplit == 1 && addend == nullptr && agb_dense_stream_ok(n_out, Cin, Cout)) {
        int rc = agb_dense_stream_launch(X, ldx, W, bias, Y, ldy, n_out, Cin, Cout, (hipStream_t)stream);
        if (rc) return rc;
        AGB_CHECK_LAUNCH("agb_spconv_fwd (dense, streaming)");
        return AGB_OK;
    }
    ConvArgs a{X, ldx, W, nbr, nbr_stride, kflip, bias, Y, ldy, n_out, K3, Cin, Cout, perm, tile_cls, cls_tab, ksplit,
               partial, cmp_mode, cmp_interleave_shift};
    a.addend = addend; a.ld_add = ld_add;
    int rc = launch_conv(a, n_tiles, (hipStream_t)stream);
    if (rc) return rc;
    AGB_CHECK_LAUNCH("agb_spconv_fwd");
    return AGB_OK;
}

// The data gradient of the generalized sparse convolution under its own name:
//   dX[q] = [addend[q] +] sum_k dY[map[k][q]] W[k]^T
// map: the transposed kernel map (kflip 0; strided layers: with the class partition perm / tile_cls / cls_tab of
// agb_parity_partition) or the forward map of a stride-1 odd kernel read backwards (kflip 1).  Wt [K3][Cout][Cin]: the
// per-offset transposes (agb_spconv_weight_transpose).  Cin = channels of dX, Cout = channels of dY.  The same kernels as
// agb_spconv_fwd_ex with the operand roles swapped; addend (optional, [n_in][ld_add], may alias dX): the other gradient of
// a residual join.  Reference: ME's ConvolutionBackward behind resnet_block.py:62-73 / senet_block.py:80-96.
int agb_spconv_bwd_data(const float* dY, int lddy, const float* Wt, const int32_t* map, long long map_stride, int kflip,
                        float* dX, int lddx, int n_in, int K3, int Cin, int Cout, const int32_t* perm,
                        const int32_t* tile_cls, const int32_t* cls_tab, int n_tiles, int ksplit, float* partial,
                        const float* addend, int ld_add, void* stream) {
    return agb_spconv_fwd_opt_add(dY, lddy, Wt, map, map_stride, kflip, nullptr, dX, lddx, n_in, K3, Cout, Cin, perm, tile_cls,
                                  cls_tab, n_tiles, ksplit, partial, 1, -1, addend, ld_add, stream);
}

// Tile geometry the pair-compacted kernel uses for this call: out[0] = tile height R (0: another kernel takes the layer),
// out[1] = rows per tile, out[2] = tiles, out[3] = log2 of the interleave block (0: contiguous tiles, no block table).
int agb_spconv_cmp_geometry(int n_out, int Cin, int Cout, int ldx, int ldy, int ksplit, int cmp_mode,
                            int cmp_interleave_shift, int32_t* out) {
    AGB_CHECK_ARG(out != nullptr, "agb_spconv_cmp_geometry: out required");
    out[0] = out[1] = out[2] = out[3] = 0;
    if (n_out <= 0 || Cin < 1 || Cout < 1 || ksplit < 1) return AGB_OK;
    ConvArgs a{};
    float dummy = 0.f;
    a.n_out = n_out; a.Cin = Cin; a.Cout = Cout; a.ldx = ldx; a.ldy = ldy; a.ksplit = ksplit;
    a.partial = ksplit > 1 ? &dummy : nullptr;
    a.cmp_mode = cmp_mode; a.cmp_il = cmp_interleave_shift;
    if (cmp_rows(a) == 0) return AGB_OK;
    int R, rpt, ntiles, nct, il;
    cmp_geometry(a, &R, &rpt, &ntiles, &nct, &il);
    out[0] = R; out[1] = rpt; out[2] = ntiles; out[3] = il;
    return AGB_OK;
}

// agb_spconv_fwd_opt with WORK-BALANCED tiles for the pair-compacted kernel: tile_blocks int32[tb_tiles][tb_blocks] from
// agb_spconv_balance_tiles for the geometry agb_spconv_cmp_geometry reports for this call (checked).  Same sums, bit for
// bit, as without the table: a row's sum does not depend on the tile it is computed in.
AGB_INTERNAL int agb_spconv_fwd_tiles_add(const float* X, int ldx, const float* W, const int32_t* nbr, long long nbr_stride,
                                          int kflip, const float* bias, float* Y, int ldy, int n_out, int K3, int Cin, int Cout,
                                          int ksplit, float* partial, int cmp_mode, int cmp_interleave_shift,
                                          const int32_t* tile_blocks, int tb_tiles, int tb_blocks, const float* addend,
                                          int ld_add, void* stream);
int agb_spconv_fwd_tiles(const float* X, int ldx, const float* W, const int32_t* nbr, long long nbr_stride, int kflip,
                         const float* bias, float* Y, int ldy, int n_out, int K3, int Cin, int Cout, int ksplit,
                         float* partial, int cmp_mode, int cmp_interleave_shift, const int32_t* tile_blocks, int tb_tiles,
                         int tb_blocks, void* stream) {
    return agb_spconv_fwd_tiles_add(X, ldx, W, nbr, nbr_stride, kflip, bias, Y, ldy, n_out, K3, Cin, Cout, ksplit, partial,
                                    cmp_mode, cmp_interleave_shift, tile_blocks, tb_tiles, tb_blocks, nullptr, 0, stream);
}
int agb_spconv_fwd_tiles_add(const float* X, int ldx, const float* W, const int32_t* nbr, long long nbr_stride, int kflip,
                             const float* bias, float* Y, int ldy, int n_out, int K3, int Cin, int Cout, int ksplit,
                             float* partial, int cmp_mode, int cmp_interleave_shift, const int32_t* tile_blocks, int tb_tiles,
                             int tb_blocks, const float* addend, int ld_add, void* stream) {
    AGB_CHECK_ARG(addend == nullptr || (ld_add >= Cout && ld_add % 4 == 0), "agb_spconv_fwd_tiles: addend ld_add %d", ld_add);
    AGB_CHECK_ARG(tile_blocks != nullptr && nbr != nullptr, "agb_spconv_fwd_tiles: tile_blocks and nbr required");
    int32_t g[4];
    int rc = agb_spconv_cmp_geometry(n_out, Cin, Cout, ldx, ldy, ksplit, cmp_mode, cmp_interleave_shift, g);
    if (rc) return rc;
    AGB_CHECK_ARG(g[0] > 0 && g[3] > 0 && g[2] == tb_tiles && (g[1] >> g[3]) == tb_blocks,
                  "agb_spconv_fwd_tiles: the table (%d tiles x %d blocks) does not match this call's geometry (%d tiles x %d "
                  "blocks of %d rows; 0 = the pair-compacted kernel does not take this layer)", tb_tiles, tb_blocks, g[2],
                  g[3] > 0 ? g[1] >> g[3] : 0, 1 << g[3]);
    AGB_CHECK_ARG(n_out >= 0 && K3 >= 1 && nbr_stride >= n_out && ldy >= Cout, "agb_spconv_fwd_tiles: bad sizes");
    AGB_CHECK_ARG(ksplit >= 1 && (ksplit == 1 || partial != nullptr), "agb_spconv_fwd_tiles: ksplit needs `partial`");
    ConvArgs a{X, ldx, W, nbr, nbr_stride, kflip, bias, Y, ldy, n_out, K3, Cin, Cout, nullptr, nullptr, nullptr, ksplit,
               partial, cmp_mode, cmp_interleave_shift, nullptr, tile_blocks};
    a.addend = addend; a.ld_add = ld_add;
    rc = launch_conv(a, 0, (hipStream_t)stream);
    if (rc) return rc;
    AGB_CHECK_LAUNCH("agb_spconv_fwd_tiles");
    return AGB_OK;
}

int agb_spconv_fwd_ex(const float* X, int ldx, const float* W, const int32_t* nbr, long long nbr_stride, int kflip,
                      const float* bias, float* Y, int ldy, int n_out, int K3, int Cin, int Cout,
                      const int32_t* perm, const int32_t* tile_cls, const int32_t* cls_tab, int n_tiles, int ksplit,
                      float* partial, void* stream) {
    return agb_spconv_fwd_opt(X, ldx, W, nbr, nbr_stride, kflip, bias, Y, ldy, n_out, K3, Cin, Cout, perm, tile_cls,
                              cls_tab, n_tiles, ksplit, partial, 1, -1, stream);
}

// Low-precision operands (precision 1 = bf16, 2 = split-bf16 x3), fp32 accumulate and fp32 I/O.  Same contract as
// agb_spconv_fwd_ex except that the weights are K-major: Wt float[K3][Cout][Cin].
int agb_spconv_fwd_lp(const float* X, int ldx, const float* Wt, const int32_t* nbr, long long nbr_stride, int kflip,
                      const float* bias, float* Y, int ldy, int n_out, int K3, int Cin, int Cout,
                      const int32_t* perm, const int32_t* tile_cls, const int32_t* cls_tab, int n_tiles, int ksplit,
                      float* partial, int precision, void* stream) {
    AGB_CHECK_ARG(precision == 1 || precision == 2, "agb_spconv_fwd_lp: precision %d (1 = bf16, 2 = bf16x3)", precision);
    AGB_CHECK_ARG(n_out >= 0 && K3 >= 1 && Cin >= 4 && Cout >= 4 && Cin % 4 == 0 && Cout % 4 == 0 && ldx % 4 == 0,
                  "agb_spconv_fwd_lp: Cin (%d), Cout (%d), ldx must be multiples of 4", Cin, Cout);
    AGB_CHECK_ARG(ksplit >= 1 && (ksplit == 1 || partial != nullptr), "agb_spconv_fwd_lp: ksplit needs `partial`");
    AGB_CHECK_ARG(nbr != nullptr || (K3 == 1 && perm == nullptr), "agb_spconv_fwd_lp: the identity map needs K3 == 1");
    if (n_out == 0) return AGB_OK;
    hipStream_t s = (hipStream_t)stream;
    ConvArgs a{X, ldx, Wt, nbr, nbr_stride, kflip, bias, Y, ldy, n_out, K3, Cin, Cout, perm, tile_cls, cls_tab, ksplit,
               partial, 0, -1};
    dim3 block(256);
    const bool x3 = precision == 2;
    if (perm) {
        dim3 grid(n_tiles, agb_cdiv(Cout, BN), ksplit);
        if (x3) AGB_LAUNCH((k_spconv_pipe_bf16<64, true, true, 64>), grid, block, 0, s, a);
        else AGB_LAUNCH((k_spconv_pipe_bf16<64, true, false, 64>), grid, block, 0, s, a);
    } else if (conv_tile_rows(n_out, Cin, Cout) == 128) {
        // 128-column tiles when the layer is wide enough and still fills the chip: the gathered and converted A tile
        // (the VALU / L2 cost of this kernel) serves twice the columns
        const bool wide = Cout >= 128 && (long long)agb_cdiv(n_out, 128) * agb_cdiv(Cout, 128) * ksplit >= 512;
        dim3 grid(agb_cdiv(n_out, 128), agb_cdiv(Cout, wide ? 128 : 64), ksplit);
        if (wide) {
            if (x3) AGB_LAUNCH((k_spconv_pipe_bf16<128, false, true, 128>), grid, block, 0, s, a);
            else AGB_LAUNCH((k_spconv_pipe_bf16<128, false, false, 128>), grid, block, 0, s, a);
        } else if (x3) AGB_LAUNCH((k_spconv_pipe_bf16<128, false, true, 64>), grid, block, 0, s, a);
        else AGB_LAUNCH((k_spconv_pipe_bf16<128, false, false, 64>), grid, block, 0, s, a);
    } else {
        dim3 grid(agb_cdiv(n_out, 64), agb_cdiv(Cout, BN), ksplit);
        if (x3) AGB_LAUNCH((k_spconv_pipe_bf16<64, false, true, 64>), grid, block, 0, s, a);
        else AGB_LAUNCH((k_spconv_pipe_bf16<64, false, false, 64>), grid, block, 0, s, a);
    }
    if (ksplit > 1) {
        long long total = (long long)n_out * (Cout / 4);
        hipLaunchKernelGGL(k_split_reduce<float>, dim3(agb_cdiv(total, 256)), dim3(256), 0, s, partial, ksplit, n_out, Cout / 4,
                           bias, Y, ldy);
    }
    AGB_CHECK_LAUNCH("agb_spconv_fwd_lp");
    return AGB_OK;
}

// bf16 twin of a row matrix (activations, gradients, K-major weights): Y16[r][c] = bf16(X[r][c]), round to nearest even.
int agb_to_bf16(const float* X, long long ldx, long long n, int C, uint16_t* Y16, long long ldy, void* stream) {
    AGB_CHECK_ARG(n >= 0 && C >= 4 && C % 4 == 0 && ldx >= C && ldy >= C && ldx % 4 == 0 && ldy % 4 == 0,
                  "agb_to_bf16: C (%d), ldx, ldy must be multiples of 4", C);
    if (n == 0) return AGB_OK;
    AGB_CHECK_ARG(X && Y16, "agb_to_bf16: null pointer");
    const long long total = n * (C / 4);
    AGB_CHECK_ARG(agb_cdiv(total, 256) <= 0x7fffffff, "agb_to_bf16: too many elements for one launch");
    hipLaunchKernelGGL(k_to_bf16, dim3((unsigned)agb_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, X, ldx, n, C / 4,
                       Y16, ldy);
    AGB_CHECK_LAUNCH("agb_to_bf16");
    return AGB_OK;
}

// agb_spconv_fwd_lp (precision 1) on bf16 STORAGE: X16 uint16 [n_in][ldx16] and K-major weights Wt16 uint16 [K3][Cout][Cin]
// are bf16 twins made by agb_to_bf16; fp32 accumulate, bias and output.  Cin % 8 == 0, ldx16 % 8 == 0.
}  // extern "C"
template <bool Y16>
static int fwd_b16_impl(const uint16_t* X16, int ldx16, const uint16_t* Wt16, const int32_t* nbr, long long nbr_stride,
                        int kflip, const float* bias, float* Y, int ldy, int n_out, int K3, int Cin, int Cout,
                        const int32_t* perm, const int32_t* tile_cls, const int32_t* cls_tab, int n_tiles, int ksplit,
                        float* partial, void* stream, const uint16_t* addend16 = nullptr, int ld_add = 0) {
    AGB_CHECK_ARG(addend16 == nullptr || (Y16 && ksplit == 1 && ld_add >= Cout && ld_add % 2 == 0),
                  "agb_spconv_bwd_data_h: an addend needs bf16 output rows, ksplit == 1 and an even ld_add >= Cout (%d)", ld_add);
    AGB_CHECK_ARG(n_out >= 0 && K3 >= 1 && Cin >= 8 && Cout >= 4 && Cin % 8 == 0 && Cout % 4 == 0 && ldx16 % 8 == 0,
                  "agb_spconv_fwd_b16: Cin (%d), ldx16 must be multiples of 8, Cout (%d) of 4", Cin, Cout);
    AGB_CHECK_ARG(!Y16 || ldy % 4 == 0, "agb_spconv_fwd_h: ldy (%d, bf16 elements) must be a multiple of 4", ldy);
    AGB_CHECK_ARG(ksplit >= 1 && (ksplit == 1 || partial != nullptr), "agb_spconv_fwd_b16: ksplit needs `partial`");
    AGB_CHECK_ARG(nbr != nullptr || (K3 == 1 && perm == nullptr), "agb_spconv_fwd_b16: the identity map needs K3 == 1");
    if (n_out == 0) return AGB_OK;
    hipStream_t s = (hipStream_t)stream;
    ConvArgs a{reinterpret_cast<const float*>(X16), ldx16, reinterpret_cast<const float*>(Wt16), nbr, nbr_stride, kflip, bias,
               Y, ldy, n_out, K3, Cin, Cout, perm, tile_cls, cls_tab, ksplit, partial, 0, -1};
    a.addend = reinterpret_cast<const float*>(addend16); a.ld_add = ld_add;
    dim3 block(256);
    if (perm) {
        hipLaunchKernelGGL((k_spconv_pipe_b16<64, true, 64, Y16>), dim3(n_tiles, agb_cdiv(Cout, BN), ksplit), block, 0, s, a);
        agb_note_kernel(Y16 ? "k_spconv_pipe_b16<64, true, 64, true>" : "k_spconv_pipe_b16<64, true, 64, false>");
    } else if (conv_tile_rows(n_out, Cin, Cout) == 128) {
        const bool wide = Cout >= 128 && (long long)agb_cdiv(n_out, 128) * agb_cdiv(Cout, 128) * ksplit >= 512;
        dim3 grid(agb_cdiv(n_out, 128), agb_cdiv(Cout, wide ? 128 : 64), ksplit);
        if (wide) hipLaunchKernelGGL((k_spconv_pipe_b16<128, false, 128, Y16>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((k_spconv_pipe_b16<128, false, 64, Y16>), grid, block, 0, s, a);
        agb_note_kernel(wide ? (Y16 ? "k_spconv_pipe_b16<128, false, 128, true>" : "k_spconv_pipe_b16<128, false, 128, false>")
                             : (Y16 ? "k_spconv_pipe_b16<128, false, 64, true>" : "k_spconv_pipe_b16<128, false, 64, false>"));
    } else {
        hipLaunchKernelGGL((k_spconv_pipe_b16<64, false, 64, Y16>), dim3(agb_cdiv(n_out, 64), agb_cdiv(Cout, BN), ksplit), block,
                           0, s, a);
        agb_note_kernel(Y16 ? "k_spconv_pipe_b16<64, false, 64, true>" : "k_spconv_pipe_b16<64, false, 64, false>");
    }
    if (ksplit > 1) {
        long long total = (long long)n_out * (Cout / 4);
        if (Y16)
            hipLaunchKernelGGL(k_split_reduce<bf16_t>, dim3(agb_cdiv(total, 256)), dim3(256), 0, s, partial, ksplit, n_out,
                               Cout / 4, bias, reinterpret_cast<bf16_t*>(Y), ldy);
        else
            hipLaunchKernelGGL(k_split_reduce<float>, dim3(agb_cdiv(total, 256)), dim3(256), 0, s, partial, ksplit, n_out,
                               Cout / 4, bias, Y, ldy);
    }
    AGB_CHECK_LAUNCH("agb_spconv_fwd_b16");
    return AGB_OK;
}

extern "C" {
int agb_spconv_fwd_b16(const uint16_t* X16, int ldx16, const uint16_t* Wt16, const int32_t* nbr, long long nbr_stride,
                       int kflip, const float* bias, float* Y, int ldy, int n_out, int K3, int Cin, int Cout,
                       const int32_t* perm, const int32_t* tile_cls, const int32_t* cls_tab, int n_tiles, int ksplit,
                       float* partial, void* stream) {
    return fwd_b16_impl<false>(X16, ldx16, Wt16, nbr, nbr_stride, kflip, bias, Y, ldy, n_out, K3, Cin, Cout, perm, tile_cls,
                               cls_tab, n_tiles, ksplit, partial, stream);
}

// The same with bf16 OUTPUT rows (the bf16-activation mode: every row matrix of the network is bf16, accumulation fp32,
// one rounding when the row leaves the kernel); Y16 uint16 [n_out][ldy16].
int agb_spconv_fwd_h(const uint16_t* X16, int ldx16, const uint16_t* Wt16, const int32_t* nbr, long long nbr_stride,
                     int kflip, const float* bias, uint16_t* Y16, int ldy16, int n_out, int K3, int Cin, int Cout,
                     const int32_t* perm, const int32_t* tile_cls, const int32_t* cls_tab, int n_tiles, int ksplit,
                     float* partial, void* stream) {
    return fwd_b16_impl<true>(X16, ldx16, Wt16, nbr, nbr_stride, kflip, bias, reinterpret_cast<float*>(Y16), ldy16, n_out, K3,
                              Cin, Cout, perm, tile_cls, cls_tab, n_tiles, ksplit, partial, stream);
}

// The data gradient on bf16 rows under its own name, like agb_spconv_bwd_data:
//   dX16[q] = bf16( [addend16[q] +] sum_k dY16[map[k][q]] W16[k] )        (fp32 sum, ONE rounding)
// W16 uint16 [K3][Cin][Cout] (Cin = channels of dX, Cout = channels of dY: for the data gradient the layer's own kernel
// [K3][Cin][Cout] IS the K-major operand); map / perm / ksplit as agb_spconv_fwd_h.  addend16 (optional, bf16
// [n_in][ld_add], ksplit == 1): the other gradient of a residual join — a block input that feeds this layer and the
// shortcut (senet_block.py:99-147, resnet_block.py:93-133).
int agb_spconv_bwd_data_h(const uint16_t* dY16, int lddy16, const uint16_t* W16, const int32_t* map, long long map_stride,
                          int kflip, uint16_t* dX16, int lddx16, int n_in, int K3, int Cin, int Cout, const int32_t* perm,
                          const int32_t* tile_cls, const int32_t* cls_tab, int n_tiles, int ksplit, float* partial,
                          const uint16_t* addend16, int ld_add, void* stream) {
    return fwd_b16_impl<true>(dY16, lddy16, W16, map, map_stride, kflip, nullptr, reinterpret_cast<float*>(dX16), lddx16, n_in,
                              K3, Cout, Cin, perm, tile_cls, cls_tab, n_tiles, ksplit, partial, stream, addend16, ld_add);
}

int agb_spconv_fwd(const float* X, int ldx, const float* W, const int32_t* nbr, long long nbr_stride, int kflip,
                   const float* bias, float* Y, int ldy, int n_out, int K3, int Cin, int Cout, void* stream) {
    return agb_spconv_fwd_ex(X, ldx, W, nbr, nbr_stride, kflip, bias, Y, ldy, n_out, K3, Cin, Cout, nullptr, nullptr,
                             nullptr, 0, 1, nullptr, stream);
}

// Rows of a level grouped by lattice-parity class for a stride-`stride` operator (see k_spconv_pipe<.., PERM>).
// perm: int32[n + stride^3 * 64] out (filled with -1 padding); tile_cls: int32[max_tiles] out with
// max_tiles = n/64 + stride^3 + 1; scratch: int32[256].  Tile size 64.
int agb_parity_partition(const int32_t* coords, int n, int ts_in, int stride, int32_t* perm, int32_t* tile_cls,
                         int max_tiles, int32_t* scratch, void* stream) {
    AGB_CHECK_ARG(stride >= 1 && stride <= 4, "agb_parity_partition: stride %d", stride);
    hipStream_t s = (hipStream_t)stream;
    const int ncls = stride * stride * stride, TMR = 64;
    (void)hipMemsetAsync(scratch, 0, sizeof(int32_t) * 256, s);
    (void)hipMemsetAsync(perm, 0xFF, sizeof(int32_t) * ((size_t)n + (size_t)ncls * TMR), s);
    if (n > 0)
        hipLaunchKernelGGL(k_parity_count, dim3(agb_cdiv(n, 256)), dim3(256), 0, s, (const int4*)coords, n, ts_in,
                           stride, scratch);
    hipLaunchKernelGGL(k_parity_layout, dim3(1), dim3(256), 0, s, scratch, ncls, TMR, scratch + 64, scratch + 160,
                       tile_cls, max_tiles);
    if (n > 0)
        hipLaunchKernelGGL(k_parity_scatter, dim3(agb_cdiv(n, 256)), dim3(256), 0, s, (const int4*)coords, n, ts_in,
                           stride, scratch + 160, perm);
    AGB_CHECK_LAUNCH("agb_parity_partition");
    return AGB_OK;
}

// dW must be zero-filled by the caller (it is accumulated into).
// precision: 0 = fp32 MFMA, 1 = bf16 operands, 2 = split-bf16 x3 (the small-Cin stem path is always fp32).
// nbr == nullptr (K3 == 1): identity map, dW = X^T dY of a 1x1 stride-1 convolution.
// True when the call runs the register-operand fp32 kernel (dwreg.hip): fp32, Cin >= 12, not one of the HBM-bound dense
// shapes the streaming kernel takes.  variant 1 forces the LDS-staged k_spconv_dw_cmp<0> (A/B measurements).
#define DW_SMALL_NSUB 4     // 1024-row sub-chunks one workgroup of the small-Cin weight gradient accumulates before it stores

// Rows per sub-chunk of the small-Cin (stem) weight gradient: ~4096 workgroups, 256..1024 rows (the kernel is latency-bound
// (index -> gather): more resident workgroups hide more of it; 512 / 1024 / 2048 rows measured within 5 % of each other on the
// 7^3 stem, 4096+ clearly slower)
static int dw_small_rows(int n_out, int K3, int Cin, int Cout) {
    const long long tiles = (long long)agb_cdiv(K3, 64 / Cin) * agb_cdiv(Cout, 64);
    long long target_chunks = 4096 / tiles;
    if (target_chunks < 1) target_chunks = 1;
    long long rows = (n_out + target_chunks - 1) / target_chunks;
    if (rows < 256) rows = 256;
    rows = (rows + 31) / 32 * 32;
    if (rows > 1024) rows = 1024;
    return (int)rows;
}

static bool dw_takes_reg_kernel(const int32_t* nbr, int n_out, int Cin, int Cout, int precision, int variant,
                                bool has_workspace) {
    if (precision != 0 || variant == 1 || Cin == 4 || Cin == 8) return false;
    // (the HBM-bound dense shapes of the streaming kernel end in cross-workgroup fp32 atomics: a caller that brings a
    // workspace asked for the reproducible sum and gets the register-operand kernel's fixed-order fold instead)
    if (nbr == nullptr && agb_dense_stream_wgrad_ok(n_out, Cin, Cout) && !has_workspace) return false;
    // automatic: the register-operand kernel is the REPRODUCIBLE form (fixed-order fold through the workspace); without a
    // workspace the LDS-staged kernel with atomic accumulation is the faster one in the training step (3.7 % of the
    // MSENet14 step, profiles/r03_bench_dw_variants.txt)
    return variant == 2 || has_workspace;
}

// Row chunks of the LDS-staged pair-compacted kernel k_spconv_dw_cmp (Cin >= 12; every operand precision).
// precision 3 = bf16 operands read from bf16 storage (agb_spconv_bwd_weight_b16).
struct DwCmpGeo { int rows, chunks, il, m_tiles, cin_tiles, n_tiles; };
static int dw_cmp_maxr(int n_out, int precision, int variant) {
    static const bool old_chunks = getenv("AGB_DW_CMP_2048") != nullptr;      // (A/B inside a training step)
    return (precision == 0 && variant != 4 && n_out >= 8192 && !old_chunks) ? DW_CMP_ROWS_F32 : DW_MAXROWS;
}
static DwCmpGeo dw_cmp_geometry(int n_out, int K3, int Cin, int Cout, bool dense, int precision, int maxr = DW_MAXROWS) {
    DwCmpGeo g;
    g.cin_tiles = agb_cdiv(Cin, 64);
    g.m_tiles = K3 * g.cin_tiles;
    g.n_tiles = agb_cdiv(Cout, 64);
    // aim for ~4096 workgroups; at least 256 rows per workgroup (multiple of 32)
    // (dense products with bf16 operands: the MFMA work of a workgroup is small against the 64 x 64 tile it ends with —
    // 4096 workgroups put 13 M atomics on the 256 x 1024 gradient of a 14 k-row layer: 71 us, 512 workgroups: 34 us;
    // 211 k x 64 x 256: 120 -> 41 us.  The 3^3 maps sit at the 2048-row cap of the pair list either way.)
    // (fp32 maps since round 5: ~6144 — with six workgroups resident per CU instead of three, 4096 were 2.7 rounds of the
    // chip; measured in the step 3072 / 4096 / 6144 / 8192 / 12288: weight gradient 1.96 / 1.91 / 1.87 / 2.02 / 2.08 ms)
    // (fp32 dense products: ~2048 — every workgroup ends in 4096 atomics on the gradient; MPointNet step 11.71 ms at 4096,
    // 11.51-11.58 at 512 .. 2048, 11.64 at 6144)
    // (bf16 operands re-measured in round 5 on MSENet50: maps 2048 / 4096 equal, 6144 +13 %, 8192 +43 %; dense 256 +8 %, 512 / 1024
    // equal, 2048 +33 %: they stay)
    const int target_wgs = (dense && (precision == 1 || precision == 3)) ? 512 : (precision == 0 && !dense) ? 6144
                           : (precision == 0 && dense) ? 2048 : 4096;
    long long target_chunks = target_wgs / ((long long)g.m_tiles * g.n_tiles);
    if (target_chunks < 1) target_chunks = 1;
    long long rows = (n_out + target_chunks - 1) / target_chunks;
    if (rows < 256) rows = 256;
    rows = (rows + 31) / 32 * 32;
    if (rows > maxr) rows = maxr;                 // (the LDS pair list)
    g.rows = (int)rows;
    g.chunks = agb_cdiv(n_out, g.rows);
    g.il = g.chunks >= 16 ? cmp_interleave(n_out, -1) : 0;
    if (g.il > 0) {
        const int nblk = agb_cdiv(n_out, 1 << g.il), bpc = g.rows >> g.il;
        g.chunks = agb_cdiv(nblk, bpc);
    }
    return g;
}

// Bytes of the workspace that makes the weight gradient of this shape a fixed-order (bitwise reproducible) sum, in EVERY
// operand precision (0 fp32, 1 bf16, 2 bf16x3; the bf16-storage entry point agb_spconv_bwd_weight_b16_ws asks with 1) and for
// the dense shapes (dense != 0: the identity map).  0 only for empty products.
size_t agb_spconv_bwd_weight_workspace_bytes(int n_out, int K3, int Cin, int Cout, int dense, int precision) {
    static const int32_t some_map = 0;
    if (n_out <= 0 || K3 < 1 || Cin < 4 || Cout < 4) return 0;
    if (!dense && (Cin == 4 || Cin == 8)) {      // the small-Cin (stem) kernel: groups of DW_SMALL_NSUB 1024-row chunks
        const int chunks = agb_cdiv(agb_cdiv(n_out, dw_small_rows(n_out, K3, Cin, Cout)), DW_SMALL_NSUB);
        size_t bytes = (size_t)chunks * K3 * Cin * Cout * sizeof(float);
        // (the pair-sparse stem kernel of stem.hip, taken when the rows are 4 floats wide: its row partitions)
        if (agb_stem_dw_ok(n_out, K3, Cin, Cout, 4, Cout)) {
            const size_t b2 = agb_stem_dw_workspace_bytes(n_out, K3);
            if (b2 > bytes) bytes = b2;
        }
        return bytes;
    }
    // fp32 maps with Cin, Cout multiples of 64: the persistent-accumulator kernel of dwa.hip (whether it takes the call also
    // depends on the row strides: the workspace serves either kernel)
    size_t dwa = 0;
    if (!dense && precision == 0 && agb_dwa_ok(n_out, K3, Cin, Cout, Cin, Cout, true)) dwa = agb_dwa_workspace_bytes(n_out, K3, Cin, Cout);
    size_t other;
    if (dw_takes_reg_kernel(dense ? nullptr : &some_map, n_out, Cin, Cout, precision, 0, true)) {
        other = agb_dwreg_workspace_bytes(n_out, K3, Cin, Cout);
    } else {
        const DwCmpGeo g = dw_cmp_geometry(n_out, K3, Cin, Cout, dense != 0, precision, dw_cmp_maxr(n_out, precision, 0));
        other = (size_t)g.chunks * K3 * Cin * Cout * sizeof(float);
    }
    return dwa > other ? dwa : other;
}

// 1 when agb_spconv_bwd_weight_ws(precision 0, variant 0, a workspace of agb_spconv_bwd_weight_workspace_bytes) runs the
// persistent-accumulator kernel (dwa.hip) for this shape: the caller then passes the workspace whether or not it asked for
// reproducible sums (the kernel's partial tiles ARE the product path; its sums are reproducible by construction).
int agb_spconv_bwd_weight_persistent(int n_out, int K3, int Cin, int Cout, int ldx, int ldy) {
    return agb_dwa_ok(n_out, K3, Cin, Cout, ldx, ldy, false) ? 1 : 0;
}

int agb_spconv_bwd_weight_lp(const float* X, int ldx, const float* dY, int ldy, const int32_t* nbr, long long nbr_stride,
                             float* dW, int n_out, int K3, int Cin, int Cout, int precision, void* stream) {
    return agb_spconv_bwd_weight_ws(X, ldx, dY, ldy, nbr, nbr_stride, dW, n_out, K3, Cin, Cout, precision, 0, nullptr, 0,
                                    stream);
}

static int bwd_weight_impl(const float* X, int ldx, const float* dY, int ldy, const int32_t* nbr, long long nbr_stride,
                           float* dW, int n_out, int K3, int Cin, int Cout, int precision, int variant, void* workspace,
                           size_t workspace_bytes, void* stream);

int agb_spconv_bwd_weight_ws(const float* X, int ldx, const float* dY, int ldy, const int32_t* nbr, long long nbr_stride,
                             float* dW, int n_out, int K3, int Cin, int Cout, int precision, int variant, void* workspace,
                             size_t workspace_bytes, void* stream) {
    AGB_CHECK_ARG(precision >= 0 && precision <= 2, "agb_spconv_bwd_weight_ws: precision %d (0 fp32, 1 bf16, 2 bf16x3)",
                  precision);
    return bwd_weight_impl(X, ldx, dY, ldy, nbr, nbr_stride, dW, n_out, K3, Cin, Cout, precision, variant, workspace,
                           workspace_bytes, stream);
}

// bf16 operands from bf16 STORAGE (see agb_spconv_fwd_b16): X16 / dY16 are agb_to_bf16 twins, ld in bf16 elements
// (multiples of 4); dW fp32, accumulated into.  Cin >= 12.
int agb_spconv_bwd_weight_b16(const uint16_t* X16, int ldx16, const uint16_t* dY16, int ldy16, const int32_t* nbr,
                              long long nbr_stride, float* dW, int n_out, int K3, int Cin, int Cout, void* stream) {
    AGB_CHECK_ARG(Cin >= 12, "agb_spconv_bwd_weight_b16: Cin %d (>= 12)", Cin);
    return bwd_weight_impl(reinterpret_cast<const float*>(X16), ldx16, reinterpret_cast<const float*>(dY16), ldy16, nbr,
                           nbr_stride, dW, n_out, K3, Cin, Cout, 3, 0, nullptr, 0, stream);
}

// The same with a caller-owned workspace of agb_spconv_bwd_weight_workspace_bytes(n_out, K3, Cin, Cout, nbr == NULL, 1)
// bytes: row chunks leave partial tiles and are folded in ascending order — the bf16 weight gradient bitwise reproducible
// from run to run (workspace == NULL: fp32 atomic accumulation, as agb_spconv_bwd_weight_b16).
int agb_spconv_bwd_weight_b16_ws(const uint16_t* X16, int ldx16, const uint16_t* dY16, int ldy16, const int32_t* nbr,
                                 long long nbr_stride, float* dW, int n_out, int K3, int Cin, int Cout, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    AGB_CHECK_ARG(Cin >= 12, "agb_spconv_bwd_weight_b16_ws: Cin %d (>= 12)", Cin);
    return bwd_weight_impl(reinterpret_cast<const float*>(X16), ldx16, reinterpret_cast<const float*>(dY16), ldy16, nbr,
                           nbr_stride, dW, n_out, K3, Cin, Cout, 3, 0, workspace, workspace_bytes, stream);
}

static int bwd_weight_impl(const float* X, int ldx, const float* dY, int ldy, const int32_t* nbr, long long nbr_stride,
                           float* dW, int n_out, int K3, int Cin, int Cout, int precision, int variant, void* workspace,
                           size_t workspace_bytes, void* stream) {
    AGB_CHECK_ARG(n_out >= 0 && K3 >= 1 && Cin >= 1 && Cout >= 1, "agb_spconv_bwd_weight: bad sizes");
    AGB_CHECK_ARG(variant >= 0 && variant <= 4, "agb_spconv_bwd_weight_ws: variant %d (0 automatic, 1 LDS-staged, 2 register "
                  "operands, 3 persistent accumulators, 4 LDS-staged with 2048-row chunks)", variant);
    // the fp32 LDS-staged kernel walks chunks of at most DW_CMP_ROWS_F32 rows (40 KB of LDS: four workgroups per CU instead of
    // three: 64->64 -4 %, 128->128 -10 %, the 3^3 stride-2 maps -3 %; levels of a few thousand rows +1 %: they keep 2048);
    // variant 4 = the 2048-row chunks of rounds 2-4 everywhere, kept for A/B measurements
    const int dw_maxr = dw_cmp_maxr(n_out, precision, variant);
    const bool dw_old = variant == 4;
    if (variant == 4) variant = 1;
    AGB_CHECK_ARG(nbr != nullptr || K3 == 1, "agb_spconv_bwd_weight: the identity map (nbr == NULL) needs K3 == 1");
    AGB_CHECK_ARG(nbr != nullptr || (Cin != 4 && Cin != 8), "agb_spconv_bwd_weight: the identity map needs Cin >= 12");
    AGB_CHECK_ARG(Cin % 4 == 0 && Cout % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0,
                  "agb_spconv_bwd_weight: Cin (%d), Cout (%d), ldx, ldy must be multiples of 4", Cin, Cout);
    if (n_out == 0) return AGB_OK;
    hipStream_t s = (hipStream_t)stream;
    if (nbr == nullptr && precision == 0 && workspace == nullptr && agb_dense_stream_wgrad_ok(n_out, Cin, Cout)) {
        int rc = agb_dense_stream_wgrad_launch(X, ldx, dY, ldy, dW, n_out, Cin, Cout, s);
        if (rc) return rc;
        AGB_CHECK_LAUNCH("agb_spconv_bwd_weight (dense, streaming)");
        return AGB_OK;
    }
    if (nbr != nullptr && precision == 0 && (variant == 0 || variant == 3) && workspace != nullptr &&
        agb_dwa_ok(n_out, K3, Cin, Cout, ldx, ldy, variant == 3) &&
        workspace_bytes >= agb_dwa_workspace_bytes(n_out, K3, Cin, Cout)) {
        // persistent accumulators, hand-scheduled main loop, fixed-order fold (dwa.hip): the fp32 product path since round 5
        int rc = agb_dwa_launch(X, ldx, dY, ldy, nbr, nbr_stride, dW, n_out, K3, Cin, Cout, workspace, workspace_bytes, s);
        if (rc) return rc;
        AGB_CHECK_LAUNCH("agb_spconv_bwd_weight (persistent accumulators)");
        return AGB_OK;
    }
    if (variant == 3) variant = 0;      // (a shape the persistent kernel does not take: the automatic choice)
    if (dw_takes_reg_kernel(nbr, n_out, Cin, Cout, precision, variant, workspace != nullptr)) {
        int rc = agb_dwreg_launch(X, ldx, dY, ldy, nbr, nbr_stride, dW, n_out, K3, Cin, Cout, workspace, workspace_bytes, s);
        if (rc) return rc;
        AGB_CHECK_LAUNCH("agb_spconv_bwd_weight (register operands)");
        return AGB_OK;
    }
    if (workspace != nullptr && variant == 0 && nbr != nullptr && agb_stem_dw_ok(n_out, K3, Cin, Cout, ldx, ldy)) {
        // the 3-channel stem with a workspace: pair-sparse, one 4x4x1 MFMA per pair (csrc/stem.hip)
        int rc = agb_stem_dw_launch(X, dY, ldy, nbr, nbr_stride, dW, n_out, K3, workspace, workspace_bytes, s);
        if (rc) return rc;
        AGB_CHECK_LAUNCH("agb_spconv_bwd_weight (stem, pair-sparse)");
        return AGB_OK;
    }
    if (Cin == 4 || Cin == 8) {
        const int m_tiles = agb_cdiv(K3, 64 / Cin), n_tiles = agb_cdiv(Cout, 64);
        const int rows = dw_small_rows(n_out, K3, Cin, Cout);
        int chunks = agb_cdiv(n_out, rows);
        dim3 grid((chunks >= 16 ? 8 * agb_cdiv(chunks, 8) : chunks) * m_tiles, n_tiles), block(256);
        // with a workspace: groups of DW_SMALL_NSUB sub-chunks per workgroup (one partial tile per group instead of one
        // atomic accumulation per sub-chunk), folded in a fixed order
        int nsub = 1;
        float* part = nullptr;
        if (workspace != nullptr && variant == 0) {
            nsub = DW_SMALL_NSUB;
            chunks = agb_cdiv(chunks, nsub);
            if ((size_t)chunks * K3 * Cin * Cout * sizeof(float) > workspace_bytes) {
                agb_set_error("agb_spconv_bwd_weight_ws: workspace of %zu bytes, %zu needed", workspace_bytes,
                              (size_t)chunks * K3 * Cin * Cout * sizeof(float));
                return AGB_EINVAL;
            }
            if (chunks > 1) part = (float*)workspace;      // (one group: a single writer per element already)
            grid = dim3((chunks >= 16 ? 8 * agb_cdiv(chunks, 8) : chunks) * m_tiles, n_tiles);
        }
        if (Cin == 4)
            AGB_LAUNCH(k_spconv_dw_small_cmp<4>, grid, block, 0, s, X, ldx, dY, ldy, nbr, nbr_stride, dW, n_out, K3,
                               Cout, (int)rows, chunks, m_tiles, nsub, part);
        else
            AGB_LAUNCH(k_spconv_dw_small_cmp<8>, grid, block, 0, s, X, ldx, dY, ldy, nbr, nbr_stride, dW, n_out, K3,
                               Cout, (int)rows, chunks, m_tiles, nsub, part);
        if (part) {
            const long long n4 = (long long)K3 * Cin * Cout / 4;
            hipLaunchKernelGGL(k_dw_fold_small, dim3((unsigned)agb_cdiv(n4, 256)), dim3(256), 0, s, (const float4*)part, chunks,
                               n4, (float4*)dW);
        }
    } else {
        // pair-compacted kernel: row chunks of at most DW_MAXROWS rows (the LDS pair list), XCD-aware 1-D grid
        const DwCmpGeo g = dw_cmp_geometry(n_out, K3, Cin, Cout, nbr == nullptr, precision, dw_maxr);
        static const bool ks64_env = getenv("AGB_DW_KS64") != nullptr;       // (A/B inside a training step)
        const bool ks64 = ks64_env || dw_old;
        const int rows = g.rows, chunks = g.chunks, il = g.il, m_tiles = g.m_tiles, cin_tiles = g.cin_tiles;
        const dim3 block(256);
        float* part = nullptr;
        if (workspace != nullptr) {
            const size_t need = (size_t)chunks * K3 * Cin * Cout * sizeof(float);
            if (need > workspace_bytes) {
                agb_set_error("agb_spconv_bwd_weight_ws: workspace of %zu bytes, %zu needed", workspace_bytes, need);
                return AGB_EINVAL;
            }
            if (chunks > 1) part = (float*)workspace;      // (one chunk: a single writer per element already)
        }
        dim3 grid1((chunks >= 16 ? 8 * agb_cdiv(chunks, 8) : chunks) * m_tiles, g.n_tiles);
        if (precision == 3 && Cin % 8 == 0 && Cout % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0)
            // bf16 twins, rows staged as they are and transposed by the LDS reads (ds_read_b64_tr_b16)
            AGB_LAUNCH((k_spconv_dw_cmp<1, true, true, 2048, 64>), grid1, block, 0, s, X, ldx, dY, ldy, nbr, nbr_stride, dW, n_out,
                               K3, Cin, Cout, rows, cin_tiles, chunks, m_tiles, il, part);
        else if (precision == 3)       // bf16 operands from bf16 twins (agb_spconv_bwd_weight_b16)
            AGB_LAUNCH((k_spconv_dw_cmp<1, true, false, 2048, 64>), grid1, block, 0, s, X, ldx, dY, ldy, nbr, nbr_stride, dW, n_out, K3,
                               Cin, Cout, rows, cin_tiles, chunks, m_tiles, il, part);
        else if (precision == 1)
            AGB_LAUNCH((k_spconv_dw_cmp<1, false, false, 2048, 64>), grid1, block, 0, s, X, ldx, dY, ldy, nbr, nbr_stride, dW, n_out, K3,
                               Cin, Cout, rows, cin_tiles, chunks, m_tiles, il, part);
        else if (precision == 2)
            AGB_LAUNCH((k_spconv_dw_cmp<2, false, false, 2048, 64>), grid1, block, 0, s, X, ldx, dY, ldy, nbr, nbr_stride, dW, n_out, K3,
                               Cin, Cout, rows, cin_tiles, chunks, m_tiles, il, part);
        // fp32, maps of >= 9 offsets: 32-pair steps — 8 + 8 KB of staged rows: six (1280-row chunks) / five (2048) workgroups
        // per CU instead of four / three; a further -1 .. -2.5 % per launch (the 2^3 maps and the dense product keep 64)
        else if (dw_maxr == DW_CMP_ROWS_F32 && !ks64 && K3 >= 9)
            AGB_LAUNCH((k_spconv_dw_cmp<0, false, false, 1280, 32>), grid1, block, 0, s, X, ldx, dY, ldy, nbr, nbr_stride,
                               dW, n_out, K3, Cin, Cout, rows, cin_tiles, chunks, m_tiles, il, part);
        else if (!ks64 && K3 >= 9)
            AGB_LAUNCH((k_spconv_dw_cmp<0, false, false, 2048, 32>), grid1, block, 0, s, X, ldx, dY, ldy, nbr, nbr_stride,
                               dW, n_out, K3, Cin, Cout, rows, cin_tiles, chunks, m_tiles, il, part);
        else if (dw_maxr == DW_CMP_ROWS_F32)
            // (literals and every template argument spelled out: the noted kernel name must equal the symbol rocprof prints)
            AGB_LAUNCH((k_spconv_dw_cmp<0, false, false, 1280, 64>), grid1, block, 0, s, X, ldx, dY, ldy, nbr, nbr_stride,
                               dW, n_out, K3, Cin, Cout, rows, cin_tiles, chunks, m_tiles, il, part);
        else
            AGB_LAUNCH((k_spconv_dw_cmp<0, false, false, 2048, 64>), grid1, block, 0, s, X, ldx, dY, ldy, nbr, nbr_stride, dW, n_out, K3,
                               Cin, Cout, rows, cin_tiles, chunks, m_tiles, il, part);
        if (part) {
            const long long n4 = (long long)K3 * Cin * Cout / 4;
            hipLaunchKernelGGL(k_dw_fold_small, dim3((unsigned)agb_cdiv(n4, 256)), dim3(256), 0, s, (const float4*)part, chunks,
                               n4, (float4*)dW);
        }
    }
    AGB_CHECK_LAUNCH("agb_spconv_bwd_weight");
    return AGB_OK;
}

int agb_spconv_bwd_weight(const float* X, int ldx, const float* dY, int ldy, const int32_t* nbr, long long nbr_stride,
                          float* dW, int n_out, int K3, int Cin, int Cout, void* stream) {
    return agb_spconv_bwd_weight_lp(X, ldx, dY, ldy, nbr, nbr_stride, dW, n_out, K3, Cin, Cout, 0, stream);
}

// Stride-1 K^3 convolution of a 3-channel input (X rows 4 floats wide) whose neighbours are probed in the level's dense
// grid instead of a pre-built kernel map.  coords int32[n_out][4]; grid / desc = the level's lookup grid ({ox, oy, oz, X,
// Y, Z, ts, B | halo << 16}), halo >= K/2.  W [K^3 * 3, Cout].  nbr_out (optional): int32 [K^3][nbr_out_stride >= n_out],
// receives the kernel map (the same values agb_grid_kernel_map writes) for the weight-gradient pass.
int agb_spconv_fwd3_grid(const float* X, int ldx, const float* W, const int32_t* coords, const int32_t* grid,
                         const int32_t* desc, int K, const float* bias, float* Y, int ldy, int n_out, int Cout,
                         int32_t* nbr_out, long long nbr_out_stride, void* stream) {
    return agb_spconv_fwd3_grid_lp(X, ldx, W, coords, grid, desc, K, bias, Y, ldy, n_out, Cout, nbr_out, nbr_out_stride, 0,
                                   stream);
}

// precision: 0 fp32 MFMA, 1 bf16 operands (k_spconv_fwd3_lp), 2 split-bf16x3 requested: served by the fp32 kernel
static int fwd3_grid_lp_impl(const float* X, int ldx, const float* W, const int32_t* coords, const int32_t* grid,
                             const int32_t* desc, int K, const float* bias, float* Y, int ldy, int n_out, int Cout,
                             int32_t* nbr_out, long long nbr_out_stride, int precision, int y16, void* stream,
                             int dense_form = 0) {
    AGB_CHECK_ARG(precision >= 0 && precision <= 2, "agb_spconv_fwd3_grid_lp: precision %d (0 fp32, 1 bf16, 2 bf16x3)",
                  precision);
    AGB_CHECK_ARG(n_out >= 0 && Cout >= 4 && Cout % 4 == 0 && ldx % 4 == 0 && ldy >= Cout, "agb_spconv_fwd3_grid: bad sizes");
    AGB_CHECK_ARG(coords && grid && desc, "agb_spconv_fwd3_grid: coords, grid and desc are required");
    AGB_CHECK_ARG(K >= 1 && K <= 9 && (K & 1), "agb_spconv_fwd3_grid: kernel size %d (odd, <= 9)", K);
    AGB_CHECK_ARG((long long)(desc[7] & 0xffff) * desc[5] * desc[4] * desc[3] < (1LL << 31),
                  "agb_spconv_fwd3_grid: grid too large for 32-bit cells");
    AGB_CHECK_ARG(((desc[7] >> 16) & 0xff) >= K / 2, "agb_spconv_fwd3_grid: the grid's halo (%d cells) is narrower than "
                  "K/2 = %d", (desc[7] >> 16) & 0xff, K / 2);
    AGB_CHECK_ARG(nbr_out == nullptr || nbr_out_stride >= n_out, "agb_spconv_fwd3_grid: nbr_out_stride < n_out");
    if (n_out == 0) return AGB_OK;
    GridProbe gp;
    gp.coords = (const int4*)coords; gp.grid = grid;
    gp.ox = desc[0]; gp.oy = desc[1]; gp.oz = desc[2]; gp.X = desc[3]; gp.Y = desc[4]; gp.Z = desc[5];
    gp.ts = desc[6]; gp.K = K; gp.nbr_out = nbr_out; gp.nbr_out_stride = nbr_out_stride;
    if (!y16 && precision != 1 && dense_form == 0 && agb_stem_fwd_ok(n_out, K, Cout, ldx)) {
        // fp32 operands, 64 output channels: the pair-sparse kernel of csrc/stem.hip (lane = row, 4x4x1 MFMA on the four-row
        // groups that have the offset): 23 GFLOP issued instead of 57 for the 9 useful
        int rc = agb_stem_fwd_launch(X, W, bias, Y, ldy, n_out, K, coords, grid, desc, nbr_out, nbr_out_stride,
                                     (hipStream_t)stream);
        if (rc) return rc;
        AGB_CHECK_LAUNCH("agb_spconv_fwd3_grid (pair-sparse)");
        return AGB_OK;
    }
    const dim3 grid3(agb_cdiv(n_out, BM), agb_cdiv(Cout, BN));
    // (split-bf16x3 measured SLOWER than the fp32 kernel here — two LDS planes to stage, three MFMAs: 740 vs 683 us —
    // so precision 2 takes the exact fp32 kernel)
    if (y16)
        AGB_LAUNCH((k_spconv_fwd3_lp<false, true>), grid3, dim3(256), 0, (hipStream_t)stream, X, ldx, W, bias, Y, ldy,
                           n_out, K * K * K, Cout, gp);
    else if (precision == 1)
        AGB_LAUNCH((k_spconv_fwd3_lp<false, false>), grid3, dim3(256), 0, (hipStream_t)stream, X, ldx, W, bias, Y, ldy, n_out,
                           K * K * K, Cout, gp);
    else
        AGB_LAUNCH(k_spconv_fwd3<true>, grid3, dim3(256), 0, (hipStream_t)stream, X, ldx, W, nullptr, 0LL, 0, bias, Y,
                           ldy, n_out, K * K * K, Cout, gp);
    AGB_CHECK_LAUNCH("agb_spconv_fwd3_grid");
    return AGB_OK;
}

int agb_spconv_fwd3_grid_lp(const float* X, int ldx, const float* W, const int32_t* coords, const int32_t* grid,
                            const int32_t* desc, int K, const float* bias, float* Y, int ldy, int n_out, int Cout,
                            int32_t* nbr_out, long long nbr_out_stride, int precision, void* stream) {
    return fwd3_grid_lp_impl(X, ldx, W, coords, grid, desc, K, bias, Y, ldy, n_out, Cout, nbr_out, nbr_out_stride, precision, 0,
                             stream);
}

// The dense-over-offsets fp32 form of the same stem (k_spconv_fwd3<true>: what agb_spconv_fwd3_grid ran until round 4 and
// still runs for Cout != 64), kept callable for A/B measurements (tools/bench_stem.py) and as the second implementation the
// parity tests compare the pair-sparse kernel with.
int agb_spconv_fwd3_grid_dense(const float* X, int ldx, const float* W, const int32_t* coords, const int32_t* grid,
                               const int32_t* desc, int K, const float* bias, float* Y, int ldy, int n_out, int Cout,
                               int32_t* nbr_out, long long nbr_out_stride, void* stream) {
    return fwd3_grid_lp_impl(X, ldx, W, coords, grid, desc, K, bias, Y, ldy, n_out, Cout, nbr_out, nbr_out_stride, 0, 0, stream,
                             1);
}

// bf16 operands and bf16 OUTPUT rows (the bf16-activation mode; the 3-channel input features stay fp32): Y16 uint16
// [n_out][ldy16]
int agb_spconv_fwd3_grid_h(const float* X, int ldx, const float* W, const int32_t* coords, const int32_t* grid,
                           const int32_t* desc, int K, const float* bias, uint16_t* Y16, int ldy16, int n_out, int Cout,
                           int32_t* nbr_out, long long nbr_out_stride, void* stream) {
    return fwd3_grid_lp_impl(X, ldx, W, coords, grid, desc, K, bias, reinterpret_cast<float*>(Y16), ldy16, n_out, Cout, nbr_out,
                             nbr_out_stride, 1, 1, stream);
}

}  // extern "C"
