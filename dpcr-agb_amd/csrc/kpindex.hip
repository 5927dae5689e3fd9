// kpindex.hip — KPConv index path on device: radius neighbours and grid subsampling, bit-exact against the
// reference's CPU C++ (which runs single-threaded in the training main loop: SURVEY.md §3.2).
//
//  A1 radius neighbours  replaces batch_nanoflann_neighbors
//       torch_points3d/modules/KPConv/cpp_wrappers/cpp_neighbors/neighbors/neighbors.cpp:211-333
//     same result: every support of the same batch element with d2 < r*r (strict), d2 = ((dx*dx)+(dy*dy))+(dz*dz)
//     in float32 without fused multiply-add (nanoflann.hpp:423-445), ascending d2 (ties: ascending index — the
//     reference's std::sort leaves tie order unspecified), global support index, rows padded with Ns.
//     Method: supports binned into a uniform cell grid (cell = r*1.001, x fastest) by a counting sort; one
//     wavefront per query walks the 9 contiguous cell runs of its 27-cell neighbourhood, compacts hits with
//     ballots into LDS and sorts (d2 bits, index) keys with a bitonic network.
//  A2 grid subsampling   replaces batch_grid_subsampling
//       .../cpp_subsampling/grid_subsampling/grid_subsampling.cpp:5-211, grid_subsampling.h:10-80
//     same cells and bit-equal barycentres: per-cell sums run sequentially in ORIGINAL point order in float32
//     and are scaled by (float)(1.0/count); emission order is canonical (cell key ascending per cloud) — the
//     reference emits in libstdc++ unordered_map iteration order, an artefact no consumer depends on.
#include "agb_common.h"
#include <stdlib.h>
#include "scan.h"
#include <float.h>
#include <limits.h>

// ------------------------------------------------------------------ float min/max via ordered ints
__device__ __forceinline__ int f2ord(float f) {
    int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }

// bbox[b][0..2] = min xyz, [3..5] = max xyz (ordered-int encoding), one block column per batch element
__global__ void k_elem_bbox(const float* __restrict__ pts, const int32_t* __restrict__ ptr, int32_t* bbox) {
    int b = blockIdx.y;
    int beg = ptr[b], end = ptr[b + 1];
    int mn[3] = {INT_MAX, INT_MAX, INT_MAX}, mx[3] = {INT_MIN, INT_MIN, INT_MIN};
    for (int i = beg + blockIdx.x * blockDim.x + threadIdx.x; i < end; i += gridDim.x * blockDim.x) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            int o = f2ord(pts[3 * (long long)i + a]);
            mn[a] = min(mn[a], o);
            mx[a] = max(mx[a], o);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            mn[a] = min(mn[a], __shfl_xor(mn[a], d, 64));
            mx[a] = max(mx[a], __shfl_xor(mx[a], d, 64));
        }
    }
    if ((threadIdx.x & 63) == 0 && mx[0] != INT_MIN) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            atomicMin(&bbox[6 * b + a], mn[a]);
            atomicMax(&bbox[6 * b + 3 + a], mx[a]);
        }
    }
}

__global__ void k_bbox_fill(int32_t* bbox, int B) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < 6 * B) bbox[t] = (t % 6) < 3 ? INT_MAX : INT_MIN;
}

// decoded float bbox for the host: out[b][6]
__global__ void k_bbox_decode(const int32_t* bbox, int B, float* out) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 6 * B) return;
    const int v = bbox[t];      // an empty cloud decodes to (+inf, -inf): min / max over clouds pass it by
    out[t] = v == INT_MAX ? INFINITY : v == INT_MIN ? -INFINITY : ord2f(v);
}

// =================================================================== radius neighbours
struct CellGrid {
    float ox, oy, oz;  // origin
    float inv_cs;      // 1 / cell size
    int X, Y, Z;       // cells per axis (shared by all batch elements)
    int B;
};

__device__ __forceinline__ int cell_coord(float v, float o, float inv_cs, int n) {
    int c = (int)floorf((v - o) * inv_cs);
    return c < 0 ? 0 : (c >= n ? n - 1 : c);
}

__global__ void k_cell_count(const float* __restrict__ sup, const int32_t* __restrict__ s_ptr, CellGrid g,
                             int32_t* cell_cnt, int32_t* cell_of) {
    int b = blockIdx.y;
    int beg = s_ptr[b], end = s_ptr[b + 1];
    for (int i = beg + blockIdx.x * blockDim.x + threadIdx.x; i < end; i += gridDim.x * blockDim.x) {
        const float* p = sup + 3 * (long long)i;
        int cx = cell_coord(p[0], g.ox, g.inv_cs, g.X), cy = cell_coord(p[1], g.oy, g.inv_cs, g.Y),
            cz = cell_coord(p[2], g.oz, g.inv_cs, g.Z);
        int c = ((b * g.Z + cz) * g.Y + cy) * g.X + cx;
        cell_of[i] = c;
        atomicAdd(&cell_cnt[c], 1);
    }
}

// scatter supports into cell order: sorted[j] = (x, y, z, global index bits)
__global__ void k_cell_scatter(const float* __restrict__ sup, int ns, const int32_t* __restrict__ cell_of,
                               const int32_t* __restrict__ cell_start, int32_t* cell_fill, float4* sorted) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns) return;
    int c = cell_of[i];
    int j = cell_start[c] + atomicAdd(&cell_fill[c], 1);
    const float* p = sup + 3 * (long long)i;
    sorted[j] = make_float4(p[0], p[1], p[2], __int_as_float(i));
}

#define BQ_CAP 1024  // hits one query can hold in LDS (the 16k-point plots of BASELINE.json peak at 265)

// The whole wave sorts one query's list of `count` keys (LDS, room for the next power of two >= max(count, 64)) by (d2, index)
// and writes the row: ragged (row_ptr) or padded with the shadow index ns.
__device__ __forceinline__ void bq_sort_emit(const int q, const int lane, unsigned long long* __restrict__ skeys, int count,
                                             int32_t* __restrict__ out, int width, int ns, const int32_t* __restrict__ row_ptr) {
    if (count <= 64) {
        // At most one key per lane (the usual case: ~20-40 neighbours): rank = number of smaller keys, counted against
        // broadcast LDS reads of the list — ~4 instructions per key instead of the 21 compare-exchange passes of a 64-key
        // bitonic network, which were 70 % of this kernel's instructions.  Keys are distinct (the index is in the low
        // word): the ranks are a permutation, the order is exactly the sorted one.
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const unsigned long long mykey = lane < count ? skeys[lane] : ~0ull;
        int rank = 0;
        for (int j = 0; j < count; ++j) rank += skeys[j] < mykey ? 1 : 0;
        int32_t* row = row_ptr ? out + row_ptr[q] : out + (long long)q * width;
        if (row_ptr) width = count;
        if (lane < count && rank < width) row[rank] = (int)(unsigned)(mykey & 0xFFFFFFFFull);
        if (!row_ptr)
            for (int j = count + lane; j < width; j += 64) row[j] = ns;
        return;
    }
    // bitonic sort of the hit keys (padded with all-ones) — wave-synchronous on the wave's own LDS slab
    int n2 = 64;
    while (n2 < count) n2 <<= 1;
    for (int i = count + lane; i < n2; i += 64) skeys[i] = ~0ull;
    __builtin_amdgcn_wave_barrier();
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = lane; i < n2; i += 64) {
                int l = i ^ j;
                if (l > i) {
                    unsigned long long a = skeys[i], c2 = skeys[l];
                    bool up = (i & k) == 0;
                    if ((a > c2) == up) {
                        skeys[i] = c2;
                        skeys[l] = a;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (row_ptr) {
        int32_t* row = out + row_ptr[q];
        for (int j = lane; j < count; j += 64) row[j] = (int)(unsigned)(skeys[j] & 0xFFFFFFFFull);
        return;
    }
    for (int j = lane; j < width; j += 64)
        out[(long long)q * width + j] = j < count ? (int)(unsigned)(skeys[j] & 0xFFFFFFFFull) : ns;
}

// one wavefront per query (`skeys`: the wave's slab of BQ_CAP keys). FILL = false: counts only.
template <bool FILL>
__device__ __forceinline__ void bq_one_query(const int q, const int lane, unsigned long long* __restrict__ skeys,
                                             const float* __restrict__ qry, const int32_t* __restrict__ q_elem, const CellGrid& g,
                                             const int32_t* __restrict__ cell_start, const float4* __restrict__ sorted, float r2,
                                             int ns, int32_t* __restrict__ counts, int32_t* __restrict__ out, int width,
                                             int32_t* status, const int32_t* __restrict__ row_ptr, const int cap = BQ_CAP) {
    // cap: keys the slab holds (BQ_CAP, or the smaller slab of k_ball_query4 when the caller knows the longest list)
    // row_ptr != NULL (FILL): ragged output — row q is out[row_ptr[q] .. row_ptr[q + 1]) (the exclusive scan of the count
    // pass), nothing is padded; NULL: the reference's padded matrix out[q * width + j], shadow index ns behind the row
    const float qx = qry[3 * (long long)q], qy = qry[3 * (long long)q + 1], qz = qry[3 * (long long)q + 2];
    const int b = q_elem[q];
    const int cx = cell_coord(qx, g.ox, g.inv_cs, g.X), cy = cell_coord(qy, g.oy, g.inv_cs, g.Y),
              cz = cell_coord(qz, g.oz, g.inv_cs, g.Z);
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.X - 1);
    // lanes 0..8 fetch the 9 (dy,dz) runs; every run covers 3 x-adjacent cells = one contiguous range
    int beg = 0, len = 0;
    if (lane < 9) {
        int yy = cy + (lane % 3) - 1, zz = cz + (lane / 3) - 1;
        if (yy >= 0 && yy < g.Y && zz >= 0 && zz < g.Z) {
            int row = ((b * g.Z + zz) * g.Y + yy) * g.X;
            beg = cell_start[row + x0];
            len = cell_start[row + x1 + 1] - beg;
        }
    }
    // inclusive prefix of the run lengths over lanes 0..8
    int pre = len;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
        int t = __shfl_up(pre, d, 64);
        if (lane >= d) pre += t;
    }
    const int total = __shfl(pre, 8, 64);
    int count = 0;
    for (int base = 0; base < total; base += 64) {
        const int c = base + lane;
        const bool valid = c < total;
        // which run holds candidate c — all 64 lanes take part in the shuffles (no divergence around them)
        const int cc = valid ? c : 0;
        int run = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) run += (cc >= __shfl(pre, j, 64)) ? 1 : 0;
        const int rbeg = __shfl(beg, run, 64), rpre = __shfl(pre, run, 64), rlen = __shfl(len, run, 64);
        bool hit = false;
        unsigned long long key = 0;
        if (valid) {
            float4 s = sorted[rbeg + (c - (rpre - rlen))];
            float dx = qx - s.x, dy = qy - s.y, dz = qz - s.z;
            float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
            hit = d2 < r2;
            key = ((unsigned long long)(unsigned)__float_as_int(d2) << 32) | (unsigned)__float_as_int(s.w);
        }
        unsigned long long m = __ballot(hit);
        if (FILL && hit) {
            int pos = count + __popcll(m & ((1ull << lane) - 1ull));
            if (pos < cap) skeys[pos] = key;
        }
        count += __popcll(m);
    }
    if (!FILL) {
        if (lane == 0) counts[q] = count;
        return;
    }
    if (count > cap) {
        if (lane == 0) atomicAdd(&status[0], 1);
        count = cap;
    }
    bq_sort_emit(q, lane, skeys, count, out, width, ns, row_ptr);
}


template <bool FILL>
__global__ __launch_bounds__(256) void k_ball_query(const float* __restrict__ qry, int nq,
                                                    const int32_t* __restrict__ q_elem,  // batch element of a query
                                                    CellGrid g, const int32_t* __restrict__ cell_start,
                                                    const float4* __restrict__ sorted, float r2, int ns,
                                                    int32_t* __restrict__ counts, int32_t* __restrict__ out,
                                                    int width, int32_t* status,
                                                    const int32_t* __restrict__ row_ptr = nullptr) {
    __shared__ unsigned long long s_keys[4][FILL ? BQ_CAP : 1];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + w;
    if (q >= nq) return;
    bq_one_query<FILL>(q, lane, s_keys[w], qry, q_elem, g, cell_start, sorted, r2, ns, counts, out, width, status, row_ptr);
}

// FOUR queries per wavefront, 16 lanes each (round 6: the one-query-per-wave form spent ~300 wave instructions per query —
// cell arithmetic, nine run look-ups and a 9-lane prefix for ONE query per wave, a run search per candidate, a rank loop
// of count iterations for one key per lane).  A quarter walks its nine (dy, dz) runs one after the other, 16 candidates per
// step (a run of three x-adjacent cells holds ~14 points: no search for "which run holds candidate c"); hits are compacted
// per quarter with one ballot; every lane ranks its keys (positions sl, sl + 16, ..) against the quarter's list.  Same
// candidates, same d2 arithmetic, same (d2, index) order as bq_one_query: bit-identical rows.  A query with more than
// cap / 4 hits is redone by the whole wave with bq_one_query on the wave's full slab (cap keys) after the
// other quarters have written their rows.
// The slab of a wave holds `cap` keys (dynamic LDS: 4 waves x cap x 8 bytes), a quarter's list cap / 4: with the longest list of
// the search known (agb_ball_query_fill_csr_m: the count pass's maximum, which the caller has read back anyway) the slab
// is the next power of two >= max(that, 256) instead of BQ_CAP = 1024 — 8-16 KB of LDS per workgroup instead of 32, twice
// the waves per CU in a kernel that waits for memory 64 % of its time (level 0 fill pass: 426 -> 286 us).
template <bool FILL>
__global__ __launch_bounds__(256) void k_ball_query4(const float* __restrict__ qry, int nq, const int32_t* __restrict__ q_elem,
                                                     CellGrid g, const int32_t* __restrict__ cell_start,
                                                     const float4* __restrict__ sorted, float r2, int ns,
                                                     int32_t* __restrict__ counts, int32_t* __restrict__ out, int width,
                                                     int32_t* status, const int32_t* __restrict__ row_ptr = nullptr,
                                                     const int cap = BQ_CAP) {
    extern __shared__ unsigned long long bq_keys_dyn[];              // [4 waves][cap]  (FILL only)
    const int BQ4_CAP = cap >> 2;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, sub = lane >> 4, sl = lane & 15;
    const int q = (blockIdx.x * 4 + w) * 4 + sub;
    if ((blockIdx.x * 4 + w) * 4 >= nq) return;                  // (whole waves leave: no workgroup barrier in this kernel)
    const bool live = q < nq;
    const int qc = live ? q : nq - 1;
    const float qx = qry[3 * (long long)qc], qy = qry[3 * (long long)qc + 1], qz = qry[3 * (long long)qc + 2];
    const int b = q_elem[qc];
    const int cx = cell_coord(qx, g.ox, g.inv_cs, g.X), cy = cell_coord(qy, g.oy, g.inv_cs, g.Y),
              cz = cell_coord(qz, g.oz, g.inv_cs, g.Z);
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.X - 1);
    // lanes 0..8 of a quarter fetch its 9 (dy, dz) runs; every run covers 3 x-adjacent cells = one contiguous range
    int beg = 0, len = 0;
    if (live && sl < 9) {
        const int yy = cy + (sl % 3) - 1, zz = cz + (sl / 3) - 1;
        if (yy >= 0 && yy < g.Y && zz >= 0 && zz < g.Z) {
            const int row = ((b * g.Z + zz) * g.Y + yy) * g.X;
            beg = cell_start[row + x0];
            len = cell_start[row + x1 + 1] - beg;
        }
    }
    unsigned long long* const slab = bq_keys_dyn + (FILL ? (size_t)cap * (threadIdx.x >> 6) : 0);
    unsigned long long* keys = slab + (FILL ? BQ4_CAP * sub : 0);
    int count = 0;
    auto take = [&](const float4 s, const bool valid) {
        bool hit = false;
        unsigned long long key = 0;
        if (valid) {
            const float dx = qx - s.x, dy = qy - s.y, dz = qz - s.z;
            const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
            hit = d2 < r2;
            key = ((unsigned long long)(unsigned)__float_as_int(d2) << 32) | (unsigned)__float_as_int(s.w);
        }
        const unsigned m16 = (unsigned)(__ballot(hit) >> (16 * sub)) & 0xFFFFu;
        if (FILL && hit) {
            const int pos = count + __popc(m16 & ((1u << sl) - 1u));
            if (pos < BQ4_CAP) keys[pos] = key;
        }
        count += __popc(m16);
    };
    // the first 16 candidates of all nine runs are requested before any is looked at (a run holds ~14 points: one memory
    // round trip per query instead of nine); longer runs continue below
    int rbs[9], rls[9];
    float4 sv[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        rbs[r] = __shfl(beg, r, 16);
        rls[r] = __shfl(len, r, 16);
        sv[r] = sl < rls[r] ? sorted[rbs[r] + sl] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int r = 0; r < 9; ++r) take(sv[r], sl < rls[r]);
#pragma unroll 1
    for (int r = 0; r < 9; ++r) {
        const int rb = __shfl(beg, r, 16), rl = __shfl(len, r, 16);
        for (int off = 16; __ballot(off < rl) != 0ull; off += 16) {
            const int c = off + sl;
            take(c < rl ? sorted[rb + c] : make_float4(0.f, 0.f, 0.f, 0.f), c < rl);
        }
    }
    if (!FILL) {
        if (live && sl == 0) counts[q] = count;
        return;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const bool over = count > BQ4_CAP;                           // (uniform in the quarter)
    const int cnt = (live && !over) ? count : 0;
    // rank of every key = number of smaller keys of the quarter's list (keys are distinct: the index is the low word)
    int32_t* row = nullptr;
    int wq = width;
    if (live) {
        row = row_ptr ? out + row_ptr[q] : out + (long long)q * width;
        if (row_ptr) wq = count;
    }
    // lists of at most 32 keys (the self-searches: ~20-35 neighbours): ranked inside the quarter; longer ones (the pooling
    // searches: up to ~270 at twice the radius) are sorted by the whole wave below, one query at a time
    const bool small = cnt <= 32;
    const int cs = small ? cnt : 0;
    int csmax = cs;
    csmax = max(csmax, __shfl_xor(csmax, 16, 64));
    csmax = max(csmax, __shfl_xor(csmax, 32, 64));
    if (csmax > 0) {
        // at most two keys per lane (the usual case: ~20-35 neighbours): every lane meets the keys of the other 15 lanes of its
        // quarter by rotating them through the 16-lane row (DPP row_ror: no LDS read, no wait)
        const unsigned long long ka = sl < cs ? keys[sl] : ~0ull, kb = 16 + sl < cs ? keys[16 + sl] : ~0ull;
        const unsigned kal = (unsigned)ka, kah = (unsigned)(ka >> 32), kbl = (unsigned)kb, kbh = (unsigned)(kb >> 32);
        int ra = kb < ka ? 1 : 0, rb2 = ka < kb ? 1 : 0;
        const bool two = csmax > 16;
#define BQ_ROT(N)                                                                                                          \
        {                                                                                                                  \
            const unsigned long long oa = ((unsigned long long)(unsigned)__builtin_amdgcn_mov_dpp((int)kah, 0x120 + N, 0xF, 0xF, false) << 32) | \
                                          (unsigned)__builtin_amdgcn_mov_dpp((int)kal, 0x120 + N, 0xF, 0xF, false);         \
            ra += oa < ka ? 1 : 0;                                                                                         \
            if (two) {                                                                                                     \
                const unsigned long long ob = ((unsigned long long)(unsigned)__builtin_amdgcn_mov_dpp((int)kbh, 0x120 + N, 0xF, 0xF, false) << 32) | \
                                              (unsigned)__builtin_amdgcn_mov_dpp((int)kbl, 0x120 + N, 0xF, 0xF, false);     \
                ra += ob < ka ? 1 : 0;                                                                                     \
                rb2 += (oa < kb ? 1 : 0) + (ob < kb ? 1 : 0);                                                              \
            }                                                                                                              \
        }
        BQ_ROT(1) BQ_ROT(2) BQ_ROT(3) BQ_ROT(4) BQ_ROT(5) BQ_ROT(6) BQ_ROT(7) BQ_ROT(8) BQ_ROT(9) BQ_ROT(10) BQ_ROT(11)
        BQ_ROT(12) BQ_ROT(13) BQ_ROT(14) BQ_ROT(15)
#undef BQ_ROT
        if (sl < cs && ra < wq) row[ra] = (int)kal;
        if (16 + sl < cs && rb2 < wq) row[rb2] = (int)kbl;
    }
    const unsigned long long big = __ballot(live && !over && !small);
    if (big != 0ull) {
        __builtin_amdgcn_wave_barrier();
        const int qbase = (blockIdx.x * 4 + w) * 4;
#pragma unroll 1
        for (int t = 0; t < 4; ++t)
            if ((big >> (16 * t)) & 1ull)
                bq_sort_emit(qbase + t, lane, slab + BQ4_CAP * t, __shfl(count, 16 * t, 64), out, width, ns, row_ptr);
    }
    if (live && small && !over && !row_ptr)
        for (int j = cnt + sl; j < width; j += 16) row[j] = ns;
    // queries with more hits than a quarter's list: the whole wave, one query at a time, on the wave's full slab
    const unsigned long long ov = __ballot(live && over);
    if (ov != 0ull) {
        __builtin_amdgcn_wave_barrier();
        const int qbase = (blockIdx.x * 4 + w) * 4;
#pragma unroll 1
        for (int t = 0; t < 4; ++t)
            if ((ov >> (16 * t)) & 1ull)
                bq_one_query<true>(qbase + t, lane, slab, qry, q_elem, g, cell_start, sorted, r2, ns, counts, out, width, status,
                                   row_ptr, cap);
    }
}

// ragged -> the reference's padded matrix (for callers of batch_neighbors; the kernels of this library walk the ragged form)
__global__ void k_csr_to_padded(const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ indices, int nq, int width,
                                int pad, int32_t* __restrict__ out) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)nq * width) return;
    const int q = (int)(t / width), j = (int)(t % width);
    const int beg = row_ptr[q], cnt = row_ptr[q + 1] - beg;
    out[t] = j < cnt ? indices[beg + j] : pad;
}

// batch element of every row from the element pointer (B is small: linear search per row)
__global__ void k_elem_of_row(const int32_t* __restrict__ ptr, int B, int n, int32_t* elem) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int b = 0;
    while (b + 1 < B && i >= ptr[b + 1]) ++b;
    elem[i] = b;
}

__global__ void k_max_i32(const int32_t* __restrict__ v, int n, int32_t* out) {
    int m = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) m = max(m, v[i]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, __shfl_xor(m, d, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

// =================================================================== grid subsampling
struct SubGrid {
    float dl;
    int cap;  // cells reserved per batch element
    int B;
};

// per element: origin (3 floats), NX, NY, NZ from the element bbox — same float arithmetic as the reference
__global__ void k_sub_geometry(const int32_t* __restrict__ bbox, SubGrid g, float* origin, int32_t* dims,
                               int32_t* status) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= g.B) return;
    if (bbox[6 * b + 3] == INT_MIN) {  // empty element
        dims[3 * b] = dims[3 * b + 1] = dims[3 * b + 2] = 0;
        return;
    }
    float inv = __fdiv_rn(1.f, g.dl);
    long long cells = 1;
    for (int a = 0; a < 3; ++a) {
        float mn = ord2f(bbox[6 * b + a]), mx = ord2f(bbox[6 * b + 3 + a]);
        float o = __fmul_rn(floorf(__fmul_rn(mn, inv)), g.dl);
        origin[3 * b + a] = o;
        int n = (int)floorf(__fdiv_rn(__fsub_rn(mx, o), g.dl)) + 1;
        dims[3 * b + a] = n;
        cells *= n;
    }
    if (cells > g.cap) atomicAdd(&status[0], 1);
}

__device__ __forceinline__ int sub_cell(const float* p, const float* o, const int32_t* d, float dl) {
    int ix = (int)floorf(__fdiv_rn(__fsub_rn(p[0], o[0]), dl));
    int iy = (int)floorf(__fdiv_rn(__fsub_rn(p[1], o[1]), dl));
    int iz = (int)floorf(__fdiv_rn(__fsub_rn(p[2], o[2]), dl));
    return ix + d[0] * (iy + d[1] * iz);
}

__global__ void k_sub_count(const float* __restrict__ pts, const int32_t* __restrict__ elem, int n, SubGrid g,
                            const float* __restrict__ origin, const int32_t* __restrict__ dims, int32_t* cell_cnt,
                            int32_t* cell_of) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int b = elem[i];
    int c = b * g.cap + sub_cell(pts + 3 * (long long)i, origin + 3 * b, dims + 3 * b, g.dl);
    cell_of[i] = c;
    atomicAdd(&cell_cnt[c], 1);
}

__global__ void k_nonzero_flag(const int32_t* __restrict__ cnt, int n, int32_t* flag) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = cnt[i] > 0;
}

__global__ void k_sub_scatter(int n, const int32_t* __restrict__ cell_of, const int32_t* __restrict__ cell_start,
                              int32_t* cell_fill, int32_t* members) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int c = cell_of[i];
    members[cell_start[c] + atomicAdd(&cell_fill[c], 1)] = i;
}

// one thread per non-empty cell: order its members by original index, sum sequentially in float, emit
__global__ void k_sub_emit(const float* __restrict__ pts, const float* __restrict__ feats, int fdim, int ncells,
                           const int32_t* __restrict__ cell_cnt, const int32_t* __restrict__ cell_start,
                           const int32_t* __restrict__ slot, int32_t* members, float* __restrict__ out_pts,
                           float* __restrict__ out_feats) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncells) return;
    int cnt = cell_cnt[c];
    if (cnt == 0) return;
    int32_t* m = members + cell_start[c];
    for (int i = 1; i < cnt; ++i) {  // insertion sort: cells hold a handful of points
        int v = m[i], j = i - 1;
        while (j >= 0 && m[j] > v) {
            m[j + 1] = m[j];
            --j;
        }
        m[j + 1] = v;
    }
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int i = 0; i < cnt; ++i) {
        const float* p = pts + 3 * (long long)m[i];
        sx = __fadd_rn(sx, p[0]);
        sy = __fadd_rn(sy, p[1]);
        sz = __fadd_rn(sz, p[2]);
    }
    float a = (float)(1.0 / (double)cnt);
    int o = slot[c];
    out_pts[3 * (long long)o] = __fmul_rn(sx, a);
    out_pts[3 * (long long)o + 1] = __fmul_rn(sy, a);
    out_pts[3 * (long long)o + 2] = __fmul_rn(sz, a);
    if (feats) {
        float fc = (float)cnt;
        for (int f = 0; f < fdim; ++f) {
            float s = 0.f;
            for (int i = 0; i < cnt; ++i) s = __fadd_rn(s, feats[(long long)m[i] * fdim + f]);
            out_feats[(long long)o * fdim + f] = __fdiv_rn(s, fc);
        }
    }
}

// out_ptr[b] = number of non-empty cells before element b (slot of its first cell); out_ptr[B] = total
__global__ void k_sub_elem_ptr(const int32_t* __restrict__ slot, const int32_t* total, SubGrid g, int32_t* out_ptr) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < g.B) out_ptr[b] = slot[(long long)b * g.cap];
    if (b == g.B) out_ptr[b] = *total;
}

// out[i] = p[i] . R[elem(i)]  i.e. out_j = (p0*R0j + p1*R1j) + p2*R2j — the random grid orientation of
// common.py:53-81 (numpy float32 products summed left to right), applied without fused multiply-add
__global__ void k_rotate_points(const float* __restrict__ pts, const int32_t* __restrict__ elem,
                                const float* __restrict__ R, int n, int transpose, float* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = R + 9 * (long long)elem[i];
    float p0 = pts[3 * (long long)i], p1 = pts[3 * (long long)i + 1], p2 = pts[3 * (long long)i + 2];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float a = transpose ? r[3 * j] : r[j], b = transpose ? r[3 * j + 1] : r[3 + j],
              c = transpose ? r[3 * j + 2] : r[6 + j];
        out[3 * (long long)i + j] = __fadd_rn(__fadd_rn(__fmul_rn(p0, a), __fmul_rn(p1, b)), __fmul_rn(p2, c));
    }
}

// =============================================================== C ABI
extern "C" {

// out = pts @ R[elem] (transpose = 0) or pts @ R[elem]^T (transpose = 1); R: float[B,3,3]
int agb_rotate_points(const float* pts, const int32_t* elem, const float* R, int n, int transpose, float* out,
                      void* stream) {
    if (n > 0)
        hipLaunchKernelGGL(k_rotate_points, dim3(agb_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, pts, elem, R, n,
                           transpose, out);
    AGB_CHECK_LAUNCH("agb_rotate_points");
    return AGB_OK;
}

// per-element bounding boxes. bbox_ord: int32[6*B] scratch; bbox: float[6*B] out (min xyz, max xyz)
int agb_elem_bbox(const float* pts, const int32_t* ptr, int B, int n, int32_t* bbox_ord, float* bbox, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_bbox_fill, dim3(agb_cdiv(6 * B, 256)), dim3(256), 0, s, bbox_ord, B);
    if (n > 0) {
        int bx = agb_cdiv(agb_cdiv(n, B > 0 ? B : 1), 256);
        if (bx > 64) bx = 64;
        if (bx < 1) bx = 1;
        hipLaunchKernelGGL(k_elem_bbox, dim3(bx, B), dim3(256), 0, s, pts, ptr, bbox_ord);
    }
    if (bbox) hipLaunchKernelGGL(k_bbox_decode, dim3(agb_cdiv(6 * B, 256)), dim3(256), 0, s, bbox_ord, B, bbox);
    AGB_CHECK_LAUNCH("agb_elem_bbox");
    return AGB_OK;
}

int agb_elem_of_row(const int32_t* ptr, int B, int n, int32_t* elem, void* stream) {
    if (n > 0)
        hipLaunchKernelGGL(k_elem_of_row, dim3(agb_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, ptr, B, n, elem);
    AGB_CHECK_LAUNCH("agb_elem_of_row");
    return AGB_OK;
}

// Build the support cell grid. grid_desc (host float[4] + int[4]): origin xyz, cell size; X, Y, Z, B.
// cell_start: int32[cells+1] out; sorted: float4[ns] out; cell_of, cell_fill: int32[ns], int32[cells] scratch;
// scan_scratch: int32[agb_scan_scratch_elems(cells+1)].
int agb_ball_grid_build(const float* supports, int ns, const int32_t* s_ptr, const float* origin_cs,
                        const int32_t* dims, int32_t* cell_start, float* sorted, int32_t* cell_of,
                        int32_t* cell_fill, int32_t* scan_scratch, int32_t* total_scratch, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    CellGrid g{origin_cs[0], origin_cs[1], origin_cs[2], 1.0f / origin_cs[3], dims[0], dims[1], dims[2], dims[3]};
    long long cells = (long long)g.X * g.Y * g.Z * g.B;
    AGB_CHECK_ARG(cells > 0 && cells < 0x7FFFFFF0LL, "agb_ball_grid_build: %lld cells out of range", cells);
    int nc = (int)cells + 1;
    (void)hipMemsetAsync(cell_fill, 0, sizeof(int32_t) * (size_t)nc, s);
    if (ns > 0) {
        int bx = agb_cdiv(agb_cdiv(ns, g.B), 256);
        if (bx > 64) bx = 64;
        if (bx < 1) bx = 1;
        hipLaunchKernelGGL(k_cell_count, dim3(bx, g.B), dim3(256), 0, s, supports, s_ptr, g, cell_fill, cell_of);
    }
    agb_launch_exclusive_scan(cell_fill, nc, cell_start, scan_scratch, total_scratch, s);
    (void)hipMemsetAsync(cell_fill, 0, sizeof(int32_t) * (size_t)nc, s);
    if (ns > 0)
        hipLaunchKernelGGL(k_cell_scatter, dim3(agb_cdiv(ns, 256)), dim3(256), 0, s, supports, ns, cell_of, cell_start,
                           cell_fill, (float4*)sorted);
    AGB_CHECK_LAUNCH("agb_ball_grid_build");
    return AGB_OK;
}

// AGB_BALL_QUERY_V1=1: the one-query-per-wave kernels of rounds 2-5 (A/B: tools/ballquery_ab.py); read once
static bool bq_use_v1() {
    static const bool v1 = [] { const char* e = getenv("AGB_BALL_QUERY_V1"); return e && e[0] == '1'; }();
    return v1;
}

// counts[nq] and *max_count (device int, zeroed here)
int agb_ball_query_count(const float* queries, int nq, const int32_t* q_elem, const float* origin_cs,
                         const int32_t* dims, const int32_t* cell_start, const float* sorted, float radius,
                         int32_t* counts, int32_t* max_count, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    CellGrid g{origin_cs[0], origin_cs[1], origin_cs[2], 1.0f / origin_cs[3], dims[0], dims[1], dims[2], dims[3]};
    (void)hipMemsetAsync(max_count, 0, sizeof(int32_t), s);
    if (nq > 0) {
        float r2 = radius * radius;
        if (bq_use_v1())
            hipLaunchKernelGGL(k_ball_query<false>, dim3(agb_cdiv(nq, 4)), dim3(256), 0, s, queries, nq, q_elem, g,
                               cell_start, (const float4*)sorted, r2, 0, counts, (int32_t*)nullptr, 0, (int32_t*)nullptr);
        else
            hipLaunchKernelGGL(k_ball_query4<false>, dim3(agb_cdiv(nq, 16)), dim3(256), 0, s, queries, nq, q_elem, g,
                               cell_start, (const float4*)sorted, r2, 0, counts, (int32_t*)nullptr, 0, (int32_t*)nullptr,
                               (const int32_t*)nullptr, BQ_CAP);
        int bx = agb_cdiv(nq, 256);
        if (bx > 256) bx = 256;
        hipLaunchKernelGGL(k_max_i32, dim3(bx), dim3(256), 0, s, counts, nq, max_count);
    }
    AGB_CHECK_LAUNCH("agb_ball_query_count");
    return AGB_OK;
}

// out: int32[nq, width], rows sorted by (d2, index), padded with ns. status[0] counts rows over the LDS capacity.
int agb_ball_query_fill(const float* queries, int nq, const int32_t* q_elem, const float* origin_cs,
                        const int32_t* dims, const int32_t* cell_start, const float* sorted, float radius, int ns,
                        int width, int32_t* out, int32_t* status, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    CellGrid g{origin_cs[0], origin_cs[1], origin_cs[2], 1.0f / origin_cs[3], dims[0], dims[1], dims[2], dims[3]};
    (void)hipMemsetAsync(status, 0, sizeof(int32_t) * 4, s);
    if (nq > 0 && width > 0) {
        float r2 = radius * radius;
        if (bq_use_v1())
            hipLaunchKernelGGL(k_ball_query<true>, dim3(agb_cdiv(nq, 4)), dim3(256), 0, s, queries, nq, q_elem, g,
                               cell_start, (const float4*)sorted, r2, ns, (int32_t*)nullptr, out, width, status);
        else
            hipLaunchKernelGGL(k_ball_query4<true>, dim3(agb_cdiv(nq, 16)), dim3(256), 4 * BQ_CAP * sizeof(unsigned long long), s,
                               queries, nq, q_elem, g, cell_start, (const float4*)sorted, r2, ns, (int32_t*)nullptr, out, width,
                               status, (const int32_t*)nullptr, BQ_CAP);
    }
    AGB_CHECK_LAUNCH("agb_ball_query_fill");
    return AGB_OK;
}

// Ragged (CSR) form of the radius search — SURVEY.md §8(d): the ball query writes sum(counts) * 4 bytes instead of the padded
// nq * max_count * 4 (the 16 k-point plots: 20 valid of 265 columns).  Call order: agb_ball_query_count (counts, max_count),
// agb_ball_query_offsets (row_ptr int32[nq + 1] = exclusive scan of counts, row_ptr[nq] = total; scratch int32[
// agb_scan_scratch_elems(nq)]), read row_ptr[nq] back, agb_ball_query_fill_csr (indices int32[total]: every row sorted by
// (d2, index) exactly like the padded rows).  agb_csr_to_padded rebuilds the reference's matrix (pad = ns) for API callers
// (cpp_neighbors/neighbors.cpp:319-325 pads to the batch-wide maximum).
int agb_ball_query_offsets(const int32_t* counts, int nq, int32_t* row_ptr, int32_t* scratch, void* stream) {
    AGB_CHECK_ARG(nq >= 0 && row_ptr != nullptr, "agb_ball_query_offsets: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (nq == 0) { (void)hipMemsetAsync(row_ptr, 0, sizeof(int32_t), s); return AGB_OK; }
    AGB_CHECK_ARG(counts && scratch, "agb_ball_query_offsets: null pointer");
    agb_launch_exclusive_scan(counts, nq, row_ptr, scratch, row_ptr + nq, s);
    AGB_CHECK_LAUNCH("agb_ball_query_offsets");
    return AGB_OK;
}

// slab of a wave for a search whose longest list is max_count: next power of two >= max(max_count, 256), at most BQ_CAP
static int bq_slab_keys(int max_count) {
    int cap = 256;
    while (cap < max_count && cap < BQ_CAP) cap <<= 1;
    return cap;
}

static int bq_fill_csr(const float* queries, int nq, const int32_t* q_elem, const float* origin_cs, const int32_t* dims,
                       const int32_t* cell_start, const float* sorted, float radius, int ns, const int32_t* row_ptr,
                       int32_t* indices, int capacity, int cap, int32_t* status, void* stream, const char* who) {
    AGB_CHECK_ARG(capacity >= 0, "%s: capacity %d", who, capacity);   // (= row_ptr[nq], read back by the caller)
    hipStream_t s = (hipStream_t)stream;
    CellGrid g{origin_cs[0], origin_cs[1], origin_cs[2], 1.0f / origin_cs[3], dims[0], dims[1], dims[2], dims[3]};
    (void)hipMemsetAsync(status, 0, sizeof(int32_t) * 4, s);
    if (nq > 0) {
        AGB_CHECK_ARG(row_ptr && indices, "%s: null pointer", who);
        float r2 = radius * radius;
        if (bq_use_v1())
            hipLaunchKernelGGL(k_ball_query<true>, dim3(agb_cdiv(nq, 4)), dim3(256), 0, s, queries, nq, q_elem, g,
                               cell_start, (const float4*)sorted, r2, ns, (int32_t*)nullptr, indices, 0, status, row_ptr);
        else
            hipLaunchKernelGGL(k_ball_query4<true>, dim3(agb_cdiv(nq, 16)), dim3(256), 4 * (size_t)cap * sizeof(unsigned long long), s,
                               queries, nq, q_elem, g, cell_start, (const float4*)sorted, r2, ns, (int32_t*)nullptr, indices, 0,
                               status, row_ptr, cap);
    }
    return AGB_OK;
}

int agb_ball_query_fill_csr(const float* queries, int nq, const int32_t* q_elem, const float* origin_cs,
                            const int32_t* dims, const int32_t* cell_start, const float* sorted, float radius, int ns,
                            const int32_t* row_ptr, int32_t* indices, int capacity, int32_t* status, void* stream) {
    int rc = bq_fill_csr(queries, nq, q_elem, origin_cs, dims, cell_start, sorted, radius, ns, row_ptr, indices, capacity, BQ_CAP,
                         status, stream, "agb_ball_query_fill_csr");
    if (rc) return rc;
    AGB_CHECK_LAUNCH("agb_ball_query_fill_csr");
    return AGB_OK;
}

// The same with the longest list of the search stated (max_count: *max_count of agb_ball_query_count, which the caller reads back
// together with row_ptr[nq]): the kernel's LDS slab shrinks to the next power of two >= max(max_count, 256) keys per wave and
// twice to four times the waves fit a CU.  A list longer than max_count is cut at the slab and counted in status[0].
int agb_ball_query_fill_csr_m(const float* queries, int nq, const int32_t* q_elem, const float* origin_cs, const int32_t* dims,
                              const int32_t* cell_start, const float* sorted, float radius, int ns, const int32_t* row_ptr,
                              int32_t* indices, int capacity, int max_count, int32_t* status, void* stream) {
    AGB_CHECK_ARG(max_count >= 0, "agb_ball_query_fill_csr_m: max_count %d", max_count);
    int rc = bq_fill_csr(queries, nq, q_elem, origin_cs, dims, cell_start, sorted, radius, ns, row_ptr, indices, capacity,
                         bq_slab_keys(max_count), status, stream, "agb_ball_query_fill_csr_m");
    if (rc) return rc;
    AGB_CHECK_LAUNCH("agb_ball_query_fill_csr_m");
    return AGB_OK;
}

int agb_csr_to_padded(const int32_t* row_ptr, const int32_t* indices, int nq, int width, int pad, int32_t* out,
                      void* stream) {
    AGB_CHECK_ARG(nq >= 0 && width >= 0, "agb_csr_to_padded: bad sizes");
    if (nq == 0 || width == 0) return AGB_OK;
    hipLaunchKernelGGL(k_csr_to_padded, dim3((unsigned)agb_cdiv((long long)nq * width, 256)), dim3(256), 0,
                       (hipStream_t)stream, row_ptr, indices, nq, width, pad, out);
    AGB_CHECK_LAUNCH("agb_csr_to_padded");
    return AGB_OK;
}

// Grid subsampling. cap = cells reserved per element (>= NX*NY*NZ of every element; status[0] counts violations).
// Scratch (int32): bbox_ord[6B], dims[3B], cell_cnt[B*cap+1], cell_start[B*cap+1], slot[B*cap+1], flag[B*cap+1],
// cell_of[n], members[n], scan_scratch[agb_scan_scratch_elems(B*cap+1)]; origin float[3B].
// Outputs: out_pts float[n*3] (upper bound), out_feats float[n*fdim] or NULL, out_ptr int32[B+1] (row range of every
// element in the output), n_out_dev.
AGB_INTERNAL int agb_grid_subsample(const float* pts, const float* feats, int fdim, int n, const int32_t* ptr,
                       const int32_t* elem, int B, float dl, int cap, int32_t* bbox_ord, float* origin,
                       int32_t* dims, int32_t* cell_cnt, int32_t* cell_start, int32_t* slot, int32_t* flag,
                       int32_t* cell_of, int32_t* members, int32_t* scan_scratch, float* out_pts, float* out_feats,
                       int32_t* out_ptr, int32_t* n_out_dev, int32_t* status, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    long long cells = (long long)B * cap;
    AGB_CHECK_ARG(cells > 0 && cells < 0x7FFFFFF0LL, "agb_grid_subsample: %lld cells out of range", cells);
    int nc = (int)cells + 1;
    SubGrid g{dl, cap, B};
    (void)hipMemsetAsync(status, 0, sizeof(int32_t) * 4, s);
    int rc = agb_elem_bbox(pts, ptr, B, n, bbox_ord, nullptr, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(k_sub_geometry, dim3(agb_cdiv(B, 64)), dim3(64), 0, s, bbox_ord, g, origin, dims, status);
    (void)hipMemsetAsync(cell_cnt, 0, sizeof(int32_t) * (size_t)nc, s);
    if (n > 0)
        hipLaunchKernelGGL(k_sub_count, dim3(agb_cdiv(n, 256)), dim3(256), 0, s, pts, elem, n, g, origin, dims,
                           cell_cnt, cell_of);
    // CSR of members per cell + output slot of every non-empty cell (cell order = canonical key order per element)
    agb_launch_exclusive_scan(cell_cnt, nc, cell_start, scan_scratch, n_out_dev, s);
    hipLaunchKernelGGL(k_nonzero_flag, dim3(agb_cdiv(nc, 256)), dim3(256), 0, s, cell_cnt, nc, flag);
    agb_launch_exclusive_scan(flag, nc, slot, scan_scratch, n_out_dev, s);
    (void)hipMemsetAsync(flag, 0, sizeof(int32_t) * (size_t)nc, s);  // reused as the per-cell fill cursor
    if (n > 0) {
        hipLaunchKernelGGL(k_sub_scatter, dim3(agb_cdiv(n, 256)), dim3(256), 0, s, n, cell_of, cell_start, flag,
                           members);
        hipLaunchKernelGGL(k_sub_emit, dim3(agb_cdiv(nc - 1, 256)), dim3(256), 0, s, pts, feats, fdim, nc - 1, cell_cnt,
                           cell_start, slot, members, out_pts, out_feats);
    }
    hipLaunchKernelGGL(k_sub_elem_ptr, dim3(agb_cdiv(B + 1, 64)), dim3(64), 0, s, slot, n_out_dev, g, out_ptr);
    AGB_CHECK_LAUNCH("agb_grid_subsample");
    return AGB_OK;
}

}  // extern "C"
