// coords.hip — sparse-voxel coordinate management on device:
//   * open-addressing hash of packed [b,x,y,z] keys            (ME SparseTensor insert;
//     reference call site torch_points3d/models/instance/minkowski.py:67-80)
//   * strided output coordinates floor(c/ts)*ts + deterministic unique
//     (ME coordinate-map stride; call sites modules/MinkowskiEngine/SENet.py:47-53,
//     resnet_block.py:48-55)
//   * kernel maps as dense neighbour tables nbr[K^3][n_out]     (ME kernel map)
//   * per-batch row ranges (rows stay batch-contiguous at every level)
//
// HBM-bound integer work: every kernel is one thread per row (or per row x offset),
// rows fastest so that table writes and coordinate reads coalesce.
#include "agb_common.h"
#include <string.h>
#include <limits.h>
#include <stdarg.h>

static thread_local char g_err[512] = "";
extern "C" const char* agb_last_error(void) { return g_err; }
static thread_local char g_kernel[128] = "";
extern "C" const char* agb_last_kernel(void) { return g_kernel; }
void agb_note_kernel(const char* name) {
    // "(k_spconv_pipe<128, false>)" -> "k_spconv_pipe<128, false>"
    size_t n = strlen(name);
    if (n >= 2 && name[0] == '(' && name[n - 1] == ')') { ++name; n -= 2; }
    if (n >= sizeof(g_kernel)) n = sizeof(g_kernel) - 1;
    memcpy(g_kernel, name, n);
    g_kernel[n] = 0;
}
void agb_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

#define TPB 256
#define AGB_PAIR_SHARDS 64  // pair_count buffers are uint64[AGB_PAIR_SHARDS * 16]; the total is the sum of all slots

__device__ __forceinline__ int eff_n(int n_bound, const int32_t* n_dev) {
    if (n_dev) {
        int v = *n_dev;
        return v < n_bound ? v : n_bound;
    }
    return n_bound;
}

// ---------------------------------------------------------------- hash table
__global__ void k_hash_clear(uint64_t* keys, int32_t* vals, int cap) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cap) {
        keys[i] = AGB_KEY_EMPTY;
        vals[i] = INT_MAX;
    }
}

__device__ __forceinline__ int hash_insert_min(uint64_t* keys, int32_t* vals, uint32_t mask, uint64_t key,
                                               int row) {
    uint32_t h = agb_hash_key(key) & mask;
    while (true) {
        unsigned long long prev =
            atomicCAS((unsigned long long*)&keys[h], (unsigned long long)AGB_KEY_EMPTY, (unsigned long long)key);
        if (prev == AGB_KEY_EMPTY || prev == key) {
            atomicMin(&vals[h], row);
            return (int)h;
        }
        h = (h + 1) & mask;
    }
}

__device__ __forceinline__ int hash_find_slot(const uint64_t* keys, uint32_t mask, uint64_t key) {
    uint32_t h = agb_hash_key(key) & mask;
    while (true) {
        uint64_t kk = keys[h];
        if (kk == key) return (int)h;
        if (kk == AGB_KEY_EMPTY) return -1;
        h = (h + 1) & mask;
    }
}

__device__ __forceinline__ bool coord_in_range(int b, int x, int y, int z) {
    return b >= 0 && b < 65536 && x >= -AGB_COORD_BIAS && x < AGB_COORD_BIAS && y >= -AGB_COORD_BIAS &&
           y < AGB_COORD_BIAS && z >= -AGB_COORD_BIAS && z < AGB_COORD_BIAS;
}

// status[0] = rows whose coordinate duplicates an earlier row
// status[1] = rows with a component outside the 16-bit packed range
// status[2] = rows whose batch index is smaller than the previous row's
__global__ void k_coords_insert(const int4* __restrict__ coords, int n, const int32_t* n_dev, uint64_t* keys,
                                int32_t* vals, uint32_t mask, int32_t* slot_of_row, int32_t* status) {
    n = eff_n(n, n_dev);
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int4 c = coords[i];  // (b, x, y, z)
    if (!coord_in_range(c.x, c.y, c.z, c.w)) {
        atomicAdd(&status[1], 1);
        slot_of_row[i] = -1;
        return;
    }
    if (i > 0 && coords[i - 1].x > c.x) atomicAdd(&status[2], 1);
    uint64_t key = agb_pack_key(c.x, c.y, c.z, c.w);
    slot_of_row[i] = hash_insert_min(keys, vals, mask, key, i);
}

__global__ void k_count_dups(int n, const int32_t* n_dev, const int32_t* vals, const int32_t* slot_of_row,
                             int32_t* status) {
    n = eff_n(n, n_dev);
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int s = slot_of_row[i];
    if (s >= 0 && vals[s] != i) atomicAdd(&status[0], 1);
}

// ------------------------------------------------------------ strided coords
// pass 1: insert floor(c / ts_out) * ts_out with the smallest input row as value
__global__ void k_stride_insert(const int4* __restrict__ coords, int n, const int32_t* n_dev, int ts_out,
                                uint64_t* keys, int32_t* vals, uint32_t mask, int32_t* slot_of_row) {
    n = eff_n(n, n_dev);
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int4 c = coords[i];
    int x = agb_floordiv(c.y, ts_out) * ts_out;
    int y = agb_floordiv(c.z, ts_out) * ts_out;
    int z = agb_floordiv(c.w, ts_out) * ts_out;
    uint64_t key = agb_pack_key(c.x, x, y, z);
    slot_of_row[i] = hash_insert_min(keys, vals, mask, key, i);
}

// pass 2: flag[i] = 1 when row i is the first occurrence of its strided key
__global__ void k_stride_flag(int n, const int32_t* n_dev, const int32_t* vals, const int32_t* slot_of_row,
                              int32_t* flags) {
    int nn = eff_n(n, n_dev);
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    flags[i] = (i < nn && vals[slot_of_row[i]] == i) ? 1 : 0;
}

#include "scan.h"

// pass 3: representatives write the output coordinate and publish their row id in the table
__global__ void k_stride_emit(const int4* __restrict__ coords, int n, const int32_t* n_dev, int ts_out,
                              const int32_t* __restrict__ flags, const int32_t* __restrict__ excl,
                              const int32_t* __restrict__ slot_of_row, int32_t* vals, int4* out_coords) {
    n = eff_n(n, n_dev);
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flags[i]) return;
    int4 c = coords[i];
    int r = excl[i];
    out_coords[r] = make_int4(c.x, agb_floordiv(c.y, ts_out) * ts_out, agb_floordiv(c.z, ts_out) * ts_out,
                              agb_floordiv(c.w, ts_out) * ts_out);
    vals[slot_of_row[i]] = r;
}

// optional: out row of every input row (valid after k_stride_emit)
__global__ void k_stride_rowmap(int n, const int32_t* n_dev, const int32_t* vals, const int32_t* slot_of_row,
                                int32_t* out_row_of_in) {
    n = eff_n(n, n_dev);
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out_row_of_in[i] = vals[slot_of_row[i]];
}

// ----------------------------------------------------------------- kernel map
// nbr[k][r] = row (in the probed table) of coordinate q_coords[r] + sign*offset_k*step, or -1.
// offset_k for k = ix + K*(iy + K*iz): odd K -> (ix - K/2, ...), even K -> (ix, ...)   (ME region HYPER_CUBE)
__global__ void k_kernel_map(const int4* __restrict__ q_coords, int n, const int32_t* n_dev, int K, int step, int sign,
                             int require_multiple_of, const uint64_t* __restrict__ keys,
                             const int32_t* __restrict__ vals, uint32_t mask, int32_t* nbr, long long nbr_stride,
                             unsigned long long* pair_count) {
    int nn = eff_n(n, n_dev);
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    int k = blockIdx.y;
    if (r >= nn) return;
    int ix = k % K, iy = (k / K) % K, iz = k / (K * K);
    int half = (K & 1) ? K / 2 : 0;
    int4 c = q_coords[r];
    int x = c.y + sign * (ix - half) * step;
    int y = c.z + sign * (iy - half) * step;
    int z = c.w + sign * (iz - half) * step;
    int res = -1;
    bool ok = coord_in_range(c.x, x, y, z);
    if (ok && require_multiple_of > 1) {
        // transposed map: the probed coordinate must lie on the (coarser) output lattice
        ok = (x % require_multiple_of == 0) && (y % require_multiple_of == 0) && (z % require_multiple_of == 0);
    }
    if (ok) {
        int s = hash_find_slot(keys, mask, agb_pack_key(c.x, x, y, z));
        if (s >= 0) res = vals[s];
    }
    nbr[(long long)k * nbr_stride + r] = res;
    if (pair_count) {
        // one atomic per wave: number of (input,output) pairs of this map (ME's kernel-map size)
        // (__ballot only sees the lanes still active here, i.e. the in-range rows of this wave)
        unsigned long long m = __ballot(res >= 0);
        unsigned long long active = __ballot(1);
        int leader = __ffsll((long long)active) - 1;
        // AGB_PAIR_SHARDS counters on separate 128-B lines: one hot address saturates at ~90 atomics/us
        if ((int)(threadIdx.x & 63) == leader && m)
            atomicAdd(pair_count + 16 * ((blockIdx.x + blockIdx.y) & (AGB_PAIR_SHARDS - 1)),
                      (unsigned long long)__popcll(m));
    }
}

// ----------------------------------------------------------------- batch rows
__global__ void k_batch_count(const int4* __restrict__ coords, int n, const int32_t* n_dev, int B, int32_t* ptr) {
    n = eff_n(n, n_dev);
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int b = coords[i].x;
    // rows are batch-contiguous: only segment heads touch memory
    if (i == 0 || coords[i - 1].x != b) {
        if (b >= 0 && b < B) ptr[b] = i;
    }
    if (i == n - 1) ptr[B] = n;
}

__global__ void k_batch_fix(int B, int32_t* ptr) {
    // empty batches (ptr == -1) take the start of the next non-empty one
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (ptr[B] < 0) ptr[B] = 0;
        for (int b = B - 1; b >= 0; --b)
            if (ptr[b] < 0) ptr[b] = ptr[b + 1];
    }
}

__global__ void k_fill_i32(int32_t* p, int n, int v) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ================================================================ dense-grid mode
// When the batch's bounding volume is small (LiDAR plots: ~81^3 cells x B), every level keeps a dense
// lookup grid int32[B][Z][Y][X] (x fastest) instead of the hash: one load per probe, no collisions, and
// neighbouring rows probe neighbouring cells (coalesced).  Empty cells hold INT_MAX.
struct GridDesc {
    int ox, oy, oz;  // origin (multiples of ts)
    int X, Y, Z;     // cells per axis
    int ts;          // tensor stride of the level
    int B;
    int halo;        // margin of cells on every side that rows may not occupy (kept empty: kernels that probe
                     // grid[cell + delta] directly never leave the grid or the plot's own block of it)
    int lg;          // log2(ts) when ts is a power of two (the usual case: shifts instead of runtime divisions in every
                     // probe), else -1
};

// strict: also -1 inside the halo margin (where rows are inserted: a coordinate outside the declared bounds must be
// reported, not parked in the halo)
__device__ __forceinline__ long long grid_cell(const GridDesc& g, int b, int x, int y, int z, bool strict = false) {
    // returns -1 when (b,x,y,z) is outside the grid or off the level's lattice
    int dx = x - g.ox, dy = y - g.oy, dz = z - g.oz;
    if (b < 0 || b >= g.B || dx < 0 || dy < 0 || dz < 0) return -1;
    if (g.lg > 0) {
        if ((dx | dy | dz) & (g.ts - 1)) return -1;
        dx >>= g.lg; dy >>= g.lg; dz >>= g.lg;
    } else if (g.ts > 1 && g.lg < 0) {
        if ((dx % g.ts) | (dy % g.ts) | (dz % g.ts)) return -1;
        dx /= g.ts; dy /= g.ts; dz /= g.ts;
    }
    if (dx >= g.X || dy >= g.Y || dz >= g.Z) return -1;
    if (strict && (dx < g.halo || dy < g.halo || dz < g.halo || dx >= g.X - g.halo || dy >= g.Y - g.halo ||
                   dz >= g.Z - g.halo))
        return -1;
    return (((long long)b * g.Z + dz) * g.Y + dy) * g.X + dx;
}

// bbox[0..2] = min x,y,z   bbox[3..5] = max x,y,z   bbox[6] = max batch   (caller pre-fills +/- extremes)
__global__ void k_coords_bbox(const int4* __restrict__ coords, int n, const int32_t* n_dev, int32_t* bbox) {
    n = eff_n(n, n_dev);
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int mnx = INT_MAX, mny = INT_MAX, mnz = INT_MAX, mxx = INT_MIN, mxy = INT_MIN, mxz = INT_MIN, mxb = INT_MIN;
    for (; i < n; i += gridDim.x * blockDim.x) {
        int4 c = coords[i];
        mnx = min(mnx, c.y); mny = min(mny, c.z); mnz = min(mnz, c.w);
        mxx = max(mxx, c.y); mxy = max(mxy, c.z); mxz = max(mxz, c.w);
        mxb = max(mxb, c.x);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        mnx = min(mnx, __shfl_xor(mnx, d, 64)); mny = min(mny, __shfl_xor(mny, d, 64));
        mnz = min(mnz, __shfl_xor(mnz, d, 64)); mxx = max(mxx, __shfl_xor(mxx, d, 64));
        mxy = max(mxy, __shfl_xor(mxy, d, 64)); mxz = max(mxz, __shfl_xor(mxz, d, 64));
        mxb = max(mxb, __shfl_xor(mxb, d, 64));
    }
    if ((threadIdx.x & 63) == 0 && mxb != INT_MIN) {
        atomicMin(&bbox[0], mnx); atomicMin(&bbox[1], mny); atomicMin(&bbox[2], mnz);
        atomicMax(&bbox[3], mxx); atomicMax(&bbox[4], mxy); atomicMax(&bbox[5], mxz);
        atomicMax(&bbox[6], mxb);
    }
}

__global__ void k_bbox_init(int32_t* bbox) {
    int t = threadIdx.x;
    if (t < 3) bbox[t] = INT_MAX;
    else if (t < 7) bbox[t] = INT_MIN;
    else if (t == 7) bbox[t] = 0;
}

// level 0: cell <- smallest row with that coordinate. cell_of_row keeps the cell (or -1) for the follow-up passes.
__global__ void k_grid_insert(const int4* __restrict__ coords, int n, const int32_t* n_dev, GridDesc g,
                              int32_t* grid, long long* cell_of_row, int32_t* status) {
    n = eff_n(n, n_dev);
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int4 c = coords[i];
    if (i > 0 && coords[i - 1].x > c.x) atomicAdd(&status[2], 1);
    long long cell = grid_cell(g, c.x, c.y, c.z, c.w, true);
    cell_of_row[i] = cell;
    if (cell < 0) {
        atomicAdd(&status[1], 1);
        return;
    }
    atomicMin(&grid[cell], i);
}

__global__ void k_grid_count_dups(int n, const int32_t* n_dev, const int32_t* grid, const long long* cell_of_row,
                                  int32_t* status) {
    n = eff_n(n, n_dev);
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    long long cell = cell_of_row[i];
    if (cell >= 0 && grid[cell] != i) atomicAdd(&status[0], 1);
}

__global__ void k_grid_stride_insert(const int4* __restrict__ coords, int n, const int32_t* n_dev, GridDesc g,
                                     int32_t* grid, long long* cell_of_row, int32_t* status) {
    n = eff_n(n, n_dev);
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int4 c = coords[i];
    int x = agb_floordiv(c.y, g.ts) * g.ts, y = agb_floordiv(c.z, g.ts) * g.ts, z = agb_floordiv(c.w, g.ts) * g.ts;
    long long cell = grid_cell(g, c.x, x, y, z, true);
    cell_of_row[i] = cell;
    if (cell < 0) {
        atomicAdd(&status[1], 1);
        return;
    }
    atomicMin(&grid[cell], i);
}

__global__ void k_grid_stride_flag(int n, const int32_t* n_dev, const int32_t* grid, const long long* cell_of_row,
                                   int32_t* flags) {
    int nn = eff_n(n, n_dev);
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int f = 0;
    if (i < nn) {
        long long cell = cell_of_row[i];
        f = (cell >= 0 && grid[cell] == i) ? 1 : 0;
    }
    flags[i] = f;
}

__global__ void k_grid_stride_emit(const int4* __restrict__ coords, int n, const int32_t* n_dev, int ts_out,
                                   const int32_t* __restrict__ flags, const int32_t* __restrict__ excl,
                                   const long long* __restrict__ cell_of_row, int32_t* grid, int4* out_coords) {
    n = eff_n(n, n_dev);
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flags[i]) return;
    int4 c = coords[i];
    int r = excl[i];
    out_coords[r] = make_int4(c.x, agb_floordiv(c.y, ts_out) * ts_out, agb_floordiv(c.z, ts_out) * ts_out,
                              agb_floordiv(c.w, ts_out) * ts_out);
    grid[cell_of_row[i]] = r;
}

// one thread per (row, iy, iz); loops the K offsets along x so the K probes of a thread are adjacent cells
__global__ void k_grid_kernel_map(const int4* __restrict__ q_coords, int n, const int32_t* n_dev, int K, int step,
                                  int sign, GridDesc g, const int32_t* __restrict__ grid, int32_t* nbr,
                                  long long nbr_stride, unsigned long long* pair_count) {
    int nn = eff_n(n, n_dev);
    int r = blockIdx.y * blockDim.x + threadIdx.x;  // row block = slow grid axis: its K*K (iy,iz) slices run together (L2 reuse of the probed grid region)
    int cnt = 0;
    if (r < nn) {
        int iy = blockIdx.x % K, iz = blockIdx.x / K;
        int half = (K & 1) ? K / 2 : 0;
        int4 c = q_coords[r];
        int y = c.z + sign * (iy - half) * step;
        int z = c.w + sign * (iz - half) * step;
        // the K probes of a thread are independent: all cells first, all loads in flight together (unconditional,
        // clamped), then the stores — a load under a branch inside a rolled loop serialised K memory latencies
        long long cell[9];
        int v[9];
#pragma unroll
        for (int ix = 0; ix < 9; ++ix) {
            const int x = c.y + sign * (ix - half) * step;
            cell[ix] = ix < K ? grid_cell(g, c.x, x, y, z) : -1;
        }
#pragma unroll
        for (int ix = 0; ix < 9; ++ix) v[ix] = grid[cell[ix] >= 0 ? cell[ix] : 0];
#pragma unroll
        for (int ix = 0; ix < 9; ++ix) {
            if (ix >= K) break;
            const int res = (cell[ix] >= 0 && v[ix] != INT_MAX) ? v[ix] : -1;
            cnt += res >= 0;
            const int k = ix + K * (iy + K * iz);
            nbr[(long long)k * nbr_stride + r] = res;
        }
    }
    if (pair_count) {
        // every lane of the wave is active here (no early return above): plain butterfly reduction
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
        if ((threadIdx.x & 63) == 0 && cnt)
            atomicAdd(pair_count + 16 * ((blockIdx.x + blockIdx.y + (threadIdx.x >> 6)) & (AGB_PAIR_SHARDS - 1)),
                      (unsigned long long)cnt);
    }
}

// =============================================================== C ABI
extern "C" {

int agb_hash_capacity(int n) {
    long long c = 1024;
    while (c < 2LL * n) c <<= 1;
    return (int)c;
}

// scratch ints needed by agb_coords_stride for n rows
int agb_scan_scratch_elems(int n) { return agb_cdiv(n, SCAN_BLOCK) + 8; }

int agb_hash_clear(uint64_t* keys, int32_t* vals, int cap, void* stream) {
    AGB_CHECK_ARG(cap > 0 && (cap & (cap - 1)) == 0, "agb_hash_clear: capacity %d is not a power of two", cap);
    hipLaunchKernelGGL(k_hash_clear, dim3(agb_cdiv(cap, TPB)), dim3(TPB), 0, (hipStream_t)stream, keys, vals, cap);
    AGB_CHECK_LAUNCH("agb_hash_clear");
    return AGB_OK;
}

int agb_coords_insert(const int32_t* coords, int n, const int32_t* n_dev, uint64_t* keys, int32_t* vals, int cap,
                      int32_t* slot_of_row, int32_t* status, void* stream) {
    AGB_CHECK_ARG(n >= 0, "agb_coords_insert: n < 0");
    AGB_CHECK_ARG(cap >= 2 * n && (cap & (cap - 1)) == 0, "agb_coords_insert: capacity %d too small for %d rows", cap,
                  n);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_hash_clear, dim3(agb_cdiv(cap, TPB)), dim3(TPB), 0, s, keys, vals, cap);
    hipLaunchKernelGGL(k_fill_i32, dim3(1), dim3(64), 0, s, status, 4, 0);
    if (n > 0) {
        hipLaunchKernelGGL(k_coords_insert, dim3(agb_cdiv(n, TPB)), dim3(TPB), 0, s, (const int4*)coords, n, n_dev,
                           keys, vals, (uint32_t)(cap - 1), slot_of_row, status);
        hipLaunchKernelGGL(k_count_dups, dim3(agb_cdiv(n, TPB)), dim3(TPB), 0, s, n, n_dev, vals, slot_of_row,
                           status);
    }
    AGB_CHECK_LAUNCH("agb_coords_insert");
    return AGB_OK;
}

int agb_coords_stride(const int32_t* in_coords, int n, const int32_t* n_dev, int ts_out, uint64_t* keys,
                      int32_t* vals, int cap, int32_t* slot_of_row, int32_t* flags, int32_t* excl, int32_t* scratch,
                      int32_t* out_coords, int32_t* n_out_dev, int32_t* out_row_of_in, void* stream) {
    AGB_CHECK_ARG(n >= 0 && ts_out > 0, "agb_coords_stride: bad n/ts");
    AGB_CHECK_ARG(cap >= 2 * n && (cap & (cap - 1)) == 0, "agb_coords_stride: capacity %d too small for %d rows", cap,
                  n);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_hash_clear, dim3(agb_cdiv(cap, TPB)), dim3(TPB), 0, s, keys, vals, cap);
    if (n == 0) {
        hipLaunchKernelGGL(k_fill_i32, dim3(1), dim3(64), 0, s, n_out_dev, 1, 0);
        AGB_CHECK_LAUNCH("agb_coords_stride");
        return AGB_OK;
    }
    dim3 g(agb_cdiv(n, TPB)), b(TPB);
    uint32_t mask = (uint32_t)(cap - 1);
    hipLaunchKernelGGL(k_stride_insert, g, b, 0, s, (const int4*)in_coords, n, n_dev, ts_out, keys, vals, mask,
                       slot_of_row);
    hipLaunchKernelGGL(k_stride_flag, g, b, 0, s, n, n_dev, vals, slot_of_row, flags);
    int nb = agb_cdiv(n, SCAN_BLOCK);
    hipLaunchKernelGGL(k_scan_block_sums, dim3(nb), b, 0, s, flags, n, scratch);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), b, 0, s, scratch, nb, n_out_dev);
    hipLaunchKernelGGL(k_scan_apply, dim3(nb), b, 0, s, flags, n, scratch, excl);
    hipLaunchKernelGGL(k_stride_emit, g, b, 0, s, (const int4*)in_coords, n, n_dev, ts_out, flags, excl, slot_of_row,
                       vals, (int4*)out_coords);
    if (out_row_of_in)
        hipLaunchKernelGGL(k_stride_rowmap, g, b, 0, s, n, n_dev, vals, slot_of_row, out_row_of_in);
    AGB_CHECK_LAUNCH("agb_coords_stride");
    return AGB_OK;
}

// Forward map of a conv/pool with kernel size K from level `in` to level `out`:
//   q_coords = out coords, table = in level, step = ts_in*dilation, sign=+1, require_multiple_of=0
// Transposed map (for data gradients of strided ops):
//   q_coords = in coords, table = out level, step = ts_in*dilation, sign=-1, require_multiple_of=ts_out
int agb_kernel_map(const int32_t* q_coords, int n, const int32_t* n_dev, int K, int step, int sign,
                   int require_multiple_of, const uint64_t* keys, const int32_t* vals, int cap, int32_t* nbr,
                   long long nbr_stride, unsigned long long* pair_count, void* stream) {
    AGB_CHECK_ARG(K >= 1 && K <= 9, "agb_kernel_map: kernel size %d unsupported", K);
    AGB_CHECK_ARG(nbr_stride >= n, "agb_kernel_map: nbr_stride < n");
    if (n == 0) return AGB_OK;
    hipLaunchKernelGGL(k_kernel_map, dim3(agb_cdiv(n, TPB), K * K * K), dim3(TPB), 0, (hipStream_t)stream,
                       (const int4*)q_coords, n, n_dev, K, step, sign, require_multiple_of, keys, vals,
                       (uint32_t)(cap - 1), nbr, nbr_stride, pair_count);
    AGB_CHECK_LAUNCH("agb_kernel_map");
    return AGB_OK;
}

int agb_batch_ptr(const int32_t* coords, int n, const int32_t* n_dev, int B, int32_t* ptr, void* stream) {
    AGB_CHECK_ARG(B >= 1, "agb_batch_ptr: B < 1");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_fill_i32, dim3(agb_cdiv(B + 1, TPB)), dim3(TPB), 0, s, ptr, B + 1, -1);
    if (n > 0)
        hipLaunchKernelGGL(k_batch_count, dim3(agb_cdiv(n, TPB)), dim3(TPB), 0, s, (const int4*)coords, n, n_dev, B,
                           ptr);
    hipLaunchKernelGGL(k_batch_fix, dim3(1), dim3(64), 0, s, B, ptr);
    AGB_CHECK_LAUNCH("agb_batch_ptr");
    return AGB_OK;
}

// ---- dense-grid mode --------------------------------------------------------------------------------
// desc = {ox, oy, oz, X, Y, Z, ts, B | halo << 16} (host ints). grid: int32[B*Z*Y*X], filled with INT_MAX by these calls.
static inline GridDesc mk_desc(const int32_t* d) {
    GridDesc g;
    g.ox = d[0]; g.oy = d[1]; g.oz = d[2]; g.X = d[3]; g.Y = d[4]; g.Z = d[5]; g.ts = d[6]; g.B = d[7] & 0xffff;
    g.halo = (d[7] >> 16) & 0xff;
    g.lg = -1;
    for (int l = 0; l < 31; ++l)
        if (g.ts == (1 << l)) g.lg = l;
    return g;
}
static inline long long desc_cells(const int32_t* d) { return (long long)(d[7] & 0xffff) * d[5] * d[4] * d[3]; }

int agb_coords_bbox(const int32_t* coords, int n, const int32_t* n_dev, int32_t* bbox, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_bbox_init, dim3(1), dim3(64), 0, s, bbox);
    if (n > 0) {
        int blocks = agb_cdiv(n, TPB);
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(k_coords_bbox, dim3(blocks), dim3(TPB), 0, s, (const int4*)coords, n, n_dev, bbox);
    }
    AGB_CHECK_LAUNCH("agb_coords_bbox");
    return AGB_OK;
}

static int fill_grid(int32_t* grid, const int32_t* desc, hipStream_t s) {
    long long cells = desc_cells(desc);
    if (cells <= 0 || cells > 0x7FFFFFFFLL) {
        agb_set_error("dense grid of %lld cells is out of range (use the hash mode)", cells);
        return AGB_ERANGE;
    }
    hipLaunchKernelGGL(k_fill_i32, dim3(agb_cdiv(cells, TPB)), dim3(TPB), 0, s, grid, (int)cells, INT_MAX);
    return AGB_OK;
}

int agb_grid_insert(const int32_t* coords, int n, const int32_t* n_dev, const int32_t* desc, int32_t* grid,
                    long long* cell_of_row, int32_t* status, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    int rc = fill_grid(grid, desc, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_fill_i32, dim3(1), dim3(64), 0, s, status, 4, 0);
    if (n > 0) {
        dim3 g(agb_cdiv(n, TPB)), b(TPB);
        hipLaunchKernelGGL(k_grid_insert, g, b, 0, s, (const int4*)coords, n, n_dev, mk_desc(desc), grid, cell_of_row,
                           status);
        hipLaunchKernelGGL(k_grid_count_dups, g, b, 0, s, n, n_dev, grid, cell_of_row, status);
    }
    AGB_CHECK_LAUNCH("agb_grid_insert");
    return AGB_OK;
}

// desc describes the OUTPUT level (ts = ts_out). status[1] counts rows that fall outside the grid.
int agb_grid_stride(const int32_t* in_coords, int n, const int32_t* n_dev, const int32_t* desc, int32_t* grid,
                    long long* cell_of_row, int32_t* flags, int32_t* excl, int32_t* scratch, int32_t* out_coords,
                    int32_t* n_out_dev, int32_t* status, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    int rc = fill_grid(grid, desc, s);
    if (rc) return rc;
    if (n == 0) {
        hipLaunchKernelGGL(k_fill_i32, dim3(1), dim3(64), 0, s, n_out_dev, 1, 0);
        AGB_CHECK_LAUNCH("agb_grid_stride");
        return AGB_OK;
    }
    dim3 g(agb_cdiv(n, TPB)), b(TPB);
    GridDesc gd = mk_desc(desc);
    hipLaunchKernelGGL(k_grid_stride_insert, g, b, 0, s, (const int4*)in_coords, n, n_dev, gd, grid, cell_of_row,
                       status);
    hipLaunchKernelGGL(k_grid_stride_flag, g, b, 0, s, n, n_dev, grid, cell_of_row, flags);
    int nb = agb_cdiv(n, SCAN_BLOCK);
    hipLaunchKernelGGL(k_scan_block_sums, dim3(nb), b, 0, s, flags, n, scratch);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), b, 0, s, scratch, nb, n_out_dev);
    hipLaunchKernelGGL(k_scan_apply, dim3(nb), b, 0, s, flags, n, scratch, excl);
    hipLaunchKernelGGL(k_grid_stride_emit, g, b, 0, s, (const int4*)in_coords, n, n_dev, gd.ts, flags, excl,
                       cell_of_row, grid, (int4*)out_coords);
    AGB_CHECK_LAUNCH("agb_grid_stride");
    return AGB_OK;
}

// Same contract as agb_kernel_map, probing a dense grid (desc/grid of the PROBED level). The lattice check of the
// transposed map is implied by the grid's own stride.
int agb_grid_kernel_map(const int32_t* q_coords, int n, const int32_t* n_dev, int K, int step, int sign,
                        const int32_t* desc, const int32_t* grid, int32_t* nbr, long long nbr_stride,
                        unsigned long long* pair_count, void* stream) {
    AGB_CHECK_ARG(K >= 1 && K <= 9, "agb_grid_kernel_map: kernel size %d unsupported", K);
    AGB_CHECK_ARG(nbr_stride >= n, "agb_grid_kernel_map: nbr_stride < n");
    if (n == 0) return AGB_OK;
    hipLaunchKernelGGL(k_grid_kernel_map, dim3(K * K, agb_cdiv(n, TPB)), dim3(TPB), 0, (hipStream_t)stream,
                       (const int4*)q_coords, n, n_dev, K, step, sign, mk_desc(desc), grid, nbr, nbr_stride,
                       pair_count);
    AGB_CHECK_LAUNCH("agb_grid_kernel_map");
    return AGB_OK;
}

}  // extern "C"
