// voxelize.hip — GridSampling3D(size, quantize_coords=True, mode="last") on device, per cloud of a batch:
// the voxelisation every sparse model's input goes through
// (torch_points3d/core/data_transform/grid_transform.py:112-128; shuffle :22-29; torch_cluster.grid_cluster +
// pyg consecutive_cluster behind it, both absent from the reference tree -> semantics restated, parity unpinned):
//   shuffled point i = original point perm[i];  c = round_half_even(pos / size) (float32);
//   key = sum_d floor(c_d - min_d) * prod_{e<d} (floor(max_e - min_e) + 1)   (x fastest, per cloud);
//   one output row per distinct key in ascending key order, taken from the LAST shuffled point of the voxel;
//   coords = int(c).  Also returns the bounding box of the integer coordinates (feeds the dense-grid
//   coordinate manager without a read-back).
// Method: per-cloud dense key grid with atomicMax of the shuffled position, flag + exclusive scan (cell order is
// key order), emit.  Integer/byte work, HBM-bound.
#include "agb_common.h"
#include "scan.h"
#include <limits.h>

__device__ __forceinline__ int vf2ord(float f) {
    int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float vord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }

__device__ __forceinline__ void rounded(const float* __restrict__ pos, long long j, float size, float c[3]) {
#pragma unroll
    for (int a = 0; a < 3; ++a) c[a] = rintf(__fdiv_rn(pos[3 * j + a], size));
}

// bbox_ord[b][0..2] = min, [3..5] = max of the rounded coordinates (ordered-int encoding)
__global__ void k_vox_bbox(const float* __restrict__ pos, const int32_t* __restrict__ ptr, float size,
                           int32_t* bbox_ord) {
    int b = blockIdx.y;
    int beg = ptr[b], end = ptr[b + 1];
    int mn[3] = {INT_MAX, INT_MAX, INT_MAX}, mx[3] = {INT_MIN, INT_MIN, INT_MIN};
    for (int i = beg + blockIdx.x * blockDim.x + threadIdx.x; i < end; i += gridDim.x * blockDim.x) {
        float c[3];
        rounded(pos, i, size, c);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            int o = vf2ord(c[a]);
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
            atomicMin(&bbox_ord[6 * b + a], mn[a]);
            atomicMax(&bbox_ord[6 * b + 3 + a], mx[a]);
        }
    }
}

__global__ void k_vox_init(int32_t* bbox_ord, int B, int32_t* bounds) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < 6 * B) bbox_ord[t] = (t % 6) < 3 ? INT_MAX : INT_MIN;
    if (t < 6) bounds[t] = t < 3 ? INT_MAX : INT_MIN;
}

// span[b][a] = floor(max - min) + 1; status[0] counts clouds whose key space exceeds the reserved cells
__global__ void k_vox_geometry(const int32_t* __restrict__ bbox_ord, int B, int cap, float* lo, int32_t* span,
                               int32_t* status) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (bbox_ord[6 * b + 3] == INT_MIN) {
        span[3 * b] = span[3 * b + 1] = span[3 * b + 2] = 0;
        return;
    }
    long long cells = 1;
    for (int a = 0; a < 3; ++a) {
        float mn = vord2f(bbox_ord[6 * b + a]), mx = vord2f(bbox_ord[6 * b + 3 + a]);
        lo[3 * b + a] = mn;
        int s = (int)floorf(__fsub_rn(mx, mn)) + 1;
        span[3 * b + a] = s;
        cells *= s;
    }
    if (cells > cap) atomicAdd(&status[0], 1);
}

// The shuffle WITHOUT a permutation tensor (perm == NULL): shuffled position i of a cloud of m points holds original point
// vox_perm(i, m, seed of the cloud) — a pseudo-random BIJECTION of [0, m): four rounds of a balanced Feistel network on the
// 2h >= log2(m) bits of the index (round function: a murmur-style mix keyed by the seed), cycle-walked into [0, m) (the
// domain 2^(2h) is below 4 m: two iterations on average at worst).  GridSampling3D(mode="last") needs SOME uniformly random
// order to pick a voxel's representative (grid_transform.py:118-121 shuffles with torch.randperm); drawing it as a keyed
// bijection costs ~40 integer instructions per point here — the host-side alternative was one int64 sort of the batch plus
// a random-number launch (2 ms of host time and 0.5 ms of device time per batch of 32 plots).
__device__ __forceinline__ unsigned vox_mix(unsigned x, unsigned key) {
    x ^= key;
    x *= 0x85ebca6bu; x ^= x >> 13;
    x *= 0xc2b2ae35u; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ unsigned vox_perm(unsigned i, unsigned m, unsigned long long seed) {
    if (m <= 1) return 0;
    int bits = 32 - __clz(m - 1);
    if (bits < 2) bits = 2;
    const int h = (bits + 1) >> 1;
    const unsigned mask = (1u << h) - 1u;
    const unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
    unsigned x = i;
    do {
        unsigned L = x >> h, R = x & mask;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned t = L ^ (vox_mix(R, (r & 1 ? k1 : k0) + 0x9e3779b9u * (unsigned)(r + 1)) & mask);
            L = R;
            R = t;
        }
        x = (L << h) | R;
    } while (x >= m);
    return x;
}
__device__ __forceinline__ unsigned long long vox_cloud_seed(unsigned long long seed, int b) {
    unsigned long long k = seed + 0x9e3779b97f4a7c15ull * (unsigned long long)(b + 1);
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return k;
}
// original position (inside its cloud) of shuffled position i (global row) of cloud b
__device__ __forceinline__ long long vox_source(const long long* __restrict__ perm, const int32_t* __restrict__ ptr, int b,
                                                int i, unsigned long long seed) {
    if (perm) return perm[i];
    const int beg = ptr[b];
    return (long long)vox_perm((unsigned)(i - beg), (unsigned)(ptr[b + 1] - beg), vox_cloud_seed(seed, b));
}

// shuffled position i (global row beg+i of cloud b) holds original point beg + perm[beg+i]
__global__ void k_vox_mark(const float* __restrict__ pos, const long long* __restrict__ perm,
                           const int32_t* __restrict__ ptr, const int32_t* __restrict__ elem, int n, float size,
                           int cap, const float* __restrict__ lo, const int32_t* __restrict__ span, int32_t* cells,
                           int32_t* cell_of, unsigned long long seed) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int b = elem[i];
    long long j = ptr[b] + vox_source(perm, ptr, b, i, seed);
    float c[3];
    rounded(pos, j, size, c);
    const float* l = lo + 3 * b;
    const int32_t* s = span + 3 * b;
    int rx = (int)floorf(__fsub_rn(c[0], l[0])), ry = (int)floorf(__fsub_rn(c[1], l[1])),
        rz = (int)floorf(__fsub_rn(c[2], l[2]));
    int cell = b * cap + rx + s[0] * (ry + s[1] * rz);
    cell_of[i] = cell;
    atomicMax(&cells[cell], i);  // last shuffled occurrence wins
}

__global__ void k_vox_flag(const int32_t* __restrict__ cells, int n, int32_t* flag) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = cells[i] >= 0;
}

__global__ void k_vox_emit(const float* __restrict__ pos, const long long* __restrict__ perm,
                           const int32_t* __restrict__ ptr, const int32_t* __restrict__ elem, int ncells, float size,
                           const int32_t* __restrict__ cells, const int32_t* __restrict__ slot,
                           int32_t* __restrict__ coords, long long* __restrict__ keep, unsigned long long seed) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    int v[3];
    if (c < ncells) {
        int i = cells[c];
        if (i >= 0) {
            int b = elem[i];
            long long j = ptr[b] + vox_source(perm, ptr, b, i, seed);
            float r[3];
            rounded(pos, j, size, r);
            int o = slot[c];
            v[0] = (int)r[0]; v[1] = (int)r[1]; v[2] = (int)r[2];
            coords[3 * (long long)o] = v[0];
            coords[3 * (long long)o + 1] = v[1];
            coords[3 * (long long)o + 2] = v[2];
            keep[o] = j;  // index into the ORIGINAL (unshuffled) stacked point order
        }
    }
}

// out_ptr; and the bounding box of the integer coordinates over the whole batch, from the per-cloud boxes of the rounded
// positions (k_vox_bbox: every occupied cell is kept, so the box of the kept voxels is the box of all points).  Until round 5
// k_vox_emit reduced it with wave-level atomics on six addresses — 1.1 M same-address atomics over the 17 M cells of a
// B = 32 batch, 5.7 ms of a 6 ms chain (profiles/r05_end2end_kernel_stats.csv).
__global__ void k_vox_out_ptr(const int32_t* __restrict__ slot, const int32_t* total, int cap, int B,
                              int32_t* out_ptr, const int32_t* __restrict__ bbox_ord, int32_t* bounds) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) out_ptr[b] = slot[(long long)b * cap];
    if (b == B) {
        out_ptr[b] = *total;
        int mn[3] = {INT_MAX, INT_MAX, INT_MAX}, mx[3] = {INT_MIN, INT_MIN, INT_MIN};
        for (int c = 0; c < B; ++c) {
            if (bbox_ord[6 * c + 3] == INT_MIN) continue;      // empty cloud
            for (int a = 0; a < 3; ++a) {
                mn[a] = min(mn[a], (int)vord2f(bbox_ord[6 * c + a]));
                mx[a] = max(mx[a], (int)vord2f(bbox_ord[6 * c + 3 + a]));
            }
        }
        for (int a = 0; a < 3; ++a) { bounds[a] = mn[a]; bounds[3 + a] = mx[a]; }
    }
}

extern "C" {

// pos float[n,3] (stacked clouds, ptr int32[B+1], elem int32[n] = cloud of every row); perm int64[n]: within-cloud
// permutation (shuffled position -> original position in the cloud), or NULL: the keyed bijection vox_perm of `seed`.
// cap = cells reserved per cloud.
// Scratch int32: bbox_ord[6B], span[3B], cells[B*cap+1], slot[B*cap+1], flag[B*cap+1], cell_of[n],
// scan_scratch[agb_scan_scratch_elems(B*cap+1)]; lo float[3B].
// Out: coords int32[n,3] (upper bound), keep int64[n], out_ptr int32[B+1], n_out_dev, bounds int32[6]
// (min xyz, max xyz of coords), status int32[4].
AGB_INTERNAL int agb_voxelize_last(const float* pos, const long long* perm, const int32_t* ptr, const int32_t* elem, int B, int n,
                      float size, int cap, int32_t* bbox_ord, float* lo, int32_t* span, int32_t* cells,
                      int32_t* slot, int32_t* flag, int32_t* cell_of, int32_t* scan_scratch, int32_t* coords,
                      long long* keep, int32_t* out_ptr, int32_t* n_out_dev, int32_t* bounds, int32_t* status,
                      void* stream, unsigned long long seed) {
    hipStream_t s = (hipStream_t)stream;
    long long total = (long long)B * cap;
    AGB_CHECK_ARG(total > 0 && total < 0x7FFFFFF0LL, "agb_voxelize_last: %lld cells out of range", total);
    AGB_CHECK_ARG(size > 0.f, "agb_voxelize_last: voxel size must be positive");
    int nc = (int)total + 1;
    (void)hipMemsetAsync(status, 0, sizeof(int32_t) * 4, s);
    hipLaunchKernelGGL(k_vox_init, dim3(agb_cdiv(6 * B > 6 ? 6 * B : 6, 256)), dim3(256), 0, s, bbox_ord, B, bounds);
    if (n > 0) {
        int bx = agb_cdiv(agb_cdiv(n, B), 256);
        if (bx > 64) bx = 64;
        if (bx < 1) bx = 1;
        hipLaunchKernelGGL(k_vox_bbox, dim3(bx, B), dim3(256), 0, s, pos, ptr, size, bbox_ord);
    }
    hipLaunchKernelGGL(k_vox_geometry, dim3(agb_cdiv(B, 64)), dim3(64), 0, s, bbox_ord, B, cap, lo, span, status);
    (void)hipMemsetAsync(cells, 0xFF, sizeof(int32_t) * (size_t)nc, s);  // -1 = empty
    if (n > 0)
        hipLaunchKernelGGL(k_vox_mark, dim3(agb_cdiv(n, 256)), dim3(256), 0, s, pos, perm, ptr, elem, n, size, cap, lo,
                           span, cells, cell_of, seed);
    hipLaunchKernelGGL(k_vox_flag, dim3(agb_cdiv(nc, 256)), dim3(256), 0, s, cells, nc, flag);
    agb_launch_exclusive_scan(flag, nc, slot, scan_scratch, n_out_dev, s);
    hipLaunchKernelGGL(k_vox_emit, dim3(agb_cdiv(nc - 1, 256)), dim3(256), 0, s, pos, perm, ptr, elem, nc - 1, size,
                       cells, slot, coords, keep, seed);
    hipLaunchKernelGGL(k_vox_out_ptr, dim3(agb_cdiv(B + 1, 64)), dim3(64), 0, s, slot, n_out_dev, cap, B, out_ptr, bbox_ord,
                       bounds);
    AGB_CHECK_LAUNCH("agb_voxelize_last");
    return AGB_OK;
}

}  // extern "C"
