// transform.hip — the per-point part of the NFI sparse transform chain on the device, for a whole batch of plots.
//
// The reference runs these transforms one sample at a time in DataLoader workers on the CPU
// (torch-points3d/conf/data/instance/NFI/transforms/sparse-xy.yaml:106-150, test_transform):
//   ScalePos(op=div)                 core/data_transform/transforms.py:590-598   pos = pos / (sx, sy, sz)
//   MoveCenterPosPerSample           transforms.py:722-739                       pos += (cx, cy, cz)
//   StartZFromZero                   transforms.py:766-769                       pos.z -= min(pos.z)  (per plot)
//   Polygon2dExtend                  transforms.py:1461-1496                     keep points inside the polygon:
//                                    matplotlib.path.Path.contains_points (crossing test, in double)
//   XYZFeature(z) / AddOnes / AddXYDistanceToCenter / AddFeatsByKeys
//                                    core/data_transform/features.py:307-334,353-383
//                                    x = [1, pos.z, || (pos.xy - c) + 1e-6 ||_2]   (torch PairwiseDistance, eps 1e-6)
// and, after GridSampling3D (voxelize.hip), the train-time coordinate augmentation
//   RandomCoordsFlip                 core/data_transform/sparse_transforms.py:49-55   c[:, ax] = max(c[:, ax]) - c[:, ax]
//   ShiftVoxels                      transforms.py:1046-1054                          c += shift (per plot)
// Random decisions (flip flags, shifts) are drawn on the host in the reference's order and passed in.
// Arithmetic is fp32 in the reference's operation order (no contraction), the polygon test runs in double on the
// fp32 coordinates like matplotlib does: results are bit-identical to the CPU chain except for points lying exactly on
// a polygon edge.  HBM-bound, a few bytes per point: N*(12 read + 12+12+8+4 written).
#include "agb_common.h"
#include <limits.h>
#include "scan.h"

struct PlotXform {
    float sx, sy, sz;     // ScalePos
    float cx, cy, cz;     // MoveCenterPosPerSample
    float fcx, fcy;       // AddXYDistanceToCenter
    int div;              // ScalePos op: 1 = div, 0 = mul
    int z0;               // StartZFromZero on/off
};

__device__ __forceinline__ float xf_scale(float v, float s, int div) { return div ? v / s : v * s; }

// zmin[b] = min over the plot of (scale(z) + cz)
__global__ void k_plot_zmin(const float* __restrict__ pos, const int32_t* __restrict__ ptr, PlotXform t,
                            float* __restrict__ zmin) {
    __shared__ float s_min[4];
    const int b = blockIdx.x;
    float m = INFINITY;
    for (int i = ptr[b] + threadIdx.x; i < ptr[b + 1]; i += blockDim.x)
        m = fminf(m, xf_scale(pos[3LL * i + 2], t.sz, t.div) + t.cz);
    for (int o = 32; o > 0; o >>= 1) m = fminf(m, __shfl_down(m, o));
    if ((threadIdx.x & 63) == 0) s_min[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) zmin[b] = fminf(fminf(s_min[0], s_min[1]), fminf(s_min[2], s_min[3]));
}

// matplotlib's point_in_path (src/_path.h, point_in_path_impl): crossing test over the implicitly closed polygon
__device__ __forceinline__ int point_in_polygon(double tx, double ty, const double* __restrict__ poly, int nv) {
    double vtx0 = poly[2 * (nv - 1)], vty0 = poly[2 * (nv - 1) + 1];
    int yflag0 = vty0 >= ty;
    int inside = 0;
    for (int j = 0; j < nv; ++j) {
        const double vtx1 = poly[2 * j], vty1 = poly[2 * j + 1];
        const int yflag1 = vty1 >= ty;
        if (yflag0 != yflag1) {
            if (((vty1 - ty) * (vtx0 - vtx1) >= (vtx1 - tx) * (vty0 - vty1)) == yflag1) inside ^= 1;
        }
        yflag0 = yflag1;
        vtx0 = vtx1;
        vty0 = vty1;
    }
    return inside;
}

__global__ void k_plot_transform(const float* __restrict__ pos, const int32_t* __restrict__ elem, int n, PlotXform t,
                                 const float* __restrict__ zmin, const double* __restrict__ poly, int nv,
                                 float* __restrict__ pos_t, int32_t* __restrict__ flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = xf_scale(pos[3LL * i], t.sx, t.div) + t.cx;
    float y = xf_scale(pos[3LL * i + 1], t.sy, t.div) + t.cy;
    float z = xf_scale(pos[3LL * i + 2], t.sz, t.div) + t.cz;
    if (t.z0) z -= zmin[elem[i]];
    pos_t[3LL * i] = x;
    pos_t[3LL * i + 1] = y;
    pos_t[3LL * i + 2] = z;
    flag[i] = nv > 0 ? point_in_polygon((double)x, (double)y, poly, nv) : 1;
}

// kept rows, order preserved: pos, features [1, z, xy distance], source row; new plot offsets
__global__ void k_plot_emit(const float* __restrict__ pos_t, const int32_t* __restrict__ flag,
                            const int32_t* __restrict__ slot, int n, PlotXform t, float* __restrict__ pos_out,
                            float* __restrict__ x_out, long long* __restrict__ src) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flag[i]) return;
    const long long o = slot[i];
    const float x = pos_t[3LL * i], y = pos_t[3LL * i + 1], z = pos_t[3LL * i + 2];
    pos_out[3 * o] = x;
    pos_out[3 * o + 1] = y;
    pos_out[3 * o + 2] = z;
    const float dx = (x - t.fcx) + 1e-6f, dy = (y - t.fcy) + 1e-6f;
    x_out[3 * o] = 1.f;
    x_out[3 * o + 1] = z;
    x_out[3 * o + 2] = sqrtf(dx * dx + dy * dy);
    src[o] = i;
}

__global__ void k_plot_out_ptr(const int32_t* __restrict__ slot, const int32_t* __restrict__ ptr,
                               const int32_t* __restrict__ total, int B, int n, int32_t* __restrict__ out_ptr) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) out_ptr[b] = ptr[b] < n ? slot[ptr[b]] : *total;
    if (b == B) out_ptr[b] = *total;
}

// Rows are ordered by plot: a wave's 64 rows almost always belong to ONE plot — it then reduces in registers and issues one
// atomic per axis.  (One atomic per row and axis was 1.1 M atomics on 96 addresses = 4.8 ms at B = 32 x 11 k voxels; same-address
// atomics retire one per ~100 ns.  profiles/r05_end2end_kernel_stats.csv)
__global__ void k_coords_max(const int32_t* __restrict__ coords, const int32_t* __restrict__ elem, int n,
                             int32_t* __restrict__ cmax) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    const int b = live ? elem[i] : -1;
    int c[3] = {INT_MIN, INT_MIN, INT_MIN};
    if (live) {
#pragma unroll
        for (int a = 0; a < 3; ++a) c[a] = coords[3LL * i + a];
    }
    const int b0 = __builtin_amdgcn_readfirstlane(b);
    if (__all(b == b0 || !live) && b0 >= 0) {     // (a wave's first lane is live whenever any of its lanes is)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) c[a] = max(c[a], __shfl_xor(c[a], d, 64));
            if ((threadIdx.x & 63) == 0) atomicMax(&cmax[3 * b0 + a], c[a]);
        }
    } else if (live) {
#pragma unroll
        for (int a = 0; a < 3; ++a) atomicMax(&cmax[3 * b + a], c[a]);
    }
}

__global__ void k_coords_augment(int32_t* __restrict__ coords, const int32_t* __restrict__ elem, int n,
                                 const int32_t* __restrict__ flip, const int32_t* __restrict__ shift,
                                 const int32_t* __restrict__ cmax) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = elem[i];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        int c = coords[3LL * i + a];
        if (flip[3 * b + a]) c = cmax[3 * b + a] - c;
        coords[3LL * i + a] = c + shift[3 * b + a];
    }
}

extern "C" {

// pos float[n,3]: stacked raw plots; ptr int32[B+1], elem int32[n] (plot of every row).
// xform float[8] = (sx, sy, sz, cx, cy, cz, fcx, fcy); scale_div: ScalePos op (1 = div); z_from_zero: StartZFromZero.
// poly double[2*nv]: polygon vertices (nv = 0: no crop).
// Scratch: zmin float[B], pos_t float[n,3], flag int32[n], slot int32[n], scan_scratch int32[agb_scan_scratch_elems(n)].
// Out (upper bound n rows): pos_out float[n,3], x_out float[n,3], src int64[n] (row in the input), out_ptr int32[B+1],
// n_out_dev int32[1].  No host synchronisation.
AGB_INTERNAL int agb_plot_prepare(const float* pos, const int32_t* ptr, const int32_t* elem, int B, int n, const float* xform,
                     int scale_div, int z_from_zero, const double* poly, int nv, float* zmin, float* pos_t,
                     int32_t* flag, int32_t* slot, int32_t* scan_scratch, float* pos_out, float* x_out, long long* src,
                     int32_t* out_ptr, int32_t* n_out_dev, void* stream) {
    AGB_CHECK_ARG(B >= 1 && n >= 0 && nv >= 0 && nv != 1 && nv != 2, "agb_plot_prepare: bad sizes (B %d, n %d, nv %d)",
                  B, n, nv);
    AGB_CHECK_ARG(xform != nullptr, "agb_plot_prepare: xform is a HOST array of 8 floats");
    hipStream_t s = (hipStream_t)stream;
    PlotXform t{xform[0], xform[1], xform[2], xform[3], xform[4], xform[5], xform[6], xform[7], scale_div, z_from_zero};
    AGB_CHECK_ARG(!scale_div || (t.sx != 0.f && t.sy != 0.f && t.sz != 0.f), "agb_plot_prepare: zero scale divisor");
    if (n == 0) {
        (void)hipMemsetAsync(out_ptr, 0, sizeof(int32_t) * (B + 1), s);
        (void)hipMemsetAsync(n_out_dev, 0, sizeof(int32_t), s);
        return AGB_OK;
    }
    hipLaunchKernelGGL(k_plot_zmin, dim3(B), dim3(256), 0, s, pos, ptr, t, zmin);
    hipLaunchKernelGGL(k_plot_transform, dim3(agb_cdiv(n, 256)), dim3(256), 0, s, pos, elem, n, t, zmin, poly, nv, pos_t,
                       flag);
    agb_launch_exclusive_scan(flag, n, slot, scan_scratch, n_out_dev, s);
    hipLaunchKernelGGL(k_plot_emit, dim3(agb_cdiv(n, 256)), dim3(256), 0, s, pos_t, flag, slot, n, t, pos_out, x_out,
                       src);
    hipLaunchKernelGGL(k_plot_out_ptr, dim3(agb_cdiv(B + 1, 64)), dim3(64), 0, s, slot, ptr, n_out_dev, B, n, out_ptr);
    AGB_CHECK_LAUNCH("agb_plot_prepare");
    return AGB_OK;
}

// RandomCoordsFlip + ShiftVoxels on voxel coordinates int32[n,3] (in place): flip int32[B,3] (0/1 per plot and axis),
// shift int32[B,3]; cmax int32[B,3] scratch.
int agb_coords_augment(int32_t* coords, const int32_t* elem, int B, int n, const int32_t* flip, const int32_t* shift,
                       int32_t* cmax, void* stream) {
    AGB_CHECK_ARG(B >= 1 && n >= 0, "agb_coords_augment: bad sizes");
    if (n == 0) return AGB_OK;
    hipStream_t s = (hipStream_t)stream;
    (void)hipMemsetAsync(cmax, 0x80, sizeof(int32_t) * 3 * B, s);   // 0x80808080: below any voxel coordinate
    hipLaunchKernelGGL(k_coords_max, dim3(agb_cdiv(n, 256)), dim3(256), 0, s, coords, elem, n, cmax);
    hipLaunchKernelGGL(k_coords_augment, dim3(agb_cdiv(n, 256)), dim3(256), 0, s, coords, elem, n, flip, shift, cmax);
    AGB_CHECK_LAUNCH("agb_coords_augment");
    return AGB_OK;
}

}  // extern "C"

// ================================================================================================================
// Train-time float augmentations of sparse-xy.yaml:4-69 on the device.  All random draws are made by the host in the
// reference's per-sample order (dpcr-agb_amd/train_transforms.py) and arrive as per-plot parameters / per-point noise:
//   RandomGroundRemoval   transforms.py:1131-1150   z -= remove_v for every point, then keep z > remove_v (host: index list)
//   RandomDropout         transforms.py:1060-1087   keep a torch.randperm prefix                      (host: index list)
//   ScalePos (div)        :590-598      RandomNoise  :482-505 (sigma * randn clamped, from the host)
//   Random3AxisRotation   features.py:12-60  pos @ M^T      RandomShiftPos :747-759     MoveCenterPosPerSample :722-739
//   StartZFromZero        :766-769
//   AddRandomPoints       :775-815  (as committed upstream max_ == min_: every added point is the per-axis minimum)
//   CopyJitterRandomPoints :818-873 (np.random.choice indices + clamped noise from the host)
//   RandomPolygon2dExtend :1502-1552 (per-plot transformed polygon from matplotlib's Affine2D on the host; points are
//                                     kept unfiltered when none falls inside)
struct PlotAug {        // one per plot, 24 floats
    float zsub;         // RandomGroundRemoval: subtracted from the raw z (0 when not applied)
    float sx, sy, sz;   // ScalePos divisors
    float M[9];         // rotation (row-major); pos_out[j] = sum_k pos[k] * M[j][k]
    float tx, ty, tz;   // RandomShiftPos (0 when not applied)
    float cx, cy, cz;   // MoveCenterPosPerSample
    float pad[5];
};

// pos1[i] = ((raw[sel[i]] - (0,0,zsub)) / scale + noise[i]) @ M^T + shift + centre
__global__ void k_plot_augment(const float* __restrict__ raw, const long long* __restrict__ sel,
                               const int32_t* __restrict__ elem, int n, const PlotAug* __restrict__ aug,
                               const float* __restrict__ noise, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const PlotAug a = aug[elem[i]];
    const long long s = sel[i];
    float x = raw[3 * s], y = raw[3 * s + 1], z = raw[3 * s + 2];
    z = z - a.zsub;
    x = x / a.sx; y = y / a.sy; z = z / a.sz;
    if (noise) { x += noise[3LL * i]; y += noise[3LL * i + 1]; z += noise[3LL * i + 2]; }
    float rx = x * a.M[0] + y * a.M[1] + z * a.M[2];
    float ry = x * a.M[3] + y * a.M[4] + z * a.M[5];
    float rz = x * a.M[6] + y * a.M[7] + z * a.M[8];
    out[3LL * i] = (rx + a.tx) + a.cx;
    out[3LL * i + 1] = (ry + a.ty) + a.cy;
    out[3LL * i + 2] = (rz + a.tz) + a.cz;
}

// per-plot minimum of x, y, z (grid B, block 256): mins[3b..3b+2]
__global__ void k_plot_min3(const float* __restrict__ pos, const int32_t* __restrict__ ptr, float* __restrict__ mins) {
    __shared__ float s_m[3][4];
    const int b = blockIdx.x;
    float m0 = INFINITY, m1 = INFINITY, m2 = INFINITY;
    for (int i = ptr[b] + threadIdx.x; i < ptr[b + 1]; i += blockDim.x) {
        m0 = fminf(m0, pos[3LL * i]);
        m1 = fminf(m1, pos[3LL * i + 1]);
        m2 = fminf(m2, pos[3LL * i + 2]);
    }
    for (int o = 32; o > 0; o >>= 1) {
        m0 = fminf(m0, __shfl_down(m0, o));
        m1 = fminf(m1, __shfl_down(m1, o));
        m2 = fminf(m2, __shfl_down(m2, o));
    }
    if ((threadIdx.x & 63) == 0) {
        s_m[0][threadIdx.x >> 6] = m0;
        s_m[1][threadIdx.x >> 6] = m1;
        s_m[2][threadIdx.x >> 6] = m2;
    }
    __syncthreads();
    if (threadIdx.x < 3)
        mins[3 * b + threadIdx.x] = fminf(fminf(s_m[threadIdx.x][0], s_m[threadIdx.x][1]),
                                          fminf(s_m[threadIdx.x][2], s_m[threadIdx.x][3]));
}

// StartZFromZero + AddRandomPoints + CopyJitterRandomPoints: output plot b = [its n1 points with z - zmin]
// ++ [n_add copies of the per-axis minimum] ++ [n_cj jittered copies pos_prev[cj_idx] + cj_noise].
// ptr1 / ptr2: offsets of the plots in the input / output; cj_ptr: offsets into cj_idx / cj_noise; n_add int32[B].
__global__ void k_plot_extend(const float* __restrict__ pos1, const int32_t* __restrict__ ptr1,
                              const float* __restrict__ mins, const int32_t* __restrict__ ptr2,
                              const int32_t* __restrict__ elem2, int n2, const int32_t* __restrict__ n_add,
                              const int32_t* __restrict__ cj_ptr, const long long* __restrict__ cj_idx,
                              const float* __restrict__ cj_noise, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    const int b = elem2[i];
    const int n1 = ptr1[b + 1] - ptr1[b];
    const int local = i - ptr2[b];
    const float zmin = mins[3 * b + 2];
    // the per-axis minimum AFTER StartZFromZero: z minimum is zmin - zmin
    const float mx = mins[3 * b], my = mins[3 * b + 1], mz = zmin - zmin;
    float x, y, z;
    if (local < n1) {
        const long long s = ptr1[b] + local;
        x = pos1[3 * s]; y = pos1[3 * s + 1]; z = pos1[3 * s + 2] - zmin;
    } else if (local < n1 + n_add[b]) {
        x = mx; y = my; z = mz;
    } else {
        const int j = cj_ptr[b] + (local - n1 - n_add[b]);
        const long long src = cj_idx[j];
        if (src < n1) {
            const long long s = ptr1[b] + src;
            x = pos1[3 * s]; y = pos1[3 * s + 1]; z = pos1[3 * s + 2] - zmin;
        } else {
            x = mx; y = my; z = mz;
        }
        x += cj_noise[3LL * j]; y += cj_noise[3LL * j + 1]; z += cj_noise[3LL * j + 2];
    }
    out[3LL * i] = x; out[3LL * i + 1] = y; out[3LL * i + 2] = z;
}

// RandomPolygon2dExtend: inside test against the plot's own polygon (double [B][2*nv]); per-plot inside counts
__global__ void k_plot_inside(const float* __restrict__ pos, const int32_t* __restrict__ elem, int n,
                              const double* __restrict__ polys, int nv, int32_t* __restrict__ flag,
                              int32_t* __restrict__ cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    const int b = live ? elem[i] : -1;
    const int in = live ? point_in_polygon((double)pos[3LL * i], (double)pos[3LL * i + 1], polys + 2LL * nv * b, nv) : 0;
    if (live) flag[i] = in;
    // one atomic per wave where its rows belong to one plot (rows are ordered by plot), else one per row
    const int b0 = __builtin_amdgcn_readfirstlane(b);
    if (__all(b == b0 || !live) && b0 >= 0) {
        const int c = __popcll(__ballot(in != 0));
        if ((threadIdx.x & 63) == 0 && c > 0) atomicAdd(&cnt[b0], c);
    } else if (in) {
        atomicAdd(&cnt[b], 1);
    }
}

__global__ void k_plot_keep_all_if_none(int32_t* __restrict__ flag, const int32_t* __restrict__ elem, int n,
                                        const int32_t* __restrict__ cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && cnt[elem[i]] == 0) flag[i] = 1;
}

extern "C" {

// raw float[n_raw,3] stacked plots; sel int64[n]: rows of raw kept after RandomGroundRemoval / RandomDropout, grouped by
// plot (elem int32[n], ptr int32[B+1]); aug: device array of B PlotAug records (24 floats each); noise float[n,3] or
// NULL.  Out: pos1 float[n,3]; mins float[3B] = per-plot minimum of pos1 (x, y, z) — StartZFromZero itself is applied by
// agb_plot_extend.
int agb_plot_augment(const float* raw, const long long* sel, const int32_t* elem, const int32_t* ptr, int B, int n,
                     const float* aug, const float* noise, float* pos1, float* mins, void* stream) {
    AGB_CHECK_ARG(B >= 1 && n >= 0, "agb_plot_augment: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    if (n > 0)
        hipLaunchKernelGGL(k_plot_augment, dim3(agb_cdiv(n, 256)), dim3(256), 0, s, raw, sel, elem, n,
                           (const PlotAug*)aug, noise, pos1);
    hipLaunchKernelGGL(k_plot_min3, dim3(B), dim3(256), 0, s, pos1, ptr, mins);
    AGB_CHECK_LAUNCH("agb_plot_augment");
    return AGB_OK;
}

// pos2 float[n2,3] = per plot [pos1 rows with z - zmin] ++ [n_add x per-axis minimum] ++ [jittered copies] (see
// k_plot_extend); ptr2 / elem2 describe the output layout (host-computed: all counts are known to the host).
int agb_plot_extend(const float* pos1, const int32_t* ptr1, const float* mins, const int32_t* ptr2,
                    const int32_t* elem2, int B, int n2, const int32_t* n_add, const int32_t* cj_ptr,
                    const long long* cj_idx, const float* cj_noise, float* pos2, void* stream) {
    AGB_CHECK_ARG(B >= 1 && n2 >= 0, "agb_plot_extend: bad sizes");
    if (n2 == 0) return AGB_OK;
    hipLaunchKernelGGL(k_plot_extend, dim3(agb_cdiv(n2, 256)), dim3(256), 0, (hipStream_t)stream, pos1, ptr1, mins, ptr2,
                       elem2, n2, n_add, cj_ptr, cj_idx, cj_noise, pos2);
    AGB_CHECK_LAUNCH("agb_plot_extend");
    return AGB_OK;
}

// Crop with one polygon per plot (polys double[B][2*nv]); a plot none of whose points falls inside is left whole
// (transforms.py:1541-1543).  Features x = [1, z, ||xy - (fcx, fcy) + 1e-6||].  Scratch: flag / slot int32[n], cnt int32[B],
// scan_scratch int32[agb_scan_scratch_elems(n)].  Out as agb_plot_prepare.
AGB_INTERNAL int agb_plot_crop(const float* pos, const int32_t* ptr, const int32_t* elem, int B, int n, const double* polys, int nv,
                  float fcx, float fcy, int32_t* flag, int32_t* slot, int32_t* cnt, int32_t* scan_scratch,
                  float* pos_out, float* x_out, long long* src, int32_t* out_ptr, int32_t* n_out_dev, void* stream) {
    AGB_CHECK_ARG(B >= 1 && n >= 0 && nv >= 3, "agb_plot_crop: bad sizes (B %d, n %d, nv %d)", B, n, nv);
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {
        (void)hipMemsetAsync(out_ptr, 0, sizeof(int32_t) * (B + 1), s);
        (void)hipMemsetAsync(n_out_dev, 0, sizeof(int32_t), s);
        return AGB_OK;
    }
    (void)hipMemsetAsync(cnt, 0, sizeof(int32_t) * B, s);
    hipLaunchKernelGGL(k_plot_inside, dim3(agb_cdiv(n, 256)), dim3(256), 0, s, pos, elem, n, polys, nv, flag, cnt);
    hipLaunchKernelGGL(k_plot_keep_all_if_none, dim3(agb_cdiv(n, 256)), dim3(256), 0, s, flag, elem, n, cnt);
    agb_launch_exclusive_scan(flag, n, slot, scan_scratch, n_out_dev, s);
    PlotXform t{1.f, 1.f, 1.f, 0.f, 0.f, 0.f, fcx, fcy, 0, 0};
    hipLaunchKernelGGL(k_plot_emit, dim3(agb_cdiv(n, 256)), dim3(256), 0, s, pos, flag, slot, n, t, pos_out, x_out, src);
    hipLaunchKernelGGL(k_plot_out_ptr, dim3(agb_cdiv(B + 1, 64)), dim3(64), 0, s, slot, ptr, n_out_dev, B, n, out_ptr);
    AGB_CHECK_LAUNCH("agb_plot_crop");
    return AGB_OK;
}

}  // extern "C"
