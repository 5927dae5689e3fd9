// abi_ws.hip — single-workspace forms of the entry points with many scratch buffers (host code only).
// A caller sizes ONE device buffer with the matching *_workspace_bytes() helper and passes it in; the carving below is
// the only place that knows the internal scratch layout (every piece 256-byte aligned).
#include "agb_common.h"

extern "C" {
int agb_scan_scratch_elems(int n);
AGB_INTERNAL int agb_grid_subsample(const float* pts, const float* feats, int fdim, int n, const int32_t* ptr, const int32_t* elem,
                       int B, float dl, int cap, int32_t* bbox_ord, float* origin, int32_t* dims, int32_t* cell_cnt,
                       int32_t* cell_start, int32_t* slot, int32_t* flag, int32_t* cell_of, int32_t* members,
                       int32_t* scan_scratch, float* out_pts, float* out_feats, int32_t* out_ptr, int32_t* n_out_dev,
                       int32_t* status, void* stream);
AGB_INTERNAL int agb_voxelize_last(const float* pos, const long long* perm, const int32_t* ptr, const int32_t* elem, int B, int n,
                      float size, int cap, int32_t* bbox_ord, float* lo, int32_t* span, int32_t* cells, int32_t* slot,
                      int32_t* flag, int32_t* cell_of, int32_t* scan_scratch, int32_t* coords, long long* keep,
                      int32_t* out_ptr, int32_t* n_out_dev, int32_t* bounds, int32_t* status, void* stream,
                      unsigned long long seed);
AGB_INTERNAL int agb_plot_prepare(const float* pos, const int32_t* ptr, const int32_t* elem, int B, int n, const float* xform,
                     int scale_div, int z_from_zero, const double* poly, int nv, float* zmin, float* pos_t, int32_t* flag,
                     int32_t* slot, int32_t* scan_scratch, float* pos_out, float* x_out, long long* src, int32_t* out_ptr,
                     int32_t* n_out_dev, void* stream);
AGB_INTERNAL int agb_plot_crop(const float* pos, const int32_t* ptr, const int32_t* elem, int B, int n, const double* polys, int nv,
                  float fcx, float fcy, int32_t* flag, int32_t* slot, int32_t* cnt, int32_t* scan_scratch, float* pos_out,
                  float* x_out, long long* src, int32_t* out_ptr, int32_t* n_out_dev, void* stream);
}

namespace {
struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(void* p) : base((char*)p) {}
    template <typename T> T* take(size_t count) {
        T* p = base ? (T*)(base + off) : nullptr;
        off += (count * sizeof(T) + 255) / 256 * 256;
        return p;
    }
};
inline size_t rows(int n) { return (size_t)(n > 0 ? n : 1); }
}  // namespace

extern "C" {

// ---- grid subsampling
struct SubWs { int32_t *bbox_ord, *dims, *cell_cnt, *cell_start, *slot, *flag, *cell_of, *members, *scan; float* origin; };
static size_t sub_carve(void* ws, int n, int B, int cap, SubWs* o) {
    Carver c(ws);
    const size_t nc = (size_t)B * cap + 1;
    o->bbox_ord = c.take<int32_t>(6 * (size_t)B); o->dims = c.take<int32_t>(3 * (size_t)B);
    o->origin = c.take<float>(3 * (size_t)B);
    o->cell_cnt = c.take<int32_t>(nc); o->cell_start = c.take<int32_t>(nc); o->slot = c.take<int32_t>(nc);
    o->flag = c.take<int32_t>(nc); o->cell_of = c.take<int32_t>(rows(n)); o->members = c.take<int32_t>(rows(n));
    o->scan = c.take<int32_t>((size_t)agb_scan_scratch_elems((int)nc));
    return c.off;
}
size_t agb_grid_subsample_workspace_bytes(int n, int B, int cap) {
    SubWs o;
    return sub_carve(nullptr, n, B, cap, &o);
}
int agb_grid_subsample_ws(const float* pts, const float* feats, int fdim, int n, const int32_t* ptr, const int32_t* elem,
                          int B, float dl, int cap, void* workspace, float* out_pts, float* out_feats, int32_t* out_ptr,
                          int32_t* n_out_dev, int32_t* status, void* stream) {
    AGB_CHECK_ARG(workspace != nullptr && B >= 1 && cap >= 1 && (long long)B * cap < 0x7FFFFFF0LL,
                  "agb_grid_subsample_ws: workspace required, B %d, cap %d", B, cap);
    SubWs o;
    sub_carve(workspace, n, B, cap, &o);
    return agb_grid_subsample(pts, feats, fdim, n, ptr, elem, B, dl, cap, o.bbox_ord, o.origin, o.dims, o.cell_cnt,
                              o.cell_start, o.slot, o.flag, o.cell_of, o.members, o.scan, out_pts, out_feats, out_ptr,
                              n_out_dev, status, stream);
}

// ---- GridSampling3D(mode = "last")
struct VoxWs { int32_t *bbox_ord, *span, *cells, *slot, *flag, *cell_of, *scan; float* lo; };
static size_t vox_carve(void* ws, int n, int B, int cap, VoxWs* o) {
    Carver c(ws);
    const size_t nc = (size_t)B * cap + 1;
    o->bbox_ord = c.take<int32_t>(6 * (size_t)B); o->span = c.take<int32_t>(3 * (size_t)B);
    o->lo = c.take<float>(3 * (size_t)B);
    o->cells = c.take<int32_t>(nc); o->slot = c.take<int32_t>(nc); o->flag = c.take<int32_t>(nc);
    o->cell_of = c.take<int32_t>(rows(n)); o->scan = c.take<int32_t>((size_t)agb_scan_scratch_elems((int)nc));
    return c.off;
}
size_t agb_voxelize_last_workspace_bytes(int n, int B, int cap) {
    VoxWs o;
    return vox_carve(nullptr, n, B, cap, &o);
}
int agb_voxelize_last_ws(const float* pos, const long long* perm, const int32_t* ptr, const int32_t* elem, int B, int n,
                         float size, int cap, void* workspace, int32_t* coords, long long* keep, int32_t* out_ptr,
                         int32_t* n_out_dev, int32_t* bounds, int32_t* status, void* stream) {
    AGB_CHECK_ARG(workspace != nullptr && B >= 1 && cap >= 1 && (long long)B * cap < 0x7FFFFFF0LL,
                  "agb_voxelize_last_ws: workspace required, B %d, cap %d", B, cap);
    VoxWs o;
    vox_carve(workspace, n, B, cap, &o);
    AGB_CHECK_ARG(perm != nullptr || n == 0, "agb_voxelize_last_ws: perm required (agb_voxelize_last_seeded_ws draws the "
                  "shuffle on the device)");
    return agb_voxelize_last(pos, perm, ptr, elem, B, n, size, cap, o.bbox_ord, o.lo, o.span, o.cells, o.slot, o.flag,
                             o.cell_of, o.scan, coords, keep, out_ptr, n_out_dev, bounds, status, stream, 0ull);
}
// The same with the shuffle drawn ON THE DEVICE from `seed` (no permutation tensor: csrc/voxelize.hip vox_perm — a keyed
// pseudo-random bijection per cloud); same workspace.  A voxel's representative is uniformly random over its points.
int agb_voxelize_last_seeded_ws(const float* pos, unsigned long long seed, const int32_t* ptr, const int32_t* elem, int B, int n,
                                float size, int cap, void* workspace, int32_t* coords, long long* keep, int32_t* out_ptr,
                                int32_t* n_out_dev, int32_t* bounds, int32_t* status, void* stream) {
    AGB_CHECK_ARG(workspace != nullptr && B >= 1 && cap >= 1 && (long long)B * cap < 0x7FFFFFF0LL,
                  "agb_voxelize_last_seeded_ws: workspace required, B %d, cap %d", B, cap);
    VoxWs o;
    vox_carve(workspace, n, B, cap, &o);
    return agb_voxelize_last(pos, nullptr, ptr, elem, B, n, size, cap, o.bbox_ord, o.lo, o.span, o.cells, o.slot, o.flag,
                             o.cell_of, o.scan, coords, keep, out_ptr, n_out_dev, bounds, status, stream, seed);
}

// ---- transform chain
struct PlotWs { float *zmin, *pos_t; int32_t *flag, *slot, *cnt, *scan; };
static size_t plot_carve(void* ws, int n, int B, PlotWs* o) {
    Carver c(ws);
    o->zmin = c.take<float>((size_t)B); o->pos_t = c.take<float>(3 * rows(n));
    o->flag = c.take<int32_t>(rows(n)); o->slot = c.take<int32_t>(rows(n)); o->cnt = c.take<int32_t>((size_t)B);
    o->scan = c.take<int32_t>((size_t)agb_scan_scratch_elems((int)rows(n)));
    return c.off;
}
size_t agb_plot_workspace_bytes(int n, int B) {
    PlotWs o;
    return plot_carve(nullptr, n, B, &o);
}
int agb_plot_prepare_ws(const float* pos, const int32_t* ptr, const int32_t* elem, int B, int n, const float* xform,
                        int scale_div, int z_from_zero, const double* poly, int nv, void* workspace, float* pos_out,
                        float* x_out, long long* src, int32_t* out_ptr, int32_t* n_out_dev, void* stream) {
    AGB_CHECK_ARG(workspace != nullptr && B >= 1, "agb_plot_prepare_ws: workspace required");
    PlotWs o;
    plot_carve(workspace, n, B, &o);
    return agb_plot_prepare(pos, ptr, elem, B, n, xform, scale_div, z_from_zero, poly, nv, o.zmin, o.pos_t, o.flag, o.slot,
                            o.scan, pos_out, x_out, src, out_ptr, n_out_dev, stream);
}
int agb_plot_crop_ws(const float* pos, const int32_t* ptr, const int32_t* elem, int B, int n, const double* polys, int nv,
                     float fcx, float fcy, void* workspace, float* pos_out, float* x_out, long long* src, int32_t* out_ptr,
                     int32_t* n_out_dev, void* stream) {
    AGB_CHECK_ARG(workspace != nullptr && B >= 1, "agb_plot_crop_ws: workspace required");
    PlotWs o;
    plot_carve(workspace, n, B, &o);
    return agb_plot_crop(pos, ptr, elem, B, n, polys, nv, fcx, fcy, o.flag, o.slot, o.cnt, o.scan, pos_out, x_out, src,
                         out_ptr, n_out_dev, stream);
}

}  // extern "C"
