// ORACLE / test infrastructure only.  extern "C" binding around the REFERENCE's own C++ (compiled from where it lies
// under /root/reference by oracle/Makefile into oracle/_ref/, never copied): lets ctypes call
//   batch_nanoflann_neighbors   torch_points3d/modules/KPConv/cpp_wrappers/cpp_neighbors/neighbors/neighbors.cpp:211-333
//   batch_grid_subsampling      .../cpp_subsampling/grid_subsampling/grid_subsampling.cpp:109-211
// exactly as the reference's CPython glue does (wrapper.cpp:188-198 / wrapper.cpp:259-293), which itself no longer
// compiles against NumPy 2.  This file contains no algorithm: it converts flat arrays to the std::vector arguments
// the reference functions take and back.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "cpp_neighbors/neighbors/neighbors.h"
#include "cpp_subsampling/grid_subsampling/grid_subsampling.h"

static std::vector<PointXYZ> to_points(const float* p, int n) {
    std::vector<PointXYZ> v(n);
    for (int i = 0; i < n; ++i) v[i] = PointXYZ(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
    return v;
}

extern "C" {

void ref_free(void* p) { free(p); }

// out: malloc'ed int[nq * width]; returns width (max neighbour count), pad value = ns
int ref_batch_neighbors(const float* queries, int nq, const float* supports, int ns, const int* q_batches,
                        const int* s_batches, int B, float radius, int** out) {
    std::vector<PointXYZ> q = to_points(queries, nq), s = to_points(supports, ns);
    std::vector<int> qb(q_batches, q_batches + B), sb(s_batches, s_batches + B), res;
    batch_nanoflann_neighbors(q, s, qb, sb, res, radius);
    int width = nq > 0 ? (int)(res.size() / (size_t)nq) : 0;
    *out = (int*)malloc(sizeof(int) * (res.size() ? res.size() : 1));
    memcpy(*out, res.data(), sizeof(int) * res.size());
    return width;
}

// out_points: malloc'ed float[m*3]; out_feats: float[m*fdim] (or NULL); out_batches: int[B]; returns m
int ref_batch_grid_subsampling(const float* points, int n, const float* feats, int fdim, const int* batches, int B,
                               float sampleDl, int max_p, float** out_points, float** out_feats, int* out_batches) {
    std::vector<PointXYZ> p = to_points(points, n), sp;
    std::vector<float> f, sf;
    if (feats && fdim > 0) f.assign(feats, feats + (size_t)n * fdim);
    std::vector<int> c, sc, b(batches, batches + B), sb;
    batch_grid_subsampling(p, sp, f, sf, c, sc, b, sb, sampleDl, max_p);
    int m = (int)sp.size();
    *out_points = (float*)malloc(sizeof(float) * 3 * (m ? m : 1));
    for (int i = 0; i < m; ++i) {
        (*out_points)[3 * i] = sp[i].x;
        (*out_points)[3 * i + 1] = sp[i].y;
        (*out_points)[3 * i + 2] = sp[i].z;
    }
    if (out_feats) {
        *out_feats = (float*)malloc(sizeof(float) * (sf.size() ? sf.size() : 1));
        memcpy(*out_feats, sf.data(), sizeof(float) * sf.size());
    }
    for (int i = 0; i < B; ++i) out_batches[i] = sb[i];
    return m;
}

}  // extern "C"
