// ORACLE (test infrastructure only — never linked or called by the product path under dpcr-agb_amd/).
//
// CPU restatement of the reference's KPConv index path, pinned against the reference's own C++ built from
// /root/reference (oracle/_ref/libref_kpconv.so) by tests/test_oracle_kpconv.py and by the committed golden
// vectors tests/golden/kpconv_index_*.npz (generated from that build by tests/golden/make_kpconv_index_golden.py).
//
//  A1  radius neighbours   torch_points3d/modules/KPConv/cpp_wrappers/cpp_neighbors/neighbors/neighbors.cpp:211-333
//      semantics restated: per batch element, every support with d2 < r*r (strict, nanoflann.hpp:220-255), where
//      d2 = ((dx*dx) + (dy*dy)) + (dz*dz) in float32, d = query - support (L2_Simple_Adaptor, nanoflann.hpp:423-445),
//      sorted ascending by d2 (nanoflann.hpp:1280-1289; ties -> ascending support index, a tie can only come from
//      exactly equidistant points), indices offset by the batch element's first support row, rows padded to the
//      batch-wide maximum count with the shadow index supports.size() (neighbors.cpp:319-325).
//  A2  grid subsampling    .../cpp_subsampling/grid_subsampling/grid_subsampling.cpp:5-211, grid_subsampling.h:10-80
//      per batch element: origin = floor(min * (1/dl)) * dl (:27), NX/NY from (max - origin)/dl (:30-31),
//      key = iX + NX*iY + NX*NY*iZ with i* = floor((p - origin)/dl) (:53-56), per-cell float sums in original
//      point order (grid_subsampling.h:58-63), barycentre = sum * (float)(1.0/count) (:87), features = sum/(float)count
//      (:90-95), optional truncation to max_p points per element (:180-199).
//      order = 1 reproduces the reference's emission order (iteration order of the libstdc++ unordered_map the
//      reference fills, :85) by filling the same container in the same sequence; order = 0 is the canonical
//      key-sorted order the HIP kernels produce.
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstring>
#include <unordered_map>
#include <utility>
#include <vector>

namespace {

inline float dist2(const float* q, const float* s) {
    float dx = q[0] - s[0], dy = q[1] - s[1], dz = q[2] - s[2];
    float r = dx * dx;
    r = r + dy * dy;
    r = r + dz * dz;
    return r;
}

struct Hit {
    float d2;
    int idx;
};

void neighbours_of(const float* q, const float* s, int s_beg, int s_len, float r2, std::vector<Hit>& hits) {
    hits.clear();
    for (int j = 0; j < s_len; ++j) {
        float d2 = dist2(q, s + 3 * (size_t)(s_beg + j));
        if (d2 < r2) hits.push_back(Hit{d2, j});
    }
    std::stable_sort(hits.begin(), hits.end(), [](const Hit& a, const Hit& b) { return a.d2 < b.d2; });
}

struct Cell {
    int count = 0;
    float x = 0.f, y = 0.f, z = 0.f;
    std::vector<float> f;
};

}  // namespace

extern "C" {

// counts[nq]; returns the maximum count
int oracle_ball_query_count(const float* queries, int nq, const float* supports, int ns, const int* q_batches,
                            const int* s_batches, int B, float radius, int* counts) {
    (void)ns;
    float r2 = radius * radius;
    std::vector<Hit> hits;
    int qi = 0, s_beg = 0, mx = 0;
    for (int b = 0; b < B; ++b) {
        for (int i = 0; i < q_batches[b]; ++i, ++qi) {
            neighbours_of(queries + 3 * (size_t)qi, supports, s_beg, s_batches[b], r2, hits);
            counts[qi] = (int)hits.size();
            mx = std::max(mx, counts[qi]);
        }
        s_beg += s_batches[b];
    }
    (void)nq;
    return mx;
}

// out[nq * width], rows padded with ns; rows longer than width are truncated (keeps the closest)
void oracle_ball_query_fill(const float* queries, int nq, const float* supports, int ns, const int* q_batches,
                            const int* s_batches, int B, float radius, int width, int* out) {
    float r2 = radius * radius;
    std::vector<Hit> hits;
    int qi = 0, s_beg = 0;
    for (int b = 0; b < B; ++b) {
        for (int i = 0; i < q_batches[b]; ++i, ++qi) {
            neighbours_of(queries + 3 * (size_t)qi, supports, s_beg, s_batches[b], r2, hits);
            for (int j = 0; j < width; ++j)
                out[(size_t)qi * width + j] = j < (int)hits.size() ? hits[j].idx + s_beg : ns;
        }
        s_beg += s_batches[b];
    }
    (void)nq;
}

// out_points[cap*3], out_feats[cap*fdim] (may be NULL), out_batches[B], out_keys[cap] (may be NULL): cap >= n.
// returns the total number of subsampled points.
int oracle_grid_subsample(const float* points, int n, const float* feats, int fdim, const int* batches, int B,
                          float dl, int max_p, int order, float* out_points, float* out_feats, int* out_batches,
                          long long* out_keys) {
    if (max_p < 1) max_p = n;
    int beg = 0, m = 0;
    for (int b = 0; b < B; ++b) {
        int len = batches[b];
        const float* p = points + 3 * (size_t)beg;
        float mn[3] = {p[0], p[1], p[2]}, mx[3] = {p[0], p[1], p[2]};
        for (int i = 0; i < len; ++i)
            for (int a = 0; a < 3; ++a) {
                float v = p[3 * (size_t)i + a];
                if (v < mn[a]) mn[a] = v;
                if (v > mx[a]) mx[a] = v;
            }
        float inv = 1 / dl;  // float division, as `1/sampleDl` with a float operand
        float org[3];
        for (int a = 0; a < 3; ++a) org[a] = std::floor(mn[a] * inv) * dl;
        size_t NX = (size_t)std::floor((mx[0] - org[0]) / dl) + 1;
        size_t NY = (size_t)std::floor((mx[1] - org[1]) / dl) + 1;

        std::unordered_map<size_t, Cell> data;  // same container + same insertion sequence as the reference
        for (int i = 0; i < len; ++i) {
            const float* q = p + 3 * (size_t)i;
            size_t iX = (size_t)std::floor((q[0] - org[0]) / dl);
            size_t iY = (size_t)std::floor((q[1] - org[1]) / dl);
            size_t iZ = (size_t)std::floor((q[2] - org[2]) / dl);
            size_t key = iX + NX * iY + NX * NY * iZ;
            if (data.count(key) < 1) {
                Cell c;
                c.f.assign(fdim, 0.f);
                data.emplace(key, c);
            }
            Cell& c = data[key];
            c.count += 1;
            c.x += q[0];
            c.y += q[1];
            c.z += q[2];
            for (int j = 0; j < fdim; ++j) c.f[j] += feats[(size_t)(beg + i) * fdim + j];
        }
        std::vector<std::pair<size_t, const Cell*>> cells;
        cells.reserve(data.size());
        for (auto& kv : data) cells.push_back({kv.first, &kv.second});
        if (order == 0)
            std::sort(cells.begin(), cells.end(),
                      [](const std::pair<size_t, const Cell*>& a, const std::pair<size_t, const Cell*>& b) {
                          return a.first < b.first;
                      });
        int keep = std::min<int>((int)cells.size(), max_p);
        for (int j = 0; j < keep; ++j) {
            const Cell& c = *cells[j].second;
            float a = (float)(1.0 / c.count);
            out_points[3 * (size_t)m] = c.x * a;
            out_points[3 * (size_t)m + 1] = c.y * a;
            out_points[3 * (size_t)m + 2] = c.z * a;
            if (out_feats) {
                float cnt = (float)c.count;
                for (int f = 0; f < fdim; ++f) out_feats[(size_t)m * fdim + f] = c.f[f] / cnt;
            }
            if (out_keys) out_keys[m] = (long long)cells[j].first;
            ++m;
        }
        out_batches[b] = keep;
        beg += len;
    }
    return m;
}

}  // extern "C"
