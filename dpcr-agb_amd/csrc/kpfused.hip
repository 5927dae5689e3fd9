// kpfused.hip — the rigid kernel-point convolution as ONE kernel per direction: neighbourhood gather + influence weights +
// feature x kernel-weight contraction, the weighted neighbourhood features wf[N, K, Cin] never in HBM.
//
// Reference: torch_points3d/modules/KPConv/blocks.py:304-400 (one expression there: neighbours gathered [N,H,Cin], influences
// [N,K,H], matmul -> [N,K,Cin], matmul with weights [K,Cin,Cout], sum over K).  The two-kernel form of this library
// (kpconv.hip gather -> wf in HBM -> spconv.hip dense product; a second dense product on the stored wf for the weight
// gradient) moves 15 * Cin * 4 bytes per row out and back in twice per layer and step: at the 16- and 32-channel levels
// of the 16 k-point plots that is ~0.4-0.5 GB per launch, 2/3 of those layers' HBM traffic.
//
// One workgroup (eight waves; sixteen in the 32-channel backward launch: KpfShape) owns a tile of 16 consecutive query rows at a
// time (persistent: it walks tiles blockIdx.x, + gridDim.x, ..):
//   gather       every wave takes TWO rows of the tile (ONE with sixteen waves; rows ranked by neighbour count, so a wave's rows
//                have similar lengths) and runs them in lockstep, 32 neighbours in all per trip: the neighbour indices go through
//                LDS, then the coordinates and feature rows of all 32 neighbours are requested before any is used (template:
//                RPW rows x UB blocks of four).  Per row the product
//                Infl^T (K x H) . X (H x C) on v_mfma_f32_16x16x4_f32 as in kpconv.hip k_kpconv_gather_mm_fwd (lane (k, j)
//                evaluates ONE influence); the (K x C) results go to the LDS tile  t[row][k * C + c]   (row stride K*C + 4
//                floats: the 16-byte operand reads of the next phase touch every bank once)
//   contraction  out[16 rows x Cout] = t[16 x K*C] . W[K*C x Cout] on the same instruction: A = 16 rows x 4 reduction indices
//                read from LDS as one ds_read_b128 per four MFMAs, B = W held in REGISTERS for the lifetime of the workgroup
//                (16 channels: 8 VGPRs per wave, 32 channels: 32): W is read from L2 once per workgroup, not once per tile.
//                The waves split (column block, reduction range) and the partial sums meet in LDS in a fixed order.
//   weight grad  (backward launch) dWt[K*Cout x Cin] += t^T (K*Cout x 16 rows) . x (16 rows x Cin) with the accumulators in
//                registers across all tiles of the workgroup; one partial per workgroup, reduced in fixed order afterwards
//                (deterministic, no atomics).
// Backward of a layer on ONE point set with a symmetric neighbour relation (kpconv_ops.KPConvSymmetricFunction): with
// wfd = the same gather applied to dy with the kernel points mirrored,
//     dx[j, c]    = sum_{k,o} wfd[j, k, o] W[k, c, o]           (the contraction above on W's transpose)
//     dW[k, c, o] = sum_j x[j, c] wfd[j, k, o]                  (the weight-gradient phase above: no stored wf needed)
// so the backward pass is ONE launch that gathers once and feeds both products from the LDS tile.
#include "agb_common.h"
#include <type_traits>

typedef float kpf_f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float kpf_influence(float rx, float ry, float rz, float kx, float ky, float kz, float inv_ext) {
    // (kpconv.hip kp_influence: the same arithmetic, so both forms of the layer agree to the summation order)
    const float dx = rx - kx, dy = ry - ky, dz = rz - kz;
    const float d = __builtin_amdgcn_sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
    return fmaxf(fmaf(-d, inv_ext, 1.f), 0.f);
}

struct KpfArgs {
    const float* pts;          // [N][3] query = support points
    const int32_t* row_ptr;    // [N + 1]
    const int32_t* indices;    // ragged neighbour rows, every row sorted by distance
    int limit, N;
    const float* feat;         // gathered rows [N][ldf], 16 * CG channels
    int ldf;
    const float* kp;           // [K][3]
    int K;
    float sign, inv_ext;       // kernel points used: sign * kp
    const float* W;            // element (k, r, col) at W[k * w_sk + r * w_sr + col * w_sc]: r gathered channel, col output channel
    int w_sk, w_sr, w_sc;
    float* out;                // [N][ldo], 16 * CO channels; NULL: no contraction
    int ldo;
    const float* xrows;        // DW: [N][ldx], 16 * CO channels
    int ldx;
    float* dw_part;            // DW: [gridDim.x][K * 16 * CG][16 * CO]
};

// NW waves per workgroup, RB blocks of 16 rows per tile (the W registers of a wave serve all RB blocks)
template <int CG, int CO, bool DW, int NW, int RB>
__global__ __launch_bounds__(64 * NW) void k_kpconv_fused(KpfArgs a) {
    static_assert(NW % CO == 0 && (16 * RB) % NW == 0 && (8 * NW) % (16 * RB) == 0, "one column block per wave, whole rows per wave");
    constexpr int RPW = 16 * RB / NW;                            // rows of a wave (in lockstep)
    // blocks of four neighbours of each row per trip: 32 row fetches in flight per wave at 16 channels (the first level: points in
    // input order, every neighbour row its own HBM sector), 16 at 32 channels (cell-sorted levels: fewer registers, more waves)
    constexpr int UB = (CG == 1 ? 8 : 4) / RPW;
    constexpr int Cg = 16 * CG, Co = 16 * CO, R = 16 * RB;
    constexpr int SPLITK = NW / CO;                              // waves sharing a column block
    constexpr int GW_MAX = (16 * CG + SPLITK - 1) / SPLITK;      // 16-index groups of the reduction per wave (K <= 16)
    constexpr int GD_MAX = (16 * CG + NW - 1) / NW;              // groups of dWt rows per wave
    extern __shared__ float kpf_lds[];
    const int K = a.K, KG = K * CG, S = K * Cg + 4;
    float* tile = kpf_lds;                                       // [R][S]
    float* red = kpf_lds + R * S;                                // [SPLITK][R][Co]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int m = lane & 15, g = lane >> 4;
    // (lanes of the padding kernel points k >= K sit 1e18 away: influence max(0, 1 - d / extent) = 0 without a select)
    const float kx = m < K ? a.sign * a.kp[3 * m] : 1e18f, ky = m < K ? a.sign * a.kp[3 * m + 1] : 0.f,
                kz = m < K ? a.sign * a.kp[3 * m + 2] : 0.f;
    // ---- this wave's share of W, for the lifetime of the workgroup
    const int cb = w % CO, kh = w / CO;
    const int GW = (KG + SPLITK - 1) / SPLITK;
    const int q0 = kh * GW, q1 = min(KG, q0 + GW);
    float breg[GW_MAX][4];
    if (a.out) {
#pragma unroll
        for (int qi = 0; qi < GW_MAX; ++qi) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kc = 16 * (q0 + qi) + 4 * g + j;       // reduction index = k * Cg + r
                const int k = kc / Cg, r = kc - k * Cg;
                breg[qi][j] = (q0 + qi < q1) ? a.W[(long long)k * a.w_sk + (long long)r * a.w_sr + (long long)(16 * cb + m) * a.w_sc] : 0.f;
            }
        }
    }
    // ---- weight-gradient accumulators: rows [16 q, 16 q + 16) of dWt for q in this wave's range, all Co columns
    const int GD = (KG + NW - 1) / NW;
    const int d0 = w * GD, d1 = min(KG, d0 + GD);
    kpf_f32x4 dacc[DW ? GD_MAX : 1][DW ? CO : 1];
    if (DW) {
#pragma unroll
        for (int qi = 0; qi < GD_MAX; ++qi)
#pragma unroll
            for (int t = 0; t < CO; ++t) dacc[qi][t] = kpf_f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int ntiles = (a.N + R - 1) / R;
    for (int tile_id = blockIdx.x; tile_id < ntiles; tile_id += gridDim.x) {
        const int n0 = tile_id * R;
        // the x rows of the weight-gradient phase: requested now, used after the barrier
        float xb[DW ? 4 * RB : 1][DW ? CO : 1];
        if (DW) {
#pragma unroll
            for (int s = 0; s < 4 * RB; ++s) {
                const int rn = n0 + 4 * s + g;
#pragma unroll
                for (int t = 0; t < CO; ++t) xb[s][t] = rn < a.N ? a.xrows[(long long)rn * a.ldx + 16 * t + m] : 0.f;
            }
        }
        // ---- gather: this wave's four rows INTERLEAVED — every trip requests the coordinates and feature rows of 8
        //      neighbours of each of the four rows before any of them is used (32 row fetches in flight per wave instead of 8:
        //      one wave per row with a trip's loads waited for at once was latency-bound), all offsets 32-bit, no per-element
        //      validity logic (ragged rows hold real neighbours only; a lane past the row's end reads row 0 with influence 0).
        {
            const int wu = __builtin_amdgcn_readfirstlane(w);
            // The four rows of a wave run in lockstep for max(length) trips: the tile's R rows are RANKED by neighbour count
            // (every wave ranks all R: ~3 instructions per row) and wave w takes ranks 4w .. 4w+3 — rows of similar length
            // (at 21 +- 9 neighbours per row, four consecutive rows ran 5.1 trips where 2.7 were needed).
            int rbeg[RPW], rlen[RPW], myrow[RPW];
            float qxs[RPW], qys[RPW], qzs[RPW];
            int maxlen = 0;
            {
                const int nl = min(n0 + (lane & (R - 1)), a.N - 1);
                const int lb = a.row_ptr[nl], le = a.row_ptr[nl + 1];
                const int len_l = (n0 + lane < a.N && lane < R) ? min(a.limit, le - lb) : 0;
                const float plx = a.pts[3 * (long long)nl], ply = a.pts[3 * (long long)nl + 1], plz = a.pts[3 * (long long)nl + 2];
                int rank = 0;
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    const int lj = __builtin_amdgcn_readlane(len_l, j);
                    rank += (lj > len_l || (lj == len_l && j < lane)) ? 1 : 0;          // longest first
                }
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    const unsigned long long hit = __ballot(lane < R && rank == RPW * wu + r);
                    const int row = __builtin_ctzll(hit);                                   // (ranks are a permutation of 0 .. R-1)
                    myrow[r] = row;
                    rbeg[r] = __builtin_amdgcn_readlane(lb, row);
                    rlen[r] = __builtin_amdgcn_readlane(len_l, row);
                    maxlen = max(maxlen, rlen[r]);
                    qxs[r] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, plx), row));
                    qys[r] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ply), row));
                    qzs[r] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, plz), row));
                }
            }
            kpf_f32x4 acc[RPW][CG];
#pragma unroll
            for (int r = 0; r < RPW; ++r)
#pragma unroll
                for (int t = 0; t < CG; ++t) acc[r][t] = kpf_f32x4{0.f, 0.f, 0.f, 0.f};
            const char* pts_b = reinterpret_cast<const char*>(a.pts);
            const char* feat_b = reinterpret_cast<const char*>(a.feat);
            const unsigned ldfb = 4u * (unsigned)a.ldf, moff = 4u * (unsigned)m;
            int* ids = reinterpret_cast<int*>(red + SPLITK * R * Co) + 64 * RPW * wu;      // [RPW rows][64] of this wave
            for (int h0 = 0; h0 < maxlen; h0 += 64) {
                // the rows' neighbour indices go through LDS: the trips below then depend on LDS reads only, and the feature
                // fetches of all four rows stay in flight together (a cross-lane move of a loaded register made every trip
                // wait for every outstanding fetch)
                {
                    int myid[RPW];
#pragma unroll
                    for (int r = 0; r < RPW; ++r) myid[r] = h0 + lane < rlen[r] ? a.indices[rbeg[r] + h0 + lane] : -1;
#pragma unroll
                    for (int r = 0; r < RPW; ++r) ids[64 * r + lane] = myid[r];
                }
                __builtin_amdgcn_wave_barrier();
                const int ntrip = (min(64, maxlen - h0) + 4 * UB - 1) / (4 * UB);
                const int* idp = ids + g;
                for (int trip = 0; trip < ntrip; ++trip, idp += 4 * UB) {
                    int id[RPW][UB];
                    float px[RPW][UB], py[RPW][UB], pz[RPW][UB], fb[RPW][UB][CG];
#pragma unroll
                    for (int r = 0; r < RPW; ++r)
#pragma unroll
                        for (int u = 0; u < UB; ++u) id[r][u] = idp[64 * r + 4 * u];
#pragma unroll
                    for (int r = 0; r < RPW; ++r) {
#pragma unroll
                        for (int u = 0; u < UB; ++u) {
                            const unsigned ic = (unsigned)max(id[r][u], 0);
                            const float* pp = reinterpret_cast<const float*>(pts_b + __umul24(ic, 12u));
                            px[r][u] = pp[0], py[r][u] = pp[1], pz[r][u] = pp[2];
                            const unsigned fo = __umul24(ic, ldfb) + moff;
#pragma unroll
                            for (int t = 0; t < CG; ++t) fb[r][u][t] = *reinterpret_cast<const float*>(feat_b + fo + 64 * t);
                        }
                    }
                    // (no per-row branch: a row that has ended runs on with influence 0 — the four rows of a wave are
                    //  neighbours in space with similar counts, and a conditional update made the compiler move all
                    //  accumulators between the two register files every trip)
#pragma unroll
                    for (int r = 0; r < RPW; ++r) {
#pragma unroll
                        for (int u = 0; u < UB; ++u) {
                            float av = kpf_influence(px[r][u] - qxs[r], py[r][u] - qys[r], pz[r][u] - qzs[r], kx, ky, kz, a.inv_ext);
                            av = id[r][u] < 0 ? 0.f : av;
#pragma unroll
                            for (int t = 0; t < CG; ++t)
                                acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, fb[r][u][t], acc[r][t], 0, 0, 0);
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                float* trow = tile + myrow[r] * S;
#pragma unroll
                for (int t = 0; t < CG; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int k = 4 * g + i;
                        if (k < K) trow[k * Cg + 16 * t + m] = acc[r][t][i];
                    }
            }
        }
        __syncthreads();
        // ---- contraction with W
        if (a.out) {
            kpf_f32x4 d[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) d[rb] = kpf_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int qi = 0; qi < GW_MAX; ++qi) {
                if (q0 + qi < q1) {
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) {
                        const float4 av = *reinterpret_cast<const float4*>(tile + (16 * rb + m) * S + 16 * (q0 + qi) + 4 * g);
                        d[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, breg[qi][0], d[rb], 0, 0, 0);
                        d[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, breg[qi][1], d[rb], 0, 0, 0);
                        d[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, breg[qi][2], d[rb], 0, 0, 0);
                        d[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, breg[qi][3], d[rb], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int i = 0; i < 4; ++i) red[(kh * R + 16 * rb + 4 * g + i) * Co + 16 * cb + m] = d[rb][i];
        }
        // ---- weight gradient: dWt[16 q + .][.] += t^T x
        if (DW) {
#pragma unroll
            for (int qi = 0; qi < GD_MAX; ++qi) {
                if (d0 + qi < d1) {
                    float av[4 * RB];
#pragma unroll
                    for (int s = 0; s < 4 * RB; ++s) av[s] = tile[(4 * s + g) * S + 16 * (d0 + qi) + m];
#pragma unroll
                    for (int s = 0; s < 4 * RB; ++s)
#pragma unroll
                        for (int t = 0; t < CO; ++t)
                            dacc[qi][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], xb[s][t], dacc[qi][t], 0, 0, 0);
                }
            }
        }
        __syncthreads();
        if (a.out) {
            for (int e = threadIdx.x; e < R * Co; e += 64 * NW) {
                const int row = e / Co, col = e - row * Co;
                float v = red[e];
#pragma unroll
                for (int s2 = 1; s2 < SPLITK; ++s2) v += red[s2 * R * Co + e];
                if (n0 + row < a.N) a.out[(long long)(n0 + row) * a.ldo + col] = v;
            }
        }
        // (the next tile's gather writes `tile` after this barrier pair; its first barrier orders the reads of `red` above
        //  against the next partial sums)
    }
    if (DW) {
        float* part = a.dw_part + (long long)blockIdx.x * (K * Cg) * Co;
#pragma unroll
        for (int qi = 0; qi < GD_MAX; ++qi) {
            if (d0 + qi < d1) {
#pragma unroll
                for (int t = 0; t < CO; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        part[(long long)(16 * (d0 + qi) + 4 * g + i) * Co + 16 * t + m] = dacc[qi][t][i];
            }
        }
    }
}

// dW[k][c][o] = sum over workgroups of part[wg][(k * Cout + o) * Cin + c], in workgroup order (16 interleaved running sums
// per element, combined in fixed order).  block = 16 elements x 16 segments.
__global__ __launch_bounds__(256) void k_kpf_reduce_dw(const float* __restrict__ part, int G, int K, int Cin, int Cout,
                                                       float* __restrict__ dW, int accumulate) {
    __shared__ float s_red[16][17];
    const int el = threadIdx.x & 15, seg = threadIdx.x >> 4;
    const int E = K * Cin * Cout;
    const int e = blockIdx.x * 16 + el;
    float v = 0.f;
    if (e < E)
        for (int wg = seg; wg < G; wg += 16) v += part[(long long)wg * E + e];
    s_red[seg][el] = v;
    __syncthreads();
    if (seg == 0 && e < E) {
        float t = s_red[0][el];
#pragma unroll
        for (int s2 = 1; s2 < 16; ++s2) t += s_red[s2][el];
        const int c = e % Cin, ko = e / Cin, o = ko % Cout, k = ko / Cout;
        float* dst = dW + ((long long)k * Cin + c) * Cout + o;
        *dst = accumulate ? *dst + t : t;
    }
}

namespace {
// the shape of a workgroup per channel width and launch kind: (waves, 16-row blocks per tile).  Measured (EXPERIMENTS.md):
// 16 channels: 8 waves x 2 rows (80 / 108 VGPRs: 6 / 4 waves per SIMD); 32 channels forward: 8 waves (127 VGPRs, two workgroups
// per CU); 32 channels backward: 16 waves x 1 row — W^T and the weight-gradient accumulators spread over twice the waves
// (128 instead of 192 VGPRs: one workgroup of 16 waves per CU instead of one of 8)
template <int CG, bool DW> struct KpfShape { static constexpr int NW = 8, RB = 1; };
template <> struct KpfShape<2, true> { static constexpr int NW = 16, RB = 1; };

template <int CG, int CO, bool DW>
constexpr size_t kpf_lds_bytes(int K) {
    return ((size_t)16 * KpfShape<CG, DW>::RB * (K * 16 * CG + 4) +
            (size_t)(KpfShape<CG, DW>::NW / CO) * 16 * KpfShape<CG, DW>::RB * 16 * CO + (size_t)16 * KpfShape<CG, DW>::RB * 64) * sizeof(float);
}

template <int CG, int CO, bool DW>
int kpf_grid(int K, int N) {
    // workgroups resident at once (registers and the K-dependent LDS tile decide): asked once per (instantiation, K)
    constexpr int NW = KpfShape<CG, DW>::NW, RB = KpfShape<CG, DW>::RB;
    static int cached[17] = {0};
    if (cached[K] == 0) {
        int dev = 0, cus = 256, per_cu = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const size_t lds = kpf_lds_bytes<CG, CO, DW>(K);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_kpconv_fused<CG, CO, DW, NW, RB>, 64 * NW, lds) != hipSuccess ||
            per_cu < 1)
            per_cu = 2;
        cached[K] = cus * per_cu;
    }
    const int ntiles = (N + 16 * RB - 1) / (16 * RB);
    return ntiles < cached[K] ? ntiles : cached[K];
}

template <int CG, int CO, bool DW>
void kpf_launch(const KpfArgs& a, int grid, hipStream_t st) {
    constexpr int NW = KpfShape<CG, DW>::NW, RB = KpfShape<CG, DW>::RB;
    const size_t lds = kpf_lds_bytes<CG, CO, DW>(a.K);
    hipLaunchKernelGGL((k_kpconv_fused<CG, CO, DW, NW, RB>), dim3(grid), dim3(64 * NW), lds, st, a);
}
}  // namespace

extern "C" {

// 1 where the fused kernels cover a layer: 15 or 16 >= K >= 1 kernel points, Cin == Cout in {16, 32}
int agb_kpconv_fused_supported(int K, int Cin, int Cout) {
    return K >= 1 && K <= 16 && Cin == Cout && (Cin == 16 || Cin == 32);
}

// workgroups the backward launch uses = partial weight gradients in its workspace
static int kpf_bwd_grid(int N, int K, int C) {
    return C == 16 ? kpf_grid<1, 1, true>(K, N) : kpf_grid<2, 2, true>(K, N);
}

size_t agb_kpconv_fused_bwd_workspace_bytes(int N, int K, int Cin, int Cout) {
    if (N <= 0 || !agb_kpconv_fused_supported(K, Cin, Cout)) return 0;
    return (size_t)kpf_bwd_grid(N, K, Cin) * K * Cin * Cout * sizeof(float);
}

// out [N][ldo] = KPConv(x) of a layer whose query and support points are the same set `pts` [N][3]: ragged neighbour rows
// (row_ptr, indices: agb_ball_query_fill_csr), every row cut at `limit` entries.  x [N][ldx] (Cin channels), kp [K][3],
// W [K][Cin][Cout].  blocks.py:304-400.
int agb_kpconv_fused_fwd(const float* pts, const int32_t* row_ptr, const int32_t* indices, int limit, int N, const float* x,
                         int ldx, const float* kp, int K, float extent, const float* W, float* out, int ldo, int Cin, int Cout,
                         void* stream) {
    AGB_CHECK_ARG(agb_kpconv_fused_supported(K, Cin, Cout), "agb_kpconv_fused_fwd: K %d, Cin %d, Cout %d not covered "
                  "(agb_kpconv_fused_supported)", K, Cin, Cout);
    // (row offsets are 24-bit x 24-bit products inside the kernel: N < 2^24 rows of at most 256 floats)
    AGB_CHECK_ARG(limit >= 1 && N >= 0 && N < (1 << 24) && ldx >= Cin && ldx <= 256 && ldo >= Cout && extent > 0.f,
                  "agb_kpconv_fused_fwd: limit %d, N %d (< 2^24), ldx %d (<= 256), ldo %d, extent %g", limit, N, ldx, ldo, (double)extent);
    if (N == 0) return AGB_OK;
    AGB_CHECK_ARG(pts && row_ptr && indices && x && kp && W && out, "agb_kpconv_fused_fwd: null pointer");
    KpfArgs a{pts, row_ptr, indices, limit, N, x, ldx, kp, K, 1.f, 1.f / extent, W, Cin * Cout, Cout, 1, out, ldo, nullptr, 0, nullptr};
    hipStream_t st = (hipStream_t)stream;
    if (Cin == 16) kpf_launch<1, 1, false>(a, kpf_grid<1, 1, false>(K, N), st);
    else kpf_launch<2, 2, false>(a, kpf_grid<2, 2, false>(K, N), st);
    AGB_CHECK_LAUNCH("agb_kpconv_fused_fwd");
    return AGB_OK;
}

// Backward of agb_kpconv_fused_fwd on a SYMMETRIC neighbour relation (j in row n <=> n in row j: an uncropped radius search of
// a point set against itself): dx [N][lddx] (NULL: not wanted) and dW [K][Cin][Cout] (NULL: not wanted; accumulate != 0: added
// to what dW holds) from dy [N][lddy] and the layer input x [N][ldx].  Nothing needs zero-filling; fixed summation order.
// workspace: agb_kpconv_fused_bwd_workspace_bytes(N, K, Cin, Cout) bytes (only used when dW is wanted).
int agb_kpconv_fused_bwd(const float* pts, const int32_t* row_ptr, const int32_t* indices, int limit, int N, const float* dy,
                         int lddy, const float* kp, int K, float extent, const float* W, const float* x, int ldx, float* dx,
                         int lddx, float* dW, int accumulate, void* workspace, size_t workspace_bytes, int Cin, int Cout,
                         void* stream) {
    AGB_CHECK_ARG(agb_kpconv_fused_supported(K, Cin, Cout), "agb_kpconv_fused_bwd: K %d, Cin %d, Cout %d not covered "
                  "(agb_kpconv_fused_supported)", K, Cin, Cout);
    AGB_CHECK_ARG(limit >= 1 && N >= 0 && N < (1 << 24) && lddy >= Cout && lddy <= 256 && extent > 0.f &&
                  (dx == nullptr || lddx >= Cin) && (dW == nullptr || ldx >= Cin),
                  "agb_kpconv_fused_bwd: limit %d, N %d (< 2^24), lddy %d (<= 256), lddx %d, ldx %d, extent %g", limit, N, lddy, lddx,
                  ldx, (double)extent);
    if (N == 0 || (dx == nullptr && dW == nullptr)) return AGB_OK;
    AGB_CHECK_ARG(pts && row_ptr && indices && dy && kp && W, "agb_kpconv_fused_bwd: null pointer");
    AGB_CHECK_ARG(dW == nullptr || (x != nullptr && workspace != nullptr &&
                                    workspace_bytes >= agb_kpconv_fused_bwd_workspace_bytes(N, K, Cin, Cout)),
                  "agb_kpconv_fused_bwd: the weight gradient needs x and a workspace of %zu bytes (%zu given)",
                  agb_kpconv_fused_bwd_workspace_bytes(N, K, Cin, Cout), workspace_bytes);
    hipStream_t st = (hipStream_t)stream;
    // gathered rows: dy (Cout channels), mirrored kernel points; contraction on W^T: element (k, r = o, col = c) = W[k][c][o]
    KpfArgs a{pts, row_ptr, indices, limit, N, dy, lddy, kp, K, -1.f, 1.f / extent, W, Cin * Cout, 1, Cout, dx, lddx, x, ldx,
              (float*)workspace};
    if (dW) {
        const int grid = kpf_bwd_grid(N, K, Cin);
        if (Cin == 16) kpf_launch<1, 1, true>(a, grid, st);
        else kpf_launch<2, 2, true>(a, grid, st);
        hipLaunchKernelGGL(k_kpf_reduce_dw, dim3(agb_cdiv((long long)K * Cin * Cout, 16)), dim3(256), 0, st, (const float*)workspace,
                           grid, K, Cin, Cout, dW, accumulate);
    } else {
        if (Cin == 16) kpf_launch<1, 1, false>(a, kpf_grid<1, 1, false>(K, N), st);
        else kpf_launch<2, 2, false>(a, kpf_grid<2, 2, false>(K, N), st);
    }
    AGB_CHECK_LAUNCH("agb_kpconv_fused_bwd");
    return AGB_OK;
}

}  // extern "C"
