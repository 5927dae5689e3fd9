/* A plain-C caller of libagbhip.so (include/agb_hip.h): coordinate hash insert -> kernel map -> sparse convolution on a
 * 3-voxel input, then Linear -> BatchNorm -> ReLU with the statistics taken from the product's epilogue, then the
 * pair-compacted kernel with a work-balanced tile table; all checked against a brute-force evaluation on the host.  Built by tests/c_abi/Makefile (gcc + the HIP
 * runtime API for device memory only), run by tests/test_c_caller.py on a GPU box.  Exit code 0 = all values match. */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "agb_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 2; } } while (0)
#define CHECK_AGB(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #x, r_, agb_last_error()); return 3; } } while (0)

int main(void) {
    enum { N = 3, K = 3, K3 = 27, CIN = 4, COUT = 4 };
    const int32_t coords[N][4] = {{0, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}};   /* (batch, x, y, z) */
    float X[N][CIN], W[K3][CIN][COUT], bias[COUT], Y[N][COUT], ref[N][COUT];
    for (int r = 0; r < N; ++r) for (int c = 0; c < CIN; ++c) X[r][c] = (float)(1 + r * CIN + c) * 0.25f;
    for (int k = 0; k < K3; ++k) for (int c = 0; c < CIN; ++c) for (int o = 0; o < COUT; ++o)
        W[k][c][o] = (float)(((k * 7 + c * 3 + o * 5) % 11) - 5) * 0.125f;
    for (int o = 0; o < COUT; ++o) bias[o] = 0.5f * (float)o;

    /* host reference: Y[r] = bias + sum_k X[row of coords[r] + offset_k] W[k], offset_k = (ix-1, iy-1, iz-1), x fastest */
    int expected_pairs = 0;
    for (int r = 0; r < N; ++r) {
        for (int o = 0; o < COUT; ++o) ref[r][o] = bias[o];
        for (int k = 0; k < K3; ++k) {
            const int dx = k % K - 1, dy = (k / K) % K - 1, dz = k / (K * K) - 1;
            for (int q = 0; q < N; ++q) {
                if (coords[q][0] == coords[r][0] && coords[q][1] == coords[r][1] + dx && coords[q][2] == coords[r][2] + dy &&
                    coords[q][3] == coords[r][3] + dz) {
                    ++expected_pairs;
                    for (int c = 0; c < CIN; ++c) for (int o = 0; o < COUT; ++o) ref[r][o] += X[q][c] * W[k][c][o];
                }
            }
        }
    }

    const int cap = agb_hash_capacity(N);
    int32_t *d_coords, *d_vals, *d_slot, *d_status, *d_nbr;
    uint64_t* d_keys;
    unsigned long long* d_pairs;
    float *d_X, *d_W, *d_b, *d_Y;
    CHECK_HIP(hipMalloc((void**)&d_coords, sizeof(coords)));
    CHECK_HIP(hipMalloc((void**)&d_keys, sizeof(uint64_t) * cap));
    CHECK_HIP(hipMalloc((void**)&d_vals, sizeof(int32_t) * cap));
    CHECK_HIP(hipMalloc((void**)&d_slot, sizeof(int32_t) * N));
    CHECK_HIP(hipMalloc((void**)&d_status, sizeof(int32_t) * 4));
    CHECK_HIP(hipMalloc((void**)&d_nbr, sizeof(int32_t) * K3 * N));
    CHECK_HIP(hipMalloc((void**)&d_pairs, sizeof(unsigned long long) * 64 * 16));
    CHECK_HIP(hipMalloc((void**)&d_X, sizeof(X)));
    CHECK_HIP(hipMalloc((void**)&d_W, sizeof(W)));
    CHECK_HIP(hipMalloc((void**)&d_b, sizeof(bias)));
    CHECK_HIP(hipMalloc((void**)&d_Y, sizeof(Y)));
    CHECK_HIP(hipMemcpy(d_coords, coords, sizeof(coords), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_X, X, sizeof(X), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_W, W, sizeof(W), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_b, bias, sizeof(bias), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemset(d_pairs, 0, sizeof(unsigned long long) * 64 * 16));

    CHECK_AGB(agb_hash_clear(d_keys, d_vals, cap, NULL));
    CHECK_AGB(agb_coords_insert(d_coords, N, NULL, d_keys, d_vals, cap, d_slot, d_status, NULL));
    CHECK_AGB(agb_kernel_map(d_coords, N, NULL, K, 1, 1, 0, d_keys, d_vals, cap, d_nbr, N, d_pairs, NULL));
    CHECK_AGB(agb_spconv_fwd(d_X, CIN, d_W, d_nbr, N, 0, d_b, d_Y, COUT, N, K3, CIN, COUT, NULL));
    CHECK_HIP(hipDeviceSynchronize());

    int32_t status[4], nbr[K3][N];
    unsigned long long pairs[64 * 16], total = 0;
    CHECK_HIP(hipMemcpy(status, d_status, sizeof(status), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(nbr, d_nbr, sizeof(nbr), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(pairs, d_pairs, sizeof(pairs), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(Y, d_Y, sizeof(Y), hipMemcpyDeviceToHost));
    for (int i = 0; i < 64 * 16; ++i) total += pairs[i];
    if (status[0] || status[1] || status[2]) { fprintf(stderr, "insert status %d %d %d\n", status[0], status[1], status[2]); return 4; }
    if ((int)total != expected_pairs) { fprintf(stderr, "kernel map holds %llu pairs, expected %d\n", total, expected_pairs); return 5; }
    if (nbr[13][0] != 0 || nbr[14][0] != 1 || nbr[16][0] != 2 || nbr[12][1] != 0 || nbr[10][2] != 0 || nbr[0][0] != -1) {
        fprintf(stderr, "kernel map entries differ\n");
        return 6;
    }
    double worst = 0.0;
    for (int r = 0; r < N; ++r) for (int o = 0; o < COUT; ++o) {
        const double e = fabs((double)Y[r][o] - (double)ref[r][o]);
        if (e > worst) worst = e;
    }
    printf("c caller: %d pairs, max |Y - ref| = %.3g\n", expected_pairs, worst);
    /* the error path is part of the contract: a bad argument returns AGB_EINVAL and leaves a message */
    if (agb_spconv_fwd(d_X, 3, d_W, d_nbr, N, 0, d_b, d_Y, COUT, N, K3, CIN, COUT, NULL) != AGB_EINVAL ||
        strlen(agb_last_error()) == 0) { fprintf(stderr, "error path\n"); return 7; }
    if (!(worst < 1e-5)) return 1;

    /* ---- Linear -> BatchNorm (training) -> ReLU through the fused entry points: the dense product leaves the partial
     * statistics of its output (agb_dense_fwd_bn), agb_bn_stats_fold turns them into mean / rstd, agb_bn_act_fwd applies */
    enum { M = 300, DI = 16, DO = 32 };
    static float A[M][DI], V[DI][DO], Z[M][DO], Zr[M][DO], Yb[M][DO], gamma[DO], beta[DO], mean_h[DO], rstd_h[DO];
    for (int r = 0; r < M; ++r) for (int c = 0; c < DI; ++c) A[r][c] = (float)(((r * 13 + c * 7) % 29) - 14) * 0.0625f;
    for (int c = 0; c < DI; ++c) for (int o = 0; o < DO; ++o) V[c][o] = (float)(((c * 5 + o * 3) % 17) - 8) * 0.03125f;
    for (int o = 0; o < DO; ++o) { gamma[o] = 1.f + 0.01f * (float)o; beta[o] = 0.1f * (float)(o % 3); }
    const int chunks = agb_dense_bn_chunks(M, DI, DO);
    if (chunks < 1) { fprintf(stderr, "agb_dense_bn_chunks(%d, %d, %d) = %d\n", M, DI, DO, chunks); return 8; }
    float *d_A, *d_V, *d_Z, *d_part, *d_mean, *d_rstd, *d_g, *d_be, *d_Yb;
    CHECK_HIP(hipMalloc((void**)&d_A, sizeof(A)));
    CHECK_HIP(hipMalloc((void**)&d_V, sizeof(V)));
    CHECK_HIP(hipMalloc((void**)&d_Z, sizeof(Z)));
    CHECK_HIP(hipMalloc((void**)&d_Yb, sizeof(Yb)));
    CHECK_HIP(hipMalloc((void**)&d_part, sizeof(float) * chunks * 3 * DO));
    CHECK_HIP(hipMalloc((void**)&d_mean, sizeof(float) * DO));
    CHECK_HIP(hipMalloc((void**)&d_rstd, sizeof(float) * DO));
    CHECK_HIP(hipMalloc((void**)&d_g, sizeof(gamma)));
    CHECK_HIP(hipMalloc((void**)&d_be, sizeof(beta)));
    CHECK_HIP(hipMemcpy(d_A, A, sizeof(A), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_V, V, sizeof(V), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_g, gamma, sizeof(gamma), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_be, beta, sizeof(beta), hipMemcpyHostToDevice));
    CHECK_AGB(agb_dense_fwd_bn(d_A, DI, d_V, NULL, d_Z, DO, M, DI, DO, d_part, NULL));
    CHECK_AGB(agb_bn_stats_fold(d_part, chunks, DO, 1e-5f, 0.f, d_mean, d_rstd, NULL, NULL, NULL, NULL));
    CHECK_AGB(agb_bn_act_fwd(d_Z, DO, M, DO, d_mean, d_rstd, d_g, d_be, 1 /* ReLU */, d_Yb, DO, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(Z, d_Z, sizeof(Z), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(Yb, d_Yb, sizeof(Yb), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(mean_h, d_mean, sizeof(mean_h), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(rstd_h, d_rstd, sizeof(rstd_h), hipMemcpyDeviceToHost));
    double worst2 = 0.0;
    for (int o = 0; o < DO; ++o) {
        double mu = 0.0, var = 0.0;
        for (int r = 0; r < M; ++r) {
            double z = 0.0;
            for (int c = 0; c < DI; ++c) z += (double)A[r][c] * (double)V[c][o];
            Zr[r][o] = (float)z;
            mu += z;
        }
        mu /= M;
        for (int r = 0; r < M; ++r) var += ((double)Zr[r][o] - mu) * ((double)Zr[r][o] - mu);
        var /= M;
        const double rs = 1.0 / sqrt(var + 1e-5);
        if (fabs(mean_h[o] - mu) > worst2) worst2 = fabs(mean_h[o] - mu);
        if (fabs(rstd_h[o] - rs) / rs > worst2) worst2 = fabs(rstd_h[o] - rs) / rs;
        for (int r = 0; r < M; ++r) {
            double y = ((double)Zr[r][o] - mu) * rs * gamma[o] + beta[o];
            if (y < 0.0) y = 0.0;
            if (fabs((double)Z[r][o] - (double)Zr[r][o]) > worst2) worst2 = fabs((double)Z[r][o] - (double)Zr[r][o]);
            if (fabs((double)Yb[r][o] - y) > worst2) worst2 = fabs((double)Yb[r][o] - y);
        }
    }
    printf("c caller: Linear -> BatchNorm -> ReLU (statistics from the product's epilogue, %d row tiles): max error %.3g\n",
           chunks, worst2);
    if (!(worst2 < 2e-5)) return 9;

    /* ---- work-balanced tiles: a 27-offset map over 20 000 rows of 64 channels whose pair density varies along the rows
     * (offset k of row i present iff (i + 3k) % 7 < 2 + (i >> 12): 29 % at the start, 86 % at the end; neighbour = a nearby
     * row), the pair-compacted kernel with the fixed interleave and with the table of agb_spconv_balance_tiles: same bits. */
    enum { TN = 20000, TC = 64 };
    int32_t geo[4];
    CHECK_AGB(agb_spconv_cmp_geometry(TN, TC, TC, TC, TC, 1, 128, -1, geo));
    if (geo[0] != 128 || geo[3] <= 0) { fprintf(stderr, "cmp geometry: R %d il %d\n", geo[0], geo[3]); return 10; }
    const int t_tiles = geo[2], t_bpt = geo[1] >> geo[3];
    int32_t* h_nbr = (int32_t*)malloc(sizeof(int32_t) * K3 * TN);
    float* h_x = (float*)malloc(sizeof(float) * TN * TC);
    float* h_w = (float*)malloc(sizeof(float) * K3 * TC * TC);
    float* h_y0 = (float*)malloc(sizeof(float) * TN * TC);
    float* h_y1 = (float*)malloc(sizeof(float) * TN * TC);
    int32_t* h_tab = (int32_t*)malloc(sizeof(int32_t) * t_tiles * t_bpt);
    long long t_pairs = 0;
    for (int k = 0; k < K3; ++k)
        for (int i = 0; i < TN; ++i) {
            const int present = (i + 3 * k) % 7 < 2 + (i >> 12);
            int j = i + (k - 13) * 5;
            if (j < 0) j += TN;
            if (j >= TN) j -= TN;
            h_nbr[k * TN + i] = present ? j : -1;
            t_pairs += present;
        }
    for (int i = 0; i < TN * TC; ++i) h_x[i] = (float)((i * 37 + 11) % 101 - 50) * 0.01f;
    for (int i = 0; i < K3 * TC * TC; ++i) h_w[i] = (float)((i * 53 + 7) % 89 - 44) * 0.002f;
    int32_t *d_tn, *d_tab;
    float *d_tx, *d_tw, *d_ty;
    void* d_ws;
    CHECK_HIP(hipMalloc((void**)&d_tn, sizeof(int32_t) * K3 * TN));
    CHECK_HIP(hipMalloc((void**)&d_tab, sizeof(int32_t) * t_tiles * t_bpt));
    CHECK_HIP(hipMalloc((void**)&d_tx, sizeof(float) * TN * TC));
    CHECK_HIP(hipMalloc((void**)&d_tw, sizeof(float) * K3 * TC * TC));
    CHECK_HIP(hipMalloc((void**)&d_ty, sizeof(float) * TN * TC));
    CHECK_HIP(hipMalloc(&d_ws, agb_spconv_balance_tiles_workspace_bytes(TN, K3, geo[3])));
    CHECK_HIP(hipMemcpy(d_tn, h_nbr, sizeof(int32_t) * K3 * TN, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_tx, h_x, sizeof(float) * TN * TC, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_tw, h_w, sizeof(float) * K3 * TC * TC, hipMemcpyHostToDevice));
    CHECK_AGB(agb_spconv_fwd_opt(d_tx, TC, d_tw, d_tn, TN, 0, NULL, d_ty, TC, TN, K3, TC, TC, NULL, NULL, NULL, 0, 1, NULL, 128, -1,
                                 NULL));
    CHECK_HIP(hipMemcpy(h_y0, d_ty, sizeof(float) * TN * TC, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemset(d_ty, 0, sizeof(float) * TN * TC));
    CHECK_AGB(agb_spconv_balance_tiles(d_tn, TN, TN, K3, geo[3], t_tiles, t_bpt, d_tab, d_ws, NULL));
    CHECK_AGB(agb_spconv_fwd_tiles(d_tx, TC, d_tw, d_tn, TN, 0, NULL, d_ty, TC, TN, K3, TC, TC, 1, NULL, 128, -1, d_tab, t_tiles,
                                   t_bpt, NULL));
    CHECK_HIP(hipMemcpy(h_y1, d_ty, sizeof(float) * TN * TC, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(h_tab, d_tab, sizeof(int32_t) * t_tiles * t_bpt, hipMemcpyDeviceToHost));
    if (memcmp(h_y0, h_y1, sizeof(float) * TN * TC) != 0) { fprintf(stderr, "balanced tiles changed the sums\n"); return 11; }
    /* a few rows against the host, and the spread of the pairs per tile with and without the table */
    double worst3 = 0.0;
    for (int i = 0; i < TN; i += 1999)
        for (int o = 0; o < TC; o += 7) {
            double acc = 0.0;
            for (int k = 0; k < K3; ++k) {
                const int j = h_nbr[k * TN + i];
                if (j >= 0) for (int c = 0; c < TC; ++c) acc += (double)h_x[j * TC + c] * (double)h_w[(k * TC + c) * TC + o];
            }
            if (fabs(acc - (double)h_y1[i * TC + o]) > worst3) worst3 = fabs(acc - (double)h_y1[i * TC + o]);
        }
    const int bs = 1 << geo[3], nblk = (TN + bs - 1) / bs;
    double mean_w = (double)t_pairs / t_tiles, max_bal = 0.0, max_fix = 0.0;
    int seen = 0;
    for (int t = 0; t < t_tiles; ++t) {
        double wb = 0.0, wf = 0.0;
        for (int j = 0; j < t_bpt; ++j) {
            const int bb = h_tab[t * t_bpt + j], bf = j * t_tiles + t;
            for (int which = 0; which < 2; ++which) {
                const int blk = which ? bf : bb;
                if (blk < 0 || blk >= nblk) continue;
                if (!which) ++seen;
                for (int r = blk * bs; r < (blk + 1) * bs && r < TN; ++r)
                    for (int k = 0; k < K3; ++k) { if (h_nbr[k * TN + r] >= 0) { if (which) wf += 1.0; else wb += 1.0; } }
            }
        }
        if (wb > max_bal) max_bal = wb;
        if (wf > max_fix) max_fix = wf;
    }
    printf("c caller: balanced tiles (%d tiles x %d blocks of %d rows): identical sums, max error vs host %.3g, pairs per tile "
           "max/mean %.3f (fixed interleave %.3f)\n", t_tiles, t_bpt, bs, worst3, max_bal / mean_w, max_fix / mean_w);
    if (seen != nblk) { fprintf(stderr, "table holds %d of %d blocks\n", seen, nblk); return 12; }
    return (worst3 < 1e-3 && max_bal / mean_w < 1.05) ? 0 : 13;
}
