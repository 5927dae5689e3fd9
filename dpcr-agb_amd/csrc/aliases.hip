// aliases.hip — the entry-point names SURVEY.md section 8(b) gave the C ABI, for a binder written from the survey:
//   agb_hash_build     = agb_coords_insert (level-0 coordinate hash; the dense-grid form is agb_grid_insert)
//   agb_kpconv_fwd/bwd = the whole rigid KPConv layer (modules/KPConv/blocks.py:264-400): neighbourhood gather + kernel-weight
//                        contraction (forward), and its three gradients (backward), as ONE call each over the entry points
//                        agb_kpconv_gather_* / agb_spconv_fwd_ex / agb_spconv_bwd_weight the Python binding drives one by one
//   (agb_spconv_bwd_data lives in spconv.hip: the data gradient with an optional addend)
// Host code only.
#include "agb_common.h"
#include "../../include/agb_hip.h"

extern "C" {

int agb_hash_build(const int32_t* coords, int n, const int32_t* n_dev, uint64_t* keys, int32_t* vals, int cap,
                   int32_t* slot_of_row, int32_t* status, void* stream) {
    return agb_coords_insert(coords, n, n_dev, keys, vals, cap, slot_of_row, status, stream);
}

// y [N][ldy] = wf [N][K*Cin] @ W [K*Cin][Cout],  wf[n,k,:] = sum_h infl(n,h,k) x[idx[n,h],:]  (blocks.py:304-400).
// wf float [N][K*Cin]: out (the backward pass takes it).  Cin, Cout multiples of 4, K*Cin >= 12.
int agb_kpconv_fwd(const float* q, const float* s, const int32_t* idx, int H, int Ns, const float* x, int ldx, const float* kp,
                   int K, float extent, const float* W, float* wf, float* y, int ldy, int N, int Cin, int Cout, void* stream) {
    AGB_CHECK_ARG(W && wf && y && K >= 1 && Cin >= 4 && Cin % 4 == 0 && Cout >= 4 && Cout % 4 == 0 && K * Cin >= 12,
                  "agb_kpconv_fwd: K %d, Cin %d, Cout %d (multiples of 4, K * Cin >= 12)", K, Cin, Cout);
    int rc = agb_kpconv_gather_fwd(q, s, idx, H, Ns, x, ldx, kp, K, extent, wf, N, Cin, stream);
    if (rc) return rc;
    return agb_spconv_fwd_ex(wf, K * Cin, W, nullptr, 0, 0, nullptr, y, ldy, N, 1, K * Cin, Cout, nullptr, nullptr, nullptr, 0, 1,
                             nullptr, stream);
}

// workspace of agb_kpconv_bwd in bytes: dwf [N][K*Cin] and W^T [Cout][K*Cin]
size_t agb_kpconv_bwd_workspace_bytes(int N, int K, int Cin, int Cout) {
    if (N < 0 || K < 1 || Cin < 1 || Cout < 1) return 0;
    const size_t a = ((size_t)(N > 0 ? N : 1) * K * Cin * sizeof(float) + 255) / 256 * 256;
    return a + (size_t)K * Cin * Cout * sizeof(float);
}

// dx [Ns][ldx] (zero-filled by the caller: fp32 atomics, as agb_kpconv_gather_bwd) and dW [K*Cin][Cout] (zero-filled by the
// caller, accumulated into) from dy [N][lddy]; wf: what agb_kpconv_fwd left.  Either of dx / dW may be NULL.
int agb_kpconv_bwd(const float* q, const float* s, const int32_t* idx, int H, int Ns, const float* wf, const float* dy, int lddy,
                   const float* kp, int K, float extent, const float* W, float* dx, int ldx, float* dW, int N, int Cin, int Cout,
                   void* workspace, size_t workspace_bytes, void* stream) {
    AGB_CHECK_ARG(workspace != nullptr && workspace_bytes >= agb_kpconv_bwd_workspace_bytes(N, K, Cin, Cout),
                  "agb_kpconv_bwd: workspace of %zu bytes, %zu needed", workspace_bytes,
                  agb_kpconv_bwd_workspace_bytes(N, K, Cin, Cout));
    AGB_CHECK_ARG(K >= 1 && Cin >= 4 && Cin % 4 == 0 && Cout >= 12 && Cout % 4 == 0 && K * Cin >= 12,
                  "agb_kpconv_bwd: K %d, Cin %d, Cout %d", K, Cin, Cout);
    const int KC = K * Cin;
    float* dwf = (float*)workspace;
    float* Wt = (float*)((char*)workspace + ((size_t)(N > 0 ? N : 1) * KC * sizeof(float) + 255) / 256 * 256);
    int rc;
    if (dW) {
        rc = agb_spconv_bwd_weight(wf, KC, dy, lddy, nullptr, 0, dW, N, 1, KC, Cout, stream);
        if (rc) return rc;
    }
    if (dx) {
        rc = agb_spconv_weight_transpose(W, Wt, 1, KC, Cout, stream);
        if (rc) return rc;
        rc = agb_spconv_fwd_ex(dy, lddy, Wt, nullptr, 0, 0, nullptr, dwf, KC, N, 1, Cout, KC, nullptr, nullptr, nullptr, 0, 1,
                               nullptr, stream);
        if (rc) return rc;
        rc = agb_kpconv_gather_bwd(q, s, idx, H, Ns, dwf, kp, K, extent, dx, ldx, N, Cin, stream);
        if (rc) return rc;
    }
    return AGB_OK;
}

}  // extern "C"
