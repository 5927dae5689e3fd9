// net.hip — ONE C entry point per network block (host orchestration + a few glue kernels).
//
// The reference's residual block is a fixed operator sequence (torch_points3d/modules/MinkowskiEngine/senet_block.py:80-96
// SEBasicBlock.forward, resnet_block.py:62-73 BasicBlock, SENet.py:47-53 the stem: conv 7^3 -> norm -> act -> max pool 3^3 s2;
// SENet.py:113-118 the forward loop over the stages).  Driving that sequence launch by launch from Python costs the enqueuing
// thread 15-25 us per launch (~195 launches per MSENet14 step).  Here the whole block — forward, or backward — is one call:
// the caller hands over a table of 64-bit fields (pointers, sizes, options; names from agb_net_fields()) and two arenas
// (`saved`: what the forward pass keeps for the backward pass; `scratch`: temporaries of the call), the library enqueues
// the same kernels the per-operator entry points of include/agb_hip.h would, in the same order, with the same launch
// geometry: results are bit-identical to the operator-by-operator path (tests/test_fused_blocks_gpu.py).
// Nothing is allocated or synchronised here; both arenas are carved by code that also answers the *_bytes queries.
#include "agb_common.h"
#include "../../include/agb_hip.h"

// ------------------------------------------------------------------------------------------------------------ fields
// One convolution (+ the BatchNorm behind it) of a block.
#define AGB_NET_CONV(X, p)                                                                                             \
    X(p##_w) X(p##_b) X(p##_K3) X(p##_cin) X(p##_cout)                               /* kernel [K3][cin][cout], bias or 0 */ \
    X(p##_g) X(p##_be) X(p##_rm) X(p##_rv) X(p##_nbt) X(p##_eps) X(p##_mom)          /* BatchNorm (eps, mom: double bits) */ \
    X(p##_nbr) X(p##_nbr_ld) X(p##_nbrT) X(p##_nbrT_ld)                              /* forward / transposed kernel map */ \
    X(p##_perm) X(p##_tile_cls) X(p##_cls_tab) X(p##_n_tiles)                        /* class partition (strided dgrad) */ \
    X(p##_tf) X(p##_tf_t) X(p##_tf_b) X(p##_tb) X(p##_tb_t) X(p##_tb_b)              /* balanced tile tables fwd / dgrad */ \
    X(p##_wt)                                                                        /* cached W^T [K3][cout][cin] or 0 */ \
    X(p##_dw) X(p##_db) X(p##_dg) X(p##_dbe)                                         /* gradients out */

#define AGB_NET_FIELDS(X)                                                                                              \
    X(x) X(ldx) X(n_in) X(n_out) X(B) X(coords) X(ptr) X(y) X(ldy)                                                     \
    X(dy) X(lddy) X(dx) X(lddx) X(need_dx) X(gzero) X(gzero_bytes)                                                     \
    X(training) X(act) X(stride) X(has_down)                                                                           \
    X(cmp_mode) X(cmp_il) X(dw_variant) X(det) X(persistent)                                                           \
    AGB_NET_CONV(X, c1) AGB_NET_CONV(X, c2) AGB_NET_CONV(X, cd)                                                        \
    X(se_act) X(se_H) X(se_w1) X(se_b1) X(se_w2) X(se_b2) X(keep)                                                      \
    X(d_se_w1) X(d_se_b1) X(d_se_w2) X(d_se_b2)                                                                        \
    /* stem only */                                                                                                    \
    X(feat) X(ldf) X(fdim) X(grid) X(desc) X(K) X(pool_nbr) X(pool_nbr_ld) X(pool_nbrT) X(pool_nbrT_ld) X(pool_K3)     \
    X(n_pool)

enum {
#define X(n) F_##n,
    AGB_NET_FIELDS(X)
#undef X
    F_COUNT
};
static const char k_field_names[] =
#define X(n) #n ","
    AGB_NET_FIELDS(X)
#undef X
    ;

namespace {
template <typename T> inline T* P(const int64_t* f, int i) { return reinterpret_cast<T*>((intptr_t)f[i]); }
inline int I(const int64_t* f, int i) { return (int)f[i]; }
inline float D(const int64_t* f, int i) {
    double d;
    memcpy(&d, &f[i], sizeof(d));
    return (float)d;
}

struct Arena {
    char* base;
    size_t off = 0, cap;
    Arena(void* p, size_t bytes) : base((char*)p), cap(bytes) {}
    template <typename T> T* take(size_t count) {
        T* p = base ? (T*)(base + off) : nullptr;
        off += (count * sizeof(T) + 255) / 256 * 256;
        return p;
    }
    bool fits() const { return base == nullptr || off <= cap; }
};
inline size_t nz(long long n) { return (size_t)(n > 0 ? n : 1); }

struct Conv {
    const float *w, *b, *g, *be;
    float *rm, *rv;
    long long* nbt;
    float eps, mom;
    int K3, cin, cout;
    const int32_t *nbr, *nbrT, *perm, *tile_cls, *cls_tab, *tf, *tb;
    long long nbr_ld, nbrT_ld;
    int n_tiles, tf_t, tf_b, tb_t, tb_b;
    const float* wt;
    float *dw, *db, *dg, *dbe;
};
Conv load_conv(const int64_t* f, int base) {
    Conv c;
    const int o = base - F_c1_w;
    c.w = P<const float>(f, F_c1_w + o); c.b = P<const float>(f, F_c1_b + o);
    c.K3 = I(f, F_c1_K3 + o); c.cin = I(f, F_c1_cin + o); c.cout = I(f, F_c1_cout + o);
    c.g = P<const float>(f, F_c1_g + o); c.be = P<const float>(f, F_c1_be + o);
    c.rm = P<float>(f, F_c1_rm + o); c.rv = P<float>(f, F_c1_rv + o); c.nbt = P<long long>(f, F_c1_nbt + o);
    c.eps = D(f, F_c1_eps + o); c.mom = D(f, F_c1_mom + o);
    c.nbr = P<const int32_t>(f, F_c1_nbr + o); c.nbr_ld = f[F_c1_nbr_ld + o];
    c.nbrT = P<const int32_t>(f, F_c1_nbrT + o); c.nbrT_ld = f[F_c1_nbrT_ld + o];
    c.perm = P<const int32_t>(f, F_c1_perm + o); c.tile_cls = P<const int32_t>(f, F_c1_tile_cls + o);
    c.cls_tab = P<const int32_t>(f, F_c1_cls_tab + o); c.n_tiles = I(f, F_c1_n_tiles + o);
    c.tf = P<const int32_t>(f, F_c1_tf + o); c.tf_t = I(f, F_c1_tf_t + o); c.tf_b = I(f, F_c1_tf_b + o);
    c.tb = P<const int32_t>(f, F_c1_tb + o); c.tb_t = I(f, F_c1_tb_t + o); c.tb_b = I(f, F_c1_tb_b + o);
    c.wt = P<const float>(f, F_c1_wt + o);
    c.dw = P<float>(f, F_c1_dw + o); c.db = P<float>(f, F_c1_db + o);
    c.dg = P<float>(f, F_c1_dg + o); c.dbe = P<float>(f, F_c1_dbe + o);
    return c;
}

struct Opts { int cmp_mode, cmp_il, dw_variant, det, persistent, training; };
Opts load_opts(const int64_t* f) {
    return Opts{I(f, F_cmp_mode), I(f, F_cmp_il), I(f, F_dw_variant), I(f, F_det), I(f, F_persistent), I(f, F_training)};
}

#define NET_TRY(call)            \
    do {                         \
        int rc__ = (call);       \
        if (rc__) return rc__;   \
    } while (0)

// ---- the three products of a convolution, as sparse_ops.py drives them (same entry points, same split rules) ----------
// split of the reduction for the forward / stride-1 data gradient: sparse_ops.spconv_forward_raw
inline int fwd_split(const Opts& o, int n_out, int K3, int cin, int cout) {
    return agb_spconv_split_hint_opt(n_out, K3, cin, cout, o.cmp_mode);
}
// class-partitioned strided data gradient: few 64-row tiles with a long reduction split four ways (sparse_ops.py)
inline int plan_split(int n_rows, int cin, int cout) {
    return ((long long)(n_rows / 64 + 1) * ((cout + 63) / 64) < 1100 && cin >= 256 && cin % 256 == 0) ? 4 : 1;
}
size_t conv_fwd_scratch(const Opts& o, const Conv& c, int n_out) {
    const int sp = fwd_split(o, n_out, c.K3, c.cin, c.cout);
    return sp > 1 ? (size_t)sp * nz(n_out) * c.cout : 0;
}
int conv_fwd(const Opts& o, const Conv& c, const float* x, int ldx, int n_out, float* y, int ldy, float* partial, void* st) {
    const int sp = fwd_split(o, n_out, c.K3, c.cin, c.cout);
    if (c.tf)
        return agb_spconv_fwd_tiles(x, ldx, c.w, c.nbr, c.nbr_ld, 0, c.b, y, ldy, n_out, c.K3, c.cin, c.cout, sp,
                                    sp > 1 ? partial : nullptr, o.cmp_mode, o.cmp_il, c.tf, c.tf_t, c.tf_b, st);
    return agb_spconv_fwd_opt(x, ldx, c.w, c.nbr, c.nbr_ld, 0, c.b, y, ldy, n_out, c.K3, c.cin, c.cout, nullptr, nullptr,
                              nullptr, 0, sp, sp > 1 ? partial : nullptr, o.cmp_mode, o.cmp_il, st);
}
// dX[n_in, cin] = sum_k dY[map] W[k]^T
int dgrad_split(const Opts& o, const Conv& c, int n_in) {
    if (c.nbrT) return c.perm ? plan_split(n_in, c.cout, c.cin) : fwd_split(o, n_in, c.K3, c.cout, c.cin);
    return fwd_split(o, n_in, c.K3, c.cout, c.cin);
}
size_t conv_dgrad_scratch(const Opts& o, const Conv& c, int n_in) {
    const int sp = dgrad_split(o, c, n_in);
    return sp > 1 ? (size_t)sp * nz(n_in) * c.cin : 0;
}
int conv_dgrad(const Opts& o, const Conv& c, const float* wt, const float* dy, int lddy, int n_in, float* dx, int lddx,
               float* partial, const float* addend, void* st) {
    const int sp = dgrad_split(o, c, n_in);
    float* part = sp > 1 ? partial : nullptr;
    const int lda = addend ? lddx : 0;
    if (c.nbrT)      // strided layer: transposed map, class partition of the input rows
        return agb_spconv_fwd_opt_add(dy, lddy, wt, c.nbrT, c.nbrT_ld, 0, nullptr, dx, lddx, n_in, c.K3, c.cout, c.cin, c.perm,
                                      c.perm ? c.tile_cls : nullptr, c.perm ? c.cls_tab : nullptr, c.perm ? c.n_tiles : 0, sp,
                                      part, o.cmp_mode, o.cmp_il, addend, lda, st);
    if (c.tb)
        return agb_spconv_fwd_tiles_add(dy, lddy, wt, c.nbr, c.nbr_ld, 1, nullptr, dx, lddx, n_in, c.K3, c.cout, c.cin, sp, part,
                                        o.cmp_mode, o.cmp_il, c.tb, c.tb_t, c.tb_b, addend, lda, st);
    return agb_spconv_fwd_opt_add(dy, lddy, wt, c.nbr, c.nbr_ld, 1, nullptr, dx, lddx, n_in, c.K3, c.cout, c.cin, nullptr,
                                  nullptr, nullptr, 0, sp, part, o.cmp_mode, o.cmp_il, addend, lda, st);
}
// workspace of the weight gradient: sparse_ops.weight_grad_raw (fixed-order sums on request, or the persistent kernel)
size_t conv_wgrad_bytes(const Opts& o, const Conv& c, int n_out, int ldx, int lddy) {
    const bool det = o.det && o.dw_variant != 1;
    const bool persistent = o.dw_variant == 3 ||
        (o.dw_variant == 0 && o.persistent && agb_spconv_bwd_weight_persistent(n_out, c.K3, c.cin, c.cout, ldx, lddy) == 1);
    return (det || persistent) ? agb_spconv_bwd_weight_workspace_bytes(n_out, c.K3, c.cin, c.cout, 0, 0) : 0;
}
int conv_wgrad(const Opts& o, const Conv& c, const float* x, int ldx, const float* dy, int lddy, int n_out, void* ws,
               size_t ws_bytes, void* st) {
    return agb_spconv_bwd_weight_ws(x, ldx, dy, lddy, c.nbr, c.nbr_ld, c.dw, n_out, c.K3, c.cin, c.cout, 0, o.dw_variant,
                                    ws_bytes ? ws : nullptr, ws_bytes, st);
}

// ---- glue kernels -----------------------------------------------------------------------------------------------------
// features [n][fdim <= 4] -> rows 4 floats wide, zero padded (the stem's input rows; sparse_ops: F.pad(feats, (0, 1)))
__global__ __launch_bounds__(256) void k_net_pad4(const float* __restrict__ X, int ldx, int fdim, int n, float4* __restrict__ Y) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const float* p = X + (long long)r * ldx;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    v.x = p[0];
    if (fdim > 1) v.y = p[1];
    if (fdim > 2) v.z = p[2];
    if (fdim > 3) v.w = p[3];
    Y[r] = v;
}
// dW [K3][3][C] <- dWp [K3][4][C] (the padded channel's gradient is dropped; sparse_ops: dwp[:, :3, :].contiguous())
__global__ __launch_bounds__(256) void k_net_take3(const float* __restrict__ dWp, int K3, int C, float* __restrict__ dW) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= K3 * 3 * C) return;
    const int k = e / (3 * C), rem = e - k * 3 * C;
    dW[e] = dWp[(long long)k * 4 * C + rem];
}
// ---- BatchNorm + activation of a convolution's output ----------------------------------------------------------------
int bn_fwd(const Opts& o, const Conv& c, const float* z, int n, int act, float* stats, float* part, float* out, void* st) {
    NET_TRY(agb_bn_stats_tracked(z, c.cout, n, c.cout, c.eps, c.mom, o.training, part, stats, stats + c.cout, c.rm, c.rv, c.nbt,
                                 st));
    return agb_bn_act_fwd(z, c.cout, n, c.cout, stats, stats + c.cout, c.g, c.be, act, out, c.cout, st);
}
// dz (and the convolution's bias gradient in closed form) from da
int bn_bwd(const Opts& o, const Conv& c, const float* z, const float* da, int n, int act, const float* stats, float* part,
           float* dz, float* colsum_scratch, void* st) {
    return agb_bn_act_bwd_colsum(z, c.cout, da, c.cout, n, c.cout, stats, stats + c.cout, c.g, c.be, act, o.training, part, dz,
                                 c.cout, c.dg, c.dbe, c.db ? c.db : colsum_scratch, st);
}

// =============================================================================================================== stem
// conv K^3 (3 -> 64 channels, stride 1, neighbours probed in the level's dense grid) -> BatchNorm -> act -> max pool
struct StemSaved { float *x4, *z, *stats; uint8_t* arg; };
struct StemFwdScratch { float *part, *a; };
struct StemBwdScratch { float *da, *part, *dz, *dwp, *colsum; void* ws; size_t ws_bytes; };

void stem_saved(Arena& A, int n, int n_pool, int C, StemSaved* s) {
    s->x4 = A.take<float>(4 * nz(n));
    s->z = A.take<float>(nz(n) * C);
    s->stats = A.take<float>(2 * (size_t)C);
    s->arg = A.take<uint8_t>(nz(n_pool) * C);
}
void stem_fwd_scratch(Arena& A, int n, int C, StemFwdScratch* s) {
    s->part = A.take<float>((size_t)agb_bn_chunks(n) * 3 * C);
    s->a = A.take<float>(nz(n) * C);
}
void stem_bwd_scratch(Arena& A, int n, int C, int K, StemBwdScratch* s) {
    s->da = A.take<float>(nz(n) * C);
    s->part = A.take<float>((size_t)agb_bn_chunks(n) * 2 * C);
    s->dz = A.take<float>(nz(n) * C);
    s->dwp = A.take<float>((size_t)K * K * K * 4 * C);
    s->colsum = A.take<float>((size_t)C);
    s->ws_bytes = agb_stem_bwd_weight_grid_workspace_bytes(n, K);
    s->ws = A.take<char>(s->ws_bytes > 0 ? s->ws_bytes : 1);
}

// ========================================================================================================== SE block
struct BlockSaved { float *z1, *a1, *z2, *zd, *r, *st1, *st2, *std_, *zbar, *p, *h_pre, *s; };
void block_saved(Arena& A, const int64_t* f, BlockSaved* s) {
    const int n = I(f, F_n_out), C = I(f, F_c1_cout), B = I(f, F_B), H = I(f, F_se_H);
    const size_t nc = nz(n) * C;
    s->z1 = A.take<float>(nc); s->a1 = A.take<float>(nc); s->z2 = A.take<float>(nc);
    s->zd = s->r = nullptr;
    if (I(f, F_has_down)) { s->zd = A.take<float>(nc); s->r = A.take<float>(nc); }
    s->st1 = A.take<float>(2 * (size_t)C); s->st2 = A.take<float>(2 * (size_t)C); s->std_ = A.take<float>(2 * (size_t)C);
    s->zbar = A.take<float>((size_t)B * C); s->p = A.take<float>((size_t)B * C);
    s->h_pre = A.take<float>((size_t)B * H); s->s = A.take<float>((size_t)B * C);
}
struct BlockFwdScratch { float *partial, *part; };
void block_fwd_scratch(Arena& A, const int64_t* f, BlockFwdScratch* s) {
    const Opts o = load_opts(f);
    const Conv c1 = load_conv(f, F_c1_w), c2 = load_conv(f, F_c2_w), cd = load_conv(f, F_cd_w);
    const int n = I(f, F_n_out), C = c1.cout, B = I(f, F_B);
    size_t pmax = conv_fwd_scratch(o, c1, n);
    const size_t p2 = conv_fwd_scratch(o, c2, n);
    if (p2 > pmax) pmax = p2;
    if (I(f, F_has_down)) {
        const size_t pd = conv_fwd_scratch(o, cd, n);
        if (pd > pmax) pmax = pd;
    }
    s->partial = A.take<float>(pmax > 0 ? pmax : 1);
    size_t chunks = (size_t)agb_bn_chunks(n);
    const size_t tc = (size_t)agb_se_tail_chunks(n, C, B);
    if (tc > chunks) chunks = tc;
    s->part = A.take<float>(chunks * 3 * C);
}
struct BlockBwdScratch {
    float *S, *spart, *ds, *dz2se, *dh, *dp, *dte, *dz2, *dr, *da1, *dz1, *dzd, *dxb, *partial, *part, *wt, *colsum;
    void* ws;
    size_t ws_bytes;
};
void block_bwd_scratch(Arena& A, const int64_t* f, BlockBwdScratch* s) {
    const Opts o = load_opts(f);
    const Conv c1 = load_conv(f, F_c1_w), c2 = load_conv(f, F_c2_w), cd = load_conv(f, F_cd_w);
    const int n = I(f, F_n_out), n_in = I(f, F_n_in), C = c1.cout, Cin = c1.cin, B = I(f, F_B), H = I(f, F_se_H);
    const bool down = I(f, F_has_down) != 0;
    const size_t nc = nz(n) * C;
    s->S = A.take<float>(2 * (size_t)B * C);
    s->spart = A.take<float>((size_t)agb_se_tail_chunks(n, C, B) * 2 * C);
    s->ds = A.take<float>((size_t)B * C);
    s->dz2se = A.take<float>((size_t)((C + 511) / 512) * B * C);
    s->dh = A.take<float>((size_t)B * H);
    s->dp = A.take<float>((size_t)B * C);
    s->dte = A.take<float>((size_t)B * C);
    s->dz2 = A.take<float>(nc);
    s->dr = A.take<float>(nc);          // gradient of the residual branch (level of the block's output)
    s->da1 = A.take<float>(nc);
    s->dz1 = A.take<float>(nc);
    s->dzd = down ? A.take<float>(nc) : nullptr;
    // the downsample branch's data gradient (level of the block's INPUT); conv1's data gradient takes it — or, without a
    // downsample, the tail's dr — as the addend of its final store: dx leaves in one pass
    s->dxb = down ? A.take<float>(nz(n_in) * Cin) : nullptr;
    size_t pmax = conv_dgrad_scratch(o, c2, n), q = conv_dgrad_scratch(o, c1, n_in);
    if (q > pmax) pmax = q;
    if (down) { q = conv_dgrad_scratch(o, cd, n_in); if (q > pmax) pmax = q; }
    s->partial = A.take<float>(pmax > 0 ? pmax : 1);
    s->part = A.take<float>((size_t)agb_bn_chunks(n) * 2 * C);
    size_t wmax = (size_t)c1.K3 * c1.cin * c1.cout;
    q = (size_t)c2.K3 * c2.cin * c2.cout; if (q > wmax) wmax = q;
    if (down) { q = (size_t)cd.K3 * cd.cin * cd.cout; if (q > wmax) wmax = q; }
    s->wt = A.take<float>(wmax);
    s->colsum = A.take<float>((size_t)C);
    size_t wb = conv_wgrad_bytes(o, c2, n, C, C);
    q = conv_wgrad_bytes(o, c1, n, Cin, C); if (q > wb) wb = q;
    if (down) { q = conv_wgrad_bytes(o, cd, n, Cin, C); if (q > wb) wb = q; }
    s->ws_bytes = wb;
    s->ws = A.take<char>(wb > 0 ? wb : 1);
}

// W^T of one convolution for its data gradient: the step's cached copy when the caller keeps one, else made here (the
// launch also clears the weight-gradient buffer, as SparseConvFunction.backward does)
int conv_wt(const Conv& c, float* wt_scratch, const float** wt_out, bool cleared, void* st) {
    if (c.wt) {
        *wt_out = c.wt;
        if (cleared) return AGB_OK;      // (the block's gradient buffer was cleared in one go: `gzero`)
        return hipMemsetAsync(c.dw, 0, sizeof(float) * (size_t)c.K3 * c.cin * c.cout, (hipStream_t)st) == hipSuccess ? AGB_OK
                                                                                                                      : AGB_ELAUNCH;
    }
    *wt_out = wt_scratch;
    return agb_spconv_weight_transpose_z(c.w, wt_scratch, c.dw, c.K3, c.cin, c.cout, st);
}
}  // namespace

extern "C" {

const char* agb_net_fields(void) { return k_field_names; }
int agb_net_field_count(void) { return F_COUNT; }

// ------------------------------------------------------------------------------------------------------------- stem
size_t agb_net_stem_bytes(const int64_t* f, int which) {
    if (!f) return 0;
    Arena A(nullptr, 0);
    const int n = I(f, F_n_in), C = I(f, F_c1_cout);
    if (which == 0) { StemSaved s; stem_saved(A, n, I(f, F_n_pool), C, &s); }
    else if (which == 1) { StemFwdScratch s; stem_fwd_scratch(A, n, C, &s); }
    else { StemBwdScratch s; stem_bwd_scratch(A, n, C, I(f, F_K), &s); }
    return A.off;
}

int agb_net_stem_fwd(const int64_t* f, void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes, void* stream) {
    AGB_CHECK_ARG(f && saved && scratch, "agb_net_stem_fwd: table, saved and scratch arenas are required");
    const Opts o = load_opts(f);
    const Conv c = load_conv(f, F_c1_w);
    const int n = I(f, F_n_in), n_pool = I(f, F_n_pool), C = c.cout, K = I(f, F_K), fdim = I(f, F_fdim);
    AGB_CHECK_ARG(fdim >= 1 && fdim <= 3 && c.cin == 3 && C == 64 && c.K3 == K * K * K && n >= 1,
                  "agb_net_stem_fwd: takes 1..3 feature channels into 64 (fdim %d, Cin %d, Cout %d, K %d, n %d)", fdim, c.cin, C,
                  K, n);
    Arena SA(saved, saved_bytes), TA(scratch, scratch_bytes);
    StemSaved S; StemFwdScratch T;
    stem_saved(SA, n, n_pool, C, &S);
    stem_fwd_scratch(TA, n, C, &T);
    AGB_CHECK_ARG(SA.fits() && TA.fits(), "agb_net_stem_fwd: arenas too small (%zu / %zu saved, %zu / %zu scratch)", saved_bytes,
                  SA.off, scratch_bytes, TA.off);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_net_pad4, dim3(agb_cdiv(n, 256)), dim3(256), 0, s, P<const float>(f, F_feat), I(f, F_ldf), fdim, n,
                       (float4*)S.x4);
    NET_TRY(agb_spconv_fwd3_grid_lp(S.x4, 4, c.w, P<const int32_t>(f, F_coords), P<const int32_t>(f, F_grid),
                                    P<const int32_t>(f, F_desc), K, c.b, S.z, C, n, C, nullptr, 0, 0, stream));
    NET_TRY(bn_fwd(o, c, S.z, n, I(f, F_act), S.stats, T.part, T.a, stream));
    NET_TRY(agb_maxpool_fwd_k(T.a, C, P<const int32_t>(f, F_pool_nbr), f[F_pool_nbr_ld], P<float>(f, F_y), I(f, F_ldy), S.arg,
                              n_pool, I(f, F_pool_K3), C, stream));
    AGB_CHECK_LAUNCH("agb_net_stem_fwd");
    return AGB_OK;
}

int agb_net_stem_bwd(const int64_t* f, void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes, void* stream) {
    AGB_CHECK_ARG(f && saved && scratch, "agb_net_stem_bwd: table, saved and scratch arenas are required");
    const Opts o = load_opts(f);
    const Conv c = load_conv(f, F_c1_w);
    const int n = I(f, F_n_in), n_pool = I(f, F_n_pool), C = c.cout, K = I(f, F_K);
    Arena SA(saved, saved_bytes), TA(scratch, scratch_bytes);
    StemSaved S; StemBwdScratch T;
    stem_saved(SA, n, n_pool, C, &S);
    stem_bwd_scratch(TA, n, C, K, &T);
    AGB_CHECK_ARG(SA.fits() && TA.fits(), "agb_net_stem_bwd: arenas too small (%zu / %zu saved, %zu / %zu scratch)", saved_bytes,
                  SA.off, scratch_bytes, TA.off);
    hipStream_t s = (hipStream_t)stream;
    NET_TRY(agb_maxpool_bwd_k(P<const float>(f, F_dy), I(f, F_lddy), S.arg, P<const int32_t>(f, F_pool_nbrT), f[F_pool_nbrT_ld],
                              T.da, C, n, I(f, F_pool_K3), C, stream));
    NET_TRY(bn_bwd(o, c, S.z, T.da, n, I(f, F_act), S.stats, T.part, T.dz, T.colsum, stream));
    if (hipMemsetAsync(T.dwp, 0, sizeof(float) * (size_t)c.K3 * 4 * C, s) != hipSuccess) return AGB_ELAUNCH;
    NET_TRY(agb_stem_bwd_weight_grid(S.x4, 4, T.dz, C, P<const int32_t>(f, F_coords), P<const int32_t>(f, F_grid),
                                     P<const int32_t>(f, F_desc), K, T.dwp, n, C, T.ws, T.ws_bytes, stream));
    hipLaunchKernelGGL(k_net_take3, dim3(agb_cdiv((long long)c.K3 * 3 * C, 256)), dim3(256), 0, s, T.dwp, c.K3, C, c.dw);
    AGB_CHECK_LAUNCH("agb_net_stem_bwd");
    return AGB_OK;
}

// --------------------------------------------------------------------------------------------------------- SE block
size_t agb_net_block_bytes(const int64_t* f, int which) {
    if (!f) return 0;
    Arena A(nullptr, 0);
    if (which == 0) { BlockSaved s; block_saved(A, f, &s); }
    else if (which == 1) { BlockFwdScratch s; block_fwd_scratch(A, f, &s); }
    else { BlockBwdScratch s; block_bwd_scratch(A, f, &s); }
    return A.off;
}

static int block_check(const int64_t* f, const char* who) {
    const Conv c1 = load_conv(f, F_c1_w), c2 = load_conv(f, F_c2_w);
    const int C = c1.cout;
    AGB_CHECK_ARG(I(f, F_n_out) >= 1 && I(f, F_n_in) >= 1 && I(f, F_B) >= 1, "%s: empty level (n_in %d, n_out %d, B %d)", who,
                  I(f, F_n_in), I(f, F_n_out), I(f, F_B));
    AGB_CHECK_ARG(c1.cin % 4 == 0 && c1.cin >= 12 && C % 4 == 0 && C >= 12 && c2.cin == C && c2.cout == C,
                  "%s: channel counts %d -> %d -> %d (multiples of 4, >= 12)", who, c1.cin, C, c2.cout);
    AGB_CHECK_ARG(I(f, F_has_down) || (c1.cin == C && I(f, F_n_in) == I(f, F_n_out)),
                  "%s: a block without a downsample branch keeps its shape", who);
    AGB_CHECK_ARG(I(f, F_ldx) == c1.cin && I(f, F_ldy) == C, "%s: rows are contiguous (ldx %d, ldy %d)", who, I(f, F_ldx),
                  I(f, F_ldy));
    return AGB_OK;
}

// y = act(BatchNorm2(conv2(act(BatchNorm1(conv1(x))))) * SE * keep + residual),  residual = BatchNorm_d(conv_d(x)) or x
int agb_net_block_fwd(const int64_t* f, void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes, void* stream) {
    AGB_CHECK_ARG(f && saved && scratch, "agb_net_block_fwd: table, saved and scratch arenas are required");
    NET_TRY(block_check(f, "agb_net_block_fwd"));
    const Opts o = load_opts(f);
    const Conv c1 = load_conv(f, F_c1_w), c2 = load_conv(f, F_c2_w), cd = load_conv(f, F_cd_w);
    const int n = I(f, F_n_out), C = c1.cout, B = I(f, F_B), H = I(f, F_se_H), act = I(f, F_act);
    const bool down = I(f, F_has_down) != 0;
    Arena SA(saved, saved_bytes), TA(scratch, scratch_bytes);
    BlockSaved S; BlockFwdScratch T;
    block_saved(SA, f, &S);
    block_fwd_scratch(TA, f, &T);
    AGB_CHECK_ARG(SA.fits() && TA.fits(), "agb_net_block_fwd: arenas too small (%zu / %zu saved, %zu / %zu scratch)", saved_bytes,
                  SA.off, scratch_bytes, TA.off);
    const float* x = P<const float>(f, F_x);
    const int32_t *coords = P<const int32_t>(f, F_coords), *ptr = P<const int32_t>(f, F_ptr);
    NET_TRY(conv_fwd(o, c1, x, c1.cin, n, S.z1, C, T.partial, stream));
    NET_TRY(bn_fwd(o, c1, S.z1, n, act, S.st1, T.part, S.a1, stream));
    NET_TRY(conv_fwd(o, c2, S.a1, C, n, S.z2, C, T.partial, stream));
    const float* r = x;
    if (down) {
        NET_TRY(conv_fwd(o, cd, x, cd.cin, n, S.zd, C, T.partial, stream));
        NET_TRY(bn_fwd(o, cd, S.zd, n, ACT_NONE, S.std_, T.part, S.r, stream));
        r = S.r;
    }
    NET_TRY(agb_se_tail_stats(S.z2, C, ptr, n, C, B, c2.eps, c2.mom, o.training, T.part, S.st2, S.st2 + C, c2.rm, c2.rv, c2.nbt,
                              stream));
    NET_TRY(agb_se_tail_pool(T.part, ptr, n, B, C, S.st2, S.st2 + C, c2.g, c2.be, S.zbar, S.p, stream));
    NET_TRY(agb_se_mlp_fwd(S.p, P<const float>(f, F_se_w1), P<const float>(f, F_se_b1), P<const float>(f, F_se_w2),
                           P<const float>(f, F_se_b2), B, C, H, I(f, F_se_act), S.h_pre, S.s, stream));
    NET_TRY(agb_se_tail_fwd(S.z2, C, r, C, coords, S.st2, S.st2 + C, c2.g, c2.be, S.s, P<const float>(f, F_keep), act, n, C,
                            P<float>(f, F_y), C, stream));
    return AGB_OK;
}

int agb_net_block_bwd(const int64_t* f, void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes, void* stream) {
    AGB_CHECK_ARG(f && saved && scratch, "agb_net_block_bwd: table, saved and scratch arenas are required");
    NET_TRY(block_check(f, "agb_net_block_bwd"));
    AGB_CHECK_ARG(I(f, F_training) != 0, "agb_net_block_bwd: the fused backward pass takes BatchNorm in batch-statistics mode");
    const Opts o = load_opts(f);
    const Conv c1 = load_conv(f, F_c1_w), c2 = load_conv(f, F_c2_w), cd = load_conv(f, F_cd_w);
    const int n = I(f, F_n_out), n_in = I(f, F_n_in), C = c1.cout, Cin = c1.cin, B = I(f, F_B), H = I(f, F_se_H);
    const int act = I(f, F_act);
    const bool down = I(f, F_has_down) != 0;
    Arena SA(saved, saved_bytes), TA(scratch, scratch_bytes);
    BlockSaved S; BlockBwdScratch T;
    block_saved(SA, f, &S);
    block_bwd_scratch(TA, f, &T);
    AGB_CHECK_ARG(SA.fits() && TA.fits(), "agb_net_block_bwd: arenas too small (%zu / %zu saved, %zu / %zu scratch)", saved_bytes,
                  SA.off, scratch_bytes, TA.off);
    hipStream_t s = (hipStream_t)stream;
    const float *x = P<const float>(f, F_x), *dy = P<const float>(f, F_dy), *keep = P<const float>(f, F_keep);
    const int lddy = I(f, F_lddy);
    const int32_t *coords = P<const int32_t>(f, F_coords), *ptr = P<const int32_t>(f, F_ptr);
    const float* r = down ? S.r : x;
    float* dx = P<float>(f, F_dx);
    const float *mean2 = S.st2, *rstd2 = S.st2 + C;
    // ---- tail: dz2 (into the last convolution), dr (into the residual branch), BatchNorm2 and excitation gradients
    NET_TRY(agb_se_tail_bwd_sums(S.z2, C, r, C, dy, lddy, ptr, B, mean2, rstd2, c2.g, c2.be, S.s, keep, act, n, C, T.spart, stream));
    NET_TRY(agb_se_tail_bwd_ds(T.spart, ptr, n, c2.g, c2.be, keep, B, C, T.S, T.S + (size_t)B * C, T.ds, stream));
    NET_TRY(agb_se_mlp_bwd(S.p, P<const float>(f, F_se_w1), P<const float>(f, F_se_w2), B, C, H, I(f, F_se_act), S.h_pre, S.s, T.ds,
                           T.dz2se, T.dh, T.dp, P<float>(f, F_d_se_w1), P<float>(f, F_d_se_b1), P<float>(f, F_d_se_w2),
                           P<float>(f, F_d_se_b2), stream));
    NET_TRY(agb_se_tail_bwd_fold(T.S, T.S + (size_t)B * C, S.zbar, ptr, T.dp, S.s, keep, mean2, rstd2, B, C, T.dte, c2.dbe, c2.dg,
                                 stream));
    NET_TRY(agb_se_tail_bwd_apply(S.z2, C, r, C, dy, lddy, coords, mean2, rstd2, c2.g, c2.be, S.s, keep, T.dte, c2.dbe, c2.dg, act,
                                  o.training, n, C, T.dz2, C, T.dr, C, stream));
    // gzero: the caller laid the buffers that must start at zero (the three weight gradients, conv2's bias gradient) out
    // contiguously: one fill for the block instead of one per buffer
    const bool cleared = P<char>(f, F_gzero) != nullptr;
    if (cleared) {
        if (hipMemsetAsync(P<char>(f, F_gzero), 0, (size_t)f[F_gzero_bytes], s) != hipSuccess) return AGB_ELAUNCH;
    } else if (c2.db && hipMemsetAsync(c2.db, 0, sizeof(float) * C, s) != hipSuccess) {
        // (bias of a convolution in front of a training-mode BatchNorm: its gradient is exactly zero, norm_ops.py)
        return AGB_ELAUNCH;
    }
    // ---- conv2: data gradient, weight gradient
    const float* wt = nullptr;
    NET_TRY(conv_wt(c2, T.wt, &wt, cleared, stream));
    NET_TRY(conv_dgrad(o, c2, wt, T.dz2, C, n, T.da1, C, T.partial, nullptr, stream));
    NET_TRY(conv_wgrad(o, c2, S.a1, C, T.dz2, C, n, T.ws, conv_wgrad_bytes(o, c2, n, C, C), stream));
    // ---- BatchNorm1 + act
    NET_TRY(bn_bwd(o, c1, S.z1, T.da1, n, act, S.st1, T.part, T.dz1, T.colsum, stream));
    // ---- residual branch: its gradient with respect to x is the addend of conv1's data gradient
    const bool need_dx = I(f, F_need_dx) != 0;
    const float* second = T.dr;      // no downsample: the residual IS x
    if (down) {
        NET_TRY(bn_bwd(o, cd, S.zd, T.dr, n, ACT_NONE, S.std_, T.part, T.dzd, T.colsum, stream));
        NET_TRY(conv_wt(cd, T.wt, &wt, cleared, stream));
        if (need_dx) NET_TRY(conv_dgrad(o, cd, wt, T.dzd, C, n_in, T.dxb, Cin, T.partial, nullptr, stream));
        NET_TRY(conv_wgrad(o, cd, x, Cin, T.dzd, C, n, T.ws, conv_wgrad_bytes(o, cd, n, Cin, C), stream));
        second = T.dxb;
    }
    // ---- conv1
    NET_TRY(conv_wt(c1, T.wt, &wt, cleared, stream));
    if (need_dx) NET_TRY(conv_dgrad(o, c1, wt, T.dz1, C, n_in, dx, Cin, T.partial, second, stream));
    NET_TRY(conv_wgrad(o, c1, x, Cin, T.dz1, C, n, T.ws, conv_wgrad_bytes(o, c1, n, Cin, C), stream));
    AGB_CHECK_LAUNCH("agb_net_block_bwd");
    return AGB_OK;
}

}  // extern "C"
