// agb_common.h — shared device/host helpers for libagbhip (gfx950 / MI355X only).
//
// Conventions of the C ABI (see include/agb_hip.h):
//   * every entry point returns 0 on success or a negative AGB_E* code and
//     records a message retrievable with agb_last_error();
//   * the caller owns every buffer (device pointers unless stated otherwise);
//   * nothing here allocates, frees or synchronises — kernels are enqueued on
//     the hipStream_t passed in (passed as void* across the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define AGB_OK 0
#define AGB_EINVAL (-1)
#define AGB_ELAUNCH (-2)
#define AGB_ERANGE (-3)
#define AGB_EUNSUPPORTED (-4)

extern "C" const char* agb_last_error(void);
void agb_set_error(const char* fmt, ...);
// Diagnostic (thread-local, like the error string): the compute kernel the last convolution / dense-product / weight-
// gradient entry point of this thread launched — bench tools name the kernel of a timed launch with it instead of
// mirroring the dispatch rules.  A leading '(' of a parenthesised template name is skipped.
extern "C" const char* agb_last_kernel(void);
void agb_note_kernel(const char* name);
#define AGB_LAUNCH(kern, ...)            \
    do {                                 \
        agb_note_kernel(#kern);          \
        hipLaunchKernelGGL(kern, __VA_ARGS__); \
    } while (0)

#define AGB_CHECK_ARG(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            agb_set_error(__VA_ARGS__);          \
            return AGB_EINVAL;                   \
        }                                        \
    } while (0)

#define AGB_CHECK_LAUNCH(name)                                                   \
    do {                                                                         \
        hipError_t e__ = hipGetLastError();                                      \
        if (e__ != hipSuccess) {                                                 \
            agb_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return AGB_ELAUNCH;                                                  \
        }                                                                        \
    } while (0)

// Entry points with C linkage that stay INSIDE the library (the many-buffer forms behind the *_ws entry points of
// abi_ws.hip): not part of the ABI of include/agb_hip.h, not exported.
#define AGB_INTERNAL __attribute__((visibility("hidden")))

static inline int agb_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// spconv.hip: the convolution entry points with an addend in the final store (Y = addend + (bias + sum); the public form is
// agb_spconv_bwd_data of include/agb_hip.h)
extern "C" AGB_INTERNAL int agb_spconv_fwd_opt_add(const float* X, int ldx, const float* W, const int32_t* nbr,
                                                  long long nbr_stride, int kflip, const float* bias, float* Y, int ldy,
                                                  int n_out, int K3, int Cin, int Cout, const int32_t* perm,
                                                  const int32_t* tile_cls, const int32_t* cls_tab, int n_tiles, int ksplit,
                                                  float* partial, int cmp_mode, int cmp_interleave_shift, const float* addend,
                                                  int ld_add, void* stream);
extern "C" AGB_INTERNAL int agb_spconv_fwd_tiles_add(const float* X, int ldx, const float* W, const int32_t* nbr,
                                                    long long nbr_stride, int kflip, const float* bias, float* Y, int ldy,
                                                    int n_out, int K3, int Cin, int Cout, int ksplit, float* partial,
                                                    int cmp_mode, int cmp_interleave_shift, const int32_t* tile_blocks,
                                                    int tb_tiles, int tb_blocks, const float* addend, int ld_add, void* stream);

// dense_stream.hip: HBM-bound dense products (many rows, small weight matrix), taken from the identity-map entry points
bool agb_dense_stream_ok(int n, int Cin, int Cout);
bool agb_dense_stream_wgrad_ok(int n, int Cin, int Cout);
int agb_dense_stream_launch(const float* X, int ldx, const float* W, const float* bias, float* Y, int ldy, int n, int Cin,
                            int Cout, hipStream_t st);
int agb_dense_stream_wgrad_launch(const float* X, int ldx, const float* dY, int ldy, float* dW, int n, int Cin, int Cout,
                                  hipStream_t st);

// dwreg.hip: fp32 weight gradient with both MFMA operands gathered into registers (no LDS staging, no barriers); row chunks
// folded in a fixed order through `workspace` (NULL: fp32 atomic accumulation)
size_t agb_dwreg_workspace_bytes(int n_out, int K3, int Cin, int Cout);
int agb_dwreg_launch(const float* X, int ldx, const float* dY, int ldy, const int32_t* nbr, long long nbr_stride, float* dW,
                     int n_out, int K3, int Cin, int Cout, void* workspace, size_t workspace_bytes, hipStream_t st);

// dwa.hip: fp32 weight gradient with persistent accumulators and a hand-scheduled main loop (Cin, Cout multiples of 64)
bool agb_dwa_ok(int n_out, int K3, int Cin, int Cout, int ldx, int ldy, bool force = false);
size_t agb_dwa_workspace_bytes(int n_out, int K3, int Cin, int Cout);
int agb_dwa_launch(const float* X, int ldx, const float* dY, int ldy, const int32_t* nbr, long long nbr_stride, float* dW,
                   int n_out, int K3, int Cin, int Cout, void* workspace, size_t workspace_bytes, hipStream_t st);

// stem.hip: pair-sparse weight gradient of the 3-channel stem (one 4x4x1 MFMA per pair, fixed-order fold)
bool agb_stem_dw_ok(int n_out, int K3, int Cin, int Cout, int ldx, int ldy);
size_t agb_stem_dw_workspace_bytes(int n_out, int K3);
int agb_stem_dw_launch(const float* X, const float* dY, int ldy, const int32_t* nbr, long long nbr_stride, float* dW, int n_out,
                       int K3, void* workspace, size_t workspace_bytes, hipStream_t st, const int32_t* coords = nullptr,
                       const int32_t* grid = nullptr, const int32_t* desc = nullptr, int K = 0);

bool agb_stem_fwd_ok(int n_out, int K, int Cout, int ldx);
int agb_stem_fwd_launch(const float* X, const float* W, const float* bias, float* Y, int ldy, int n_out, int K,
                        const int32_t* coords, const int32_t* grid, const int32_t* desc, int32_t* nbr_out,
                        long long nbr_out_stride, hipStream_t st);

// ---- coordinate key packing -------------------------------------------------
// [b | z | y | x], 16 bits each, spatial components biased by 32768 so that
// negative voxel coordinates (ME allows them) order correctly.
#define AGB_COORD_BIAS 32768
#define AGB_KEY_EMPTY 0xFFFFFFFFFFFFFFFFull

__host__ __device__ static inline uint64_t agb_pack_key(int b, int x, int y, int z) {
    return ((uint64_t)(uint16_t)b << 48) | ((uint64_t)(uint16_t)(z + AGB_COORD_BIAS) << 32) |
           ((uint64_t)(uint16_t)(y + AGB_COORD_BIAS) << 16) | (uint64_t)(uint16_t)(x + AGB_COORD_BIAS);
}

__host__ __device__ static inline uint32_t agb_hash_key(uint64_t k) {
    // 64-bit finaliser (murmur3 fmix64); table capacity is a power of two.
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return (uint32_t)k;
}

// floor division for possibly negative numerators, positive divisor
__host__ __device__ static inline int agb_floordiv(int a, int d) {
    int q = a / d;
    int r = a % d;
    return (r != 0 && ((r < 0) != (d < 0))) ? q - 1 : q;
}

// ---- row-matrix storage types -----------------------------------------------------
// Activation / gradient row matrices [N, C] are fp32, or — the bf16-activation mode of BASELINE config 5 — bf16 (uint16
// bit patterns, round to nearest even on store; all arithmetic stays fp32).  The HBM-bound kernels of norm.hip / pool.hip
// are templates over the storage type and move 4 channels per lane either way (16-B / 8-B pieces); entry points of the
// bf16 form carry the suffix _h.
typedef unsigned short bf16_t;
#if defined(__HIPCC__)
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const bf16_t* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xFFFF0000u));
}
__device__ __forceinline__ unsigned agb_pack2_bf16(float a, float b) {      // v_cvt_pk_bf16_f32, round to nearest even
    typedef float agb_f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 agb_bf16x2 __attribute__((ext_vector_type(2)));
    const agb_f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, agb_bf16x2));
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(bf16_t* p, float4 v) {
    *reinterpret_cast<uint2*>(p) = make_uint2(agb_pack2_bf16(v.x, v.y), agb_pack2_bf16(v.z, v.w));
}
// value of the neighbouring lane (lane ^ 1) through a DPP quad permutation [1, 0, 3, 2]: a VALU move, no LDS traffic
__device__ __forceinline__ float agb_lane_xor1(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
}
// One bf16 pair per lane from two accumulator registers that hold rows r (v0) and r + 1 (v1) of column `lane`: the even
// lane of a pair returns row r, columns (lane, lane + 1), the odd lane row r + 1, columns (lane - 1, lane) — one 4-byte
// store each instead of two 2-byte stores.  (Written with the select on the PACKED words: a select between the two
// accumulator registers themselves is folded into a dynamically indexed vector read, ~16 compare/select pairs each.)
__device__ __forceinline__ unsigned agb_bf16_pair_rows(float v0, float v1, bool odd) {
    const float n0 = agb_lane_xor1(v0), n1 = agb_lane_xor1(v1);
    const unsigned even_w = agb_pack2_bf16(v0, n0), odd_w = agb_pack2_bf16(n1, v1);
    return odd ? odd_w : even_w;
}
// The same with the packed bf16 pair `adw` of another matrix at the lane's STORE position added in fp32 before the one
// rounding (the addend of a residual join, ConvArgs.addend on bf16 rows).
__device__ __forceinline__ unsigned agb_bf16_pair_rows_add(float v0, float v1, bool odd, unsigned adw) {
    const float n0 = agb_lane_xor1(v0), n1 = agb_lane_xor1(v1);
    const float alo = __uint_as_float(adw << 16), ahi = __uint_as_float(adw & 0xffff0000u);
    const unsigned even_w = agb_pack2_bf16(v0 + alo, n0 + ahi), odd_w = agb_pack2_bf16(n1 + alo, v1 + ahi);
    return odd ? odd_w : even_w;
}
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(bf16_t* p, float v) {
    const __bf16 h = (__bf16)v;
    *p = __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const bf16_t* p) { return __uint_as_float((unsigned)*p << 16); }
#endif

// ---- activations fused into the BatchNorm / pooling kernels ---------------------
#define ACT_NONE 0
#define ACT_RELU 1
#define ACT_GELU 2

__device__ __forceinline__ float act_fwd(float z, int act) {
    if (act == ACT_RELU) return z > 0.f ? z : 0.f;
    if (act == ACT_GELU) return 0.5f * z * (1.f + erff(z * 0.70710678118654752440f));
    return z;
}
// value and derivative at once (the GELU pair shares its erf)
// (Round 4 measured a cheaper GELU — Phi through the five-term erfc of Abramowitz & Stegun 7.1.26, 24 instead of 46 VALU
// instructions for value + derivative, 3e-7 absolute error — on the premise that the bf16-row element-wise kernels were at
// the VALU roof: MSENet50 bf16 rows 1784 -> 1800, MPointNet 5525 -> 5531 plots/s: nothing, those passes are bound by
// launch count and size (16-65 us kernels), not by erff.  Dropped: the exact erf stays — EXPERIMENTS.md.)
__device__ __forceinline__ void act_fwd_grad(float z, int act, float* val, float* grad) {
    if (act == ACT_RELU) { *val = z > 0.f ? z : 0.f; *grad = z > 0.f ? 1.f : 0.f; return; }
    if (act == ACT_GELU) {
        const float e = erff(z * 0.70710678118654752440f);
        const float pdf = 0.39894228040143267794f * __expf(-0.5f * z * z);   // (v_exp_f32: 2 ulp, the sums take 1e-7)
        *val = 0.5f * z * (1.f + e);                  // (the expressions of act_fwd / act_grad: same rounding)
        *grad = 0.5f * (1.f + e) + z * pdf;
        return;
    }
    *val = z; *grad = 1.f;
}
__device__ __forceinline__ float act_grad(float z, int act) {
    if (act == ACT_RELU) return z > 0.f ? 1.f : 0.f;
    if (act == ACT_GELU) {
        float cdf = 0.5f * (1.f + erff(z * 0.70710678118654752440f));
        float pdf = 0.39894228040143267794f * __expf(-0.5f * z * z);   // (v_exp_f32, 2 ulp)
        return cdf + z * pdf;
    }
    return 1.f;
}
