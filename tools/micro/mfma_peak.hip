// Sustained fp32 / bf16 matrix-core rate and shader clock of an MI355X under MFMA load: every SIMD issues back-to-back
// independent MFMAs for a few hundred microseconds to a few milliseconds; the kernel reads s_memtime (shader clocks) and
// s_memrealtime (100 MHz) at both ends, so the clock the chip actually holds under this load comes out beside the TFLOP/s.
// Build: hipcc -O3 --offload-arch=gfx950 -Wno-unused-result -o mfma_peak mfma_peak.hip      Run: ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

// MODE 0: v_mfma_f32_16x16x4_f32 (2048 flop), 1: v_mfma_f32_32x32x2_f32 (4096 flop), 2: v_mfma_f32_32x32x16_bf16 (32768 flop)
template <int MODE, bool VARY>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* clk, int iters) {
    // operands that differ per lane and change every trip (toggling inputs draw more power than constants)
    float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f + threadIdx.x * 3e-3f;
    unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    float keep = 0.f;
    if (MODE == 0) {
        f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
            }
            if (VARY) { a = -a * 1.0001f; b = b * 0.9999f + 1e-3f; }
        }
        keep = c0[0] + c1[1] + c2[2] + c3[3];
    } else if (MODE == 1) {
        f32x16 c0, c1;
        for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; }
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
            }
            if (VARY) { a = -a * 1.0001f; b = b * 0.9999f + 1e-3f; }
        }
        keep = c0[0] + c1[5];
    } else {
        f32x16 c0, c1;
        for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; }
        bf16x8 av, bv;
        for (int i = 0; i < 8; ++i) { av[i] = (short)(0x3f80 + threadIdx.x % 7); bv[i] = 0x3f00; }
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c1, 0, 0, 0);
            }
            if (VARY) { av = av * (short)-3 + (short)i; bv = bv + (short)(i * 7); }
        }
        keep = c0[0] + c1[5];
    }
    unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = keep;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE, bool VARY>
void run(const char* name, double flop_per_mfma, int mfma_per_iter, int waves_per_simd, int iters) {
    const int blocks = 256 * waves_per_simd;       // 256 CUs x 4 SIMDs: one 256-thread workgroup = one wave per SIMD
    float* d; unsigned long long* c;
    hipMalloc(&d, blocks * 256 * sizeof(float)); hipMalloc(&c, blocks * 2 * sizeof(unsigned long long));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, VARY>), dim3(blocks), dim3(256), 0, 0, d, c, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, VARY>), dim3(blocks), dim3(256), 0, 0, d, c, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 2);
    hipMemcpy(h.data(), c, blocks * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (int i = 0; i < blocks; ++i) { cyc += h[2 * i]; rt += h[2 * i + 1]; }
    cyc /= blocks; rt /= blocks;
    const double flops = (double)blocks * 4 * iters * mfma_per_iter * flop_per_mfma;
    printf("%-26s %s %d wave/SIMD  %8.3f ms  %7.1f TFLOP/s   in-kernel: %.0f s_memtime ticks per wave over %.1f us (100 MHz clock) = %.3f GHz if s_memtime counts shader clocks;"
           " %.2f ticks per MFMA per wave\n", name, VARY ? "varying " : "constant", waves_per_simd, ms, flops / ms / 1e9, cyc, rt / 100.0, cyc / (rt / 100.0) / 1e3,
           cyc / ((double)iters * mfma_per_iter));
    hipFree(d); hipFree(c);
}

int main() {
    for (int rep = 0; rep < 2; ++rep) {
        const int it = 100000;
        run<0, false>("v_mfma_f32_16x16x4_f32", 2048, 16, 1, it);
        run<0, true>("v_mfma_f32_16x16x4_f32", 2048, 16, 1, it);
        run<1, false>("v_mfma_f32_32x32x2_f32", 4096, 8, 1, it);
        run<1, true>("v_mfma_f32_32x32x2_f32", 4096, 8, 1, it);
        run<2, false>("v_mfma_f32_32x32x16_bf16", 32768, 8, 1, it);
        run<2, true>("v_mfma_f32_32x32x16_bf16", 32768, 8, 1, it);
        run<2, true>("v_mfma_f32_32x32x16_bf16", 32768, 8, 2, it);
    }
    return 0;
}
