// LDS update throughput on gfx950: ds_add_f32 vs ds_add_u32 vs a plain read-add-write, conflict-free addresses,
// and global fp32 atomics with many lanes per 64-byte piece.  Build: hipcc -O3 --offload-arch=gfx950 -o lds_atomic_bench lds_atomic_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float s[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) s[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float v = 1.0f + lane;
    for (int it = 0; it < iters; ++it) {
        int a = wave * 1024 + ((it * 64) & 1023) + lane;     // conflict-free, each wave its own region
        if (MODE == 0) atomicAdd(&s[a], v);
        else if (MODE == 1) atomicAdd(reinterpret_cast<unsigned*>(&s[a]), 3u);
        else if (MODE == 2) { float t = s[a]; s[a] = t + v; }
        else if (MODE == 3) { float t = atomicAdd(&s[a], v); v += t * 1e-30f; }
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = s[threadIdx.x] + v;
}

template <int MODE>
float run(float* d, int iters, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* d; hipMalloc(&d, 256 * 4096 * 4);
    const int iters = 20000, blocks = 256 * 2;      // 2 workgroups (8 waves) per CU
    const char* names[] = {"ds_add_f32", "ds_add_u32", "read+add+write", "ds_add_rtn_f32"};
    float ms[4] = {run<0>(d, iters, blocks), run<1>(d, iters, blocks), run<2>(d, iters, blocks), run<3>(d, iters, blocks)};
    for (int i = 0; i < 4; ++i) {
        // per CU: 8 waves x iters wave-instructions
        double cyc = ms[i] * 1e-3 * 2.4e9 / (8.0 * iters);
        printf("%-16s %8.3f ms  -> %.1f CU cycles per 64-lane update (at 2.4 GHz)\n", names[i], ms[i], cyc);
    }
    return 0;
}
