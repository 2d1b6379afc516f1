// Microbenchmark: issue rate of dependent vs independent v_mfma_f32_32x32x2_f32 / 32x32x16_bf16 chains,
// one wave per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 mfma_chain.hip -o mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int CHAINS, int DS>
__global__ void __launch_bounds__(256) k_f32(float *out, unsigned long long *ts, int iters) {
    __shared__ float lds[4096];
    f32x16 c[CHAINS];
    for (int i = 0; i < CHAINS; ++i)
        for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    lds[threadIdx.x] = a;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (DS) a = lds[(threadIdx.x + u * 64 + it) & 4095];
#pragma unroll
            for (int i = 0; i < CHAINS; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[i], 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < CHAINS; ++i)
        for (int r = 0; r < 16; ++r) s += c[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { ts[0] = t1 - t0; }
}

template <int CHAINS>
__global__ void __launch_bounds__(256) k_bf16(float *out, unsigned long long *ts, int iters) {
    f32x16 c[CHAINS];
    for (int i = 0; i < CHAINS; ++i)
        for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
    bf16x8 a, b;
    for (int r = 0; r < 8; ++r) { a[r] = (__bf16)(threadIdx.x * 1e-3f); b[r] = (__bf16)1.0f; }
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < CHAINS; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < CHAINS; ++i)
        for (int r = 0; r < 16; ++r) s += c[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { ts[0] = t1 - t0; }
}

template <class F>
void run(const char *name, F launch, int mfma_per_iter, int iters) {
    float *out; unsigned long long *ts, h;
    hipMalloc(&out, 256 * 256 * 4 * 8); hipMalloc(&ts, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(out, ts, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch(out, ts, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, ts, 8, hipMemcpyDeviceToHost);
    double n = (double)mfma_per_iter * iters;
    printf("%-28s ticks/MFMA %.2f   ns/MFMA %.3f  (kernel %.3f ms)\n", name, h / n, ms * 1e6 / n, ms);
    hipFree(out); hipFree(ts);
}

int main() {
    const int iters = 2000;
    const dim3 g(256), b(256);   // one workgroup of 4 waves per CU
    run("f32 1 chain", [&](float *o, unsigned long long *t, int it) { hipLaunchKernelGGL((k_f32<1, 0>), g, b, 0, 0, o, t, it); }, 16, iters);
    run("f32 2 chains", [&](float *o, unsigned long long *t, int it) { hipLaunchKernelGGL((k_f32<2, 0>), g, b, 0, 0, o, t, it); }, 32, iters);
    run("f32 4 chains", [&](float *o, unsigned long long *t, int it) { hipLaunchKernelGGL((k_f32<4, 0>), g, b, 0, 0, o, t, it); }, 64, iters);
    run("f32 1 chain + ds_read", [&](float *o, unsigned long long *t, int it) { hipLaunchKernelGGL((k_f32<1, 1>), g, b, 0, 0, o, t, it); }, 16, iters);
    run("f32 2 chains + ds_read", [&](float *o, unsigned long long *t, int it) { hipLaunchKernelGGL((k_f32<2, 1>), g, b, 0, 0, o, t, it); }, 32, iters);
    run("bf16 1 chain", [&](float *o, unsigned long long *t, int it) { hipLaunchKernelGGL((k_bf16<1>), g, b, 0, 0, o, t, it); }, 16, iters);
    run("bf16 2 chains", [&](float *o, unsigned long long *t, int it) { hipLaunchKernelGGL((k_bf16<2>), g, b, 0, 0, o, t, it); }, 32, iters);
    run("bf16 4 chains", [&](float *o, unsigned long long *t, int it) { hipLaunchKernelGGL((k_bf16<4>), g, b, 0, 0, o, t, it); }, 64, iters);
    // same with a single workgroup (clock not power-limited)
    const dim3 g1(1);
    run("f32 1 chain, 1 WG", [&](float *o, unsigned long long *t, int it) { hipLaunchKernelGGL((k_f32<1, 0>), g1, b, 0, 0, o, t, it); }, 16, iters);
    run("f32 2 chains, 1 WG", [&](float *o, unsigned long long *t, int it) { hipLaunchKernelGGL((k_f32<2, 0>), g1, b, 0, 0, o, t, it); }, 32, iters);
    run("bf16 1 chain, 1 WG", [&](float *o, unsigned long long *t, int it) { hipLaunchKernelGGL((k_bf16<1>), g1, b, 0, 0, o, t, it); }, 16, iters);
    run("bf16 2 chains, 1 WG", [&](float *o, unsigned long long *t, int it) { hipLaunchKernelGGL((k_bf16<2>), g1, b, 0, 0, o, t, it); }, 32, iters);
    return 0;
}
