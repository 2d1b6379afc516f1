// Microbenchmark: do a SIMD's matrix instructions overlap with VECTOR instructions of ANOTHER wave on the same SIMD?
// (mfma_valu.hip answers the question for one wave: fp32 MFMAs do not overlap with the wave's own VALU work.)
// One workgroup of 8 waves per CU = two waves per SIMD.  Waves 0-3 ("M") run a chain of dependent MFMAs, waves 4-7 ("V")
// run a chain of dependent v_fma_f32; each role is timed alone and together.  If the matrix instruction has its own pipe
// the two finish in max(M, V); if it runs on the vector ALUs, in M + V.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_two_waves.hip -o mfma_two_waves
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MKIND, int NACC = 1>     // NACC independent accumulators, round robin;  MKIND 0: v_mfma_f32_32x32x2_f32   1: v_mfma_f32_32x32x16_bf16   2: v_mfma_f32_16x16x4_f32
__global__ void __launch_bounds__(512) k(float *out, unsigned long long *ts, int m_iters, int v_iters) {
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // scalar: the role branch must be a real branch
    const bool is_m = wid < 4;
    f32x16 c, cb, cc, cd;
    f32x4 c4 = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < 16; ++r) c[r] = cb[r] = cc[r] = cd[r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    f32x4 av = {a, a, a, a}, bv = {b, b, b, b};
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = a + i;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (is_m) {
        for (int it = 0; it < m_iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (MKIND == 0 && NACC == 4) {
                    if ((u & 3) == 0) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
                    if ((u & 3) == 1) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(cb) : "v"(a), "v"(b));
                    if ((u & 3) == 2) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(cc) : "v"(a), "v"(b));
                    if ((u & 3) == 3) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(cd) : "v"(a), "v"(b));
                }
                if (MKIND == 0 && NACC == 1) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
                if (MKIND == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(av), "v"(bv));
                if (MKIND == 2) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(c4) : "v"(a), "v"(b));
            }
        }
    } else {
        for (int it = 0; it < v_iters; ++it) {
#pragma unroll
            for (int u = 0; u < 64; ++u) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[u % 8]) : "v"(b));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c[r] + cb[r] + cc[r] + cd[r] + x[r & 7] + c4[r & 3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) ts[wid] = t1 - t0;
}

template <int MKIND, int NACC = 1>
void run(const char *name, int m_iters, int v_iters) {
    float *out; unsigned long long *ts, h[8];
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&ts, 64);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<MKIND, NACC>), dim3(256), dim3(512), 0, 0, out, ts, m_iters, v_iters);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    (void)hipMemcpy(h, ts, 64, hipMemcpyDeviceToHost);
    printf("%-36s m_iters %5d v_iters %5d   kernel %7.3f ms   M wave %9llu ticks   V wave %9llu ticks\n", name, m_iters, v_iters,
           ms, h[0], h[4]);
    (void)hipFree(out); (void)hipFree(ts);
}

template <int MKIND, int NACC = 1>
void trio(const char *name, int m_iters, int v_iters) {
    run<MKIND, NACC>(name, m_iters, 0);
    run<MKIND, NACC>(name, 0, v_iters);
    run<MKIND, NACC>(name, m_iters, v_iters);
}

int main() {
    // 16 MFMAs x 1000 = 16 000 fp32 32x32x2 MFMAs (64 cycles each) ~ 1.02 M cycles; 64 x 4000 v_fma (4 cycles each) ~ 1.02 M
    printf("alone / alone / together (ticks of the shader clock counter; equal work in all three lines of a group)\n");
    trio<0>("fp32 32x32x2 MFMA | v_fma_f32", 1000, 4000);
    trio<0, 4>("fp32 32x32x2, 4 accumulators | v_fma", 1000, 4000);
    trio<2>("fp32 16x16x4 MFMA | v_fma_f32", 2000, 4000);
    trio<1>("bf16 32x32x16 MFMA | v_fma_f32", 2000, 4000);
    return 0;
}
