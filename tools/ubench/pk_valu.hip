// Microbenchmark (round 3): what do PACKED fp32 vector instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two values
// per lane per instruction) and quarter-rate transcendentals cost next to a chain of fp32 MFMAs?  One wave per SIMD, a
// dependent chain of v_mfma_f32_32x32x2_f32 with V vector instructions of a kind after each, NCH independent register
// chains for the vector work (NCH = 1: every instruction consumes the previous one's result).
// Build: hipcc --offload-arch=gfx950 -O3 pk_valu.hip -o pk_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND, int V, int NCH>
__global__ void __launch_bounds__(256) k(float *out, unsigned long long *ts, int iters) {
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    f32x2 x[8];
    for (int i = 0; i < 8; ++i) x[i] = f32x2{a + i, a - i};
    const f32x2 bb = {b, b * 0.5f};
    unsigned mk = threadIdx.x;
    float *gaddr = out + blockIdx.x * 1024 + threadIdx.x * 4;
    f32x4 st4 = {a, b, a, b};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
#pragma unroll
            for (int v = 0; v < V; ++v) {
                f32x2 &y = x[(u * V + v) % NCH];
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(y[0]) : "v"(b));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(y) : "v"(bb));
                if (KIND == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y) : "v"(bb));
                if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y) : "v"(bb));
                if (KIND == 4) asm volatile("v_sqrt_f32 %0, %0" : "+v"(y[0]));
                if (KIND == 5) asm volatile("v_lshl_or_b32 %0, %1, 31, %0" : "+v"(y[0]) : "v"(mk));
                if (KIND == 6) asm volatile("v_accvgpr_read_b32 %0, a3" : "=v"(y[0]));
                if (KIND == 7) asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(gaddr), "v"(st4) : "memory");
                if (KIND == 8) asm volatile("global_store_dword %0, %1, off" : : "v"(gaddr), "v"(a) : "memory");
                if (KIND == 9) asm volatile("global_load_dword %0, %1, off" : "=v"(y[0]) : "v"(gaddr) : "memory");
                if (KIND == 10) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(st4) : "v"(gaddr) : "memory");
                if (KIND == 11) asm volatile("v_sin_f32 %0, %0" : "+v"(y[0]));
                if (KIND == 12) asm volatile("v_rndne_f32 %0, %0" : "+v"(y[0]));
                if (KIND == 13) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(y[0]) : "v"(mk));
                if (KIND == 14) asm volatile("v_alignbit_b32 %0, %1, %0, 31" : "+v"(y[0]) : "v"(mk));
                if (KIND == 15) asm volatile("v_bfi_b32 %0, 1, %1, %0" : "+v"(y[0]) : "v"(mk));
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c[r];
    for (int i = 0; i < 8; ++i) s += x[i][0] + x[i][1];
    s += st4[0] + st4[3];
    out[256 * 1024 + blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ts[0] = t1 - t0;
}

template <int KIND, int V, int NCH>
void run(const char *name) {
    const int iters = 500;
    float *out; unsigned long long *ts, h;
    (void)hipMalloc(&out, (256 * 1024 + 256 * 256) * 4); (void)hipMalloc(&ts, 8);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<KIND, V, NCH>), dim3(256), dim3(256), 0, 0, out, ts, iters);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(&h, ts, 8, hipMemcpyDeviceToHost);
    printf("%-22s chains=%d V=%2d  ticks per MFMA(+V) %.2f\n", name, NCH, V, h / (16.0 * iters));
    (void)hipFree(out); (void)hipFree(ts);
}
template <int KIND, int NCH> void sweep(const char *name) {
    run<KIND, 0, NCH>(name); run<KIND, 2, NCH>(name); run<KIND, 4, NCH>(name); run<KIND, 8, NCH>(name); run<KIND, 16, NCH>(name);
}
int main() {
    sweep<0, 8>("v_fma_f32"); sweep<0, 2>("v_fma_f32"); sweep<0, 1>("v_fma_f32");
    sweep<1, 8>("v_pk_fma_f32"); sweep<1, 4>("v_pk_fma_f32"); sweep<1, 2>("v_pk_fma_f32"); sweep<1, 1>("v_pk_fma_f32");
    sweep<2, 8>("v_pk_mul_f32"); sweep<3, 8>("v_pk_add_f32");
    sweep<4, 8>("v_sqrt_f32"); sweep<11, 8>("v_sin_f32"); sweep<12, 8>("v_rndne_f32");
    sweep<5, 8>("v_lshl_or_b32"); sweep<13, 8>("v_xor_b32"); sweep<14, 8>("v_alignbit_b32"); sweep<15, 8>("v_bfi_b32");
    sweep<6, 8>("v_accvgpr_read");
    run<7, 1, 8>("global_store_dwordx4"); run<7, 2, 8>("global_store_dwordx4"); run<7, 4, 8>("global_store_dwordx4");
    run<8, 1, 8>("global_store_dword"); run<8, 2, 8>("global_store_dword"); run<8, 4, 8>("global_store_dword");
    run<9, 1, 8>("global_load_dword"); run<9, 2, 8>("global_load_dword"); run<9, 4, 8>("global_load_dword");
    run<10, 1, 8>("global_load_dwordx4"); run<10, 2, 8>("global_load_dwordx4"); run<10, 4, 8>("global_load_dwordx4");
    return 0;
}
