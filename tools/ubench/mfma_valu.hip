// Microbenchmark: what does vector work placed between dependent MFMAs cost?  One wave per SIMD, a chain of
// v_mfma_f32_32x32x2_f32 on one accumulator with V independent VALU instructions of a given kind after each.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_valu.hip -o mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND, int V, int CHAINS = 1>
__global__ void __launch_bounds__(256) k(float *out, unsigned long long *ts, int iters) {
    f32x16 c, c2;
    for (int r = 0; r < 16; ++r) c[r] = 0.f, c2[r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = a + i;
    f32x4 av = {a, a, a, a}, bv = {b, b, b, b};
    __shared__ float lds[4096];
    lds[threadIdx.x] = a;
    f32x4 ld[4] = {av, av, av, av};
    const unsigned laddr = (threadIdx.x & 63) * 16;
    const float *gaddr = out + blockIdx.x * 256 + threadIdx.x;
    float *sdst = out + 256 * 256 * 2 + blockIdx.x * 64;            // wave-uniform: lives in an SGPR pair
    unsigned mkw = 0;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (CHAINS >= 3) {
                if (CHAINS == 3 || (u & 1) == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(av), "v"(bv));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c2) : "v"(av), "v"(bv));
            } else if (CHAINS == 1 || (u & 1) == 0) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c2) : "v"(a), "v"(b));
#pragma unroll
            for (int v = 0; v < V; ++v) {
                float &y = x[(u * V + v) % 16];
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(y) : "v"(b));
                if (KIND == 1) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(y));
                if (KIND == 2) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(y) : "v"(b), "s"(__builtin_amdgcn_read_exec()));
                if (KIND == 3) asm volatile("v_max_f32_e32 %0, %0, %1" : "+v"(y) : "v"(b));
                if (KIND == 5) asm volatile("s_nop 0");
                if (KIND == 11) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(y) : "v"(b));
                if (KIND == 12) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(y));
                if (KIND == 13) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(y) : "v"(b));
                if (KIND == 14) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(y));
                if (KIND == 15) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(y) : "v"(b), "v"(a));
                if (KIND == 16) asm volatile("v_accvgpr_read_b32 %0, a5" : "=v"(y));
                if (KIND == 17) asm volatile("v_max_i32 %0, 0, %0" : "+v"(y));
                // ReLU sign masks as LANE masks: one compare into an SGPR pair (18), plus a scalar store of the pair (19)
                if (KIND == 18 || KIND == 19) asm volatile("v_cmp_gt_f32_e64 s[20:21], %0, %1" : : "v"(y), "v"(b) : "s20", "s21");
                if (KIND == 19) asm volatile("s_store_dwordx2 s[20:21], %0, 0x0" : : "s"(sdst) : "memory");
                // ... against today's form: v_min_u32 + v_lshl_or_b32 into a packed word (20)
                if (KIND == 21) asm volatile("v_min_u32 %0, 1, %0" : "+v"(y));
                if (KIND == 22) asm volatile("v_lshl_or_b32 %0, %1, %2, %0" : "+v"(mkw) : "v"(y), "s"(u));
                if (KIND == 23) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(mkw) : "v"(y));
                if (KIND == 24) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(mkw) : "v"(y));
                if (KIND == 25) asm volatile("v_sub_u32 %0, 0, %0" : "+v"(y));
                if (KIND == 26) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(mkw) : "v"(y));
                if (KIND == 27) {      // candidate: integer negate (MSB = value != +0) + alignbit
                    unsigned t;
                    asm volatile("v_sub_u32 %0, 0, %1" : "=v"(t) : "v"(y));
                    asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(mkw) : "v"(t));
                }
                if (KIND == 20) {
                    unsigned one;
                    asm volatile("v_min_u32 %0, 1, %1" : "=v"(one) : "v"(y));
                    asm volatile("v_lshl_or_b32 %0, %1, %2, %0" : "+v"(mkw) : "v"(one), "s"(u));
                }
                if (KIND == 7) asm volatile("ds_read_b128 %0, %1" : "=v"(ld[(u * V + v) % 4]) : "v"(laddr));
                if (KIND == 8) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ld[(u * V + v) % 4]) : "v"(gaddr));
                if (KIND == 9) asm volatile("ds_write_b128 %0, %1" : : "v"(laddr), "v"(ld[0]));
                if (KIND == 10) asm volatile("global_store_dword %0, %1, off" : : "v"(gaddr), "v"(y));
                if (KIND == 6) asm volatile("s_add_u32 %0, %0, 1" : "+s"(it));
                if (KIND == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double *>(&x[2 * ((u * V + v) % 8)])) : "v"(*reinterpret_cast<double *>(&x[0])));
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
    if (KIND == 19) asm volatile("s_dcache_wb\n s_waitcnt lgkmcnt(0)");
    for (int r = 0; r < 16; ++r) s += c[r] + c2[r] + x[r] + ld[r & 3][r >> 2];
    s += (float)mkw;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ts[0] = t1 - t0;
}

template <int KIND, int V, int CHAINS = 1>
void run(const char *name) {
    const int iters = 1000;
    float *out; unsigned long long *ts, h;
    (void)hipMalloc(&out, 256 * 256 * 4 * 4); (void)hipMalloc(&ts, 8);
    hipLaunchKernelGGL((k<KIND, V, CHAINS>), dim3(256), dim3(256), 0, 0, out, ts, iters);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL((k<KIND, V, CHAINS>), dim3(256), dim3(256), 0, 0, out, ts, iters);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h, ts, 8, hipMemcpyDeviceToHost);
    printf("%-14s V=%2d  ticks per MFMA(+V) %.2f\n", name, V, h / (16.0 * iters));
    (void)hipFree(out); (void)hipFree(ts);
}
template <int KIND> void sweep(const char *name) {
    run<KIND, 0>(name); run<KIND, 2>(name); run<KIND, 4>(name); run<KIND, 8>(name); run<KIND, 12>(name); run<KIND, 16>(name); run<KIND, 24>(name);
}
int main() {
    sweep<0>("v_fma_f32");
    sweep<5>("s_nop 0");
    printf("two independent chains, alternating:\n");
    run<0, 0, 2>("v_fma_f32"); run<0, 2, 2>("v_fma_f32"); run<0, 4, 2>("v_fma_f32"); run<0, 8, 2>("v_fma_f32"); run<0, 12, 2>("v_fma_f32"); run<0, 16, 2>("v_fma_f32"); run<0, 24, 2>("v_fma_f32");
    printf("fp32 MFMA chain + memory instructions:\n");
    run<7, 1>("ds_read_b128"); run<7, 2>("ds_read_b128"); run<7, 4>("ds_read_b128");
    run<8, 1>("global_load_x4"); run<8, 2>("global_load_x4"); run<8, 4>("global_load_x4");
    run<9, 1>("ds_write_b128"); run<9, 2>("ds_write_b128"); run<9, 4>("ds_write_b128");
    run<10, 1>("global_store_dw"); run<10, 2>("global_store_dw"); run<10, 4>("global_store_dw");
    printf("bf16 MFMA chain + memory instructions:\n");
    run<7, 1, 3>("ds_read_b128"); run<7, 2, 3>("ds_read_b128"); run<7, 3, 3>("ds_read_b128");
    run<8, 1, 3>("global_load_x4"); run<8, 2, 3>("global_load_x4");
    run<9, 1, 3>("ds_write_b128"); run<9, 2, 3>("ds_write_b128");
    run<10, 1, 3>("global_store_dw"); run<10, 2, 3>("global_store_dw");
    printf("fp32 MFMA chain + ReLU-mask forms (per value): lane mask by v_cmp into SGPRs, the same + s_store_dwordx2, today's v_min+v_lshl_or:\n");
    run<18, 2>("v_cmp->sgpr"); run<18, 4>("v_cmp->sgpr"); run<18, 8>("v_cmp->sgpr");
    run<19, 2>("v_cmp+s_store"); run<19, 4>("v_cmp+s_store"); run<19, 8>("v_cmp+s_store");
    run<20, 2>("v_min+v_lshl_or"); run<20, 4>("v_min+v_lshl_or"); run<20, 8>("v_min+v_lshl_or");
    printf("single instructions after an fp32 MFMA (V = 4, 8):\n");
    run<21, 4>("v_min_u32"); run<21, 8>("v_min_u32");
    run<22, 4>("v_lshl_or sgpr"); run<22, 8>("v_lshl_or sgpr");
    run<23, 4>("v_lshl_or const"); run<23, 8>("v_lshl_or const");
    run<24, 4>("v_alignbit"); run<24, 8>("v_alignbit");
    run<25, 4>("v_sub_u32"); run<25, 8>("v_sub_u32");
    run<26, 4>("v_lshl_add_u32"); run<26, 8>("v_lshl_add_u32");
    run<27, 2>("v_sub+v_alignbit"); run<27, 4>("v_sub+v_alignbit"); run<27, 8>("v_sub+v_alignbit");
    printf("bf16 32x32x16 MFMA, one dependent chain:\n");
    run<0, 0, 3>("v_fma_f32"); run<0, 1, 3>("v_fma_f32"); run<0, 2, 3>("v_fma_f32"); run<0, 4, 3>("v_fma_f32"); run<0, 6, 3>("v_fma_f32"); run<0, 8, 3>("v_fma_f32"); run<0, 12, 3>("v_fma_f32");
    printf("bf16 32x32x16 MFMA, two chains:\n");
    run<0, 0, 4>("v_fma_f32"); run<0, 1, 4>("v_fma_f32"); run<0, 2, 4>("v_fma_f32"); run<0, 4, 4>("v_fma_f32"); run<0, 6, 4>("v_fma_f32"); run<0, 8, 4>("v_fma_f32"); run<0, 12, 4>("v_fma_f32");
    printf("bf16 one chain + instruction kinds (V = 2, 4, 6):\n");
    run<11, 2, 3>("v_cvt_pk_bf16"); run<11, 4, 3>("v_cvt_pk_bf16"); run<11, 6, 3>("v_cvt_pk_bf16");
    run<12, 2, 3>("v_and_b32 lit"); run<12, 4, 3>("v_and_b32 lit"); run<12, 6, 3>("v_and_b32 lit");
    run<13, 2, 3>("v_sub_f32"); run<13, 4, 3>("v_sub_f32"); run<13, 6, 3>("v_sub_f32");
    run<14, 2, 3>("v_lshlrev_b32"); run<14, 4, 3>("v_lshlrev_b32"); run<14, 6, 3>("v_lshlrev_b32");
    run<15, 2, 3>("v_perm_b32"); run<15, 4, 3>("v_perm_b32"); run<15, 6, 3>("v_perm_b32");
    run<16, 2, 3>("v_accvgpr_read"); run<16, 4, 3>("v_accvgpr_read"); run<16, 6, 3>("v_accvgpr_read");
    run<17, 2, 3>("v_max_i32"); run<17, 4, 3>("v_max_i32"); run<17, 6, 3>("v_max_i32");
    printf("bf16, cvt_pk / dpp kinds, one chain:\n");
    run<1, 2, 3>("v_mov_dpp"); run<1, 4, 3>("v_mov_dpp"); run<1, 6, 3>("v_mov_dpp"); run<4, 2, 3>("v_pk_add_f32"); run<4, 4, 3>("v_pk_add_f32"); run<4, 6, 3>("v_pk_add_f32");
    return 0;
}
