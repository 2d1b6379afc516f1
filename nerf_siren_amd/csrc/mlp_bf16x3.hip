// OPT-IN fast math for the NeRF MLP forward (inference, and training with saved activations) on gfx950:
// fp32-equivalent products on the bf16 matrix cores.  (Backward: mlp_bwd.hip sections 1b and 2b; layer: bf16x3_core.h.)
//
// gfx950 has no TF32/xf32; its exact-fp32 MFMA runs at 1/16 of the bf16 rate.  An fp32 value splits EXACTLY
// into three bf16 terms (x = x1 + x2 + x3, 3 x 8 significant bits, round-to-nearest at each step), and a
// product of two bf16 numbers is exact in fp32, so
//     a*b = a1b1 + a1b2 + a2b1 + a2b2 + a1b3 + a3b1   (+ terms <= 2^-24 |ab|)
// accumulated in fp32 by six v_mfma_f32_32x32x16_bf16 is as accurate as an fp32 fma chain, at 16/6 = 2.7x the
// fp32-MFMA rate.  Same transposed, register-resident scheme as mlp.hip: the accumulator row map is the next
// layer's B operand order (two k-steps of 8 units per 32-unit block and lane half), so activations are split
// once per layer in registers; weights are pre-split at pack time and shared through the LDS ring.
// The default path stays the exact-fp32 MFMA (mlp.hip); this one is selected explicitly and is held to the same
// parity tests.
#include "bf16x3_core.h"

namespace nerfmi {


static FastTable nerf_fast_table() {
    FastTable T;
    int n = 0;
    auto add = [&](int off, int JB, int KB, int unit0) { T.l[n++] = FastLayer{off, JB, KB, unit0}; };
    const int fwd[NL_FWD][3] = {{OFF_L1, 8, 2}, {OFF_L2, 8, 8}, {OFF_L3, 8, 8}, {OFF_L4, 8, 8}, {OFF_L5, 8, 10},
                                {OFF_L6, 8, 8}, {OFF_L7, 8, 8}, {OFF_L8, 8, 8}, {OFF_FINAL, 8, 8}, {OFF_DIR, 4, 9}};
    for (auto &f : fwd) add(f[0], f[1], f[2], f[0] / 512);
    // transposed images of the backward chain, in the order it walks them: (output blocks of dX, contraction blocks)
    const int bwd[9][3] = {{OFF_TDIR, 8, 4}, {OFF_TFINAL, 8, 8}, {OFF_T8, 8, 8}, {OFF_T7, 8, 8}, {OFF_T6, 8, 8},
                           {OFF_T5, 8, 8}, {OFF_T4, 8, 8}, {OFF_T3, 8, 8}, {OFF_T2, 8, 8}};
    for (auto &f : bwd) add(f[0], f[1], f[2], FAST_FWD_UNITS + (f[0] - OFF_TRANS) / 512);
    T.n = n;
    T.n_units = FAST_FWD_UNITS + FAST_T_UNITS;
    return T;
}

__device__ __forceinline__ void embed_xyz_blocks_f(float x, float y, float z, int half, f32x16 *e) {
    float v[64];
    v[0] = x; v[1] = y; v[2] = z; v[63] = 0.f;
    const float xyz[3] = {x, y, z};
#pragma unroll
    for (int f = 0; f < 10; ++f)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            float s, c;
            sincos_cw(__fmul_rn(xyz[d], (float)(1 << f)), s, c);
            v[3 + 6 * f + d] = s;
            v[3 + 6 * f + 3 + d] = c;
        }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c0 = 32 * kb + 8 * (r >> 2) + (r & 3);
            e[kb][r] = half ? v[c0 + 4] : v[c0];
        }
}
__device__ __forceinline__ void embed_dir_block_f(float x, float y, float z, int half, f32x16 &e) {
    float v[32];
    v[0] = x; v[1] = y; v[2] = z;
#pragma unroll
    for (int c = 27; c < 32; ++c) v[c] = 0.f;
    const float xyz[3] = {x, y, z};
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            float s, c;
            sincos_cw(__fmul_rn(xyz[d], (float)(1 << f)), s, c);
            v[3 + 6 * f + d] = s;
            v[3 + 6 * f + 3 + d] = c;
        }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int c0 = 8 * (r >> 2) + (r & 3);
        e[r] = half ? v[c0 + 4] : v[c0];
    }
}

// Experiment builds (-DNERFMI_TIMING) stamp the shader clock at layer boundaries for a few workgroups.
#ifdef NERFMI_TIMING
__device__ unsigned long long nerfmi_dbg_ts_fast[64 * 16];
#define NERFMI_TSF(i)                                                                                       \
    do {                                                                                                    \
        if ((threadIdx.x == 0) && (blockIdx.x % 32 == 0) && (blockIdx.x / 32 < 32))                         \
            nerfmi_dbg_ts_fast[(blockIdx.x / 32) * 16 + (i)] = __builtin_readcyclecounter();                \
    } while (0)
#else
#define NERFMI_TSF(i) do { } while (0)
#endif

template <bool SIGMA_ONLY, bool SAVE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
nerf_forward_bf16x3_kernel(const float *__restrict__ packed, const __bf16 *__restrict__ fast,
                           const float *__restrict__ rays, const float *__restrict__ z, int64_t n_points,
                           int n_per_ray, float *__restrict__ out, float *__restrict__ saved, int64_t ld) {
    extern __shared__ __attribute__((aligned(16))) char wlds[];
    const int lane = threadIdx.x & 63, half = lane >> 5, wid = threadIdx.x >> 6;
    NERFMI_TSF(0);
    const int64_t wave = (int64_t)blockIdx.x * 4 + wid;
    const int64_t praw = wave * 32 + (lane & 31);
    const bool ok = praw < n_points;
    const int64_t p = ok ? praw : n_points - 1;
    RowImage S;
    S.init(saved, wave, ld / 32, SAVED_ROWS, lane, ok, wave * 32 < n_points);
    const float *rr = rays + (p / n_per_ray) * 8;
    const float zz = z[p];
    const float x = __fadd_rn(rr[0], __fmul_rn(rr[3], zz));
    const float y = __fadd_rn(rr[1], __fmul_rn(rr[4], zz));
    const float w = __fadd_rn(rr[2], __fmul_rn(rr[5], zz));
    f32x16 e[2];
    embed_xyz_blocks_f(x, y, w, half, e);
    if (SAVE) {
        store_block(S, S_EMB, e[0]);
        store_block(S, S_EMB + 32, e[1]);
    }
    const float *bias = packed + OFF_BIAS + 4 * half;
    auto img = [&](int off) { return fast + fast_fwd_elems(off); };
    f32x16 hA[8], hB[8];                                 // alternate: a layer reads one, accumulates into the other
    FastStage fs;
    // Training: a layer's post-ReLU output is written to the saved image (and its sign bits to the mask words) by
    // the layer that CONSUMES it, pair by pair between its MFMAs (mlp_layout.h S_H / S_MASK; same images as mlp.hip).
    unsigned mk[4] = {0u, 0u, 0u, 0u};
    auto save_h = [&](int row0) {
        return [&S, &mk, row0](int kb, int pr, float &x0, float &x1) {
            if (!SAVE) return;
            const int r = 2 * pr;
            float *dst = S.tile + (row0 + 32 * kb + 8 * (r >> 2) + 4 * (S.lane >> 5) + (r & 3)) * 32 + (S.lane & 31);
            __builtin_nontemporal_store(x0, dst);
            __builtin_nontemporal_store(x1, dst + 32);
#ifdef NERFMI_EXP_MASK_CHAIN
            unsigned one;
            asm("v_min_u32 %0, 1, %1" : "=v"(one) : "v"(x0));
            asm("v_lshl_or_b32 %0, %1, %2, %0" : "+v"(mk[kb >> 1]) : "v"(one), "s"(16 * (kb & 1) + r));
            asm("v_min_u32 %0, 1, %1" : "=v"(one) : "v"(x1));
            asm("v_lshl_or_b32 %0, %1, %2, %0" : "+v"(mk[kb >> 1]) : "v"(one), "s"(16 * (kb & 1) + r + 1));
#else
            unsigned b0, b1;                             // one update of the mask word per pair (mlp_core.h mask_or)
            asm("v_min_u32 %0, 1, %1" : "=v"(b0) : "v"(x0));
            asm("v_min_u32 %0, 1, %1" : "=v"(b1) : "v"(x1));
            mk[kb >> 1] |= ((b1 << 1) | b0) << (16 * (kb & 1) + r);
#endif
        };
    };
    auto save_raw = [&](int row0) {                      // no activation: values only
        return [&S, row0](int kb, int pr, float &x0, float &x1) {
            if (!SAVE) return;
            const int r = 2 * pr;
            float *dst = S.tile + (row0 + 32 * kb + 8 * (r >> 2) + 4 * (S.lane >> 5) + (r & 3)) * 32 + (S.lane & 31);
            __builtin_nontemporal_store(x0, dst);
            __builtin_nontemporal_store(x1, dst + 32);
        };
    };
    NoHook none;
    NERFMI_TSF(1);
    layer_bf16x3<2, 0, 8, false, true>(img(OFF_L1), bias, e, nullptr, hA, wlds, fs, wid, lane);
    NERFMI_TSF(2);
    layer_bf16x3<0, 8, 8, true, false>(img(OFF_L2), bias + 256 * 1, nullptr, hA, hB, wlds, fs, wid, lane, none, save_h(S_H));
    if (SAVE) store_mask(S, 0, mk);
    NERFMI_TSF(3);
    layer_bf16x3<0, 8, 8, true, false>(img(OFF_L3), bias + 256 * 2, nullptr, hB, hA, wlds, fs, wid, lane, none, save_h(S_H + 256));
    if (SAVE) store_mask(S, 1, mk);
    NERFMI_TSF(4);
    layer_bf16x3<0, 8, 8, true, false>(img(OFF_L4), bias + 256 * 3, nullptr, hA, hB, wlds, fs, wid, lane, none, save_h(S_H + 256 * 2));
    if (SAVE) store_mask(S, 2, mk);
    NERFMI_TSF(5);
    layer_bf16x3<2, 8, 8, true, false>(img(OFF_L5), bias + 256 * 4, e, hB, hA, wlds, fs, wid, lane, none, save_h(S_H + 256 * 3));
    if (SAVE) store_mask(S, 3, mk);
    NERFMI_TSF(6);
    layer_bf16x3<0, 8, 8, true, false>(img(OFF_L6), bias + 256 * 5, nullptr, hA, hB, wlds, fs, wid, lane, none, save_h(S_H + 256 * 4));
    if (SAVE) store_mask(S, 4, mk);
    NERFMI_TSF(7);
    layer_bf16x3<0, 8, 8, true, false>(img(OFF_L7), bias + 256 * 6, nullptr, hB, hA, wlds, fs, wid, lane, none, save_h(S_H + 256 * 5));
    if (SAVE) store_mask(S, 5, mk);
    NERFMI_TSF(8);
    layer_bf16x3<0, 8, 8, true, false>(img(OFF_L8), bias + 256 * 7, nullptr, hA, hB, wlds, fs, wid, lane, none, save_h(S_H + 256 * 6));
    if (SAVE) store_mask(S, 6, mk);
    NERFMI_TSF(9);
    // hB = raw outputs of xyz_encoding_8; sigma = w_sigma . relu(h) + b (nerf.py:112)
    float sigma;
    {
        f32x16 h8[8];
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) h8[b][r] = relu1(hB[b][r]);
        sigma = dot_blocks<8>(h8, packed + OFF_W_SIGMA + 4 * half) + packed[OFF_B_SIGMA];
    }
    if (SIGMA_ONLY) {
        if (ok && half == 0) out[p] = sigma;
        return;
    }
    // xyz_encoding_final: no activation on its output
    layer_bf16x3<0, 8, 8, true, false>(img(OFF_FINAL), bias + 256 * 8, nullptr, hB, hA, wlds, fs, wid, lane, none, save_h(S_H + 256 * 7));
    if (SAVE) store_mask(S, 7, mk);
    NERFMI_TSF(10);
    f32x16 de[1], dh[4];
    embed_dir_block_f(rr[3], rr[4], rr[5], half, de[0]);    // computed here, not up front: 16 registers fewer held
    if (SAVE) store_block(S, S_DEMB, de[0]);
    // dir_encoding input = [final (no ReLU) | dir embedding]: the packed order is final first (mlp_layout.h)
    layer_bf16x3<8, 1, 4, false, false>(img(OFF_DIR), packed + OFF_BIAS_DIR + 4 * half, hA, de, dh, wlds, fs, wid, lane, save_raw(S_FINAL));
    NERFMI_TSF(11);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
#pragma unroll
        for (int r = 0; r < 16; ++r) dh[b][r] = relu1(dh[b][r]);
        if (SAVE) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                mask_or(mk, b, q, f32x4{dh[b][4 * q], dh[b][4 * q + 1], dh[b][4 * q + 2], dh[b][4 * q + 3]});
            store_block(S, S_DIRH + 32 * b, dh[b]);
        }
    }
    if (SAVE) store_mask(S, 8, mk);
    float rgb[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float pre = dot_blocks<4>(dh, packed + OFF_W_RGB + 128 * c + 4 * half) + packed[OFF_B_RGB + c];
        rgb[c] = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-pre)));
    }
    if (ok && half == 0) {
        float4 o;
        o.x = rgb[0]; o.y = rgb[1]; o.z = rgb[2]; o.w = sigma;
        reinterpret_cast<float4 *>(out)[p] = o;
    }
    if (SAVE && half == 0) {
        *S.at(S_RGB + 0) = rgb[0];
        *S.at(S_RGB + 1) = rgb[1];
        *S.at(S_RGB + 2) = rgb[2];
    }
    NERFMI_TSF(12);
}

}  // namespace nerfmi

using namespace nerfmi;

extern "C" {

#ifdef NERFMI_TIMING
int nerfmi_debug_timing_fast(unsigned long long *host) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(nerfmi_dbg_ts_fast), sizeof(unsigned long long) * 64 * 16) == hipSuccess ? 0 : 1;
}
#endif

size_t nerfmi_nerf_fast_bytes(void) { return (size_t)(FAST_FWD_UNITS + FAST_T_UNITS) * 3072 + FAST_TAIL_BYTES; }

int nerfmi_nerf_pack_fast(const float *packed, void *fast, nerfmi_stream_t stream) {
    NERFMI_ENTER();
    NERFMI_REQUIRE(packed && fast, "nerf_pack_fast: null pointer");
    hipLaunchKernelGGL(pack_bf16x3_table_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, nerf_fast_table(), packed,
                       (__bf16 *)fast);
    return check_launch("nerf_pack_fast");
}

int nerfmi_nerf_forward_rays_fast(const float *packed, const void *fast, const float *rays, const float *z, int n_rays,
                                  int n_per_ray, int sigma_only, float *out, float *saved, nerfmi_stream_t stream) {
    NERFMI_ENTER();
    NERFMI_REQUIRE(n_rays >= 0 && n_per_ray >= 1, "nerf_forward_rays_fast: bad sizes");
    const int64_t n_points = (int64_t)n_rays * n_per_ray;
    if (n_points == 0) return NERFMI_OK;
    NERFMI_REQUIRE(packed && fast && rays && z && out, "nerf_forward_rays_fast: null pointer");
    NERFMI_REQUIRE(!(saved && sigma_only), "nerf_forward_rays_fast: saved activations need the full (rgb,sigma) pass");
    const void *kernels[3] = {reinterpret_cast<const void *>(nerf_forward_bf16x3_kernel<false, false>),
                              reinterpret_cast<const void *>(nerf_forward_bf16x3_kernel<true, false>),
                              reinterpret_cast<const void *>(nerf_forward_bf16x3_kernel<false, true>)};
    static PerDeviceOnce attr_set;
    int attr_dev;
    if (attr_set.needed(attr_dev)) {
        for (const void *k : kernels)
            if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, FLDS_BYTES) != hipSuccess) {
                (void)hipGetLastError();
                set_error("nerf_forward_rays_fast: cannot raise the dynamic LDS limit");
                return NERFMI_E_LAUNCH;
            }
        attr_set.mark(attr_dev);
    }
    const int64_t waves = (n_points + 31) / 32;
    const int64_t ld = waves * 32;
    const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    KernelSpan span(sigma_only ? "nerf_forward_bf16x3_kernel<sigma_only>" : (saved ? "nerf_forward_bf16x3_kernel<save>"
                                                                                      : "nerf_forward_bf16x3_kernel"), n_points, st);
    if (sigma_only)
        hipLaunchKernelGGL((nerf_forward_bf16x3_kernel<true, false>), grid, block, FLDS_BYTES, st, packed,
                           (const __bf16 *)fast, rays, z, n_points, n_per_ray, out, nullptr, ld);
    else if (saved)
        hipLaunchKernelGGL((nerf_forward_bf16x3_kernel<false, true>), grid, block, FLDS_BYTES, st, packed,
                           (const __bf16 *)fast, rays, z, n_points, n_per_ray, out, saved, ld);
    else
        hipLaunchKernelGGL((nerf_forward_bf16x3_kernel<false, false>), grid, block, FLDS_BYTES, st, packed,
                           (const __bf16 *)fast, rays, z, n_points, n_per_ray, out, nullptr, ld);
    return check_launch("nerf_forward_rays_fast");
}

}  // extern "C"
