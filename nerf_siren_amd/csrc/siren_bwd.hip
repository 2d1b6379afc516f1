// FiLM-SIREN field, training path for gfx950 (MI355X): autograd of models/nerf.py:142-151 (FiLMLayer) and :201-216
// (SemanticNeRF.forward_with_frequencies_phase_shifts) w.r.t. the 22 parameters.
//
//  0. siren_forward_kernel<., false, SAVE = true> (siren_core.h): the forward that also writes, per 32-point tile, the
//     layer inputs the dW GEMM needs (the sines themselves) and cos(arg) of every unit.
//  1. siren_backward_chain_kernel -- the dX chain.  Same register-resident scheme as the NeRF chain (mlp_bwd.hip): a
//     wave owns 32 points, dZ_l^T lives in the accumulator layout and is the B operand of
//     dH_{l-1}^T = W_l^T . dZ_l^T (A = the transposed packed image).  The activation derivative is
//         d sin(fr * pre + ph) / d pre = fr * cos(arg),
//     the cosine read back from the forward's lane-private cosine image (siren_core.h SS_COS: one 16-byte load per slice).
//     Every dZ_l goes to the workspace as a tile-major image.
//  2. dW_l = dZ_l^T . X_l: the NeRF dW GEMM (dw_core.h) on a 12-task plan of exactly 256 workgroups.
//  3. deterministic slab reduction (siren_dw_reduce_kernel below): bit-reproducible gradients -- and, when the launch
//     shares ONE conditioning row, the gradients of that row (frequencies, phase_shifts: nerf.py:147-151, :201-216 are
//     differentiable in them) from the same slabs.  With  arg = fr (W h + b) + ph,  fr = 15 f + 30,  G = dH cos(arg):
//         dZ = fr G,   dW = fr (G^T X),   db = fr sum_p G,   d ph = sum_p G,
//         d f = 15 sum_p G (W h + b) = 15 (<W_j, (G^T X)_j> + b_j sum_p G_j)
//     -- every term is a row operation on the UNSCALED slabs G^T X.  So in that mode the chain kernel writes G (not dZ) to
//     the workspace (it still hands dZ = fr G to the next layer's MFMAs: same instruction count), the reduction scales the
//     rows by fr, and the conditioning gradients cost one dot product per unit: no division by fr (fr = 0 is legal), no
//     second pass over the activations.
#include <stdlib.h>

#include "dw_bf16x3.h"
#include "siren_core.h"

namespace nerfmi {

// workspace row map (tile-major images written by the chain kernel)
constexpr int SW_DZ = 0;                   // 9 x 256: dZ of network.0..7, then color_layer_sine (G = dZ / fr in one-row launches)
// head rows: [row][32 points] order in the split-bf16 chain (rows SW_DRGB + c, SW_DSIG); x4 order in the fp32 chain, where they
// sit in unit 0 of their own row groups (rows SW_DRGB + 4c and SW_DSIG4) for dw_task4g's narrow-A form
constexpr int SW_DRGB = 9 * 256;           // d rgb pre-sigmoid
constexpr int SW_DSIG = SW_DRGB + 4;       // d sigma ([row][point] order)
constexpr int SW_DSIG4 = SW_DRGB + 16;     // d sigma (x4 order: the narrow-A task reads row groups 0..3 from its first row)
constexpr int SW_ROWS = SW_DRGB + 32;

// the 16 saved cosines of block jb of `layer` that belong to this lane's accumulator registers: four 16-byte loads
__device__ __forceinline__ f32x16 load_cos_block(const RowImage &im, int layer, int jb) {
    f32x16 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 c = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(cos_slice(im, layer, jb, q)));   // read once
        v[4 * q] = c[0]; v[4 * q + 1] = c[1]; v[4 * q + 2] = c[2]; v[4 * q + 3] = c[3];
    }
    return v;
}

// d pre = d h * fr * cos(arg) for the four units of slice q of block jb (layer `layer`): cs = the forward's saved cosines
// (this lane's 16 of the block), fr from LDS (COND_LDS) or memory.  Two packed multiplies per pair.
// Returns dZ = fr * G (the next layer's MFMA operand); `g` receives what goes to the workspace image: G = dh * cos(arg) when
// the launch shares one conditioning row (COND_LDS: the reduction scales the rows by fr), dZ otherwise.
template <bool COND_LDS>
__device__ __forceinline__ f32x4 film_grad(f32x4 dh, const f32x16 &cs, const float *fq, const float *lfr, int layer, int jb,
                                           int q, f32x4 &g) {
    f32x4 fr;
    if (COND_LDS) {
        fr = *reinterpret_cast<const f32x4 *>(lfr + 256 * layer + 32 * jb + 8 * q);
    } else {
        const f32x4 f = ldg4(fq + 256 * layer + 32 * jb + 8 * q);
#pragma unroll
        for (int t = 0; t < 4; ++t) fr[t] = __fadd_rn(__fmul_rn(f[t], 15.0f), 30.0f);                  // nerf.py:202
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x2 gg = f32x2{dh[2 * h], dh[2 * h + 1]} * f32x2{cs[4 * q + 2 * h], cs[4 * q + 2 * h + 1]};
        const f32x2 dz = gg * f32x2{fr[2 * h], fr[2 * h + 1]};
        dh[2 * h] = dz[0]; dh[2 * h + 1] = dz[1];
        g[2 * h] = COND_LDS ? gg[0] : dz[0]; g[2 * h + 1] = COND_LDS ? gg[1] : dz[1];
    }
    return dh;
}

template <bool COND_LDS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
siren_backward_chain_kernel(const float *__restrict__ packed, const float *__restrict__ saved,
                            const float *__restrict__ grad_out, const float *__restrict__ freq, int64_t n_points,
                            int64_t points_per_cond, int64_t ld, float *__restrict__ work) {
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t p0 = wave * 32;
    const bool live = p0 < n_points;          // no early exit: the workgroup's waves share barriers (layer_mfma_lds)
    const int64_t praw = p0 + (lane & 31);
    const bool ok = praw < n_points;
    const int64_t p = ok ? praw : n_points - 1;
    RowImage S, Wk;
    S.init(const_cast<float *>(saved), wave, ld / 32, SIREN_SAVED_ROWS, lane, ok, live);
    Wk.init(work, wave, ld / 32, SW_ROWS, lane, ok, live);
    // fr = 15 f + 30 of the launch's one conditioning row staged in LDS (COND_LDS), else this lane's row from memory
    __shared__ __attribute__((aligned(16))) float film[COND_LDS ? 2304 : 4];
    if (COND_LDS) {
        for (int i = threadIdx.x; i < 2304; i += blockDim.x) film[i] = __fadd_rn(__fmul_rn(freq[i], 15.0f), 30.0f);
        __syncthreads();
    }
    const float *fq = freq + (COND_LDS ? 0 : (p / points_per_cond) * 2304) + 4 * half;
    const float *lfr = film + 4 * half;

    float4 go = make_float4(0.f, 0.f, 0.f, 0.f);       // points past the end: every dZ below is exactly 0
    if (ok) go = reinterpret_cast<const float4 *>(grad_out)[praw];
    // rgb = sigmoid(pre): d pre = d rgb * rgb * (1 - rgb)      (nerf.py:214)
    float dpre[3];
    {
        const float g3[3] = {go.x, go.y, go.z};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float rgb = *S.at(SS_RGB + c);
            dpre[c] = g3[c] * rgb * (1.0f - rgb);
            if (half == 0) *at4(Wk, SW_DRGB + 4 * c) = dpre[c];
        }
        if (half == 0) *at4(Wk, SW_DSIG4) = go.w;
    }
    const float dsig = go.w;

    f32x16 dzA[8], dzB[8];
    // d h_c = W_rgb^T d pre;  dZ_c = d h_c * fr_8 * cos(arg_c)          (nerf.py:213-214)
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const f32x16 sv = load_cos_block(S, 8, b);
        f32x16 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 w[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) w[c] = ldg4(packed + SOFF_W_RGB + 256 * c + 32 * b + 8 * q + 4 * half);
            f32x4 dh;
#pragma unroll
            for (int t = 0; t < 4; ++t)
                dh[t] = __builtin_fmaf(w[2][t], dpre[2], __builtin_fmaf(w[1][t], dpre[1], w[0][t] * dpre[0]));
            f32x4 g;
            dh = film_grad<COND_LDS>(dh, sv, fq, lfr, 8, b, q, g);
#pragma unroll
            for (int t = 0; t < 4; ++t) v[4 * q + t] = dh[t];
            store_slice4(Wk, SW_DZ + 8 * 256 + 32 * b, q, g);
        }
        dzA[b] = v;
    }
    extern __shared__ __attribute__((aligned(16))) float wlds[];       // SIREN_WLDS_BYTES
    const int wid = threadIdx.x >> 6;
    WeightStageT<SIREN_GS> ws;
    // d h_7 = W_c[:, 3:]^T dZ_c + w_sigma d sigma;  dZ_7 = d h_7 * fr_7 * cos(arg_7)      (nerf.py:212-213)
    layer_mfma_lds<8, 0, 8, 0, true, SIREN_GS, SIREN_DMA>(packed + SOFF_TCOLOR, nullptr, dzA, nullptr, dzB,
                                     [&S](int jb) { return load_cos_block(S, 7, jb); },
                                     [&](int jb, int q, f32x4 c, const f32x16 &sv) {
                                         const f32x4 w = ldg4(packed + SOFF_W_SIGMA + 32 * jb + 8 * q + 4 * half);
#pragma unroll
                                         for (int t = 0; t < 4; ++t) c[t] = __builtin_fmaf(w[t], dsig, c[t]);
                                         f32x4 g;
                                         c = film_grad<COND_LDS>(c, sv, fq, lfr, 7, jb, q, g);
                                         store_slice4(Wk, SW_DZ + 7 * 256 + 32 * jb, q, g);
                                         return c;
                                     }, wlds, ws, wid, lane);
    // network.7 .. network.1: d h_{l-1} = W_l^T dZ_l;  dZ_{l-1} = d h_{l-1} * fr_{l-1} * cos(arg_{l-1})
    auto back = [&](int l, const f32x16 *in, f32x16 *out_dz) __attribute__((always_inline)) {
        layer_mfma_lds<8, 0, 8, 0, false, SIREN_GS, SIREN_DMA>(packed + SOFF_T7 + (7 - l) * SZ_HID, nullptr, in, nullptr, out_dz,
                                          [&S, l](int jb) { return load_cos_block(S, l - 1, jb); },
                                          [&, l](int jb, int q, f32x4 c, const f32x16 &sv) {
                                              f32x4 g;
                                              c = film_grad<COND_LDS>(c, sv, fq, lfr, l - 1, jb, q, g);
                                              store_slice4(Wk, SW_DZ + (l - 1) * 256 + 32 * jb, q, g);
                                              return c;
                                          }, wlds, ws, wid, lane);
    };
    back(7, dzB, dzA);
    back(6, dzA, dzB);
    back(5, dzB, dzA);
    back(4, dzA, dzB);
    back(3, dzB, dzA);
    back(2, dzA, dzB);
    back(1, dzB, dzA);
    if (SIREN_DMA) ring_drain();
}

// ---------------------------------------------------------------------------
// 1b. the dX chain on the bf16 matrix cores (opt-in split-bf16 math, bf16x3_core.h): same images in, same images out.
//     The eight 256 x 256 products run as six bf16 MFMAs per fp32-equivalent product block; the FiLM derivative of a
//     layer's raw output (two packed multiplies per pair with the saved cosines), the G / dZ stores and the next layer's
//     operand are finished BETWEEN the layers (not in consuming-layer hooks as in the NeRF chain: the hooks would have to
//     prefetch 16 cosines per block inside a kernel that already uses 500 registers).
// ---------------------------------------------------------------------------
template <bool COND_LDS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
siren_backward_chain_bf16x3_kernel(const float *__restrict__ packed, const __bf16 *__restrict__ fast,
                                   const float *__restrict__ saved, const float *__restrict__ grad_out,
                                   const float *__restrict__ freq, int64_t n_points, int64_t points_per_cond, int64_t ld,
                                   float *__restrict__ work) {
    extern __shared__ __attribute__((aligned(16))) char wlds_fast[];
    const int lane = threadIdx.x & 63, half = lane >> 5, wid = threadIdx.x >> 6;
    const int64_t wave = (int64_t)blockIdx.x * 4 + wid;
    const int64_t p0 = wave * 32;
    const bool live = p0 < n_points;
    const int64_t praw = p0 + (lane & 31);
    const bool ok = praw < n_points;
    const int64_t p = ok ? praw : n_points - 1;
    RowImage S, Wk;
    S.init(const_cast<float *>(saved), wave, ld / 32, SIREN_SAVED_ROWS, lane, ok, live);
    Wk.init(work, wave, ld / 32, SW_ROWS, lane, ok, live);
    __shared__ __attribute__((aligned(16))) float film[COND_LDS ? 2304 : 4];
    if (COND_LDS) {
        for (int i = threadIdx.x; i < 2304; i += blockDim.x) film[i] = __fadd_rn(__fmul_rn(freq[i], 15.0f), 30.0f);
        __syncthreads();
    }
    const float *fq = freq + (COND_LDS ? 0 : (p / points_per_cond) * 2304) + 4 * half;
    const float *lfr = film + 4 * half;

    float4 go = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) go = reinterpret_cast<const float4 *>(grad_out)[praw];
    float dpre[3];
    {
        const float g3[3] = {go.x, go.y, go.z};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float rgb = *S.at(SS_RGB + c);
            dpre[c] = g3[c] * rgb * (1.0f - rgb);
            if (half == 0) *Wk.at(SW_DRGB + c) = dpre[c];
        }
        if (half == 0) *Wk.at(SW_DSIG) = go.w;
    }
    const float dsig = go.w;

    f32x16 dzA[8], dzB[8];
    // d h_c = W_rgb^T d pre;  dZ_c = d h_c * fr_8 * cos(arg_c)          (nerf.py:213-214)
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const f32x16 sv = load_cos_block(S, 8, b);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 w[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) w[c] = ldg4(packed + SOFF_W_RGB + 256 * c + 32 * b + 8 * q + 4 * half);
            f32x4 dh, g;
#pragma unroll
            for (int t = 0; t < 4; ++t)
                dh[t] = __builtin_fmaf(w[2][t], dpre[2], __builtin_fmaf(w[1][t], dpre[1], w[0][t] * dpre[0]));
            dh = film_grad<COND_LDS>(dh, sv, fq, lfr, 8, b, q, g);
#pragma unroll
            for (int t = 0; t < 4; ++t) dzA[b][4 * q + t] = dh[t];
            store_slice(Wk, SW_DZ + 8 * 256 + 32 * b, q, g);
        }
    }
    // raw d h_layer (accumulator layout) -> dZ_layer in place; G (one-row launches) or dZ to the workspace image
    auto finish = [&](int layer, f32x16 *dh) __attribute__((always_inline)) {
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const f32x16 sv = load_cos_block(S, layer, b);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 c = {dh[b][4 * q], dh[b][4 * q + 1], dh[b][4 * q + 2], dh[b][4 * q + 3]}, g;
                c = film_grad<COND_LDS>(c, sv, fq, lfr, layer, b, q, g);
                store_slice(Wk, SW_DZ + layer * 256 + 32 * b, q, g);
                dh[b][4 * q] = c[0]; dh[b][4 * q + 1] = c[1]; dh[b][4 * q + 2] = c[2]; dh[b][4 * q + 3] = c[3];
            }
        }
    };
    FastStage fs;
    NoHook none;
    auto timg = [&](int t) { return fast + (int64_t)(SIREN_FAST_FWD_UNITS + t * (SZ_HID / 512)) * 1536; };
    // d h_7 = W_c[:, 3:]^T dZ_c + w_sigma d sigma   (nerf.py:212-213)
    layer_bf16x3<8, 0, 8, false, true>(timg(0), packed + SOFF_W_SIGMA + 4 * half, dzA, nullptr, dzB, wlds_fast, fs, wid, lane, none,
                                       none, dsig);
    finish(7, dzB);
    // network.7 .. network.1: d h_{l-1} = W_l^T dZ_l
    layer_bf16x3<8, 0, 8, false, false>(timg(1), nullptr, dzB, nullptr, dzA, wlds_fast, fs, wid, lane);
    finish(6, dzA);
    layer_bf16x3<8, 0, 8, false, false>(timg(2), nullptr, dzA, nullptr, dzB, wlds_fast, fs, wid, lane);
    finish(5, dzB);
    layer_bf16x3<8, 0, 8, false, false>(timg(3), nullptr, dzB, nullptr, dzA, wlds_fast, fs, wid, lane);
    finish(4, dzA);
    layer_bf16x3<8, 0, 8, false, false>(timg(4), nullptr, dzA, nullptr, dzB, wlds_fast, fs, wid, lane);
    finish(3, dzB);
    layer_bf16x3<8, 0, 8, false, false>(timg(5), nullptr, dzB, nullptr, dzA, wlds_fast, fs, wid, lane);
    finish(2, dzA);
    layer_bf16x3<8, 0, 8, false, false>(timg(6), nullptr, dzA, nullptr, dzB, wlds_fast, fs, wid, lane);
    finish(1, dzB);
    layer_bf16x3<8, 0, 8, false, false>(timg(7), nullptr, dzB, nullptr, dzA, wlds_fast, fs, wid, lane);
    finish(0, dzA);
}

// ---------------------------------------------------------------------------
// 2. dW: same GEMM as the NeRF backward on the SIREN images
// ---------------------------------------------------------------------------
#ifdef NERFMI_TIMING
__device__ unsigned long long nerfmi_dbg_siren_dw[512];   // per-workgroup shader-clock duration (experiment builds)
#endif

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
siren_dw_kernel(DwPlan plan, const float *__restrict__ work, const float *__restrict__ saved, int64_t ld,
                float *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef NERFMI_TIMING
    const unsigned long long t_start = __builtin_readcyclecounter();
#endif
    int ti = 0;
    for (int i = 1; i < plan.n_tasks; ++i)
        if ((int)blockIdx.x >= plan.t[i].wg0) ti = i;
    const DwTask T = plan.t[ti];
    const int chunk = blockIdx.x - T.wg0;
    switch (T.kind) {
        // dw_core.h dw_task4g<IA, JB4, WA, WB, WP>
        case 0: dw_task4g<4, 4, 2, 2, 1, 32, 32, SW_ROWS, SIREN_SAVED_ROWS>(T, chunk, work, saved, ld, partial, lds); break;   // 256 x 256
        // the narrow tasks on the 16 x 16 x 4 MFMA (dw_task4g16: 16 row groups per operand set)
        case 1: dw_task4g16<4, 1, 4, 1, 1, 32, 2, 3, SW_ROWS, SIREN_SAVED_ROWS>(T, chunk, work, saved, ld, partial, lds); break;   // 256 x 3 (x, y, z / direction)
        default: dw_task4g16<1, 4, 1, 4, 1, 2, 32, 3, SW_ROWS, SIREN_SAVED_ROWS>(T, chunk, work, saved, ld, partial, lds); break;  // 3 x 256 (heads)
    }
#ifdef NERFMI_TIMING
    if (threadIdx.x == 0) nerfmi_dbg_siren_dw[blockIdx.x] = __builtin_readcyclecounter() - t_start;
#endif
}

// the 256 x 256 tasks on the bf16 matrix cores (dw_bf16x3.h); the narrow tasks stay on the fp32 path
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
siren_dw_bf16x3_kernel(DwPlan plan, const float *__restrict__ work, const float *__restrict__ saved, int64_t ld,
                       float *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int ti = 0;
    for (int i = 1; i < plan.n_tasks; ++i)
        if ((int)blockIdx.x >= plan.t[i].wg0) ti = i;
    const DwTask T = plan.t[ti];
    const int chunk = blockIdx.x - T.wg0;
    switch (T.kind) {
        case 0: dw_task_bf16x3<8, 8, 4, 4, 2, SW_ROWS, SIREN_SAVED_ROWS>(T, chunk, work, saved, ld, partial, reinterpret_cast<char *>(lds)); break;
        case 1: dw_task<2, 1, 4, 1, SW_ROWS, SIREN_SAVED_ROWS>(T, chunk, work, saved, ld, partial, lds); break;
        default: dw_task<1, 2, 1, 4, SW_ROWS, SIREN_SAVED_ROWS>(T, chunk, work, saved, ld, partial, lds); break;
    }
}

// ---------------------------------------------------------------------------
// 3. slab reduction, one workgroup per (parameter row, layer group): fixed summation order (bit-reproducible), the FiLM
//    row scale of one-row launches, and the conditioning-row gradients (header of this file).
//    Layer groups: 0..7 = network.l (task l), 8 = color_layer_sine (tasks 8 + 9: direction and hidden columns),
//    9 = color_layer_linear (task 10), 10 = final_layer (task 11).
// ---------------------------------------------------------------------------
constexpr int SIREN_GROUPS = 11;

// W_g[j][col] read back from the packed forward image (siren_pack_kernel's map inverted)
__device__ __forceinline__ float packed_weight(const float *__restrict__ packed, int g, int j, int col) {
    int base, KB, kc;
    if (g == 0) { base = SOFF_L1; KB = 1; kc = col; }
    else if (g < 8) { base = SOFF_L2 + (g - 1) * SZ_HID; KB = 8; kc = col; }
    else { base = SOFF_COLOR; KB = 9; kc = (col < 3) ? col : 32 + (col - 3); }
    const int jb = j >> 5, kb = kc >> 5, r = kc & 31;
    const int lane = ((r >> 2) & 1) * 32 + (j & 31);
    return packed[base + ((jb * KB + kb) * 4 + (r >> 3)) * 256 + lane * 4 + (r & 3)];
}

__global__ void __launch_bounds__(256)
siren_dw_reduce_kernel(DwPlan plan, const float *__restrict__ partial, GradPtrs G, const float *__restrict__ packed,
                       const float *__restrict__ freq, int row_scale, float *__restrict__ d_freq,
                       float *__restrict__ d_phase) {
    const int j = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
    const int t_first = (g < 8) ? g : (g == 8 ? 8 : g + 1), n_t = (g == 8) ? 2 : 1;
    if (j >= plan.t[t_first].a_valid) return;
    const bool film = g <= 8;
    const float fr = (film && row_scale) ? __fadd_rn(__fmul_rn(freq[256 * g + j], 15.0f), 30.0f) : 1.0f;   // nerf.py:202
    float dot = 0.f, sb = 0.f;
    for (int ti = t_first; ti < t_first + n_t; ++ti) {
        const DwTask T = plan.t[ti];
        const int cols = T.KB * 32, slab = T.JB * 32 * (cols + 1);
        const float *src = partial + T.part_off + j * cols;
        for (int k = tid; k < cols; k += 256) {
            float s = 0.f;
            int c = 0;
            for (; c + 8 <= T.chunks; c += 8) {           // eight slab loads in flight instead of one HBM round trip per chunk
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(c + u) * slab + k];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += v[u];
            }
            for (; c < T.chunks; ++c) s += src[(int64_t)c * slab + k];
            if (k < T.b_valid) {
                G.p[T.param][j * T.in_f + T.out_col0 + k] = fr * s;
                if (film && d_freq) dot = __builtin_fmaf(packed_weight(packed, g, j, T.out_col0 + k), s, dot);
            }
        }
        if (T.bias_param >= 0) {                          // every thread: same addresses, one broadcast load per chunk
            const float *bsrc = partial + T.part_off + T.JB * 32 * cols + j;
            for (int c = 0; c < T.chunks; ++c) sb += bsrc[(int64_t)c * slab];
            if (tid == 0) G.p[T.bias_param][j] = fr * sb;
        }
    }
    if (!(film && d_freq)) return;
    // <W_j, (G^T X)_j> over the workgroup in a fixed tree
    __shared__ float red[4];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) dot += __shfl_xor(dot, m, WAVE);
    if ((tid & 63) == 0) red[tid >> 6] = dot;
    __syncthreads();
    if (tid == 0) {
        const float d = (red[0] + red[1]) + (red[2] + red[3]);
        d_freq[256 * g + j] = 15.0f * __builtin_fmaf(packed[SOFF_BIAS + 256 * g + j], sb, d);
        d_phase[256 * g + j] = sb;
    }
}

static const int SKIND_JB[3] = {8, 8, 1};
static const int SKIND_KB[3] = {8, 1, 8};

static DwPlan siren_plan(int64_t ld, bool fast = false) {
    DwPlan P;
    int n = 0;
    auto add = [&](int kind, int a_row0, int a_valid, int b_row0, int b_valid, int param, int col0, int in_f, int bias) {
        DwTask &t = P.t[n++];
        t.kind = kind; t.a_row0 = a_row0; t.a_valid = a_valid; t.b_row0 = b_row0; t.b_valid = b_valid;
        t.param = param; t.out_col0 = col0; t.in_f = in_f; t.bias_param = bias;
        t.wp = 1;                                       // one slab per workgroup in every form this plan uses
    };
    // B rows past b_valid are whatever follows in the tile (the GEMM's extra columns are dropped by the reduction)
    add(1, SW_DZ, 256, SS_X, 3, 0, 0, 3, 1);                                                    // network.0: X = warped xyz
    for (int l = 1; l < 8; ++l) add(0, SW_DZ + 256 * l, 256, SS_H + 256 * (l - 1), 256, 2 * l, 0, 256, 2 * l + 1);
    add(1, SW_DZ + 256 * 8, 256, SS_D, 3, 18, 0, 259, -1);                                      // colour layer, dir columns 0..2
    add(0, SW_DZ + 256 * 8, 256, SS_H + 256 * 7, 256, 18, 3, 259, 19);                          // colour layer, hidden columns
    add(2, SW_DRGB, 3, SS_HC, 256, 20, 0, 256, 21);                                             // color_layer_linear.0
    add(2, fast ? SW_DSIG : SW_DSIG4, 1, SS_H + 256 * 7, 256, 16, 0, 256, 17);                  // final_layer
    P.n_tasks = n;
    // 8 x 30 + 2 x 4 + 2 x 4 = 256 workgroups, one per CU (shares from per-task workgroup stamps, tools/exp_siren_dw_timing.py:
    // 71.6 / 7.5 / 8.4 M cycles per task of kinds 0..2 at 4096 tiles; {29,6,6}, {30,5,3}, {30,3,5} measured slower).  The four
    // narrow tasks cost 11.4-12.1 M cycles each while a tile of theirs took the latency of its loads; with the three-buffer ring
    // of dw_task4g16 they fit into 16 workgroups instead of 24.
    static const int chunks_default[3] = {30, 4, 4};
    // split-bf16 variant: the 256 x 256 tasks run ~2x faster per tile, so the fp32 narrow tasks get the larger share of CUs:
    // 8 x 24 + 2 x 16 + 2 x 16 = 256
    static const int chunks_fast[3] = {24, 16, 16};
    const int *chunks = fast ? chunks_fast : chunks_default;
#ifdef NERFMI_TIMING
    // experiment builds only (tools/exp_siren_dw_timing.py): "c0,c1,c2", validated -- the product never reads the environment
    int chunks_env[3];
    if (const char *e = getenv("NERFMI_SIREN_DW_CHUNKS")) {
        if (sscanf(e, "%d,%d,%d", chunks_env, chunks_env + 1, chunks_env + 2) == 3 && chunks_env[0] >= 1 && chunks_env[1] >= 1 &&
            chunks_env[2] >= 1 && 8 * chunks_env[0] + 2 * chunks_env[1] + 2 * chunks_env[2] <= 512)
            chunks = chunks_env;
    }
#endif
    dw_finish_plan(P, SKIND_JB, SKIND_KB, chunks, ld);
    return P;
}

}  // namespace nerfmi

using namespace nerfmi;

extern "C" {

#ifdef NERFMI_TIMING
int nerfmi_debug_timing_siren_dw(unsigned long long *host) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(nerfmi_dbg_siren_dw), sizeof(unsigned long long) * 512) == hipSuccess ? 0 : 1;
}
#endif

// + one dump tile for waves past the end (mlp_core.h RowImage)
size_t nerfmi_siren_saved_floats(int64_t n_points) {
    return (size_t)SIREN_SAVED_ROWS * (size_t)(siren_pad_points(n_points < 1 ? 1 : n_points) + 32);
}

size_t nerfmi_siren_backward_workspace_floats(int64_t n_points) {
    const int64_t ld = siren_pad_points(n_points < 1 ? 1 : n_points);
    const size_t p32 = dw_partial_floats(siren_plan(ld)), pfast = dw_partial_floats(siren_plan(ld, true));
    return (size_t)SW_ROWS * (size_t)(ld + 32) + (p32 > pfast ? p32 : pfast);
}

int nerfmi_siren_forward_rays_train(const float *packed, const float *rays, const float *z, const float *frequencies,
                                    const float *phase_shifts, int n_rays, int n_per_ray, int64_t rays_per_cond,
                                    float *out, float *saved, nerfmi_stream_t stream) {
    NERFMI_ENTER();
    NERFMI_REQUIRE(n_rays >= 0 && n_per_ray >= 1 && rays_per_cond >= 1, "siren_forward_rays_train: bad sizes");
    const int64_t n_points = (int64_t)n_rays * n_per_ray;
    if (n_points == 0) return NERFMI_OK;
    NERFMI_REQUIRE(packed && rays && z && frequencies && phase_shifts && out && saved, "siren_forward_rays_train: null pointer");
    const int64_t waves = (n_points + 31) / 32;
    const bool one_cond = rays_per_cond >= n_rays;
    KernelSpan span("siren_forward_kernel<save>", n_points, (hipStream_t)stream);
    SIREN_FORWARD_LAUNCH(true, false, true, dim3((unsigned)((waves + 3) / 4)), dim3(256),
                       (hipStream_t)stream, packed, rays, z, nullptr, nullptr, frequencies, phase_shifts, n_points,
                       n_per_ray, rays_per_cond * n_per_ray, out, saved, siren_pad_points(n_points));
    return check_launch("siren_forward_rays_train");
}

int nerfmi_siren_forward_points_train(const float *packed, const float *points, const float *ray_directions,
                                      const float *frequencies, const float *phase_shifts, int64_t n_points,
                                      int64_t points_per_cond, float *out, float *saved, nerfmi_stream_t stream) {
    NERFMI_ENTER();
    NERFMI_REQUIRE(n_points >= 0 && points_per_cond >= 1, "siren_forward_points_train: bad sizes");
    if (n_points == 0) return NERFMI_OK;
    NERFMI_REQUIRE(packed && points && ray_directions && frequencies && phase_shifts && out && saved,
                   "siren_forward_points_train: null pointer");
    const int64_t waves = (n_points + 31) / 32;
    const bool one_cond = points_per_cond >= n_points;
    SIREN_FORWARD_LAUNCH(false, false, true, dim3((unsigned)((waves + 3) / 4)), dim3(256),
                       (hipStream_t)stream, packed, nullptr, nullptr, points, ray_directions, frequencies, phase_shifts,
                       n_points, 1, points_per_cond, out, saved, siren_pad_points(n_points));
    return check_launch("siren_forward_points_train");
}

static int siren_backward_impl(const char *who, const float *packed, const void *fast, const float *saved,
                               const float *grad_out, const float *frequencies, int64_t n_points, int64_t points_per_cond,
                               float *const *grad_params, float *grad_frequencies, float *grad_phase_shifts,
                               float *workspace, nerfmi_stream_t stream) {
    NERFMI_ENTER();
    NERFMI_REQUIRE(n_points >= 1 && points_per_cond >= 1, "%s: bad sizes", who);
    NERFMI_REQUIRE(packed && saved && grad_out && frequencies && grad_params && workspace, "%s: null pointer", who);
    const bool one_cond = points_per_cond >= n_points;
    const bool want_cond = grad_frequencies || grad_phase_shifts;
    NERFMI_REQUIRE(!want_cond || (grad_frequencies && grad_phase_shifts),
                   "%s: grad_frequencies and grad_phase_shifts come together", who);
    NERFMI_REQUIRE(!want_cond || one_cond,
                   "%s: conditioning gradients need a launch that shares one conditioning row (points_per_cond >= n_points); "
                   "call once per row", who);
    const int64_t ld = siren_pad_points(n_points);
    GradPtrs G;
    for (int i = 0; i < N_PARAMS; ++i) G.p[i] = nullptr;
    for (int i = 0; i < SIREN_N_PARAMS; ++i) {
        NERFMI_REQUIRE(grad_params[i], "%s: grad_params[%d] is null", who, i);
        G.p[i] = grad_params[i];
    }
    hipStream_t st = (hipStream_t)stream;
    float *work = workspace;
    float *partial = workspace + (size_t)SW_ROWS * (ld + 32);
    const int64_t waves = (n_points + 31) / 32;
    const DwPlan P = siren_plan(ld, fast != nullptr);       // (the split-bf16 plan's slabs fit the fp32 plan's workspace)
    // split-bf16 / [row][point] tasks: two 73 728-B tile buffers; dw_task4g: two 65 536-B ones (> the 64 KiB default limit)
    const size_t lds = sizeof(float) * 2 * 512 * LROW;
    static_assert(2 * DWF_KSTEP_BYTES <= sizeof(float) * 2 * 512 * LROW, "the split-bf16 k-step buffers fit in the same allocation");
    static PerDeviceOnce attr_set;
    int attr_dev;
    if (attr_set.needed(attr_dev)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(siren_dw_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(siren_dw_bf16x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(siren_backward_chain_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, SIREN_WLDS_BYTES) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(siren_backward_chain_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, SIREN_WLDS_BYTES) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(siren_backward_chain_bf16x3_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, FLDS_BYTES) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(siren_backward_chain_bf16x3_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, FLDS_BYTES) != hipSuccess) {
            (void)hipGetLastError();
            set_error("%s: cannot raise the dynamic LDS limit", who);
            return NERFMI_E_LAUNCH;
        }
        attr_set.mark(attr_dev);
    }
    const dim3 cgrid((unsigned)((waves + 3) / 4));
    if (fast) {
        {
            KernelSpan span("siren_backward_chain_bf16x3_kernel", n_points, st);
            if (one_cond)
                hipLaunchKernelGGL(siren_backward_chain_bf16x3_kernel<true>, cgrid, dim3(256), FLDS_BYTES, st, packed,
                                   (const __bf16 *)fast, saved, grad_out, frequencies, n_points, points_per_cond, ld, work);
            else
                hipLaunchKernelGGL(siren_backward_chain_bf16x3_kernel<false>, cgrid, dim3(256), FLDS_BYTES, st, packed,
                                   (const __bf16 *)fast, saved, grad_out, frequencies, n_points, points_per_cond, ld, work);
        }
        KernelSpan span("siren_dw_bf16x3_kernel", n_points, st);
        hipLaunchKernelGGL(siren_dw_bf16x3_kernel, dim3(P.n_wg), dim3(256), lds, st, P, work, saved, ld, partial);
    } else {
        {
            KernelSpan span("siren_backward_chain_kernel", n_points, st);
            if (one_cond)   // the workspace images hold G = dZ / fr; the reduction scales the rows (header of this file)
                hipLaunchKernelGGL(siren_backward_chain_kernel<true>, cgrid, dim3(256), SIREN_WLDS_BYTES, st, packed, saved, grad_out, frequencies,
                                   n_points, points_per_cond, ld, work);
            else
                hipLaunchKernelGGL(siren_backward_chain_kernel<false>, cgrid, dim3(256), SIREN_WLDS_BYTES, st, packed, saved, grad_out,
                                   frequencies, n_points, points_per_cond, ld, work);
        }
        KernelSpan span("siren_dw_kernel", n_points, st);
        hipLaunchKernelGGL(siren_dw_kernel, dim3(P.n_wg), dim3(256), lds, st, P, work, saved, ld, partial);
    }
    {
        KernelSpan span("siren_dw_reduce_kernel", n_points, st);
        hipLaunchKernelGGL(siren_dw_reduce_kernel, dim3(256, SIREN_GROUPS), dim3(256), 0, st, P, partial, G, packed, frequencies,
                           one_cond ? 1 : 0, grad_frequencies, grad_phase_shifts);
    }
    return check_launch(who);
}

int nerfmi_siren_backward(const float *packed, const float *saved, const float *grad_out, const float *frequencies,
                          int64_t n_points, int64_t points_per_cond, float *const *grad_params, float *workspace,
                          nerfmi_stream_t stream) {
    return siren_backward_impl("siren_backward", packed, nullptr, saved, grad_out, frequencies, n_points, points_per_cond,
                               grad_params, nullptr, nullptr, workspace, stream);
}

int nerfmi_siren_backward_cond(const float *packed, const float *saved, const float *grad_out, const float *frequencies,
                               int64_t n_points, float *const *grad_params, float *grad_frequencies,
                               float *grad_phase_shifts, float *workspace, nerfmi_stream_t stream) {
    NERFMI_REQUIRE(grad_frequencies && grad_phase_shifts, "siren_backward_cond: null conditioning-gradient pointer");
    return siren_backward_impl("siren_backward_cond", packed, nullptr, saved, grad_out, frequencies, n_points, n_points,
                               grad_params, grad_frequencies, grad_phase_shifts, workspace, stream);
}

// The same backward with the dX chain and the 256 x 256 tasks of the dW GEMM on the bf16 matrix cores (opt-in split-bf16 math;
// `fast` = nerfmi_siren_pack_fast's image).  grad_frequencies / grad_phase_shifts: both NULL, or both set for a launch that
// shares one conditioning row (points_per_cond >= n_points).
int nerfmi_siren_backward_fast(const float *packed, const void *fast, const float *saved, const float *grad_out,
                               const float *frequencies, int64_t n_points, int64_t points_per_cond, float *const *grad_params,
                               float *grad_frequencies, float *grad_phase_shifts, float *workspace, nerfmi_stream_t stream) {
    NERFMI_REQUIRE(fast, "siren_backward_fast: null pointer");
    return siren_backward_impl("siren_backward_fast", packed, fast, saved, grad_out, frequencies, n_points, points_per_cond,
                               grad_params, grad_frequencies, grad_phase_shifts, workspace, stream);
}

}  // extern "C"
