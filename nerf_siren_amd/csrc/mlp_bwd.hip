// NeRF MLP backward for gfx950 (MI355X): autograd of models/nerf.py:83-124
// w.r.t. the 24 parameters (SURVEY section 8a backward contract).
//
//  1. nerf_backward_chain_kernel -- the dX chain.  Same register-resident scheme
//     as the forward (mlp_core.h): a wave owns 32 points, dZ_l^T lives in the
//     accumulator layout and is the B operand of  dH_{l-1}^T = W_l^T . dZ_l^T
//     (A = the transposed packed image).  ReLU masks come from the saved
//     post-activation images; every dZ_l is written to the workspace as a
//     tile-major [tile][unit][32 points] image (mlp_core.h RowImage).
//  2. nerf_dw_kernel -- dW_l = dZ_l^T . X_l, a (units x units x points) GEMM whose
//     contraction runs over POINTS.  Both operands are tile-major images, so a
//     32-point tile of A and B row ranges (contiguous, 128 B per row) is staged
//     through LDS and consumed by v_mfma_f32_32x32x2_f32; the point range is split
//     into chunks (split-K) and every workgroup writes its partial slab.
//  3. nerf_dw_reduce_kernel -- deterministic slab reduction into the (out,in)
//     gradient tensors (no float atomics: results are bit-reproducible).
//  1b / 2b. the same chain and GEMM on the opt-in split-bf16 math (bf16x3_core.h): same images in and out.
#include <stdlib.h>

#include "dw_bf16x3.h"

namespace nerfmi {

// workspace row map (tile-major images written by the chain kernel)
constexpr int W_DZ = 0;                  // 8 x 256: dZ of xyz_encoding_1..8 (masked by ReLU)
constexpr int W_DFINAL = 8 * 256;        // 256: d xyz_encoding_final output
constexpr int W_DDIR = W_DFINAL + 256;   // 128: dZ of dir_encoding
// head rows: [row][32 points] order in the split-bf16 chain (rows W_DRGB + c, W_DSIG); x4 order in the fp32 chain, where they sit
// in unit 0 of their own row groups (rows W_DRGB + 4c, W_DSIG4) for dw_task4g's narrow-A form (two pieces = 16 rows each)
constexpr int W_DRGB = W_DDIR + 128;     // d rgb pre-sigmoid
constexpr int W_DSIG = W_DRGB + 4;       // d sigma ([row][point] order)
constexpr int W_DSIG4 = W_DRGB + 16;     // d sigma (x4 order)
constexpr int W_ROWS = W_DRGB + 32;

// ---------------------------------------------------------------------------
// 1. dX chain
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
nerf_backward_chain_kernel(const float *__restrict__ packed, const float *__restrict__ saved,
                           const float *__restrict__ grad_out, int64_t n_points, int64_t ld,
                           float *__restrict__ work) {
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t p0 = wave * 32;
    const bool live = p0 < n_points;          // no early exit: the workgroup's waves share barriers (layer_mfma_lds)
    const int64_t praw = p0 + (lane & 31);
    const bool ok = praw < n_points;
    RowImage S, Wk;
    S.init(const_cast<float *>(saved), wave, ld / 32, SAVED_ROWS, lane, ok, live);
    Wk.init(work, wave, ld / 32, W_ROWS, lane, ok, live);

    float4 go = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) go = reinterpret_cast<const float4 *>(grad_out)[praw];
    // rgb = sigmoid(pre): d pre = d rgb * rgb * (1 - rgb)      (nerf.py:78-80)
    float dpre[3];
    {
        const float g3[3] = {go.x, go.y, go.z};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float rgb = *S.at(S_RGB + c);
            dpre[c] = g3[c] * rgb * (1.0f - rgb);
            if (half == 0) *at4(Wk, W_DRGB + 4 * c) = dpre[c];
        }
        if (half == 0) *at4(Wk, W_DSIG4) = go.w;
    }
    const float dsig = go.w;

    // two gradient buffers used alternately (a layer reads one, its epilogue writes the other: no copies)
    f32x16 dzA[8], dzB[8];
    unsigned mk[4];
    // d dir_h = W_rgb^T d pre, masked by the saved ReLU sign bits  (nerf.py:119-120)
    load_mask(S, 8, mk);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        f32x16 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 w[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) w[c] = ldg4(packed + OFF_W_RGB + 128 * c + 32 * b + 8 * q + 4 * half);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float s = __builtin_fmaf(w[2][t], dpre[2], __builtin_fmaf(w[1][t], dpre[1], w[0][t] * dpre[0]));
                v[4 * q + t] = mask_keep(mk, b, q, t, s);
            }
        }
        dzA[b] = v;
        store_block4(Wk, W_DDIR + 32 * b, v);
    }
    __shared__ __attribute__((aligned(16))) float wlds[WLDS_FLOATS];
    const int wid = threadIdx.x >> 6;
    WeightStage ws;
    auto zero_pre = [](int) { return 0; };
    // d final = W_dir[:, :256]^T dZ_dir                          (nerf.py:116-118; no activation on final)
    layer_mfma_lds<4, 0, 8, 0, true>(packed + OFF_TDIR, nullptr, dzA, nullptr, dzB, zero_pre,
                            [&Wk](int jb, int q, f32x4 c, int) {
                                store_slice4(Wk, W_DFINAL + 32 * jb, q, c);
                                return c;
                            }, wlds, ws, wid, lane);
    // d h8 = W_final^T d final + w_sigma d sigma, masked by h8 > 0 (nerf.py:112-116)
    load_mask(S, 7, mk);
    layer_mfma_lds<8, 0, 8, 0, false>(packed + OFF_TFINAL, nullptr, dzB, nullptr, dzA, zero_pre,
                             [&](int jb, int q, f32x4 c, int) {
                                 const f32x4 w = ldg4(packed + OFF_W_SIGMA + 32 * jb + 8 * q + 4 * half);
#pragma unroll
                                 for (int t = 0; t < 4; ++t) {
                                     const float v = __builtin_fmaf(w[t], dsig, c[t]);
                                     c[t] = mask_keep(mk, jb, q, t, v);
                                 }
                                 store_slice4(Wk, W_DZ + 7 * 256 + 32 * jb, q, c);
                                 return c;
                             }, wlds, ws, wid, lane);
    // xyz_encoding_8 .. xyz_encoding_2: d h_{l-1} = W_l[:, hidden]^T dZ_l, masked by h_{l-1} > 0
    auto back = [&](int li, const f32x16 *in, f32x16 *out_dz) __attribute__((always_inline)) {
        const int wrow = W_DZ + (li - 1) * 256;
        load_mask(S, li - 1, mk);
        layer_mfma_lds<8, 0, 8, 0, false>(packed + OFF_T8 + (7 - li) * SZ_HID, nullptr, in, nullptr, out_dz, zero_pre,
                                 [&](int jb, int q, f32x4 c, int) {
#pragma unroll
                                     for (int t = 0; t < 4; ++t) c[t] = mask_keep(mk, jb, q, t, c[t]);
                                     store_slice4(Wk, wrow + 32 * jb, q, c);
                                     return c;
                                 }, wlds, ws, wid, lane);
    };
    back(7, dzA, dzB);
    back(6, dzB, dzA);
    back(5, dzA, dzB);
    back(4, dzB, dzA);
    back(3, dzA, dzB);
    back(2, dzB, dzA);
    back(1, dzA, dzB);
}

// ---------------------------------------------------------------------------
// 1b. dX chain on the bf16 matrix cores (opt-in split-bf16 math, bf16x3_core.h): same images in, same images out.
//     A layer's raw output is finished -- ReLU-masked with the forward's sign bits, stored as the dZ image the dW GEMM
//     reads -- by the layer that CONSUMES it, pair by pair between its MFMAs; d h8 starts from w_sigma * d sigma.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
nerf_backward_chain_bf16x3_kernel(const float *__restrict__ packed, const __bf16 *__restrict__ fast,
                                  const float *__restrict__ saved, const float *__restrict__ grad_out, int64_t n_points,
                                  int64_t ld, float *__restrict__ work) {
    extern __shared__ __attribute__((aligned(16))) char wlds_fast[];
    const int lane = threadIdx.x & 63, half = lane >> 5, wid = threadIdx.x >> 6;
    const int64_t wave = (int64_t)blockIdx.x * 4 + wid;
    const int64_t p0 = wave * 32;
    const bool live = p0 < n_points;
    const int64_t praw = p0 + (lane & 31);
    const bool ok = praw < n_points;
    RowImage S, Wk;
    S.init(const_cast<float *>(saved), wave, ld / 32, SAVED_ROWS, lane, ok, live);
    Wk.init(work, wave, ld / 32, W_ROWS, lane, ok, live);

    float4 go = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) go = reinterpret_cast<const float4 *>(grad_out)[praw];
    float dpre[3];
    {
        const float g3[3] = {go.x, go.y, go.z};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float rgb = *S.at(S_RGB + c);
            dpre[c] = g3[c] * rgb * (1.0f - rgb);
            if (half == 0) *Wk.at(W_DRGB + c) = dpre[c];
        }
        if (half == 0) *Wk.at(W_DSIG) = go.w;
    }
    const float dsig = go.w;

    f32x16 dzA[8], dzB[8];
    unsigned mk[4];
    load_mask(S, 8, mk);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        f32x16 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 w[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) w[c] = ldg4(packed + OFF_W_RGB + 128 * c + 32 * b + 8 * q + 4 * half);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float s = __builtin_fmaf(w[2][t], dpre[2], __builtin_fmaf(w[1][t], dpre[1], w[0][t] * dpre[0]));
                v[4 * q + t] = mask_keep(mk, b, q, t, s);
            }
        }
        dzA[b] = v;
        store_block(Wk, W_DDIR + 32 * b, v);
    }
    FastStage fs;
    NoHook none;
    // consuming hooks: store the pair's rows of the dZ image at `row0`, optionally after the ReLU mask
    auto put = [&Wk](int row0, int kb, int pr, float x0, float x1) {
        const int r = 2 * pr;
        float *dst = Wk.tile + (row0 + 32 * kb + 8 * (r >> 2) + 4 * (Wk.lane >> 5) + (r & 3)) * 32 + (Wk.lane & 31);
        __builtin_nontemporal_store(x0, dst);
        __builtin_nontemporal_store(x1, dst + 32);
    };
    auto plain = [&](int row0) {
        return [&put, row0](int kb, int pr, float &x0, float &x1) { put(row0, kb, pr, x0, x1); };
    };
    auto masked = [&](int row0) {
        return [&put, &mk, row0](int kb, int pr, float &x0, float &x1) {
            const int r = 2 * pr;
            x0 = mask_keep(mk, kb, r >> 2, r & 3, x0);
            x1 = mask_keep(mk, kb, r >> 2, (r & 3) + 1, x1);
            put(row0, kb, pr, x0, x1);
        };
    };
    auto timg = [&](int off) { return fast + fast_t_elems(off); };
    // d final = W_dir[:, :256]^T dZ_dir   (dZ_dir is already masked and stored above)
    layer_bf16x3<4, 0, 8, false, true>(timg(OFF_TDIR), nullptr, dzA, nullptr, dzB, wlds_fast, fs, wid, lane);
    // d h8 = W_final^T d final + w_sigma d sigma: consumes d final (stored as it goes, no activation on final)
    layer_bf16x3<8, 0, 8, false, false>(timg(OFF_TFINAL), packed + OFF_W_SIGMA + 4 * half, dzB, nullptr, dzA, wlds_fast, fs,
                                        wid, lane, plain(W_DFINAL), none, dsig);
    // xyz_encoding_8 .. 2: the layer consuming d h_l masks it with h_l > 0 and stores it as dZ_l
    load_mask(S, 7, mk);
    layer_bf16x3<8, 0, 8, false, false>(timg(OFF_T8), nullptr, dzA, nullptr, dzB, wlds_fast, fs, wid, lane, masked(W_DZ + 7 * 256));
    load_mask(S, 6, mk);
    layer_bf16x3<8, 0, 8, false, false>(timg(OFF_T7), nullptr, dzB, nullptr, dzA, wlds_fast, fs, wid, lane, masked(W_DZ + 6 * 256));
    load_mask(S, 5, mk);
    layer_bf16x3<8, 0, 8, false, false>(timg(OFF_T6), nullptr, dzA, nullptr, dzB, wlds_fast, fs, wid, lane, masked(W_DZ + 5 * 256));
    load_mask(S, 4, mk);
    layer_bf16x3<8, 0, 8, false, false>(timg(OFF_T5), nullptr, dzB, nullptr, dzA, wlds_fast, fs, wid, lane, masked(W_DZ + 4 * 256));
    load_mask(S, 3, mk);
    layer_bf16x3<8, 0, 8, false, false>(timg(OFF_T4), nullptr, dzA, nullptr, dzB, wlds_fast, fs, wid, lane, masked(W_DZ + 3 * 256));
    load_mask(S, 2, mk);
    layer_bf16x3<8, 0, 8, false, false>(timg(OFF_T3), nullptr, dzB, nullptr, dzA, wlds_fast, fs, wid, lane, masked(W_DZ + 2 * 256));
    load_mask(S, 1, mk);
    layer_bf16x3<8, 0, 8, false, false>(timg(OFF_T2), nullptr, dzA, nullptr, dzB, wlds_fast, fs, wid, lane, masked(W_DZ + 1 * 256));
    // d h1 has no consumer in the chain (the inputs carry no gradient): mask and store it here
    load_mask(S, 0, mk);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        f32x16 v = dzB[b];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = mask_keep(mk, b, r >> 2, r & 3, v[r]);
        store_block(Wk, W_DZ + 32 * b, v);
    }
}

// 2. dW = dZ^T . X over points: dw_core.h (shared with the FiLM-SIREN backward, siren_bwd.hip)

// 2b. the 256 x 256 tasks of the dW GEMM on the bf16 matrix cores (opt-in split-bf16 math): dw_bf16x3.h

#ifdef NERFMI_TIMING
__device__ unsigned long long nerfmi_dbg_dw[512];        // per-workgroup shader-clock duration (experiment builds)
#endif

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
nerf_dw_bf16x3_kernel(DwPlan plan, const float *__restrict__ work, const float *__restrict__ saved, int64_t ld,
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
        case 0: dw_task_bf16x3<8, 8, 4, 4, 2, W_ROWS, SAVED_ROWS>(T, chunk, work, saved, ld, partial, reinterpret_cast<char *>(lds)); break;
        case 1: dw_task_bf16x3<8, 2, 2, 2, 1, W_ROWS, SAVED_ROWS>(T, chunk, work, saved, ld, partial, reinterpret_cast<char *>(lds)); break;
        case 2: dw_task_bf16x3<4, 8, 2, 4, 2, W_ROWS, SAVED_ROWS>(T, chunk, work, saved, ld, partial, reinterpret_cast<char *>(lds)); break;
        case 3: dw_task<1, 1, 4, 1, W_ROWS, SAVED_ROWS>(T, chunk, work, saved, ld, partial, lds); break;
        case 4: dw_task<1, 1, 1, 4, W_ROWS, SAVED_ROWS>(T, chunk, work, saved, ld, partial, lds); break;
        default: dw_task<1, 2, 1, 4, W_ROWS, SAVED_ROWS>(T, chunk, work, saved, ld, partial, lds); break;
    }
#ifdef NERFMI_TIMING
    if (threadIdx.x == 0) nerfmi_dbg_dw[blockIdx.x] = __builtin_readcyclecounter() - t_start;
#endif
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
nerf_dw_kernel(DwPlan plan, const float *__restrict__ work, const float *__restrict__ saved, int64_t ld,
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
        // dw_core.h dw_task4g<IA, JB4, WA, WB, WP, NA_P, NB_P> / dw_task4g16<..., NBUF> on the x4 images of the fp32 forward / chain
        case 0: dw_task4g<4, 4, 2, 2, 1, 32, 32, W_ROWS, SAVED_ROWS>(T, chunk, work, saved, ld, partial, lds); break;   // 256 x 256
        case 1: dw_task4g16<4, 4, 4, 1, 1, 32, 8, 3, W_ROWS, SAVED_ROWS>(T, chunk, work, saved, ld, partial, lds); break;  // 256 x 63 (xyz embedding)
        case 2: dw_task4g<4, 4, 1, 2, 2, 16, 32, W_ROWS, SAVED_ROWS>(T, chunk, work, saved, ld, partial, lds); break;   // 128 x 256
        case 3: dw_task4g16<4, 4, 2, 1, 2, 16, 4, 5, W_ROWS, SAVED_ROWS>(T, chunk, work, saved, ld, partial, lds); break;  // 128 x 27 (dir embedding)
        case 4: dw_task4g16<1, 4, 1, 2, 2, 2, 16, 5, W_ROWS, SAVED_ROWS>(T, chunk, work, saved, ld, partial, lds); break;  // rgb 3 x 128
        default: dw_task4g16<1, 4, 1, 4, 1, 2, 32, 3, W_ROWS, SAVED_ROWS>(T, chunk, work, saved, ld, partial, lds); break; // sigma 1 x 256
    }
#ifdef NERFMI_TIMING
    if (threadIdx.x == 0) nerfmi_dbg_dw[blockIdx.x] = __builtin_readcyclecounter() - t_start;
#endif
}

// 3. slab reduction: dw_core.h

// slab = 32 JB rows x 32 KB columns per kind: fp32 plan (dw_task4g forms above) / split-bf16 plan ([row][point] tasks)
static const int KIND_JB[6] = {8, 8, 4, 4, 1, 1};
static const int KIND_KB[6] = {8, 2, 8, 2, 4, 8};
static const int KIND_WP[6] = {1, 1, 2, 2, 2, 1};      // slabs per workgroup (point ranges of a tile handled by different waves)
static const int KIND_KB_FAST[6] = {8, 2, 8, 1, 4, 8};

static DwPlan make_plan(int64_t ld, bool fast = false) {
    DwPlan P;
    int n = 0;
    auto add = [&](int kind, int a_row0, int a_valid, int b_row0, int b_valid, int param, int col0, int in_f, int bias) {
        DwTask &t = P.t[n++];
        t.kind = kind; t.a_row0 = a_row0; t.a_valid = a_valid; t.b_row0 = b_row0; t.b_valid = b_valid;
        t.param = param; t.out_col0 = col0; t.in_f = in_f; t.bias_param = bias; t.wp = fast ? 1 : KIND_WP[kind];
        t.JB = KIND_JB[kind]; t.KB = fast ? KIND_KB_FAST[kind] : KIND_KB[kind];
    };
    add(1, W_DZ, 256, S_EMB, 63, 0, 0, 63, 1);                                       // xyz_encoding_1
    for (int li = 1; li <= 3; ++li) add(0, W_DZ + 256 * li, 256, S_H + 256 * (li - 1), 256, 2 * li, 0, 256, 2 * li + 1);
    add(1, W_DZ + 256 * 4, 256, S_EMB, 63, 8, 0, 319, -1);                           // xyz_encoding_5, skip part
    add(0, W_DZ + 256 * 4, 256, S_H + 256 * 3, 256, 8, 63, 319, 9);                  // xyz_encoding_5, hidden part
    for (int li = 5; li <= 7; ++li) add(0, W_DZ + 256 * li, 256, S_H + 256 * (li - 1), 256, 2 * li, 0, 256, 2 * li + 1);
    add(0, W_DFINAL, 256, S_H + 256 * 7, 256, 16, 0, 256, 17);                       // xyz_encoding_final
    add(2, W_DDIR, 128, S_FINAL, 256, 18, 0, 283, 19);                               // dir_encoding, final part
    add(3, W_DDIR, 128, S_DEMB, 27, 18, 256, 283, -1);                               // dir_encoding, dir-emb part
    add(4, W_DRGB, 3, S_DIRH, 128, PARAM_RGB_W, 0, 128, PARAM_RGB_B);                // rgb
    add(5, fast ? W_DSIG : W_DSIG4, 1, S_H + 256 * 7, 256, PARAM_SIGMA_W, 0, 256, PARAM_SIGMA_B);     // sigma
    P.n_tasks = n;
    // chunks proportional to the measured cost of a task (per-workgroup shader-clock stamps, tools/exp_dw_timing.py ... fp32:
    // 71.7 / 21.3 / 36.9 / 12.6 / 9.1 / 12.5 M cycles per task of kinds 0..5 at 4096 tiles), so that every workgroup costs about
    // the same and the grid is exactly 256 (one per CU: the LDS tiles of a workgroup fill a CU, so the kernel takes one
    // workgroup's time, and leaving CUs without a chunk costs their share outright): 8 x 26 + 2 x 9 + 14 + 6 + 4 + 6 = 256.
    // The 256 x 256 tasks set the time (2.76 M cycles per workgroup, 93 % of it MFMAs); the two heads are bound by the latency
    // of their tile loads, not by their 16-32 MFMAs per tile.  (With the embedding tasks in the 32-lane form, half / a quarter of
    // whose B lanes are empty, those cost 36.6 / 18.5 M cycles and the best split was 8 x 25 + ...: kernel +4 %.)
    static const int base_fp32[6] = {26, 9, 14, 6, 4, 6};
    // split-bf16 variant: the 256 x 256 tasks run ~2x faster per tile, so the fp32 tasks get the larger share of CUs:
    // 8 x 24 + 2 x 12 + 18 + 6 + 6 + 10 = 256, from measured per-task workgroup times (tools/exp_dw_timing.py:
    // 38.2 / 18.5 / 25.5 / 8.4 / 8.5 / 12.6 M cycles per task at 4096 tiles); the tiny tasks 3..5 stay on the fp32 path
    static const int base_fast[6] = {24, 12, 18, 6, 6, 10};
    const int *base = fast ? base_fast : base_fp32;
#ifdef NERFMI_TIMING
    // experiment builds only (tools/exp_dw_timing.py): "c0,c1,c2,c3,c4,c5", validated -- the product never reads the environment
    int base_env[6];
    if (const char *e = getenv("NERFMI_DW_CHUNKS")) {
        if (sscanf(e, "%d,%d,%d,%d,%d,%d", base_env, base_env + 1, base_env + 2, base_env + 3, base_env + 4, base_env + 5) == 6) {
            bool ok = true;
            for (int i = 0; i < 6; ++i) ok = ok && base_env[i] >= 1 && base_env[i] <= 64;
            if (ok) base = base_env;
        }
    }
#endif
    const int64_t tiles = ld / 32;
    int wg = 0, off = 0;
    for (int i = 0; i < n; ++i) {
        DwTask &t = P.t[i];
        int c = base[t.kind];
        if (c > tiles) c = (int)(tiles < 1 ? 1 : tiles);
        t.chunks = c * t.wp; t.wg0 = wg; t.part_off = off;
        wg += c;
        off += c * t.wp * (t.JB * 32 * (t.KB * 32 + 1));
    }
    P.n_wg = wg;
    return P;
}

static size_t plan_partial_floats(const DwPlan &P) {
    const DwTask &t = P.t[P.n_tasks - 1];
    return (size_t)t.part_off + (size_t)t.chunks * (t.JB * 32 * (t.KB * 32 + 1));
}

static inline int64_t pad_points(int64_t n) { return (n + 31) / 32 * 32; }

}  // namespace nerfmi

using namespace nerfmi;

extern "C" {

size_t nerfmi_nerf_backward_workspace_floats(int64_t n_points) {
    const int64_t ld = pad_points(n_points < 1 ? 1 : n_points);
    const size_t pa = plan_partial_floats(make_plan(ld, false)), pb = plan_partial_floats(make_plan(ld, true));
    return (size_t)W_ROWS * (size_t)(ld + 32) + (pa > pb ? pa : pb);   // + the dump tile (mlp_core.h RowImage)
}

#ifdef NERFMI_TIMING
extern "C" int nerfmi_debug_timing_dw(unsigned long long *host) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(nerfmi_dbg_dw), sizeof(unsigned long long) * 512) == hipSuccess ? 0 : 1;
}
#endif

static int backward_impl(const char *who, const float *packed, const void *fast, int n_rays, int n_per_ray,
                         const float *saved, const float *grad_out, float *const *grad_params, float *workspace,
                         nerfmi_stream_t stream) {
    NERFMI_ENTER();
    NERFMI_REQUIRE(n_rays >= 1 && n_per_ray >= 1, "%s: bad sizes", who);
    NERFMI_REQUIRE(packed && saved && grad_out && grad_params && workspace, "%s: null pointer", who);
    const int64_t n_points = (int64_t)n_rays * n_per_ray;
    const int64_t ld = pad_points(n_points);
    GradPtrs G;
    for (int i = 0; i < N_PARAMS; ++i) {
        NERFMI_REQUIRE(grad_params[i], "%s: grad_params[%d] is null", who, i);
        G.p[i] = grad_params[i];
    }
    hipStream_t st = (hipStream_t)stream;
    float *work = workspace;
    float *partial = workspace + (size_t)W_ROWS * (ld + 32);
    const int64_t waves = (n_points + 31) / 32;
    const DwPlan P = make_plan(ld, fast != nullptr);
    const size_t lds = sizeof(float) * 2 * 512 * LROW;  // two 73 728-B tile buffers (> the 64 KiB default dynamic-LDS limit)
    static_assert(2 * DWF_KSTEP_BYTES <= sizeof(float) * 2 * 512 * LROW, "the split-bf16 k-step buffers fit in the same allocation");
    static PerDeviceOnce lds_attr_set;
    int attr_dev;
    if (lds_attr_set.needed(attr_dev)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(nerf_dw_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(nerf_dw_bf16x3_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(nerf_backward_chain_bf16x3_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, FLDS_BYTES) != hipSuccess) {
            (void)hipGetLastError();
            set_error("%s: cannot raise the dynamic LDS limit", who);
            return NERFMI_E_LAUNCH;
        }
        lds_attr_set.mark(attr_dev);
    }
    {
        KernelSpan span(fast ? "nerf_backward_chain_bf16x3_kernel" : "nerf_backward_chain_kernel", n_points, st);
        if (fast)
            hipLaunchKernelGGL(nerf_backward_chain_bf16x3_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), FLDS_BYTES, st,
                               packed, (const __bf16 *)fast, saved, grad_out, n_points, ld, work);
        else
            hipLaunchKernelGGL(nerf_backward_chain_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, packed, saved,
                               grad_out, n_points, ld, work);
    }
    {
        KernelSpan span(fast ? "nerf_dw_bf16x3_kernel" : "nerf_dw_kernel", n_points, st);
        if (fast)
            hipLaunchKernelGGL(nerf_dw_bf16x3_kernel, dim3(P.n_wg), dim3(256), lds, st, P, work, saved, ld, partial);
        else
            hipLaunchKernelGGL(nerf_dw_kernel, dim3(P.n_wg), dim3(256), lds, st, P, work, saved, ld, partial);
    }
    {
        KernelSpan span("dw_reduce_kernel<nerf>", n_points, st);
        hipLaunchKernelGGL(dw_reduce_kernel<0>, dim3(128, P.n_tasks), dim3(256), 0, st, P, partial, G);
    }
    return check_launch(who);
}

int nerfmi_nerf_backward_rays(const float *packed, const float *rays, const float *z, int n_rays, int n_per_ray,
                              const float *saved, const float *grad_out, float *const *grad_params,
                              float *workspace, nerfmi_stream_t stream) {
    (void)rays; (void)z;   // inputs carry no gradient (rendering.py:54, :244); kept for ABI symmetry
    return backward_impl("nerf_backward_rays", packed, nullptr, n_rays, n_per_ray, saved, grad_out, grad_params, workspace,
                         stream);
}

int nerfmi_nerf_backward_rays_fast(const float *packed, const void *fast, int n_rays, int n_per_ray, const float *saved,
                                   const float *grad_out, float *const *grad_params, float *workspace,
                                   nerfmi_stream_t stream) {
    NERFMI_REQUIRE(fast, "nerf_backward_rays_fast: null pointer");
    return backward_impl("nerf_backward_rays_fast", packed, fast, n_rays, n_per_ray, saved, grad_out, grad_params, workspace,
                         stream);
}

}  // extern "C"
