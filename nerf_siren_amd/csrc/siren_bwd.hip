// FiLM-SIREN field, training path for gfx950 (MI355X): autograd of models/nerf.py:142-151 (FiLMLayer) and :201-216
// (SemanticNeRF.forward_with_frequencies_phase_shifts) w.r.t. the 22 parameters.
//
//  0. siren_forward_kernel<., false, SAVE = true> (siren_core.h): the forward that also writes, per 32-point tile, the
//     layer inputs the dW GEMM needs (the sines themselves) and one sign bit of cos(arg) per unit.
//  1. siren_backward_chain_kernel -- the dX chain.  Same register-resident scheme as the NeRF chain (mlp_bwd.hip): a
//     wave owns 32 points, dZ_l^T lives in the accumulator layout and is the B operand of
//     dH_{l-1}^T = W_l^T . dZ_l^T (A = the transposed packed image).  The activation derivative is
//         d sin(fr * pre + ph) / d pre = fr * cos(arg),   |cos(arg)| = sqrt(1 - s^2),  s = the saved sine,
//     with the sign from the forward's bitmask: nothing but the sines is re-read.  Every dZ_l goes to the workspace as
//     a tile-major image.
//  2. dW_l = dZ_l^T . X_l: the NeRF dW GEMM (dw_core.h) on a 12-task plan of exactly 256 workgroups.
//  3. deterministic slab reduction (dw_core.h): bit-reproducible gradients.
#include <stdlib.h>

#include "dw_core.h"
#include "siren_core.h"

namespace nerfmi {

// workspace row map (tile-major images written by the chain kernel)
constexpr int SW_DZ = 0;                   // 9 x 256: dZ of network.0..7, then color_layer_sine
constexpr int SW_DRGB = 9 * 256;           // 3 (+1 pad): d rgb pre-sigmoid
constexpr int SW_DSIG = SW_DRGB + 4;       // 1 (+3 pad): d sigma
constexpr int SW_ROWS = SW_DSIG + 4;

// the 16 saved sines of block (row0 .. row0+31) that belong to this lane's accumulator registers
__device__ __forceinline__ f32x16 load_block(const RowImage &im, int row0) {
    f32x16 v;
    const float *src = im.tile + (row0 + 4 * (im.lane >> 5)) * 32 + (im.lane & 31);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int t = 0; t < 4; ++t) v[4 * q + t] = __builtin_nontemporal_load(src + (8 * q + t) * 32);   // read once: keep L2 for the weights
    return v;
}

// d pre = d h * fr * cos(arg) for the four units of slice q of block jb (layer `layer`):
// s = saved sines (this lane's 16 of the block), sign bits of the cosines in mk, fr from LDS (COND_LDS) or memory
template <bool COND_LDS>
__device__ __forceinline__ f32x4 film_grad(f32x4 dh, const f32x16 &s, const unsigned (&mk)[4], const float *fq,
                                           const float *lfr, int layer, int jb, int q) {
    f32x4 fr;
    if (COND_LDS) {
        fr = *reinterpret_cast<const f32x4 *>(lfr + 256 * layer + 32 * jb + 8 * q);
    } else {
        const f32x4 f = ldg4(fq + 256 * layer + 32 * jb + 8 * q);
#pragma unroll
        for (int t = 0; t < 4; ++t) fr[t] = __fadd_rn(__fmul_rn(f[t], 15.0f), 30.0f);                  // nerf.py:202
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float sv = s[4 * q + t];
        const float c2 = __builtin_fmaf(-sv, sv, 1.0f);         // cos^2 = 1 - s^2, one rounding (no worse than the sine's own)
        // the raw v_sqrt_f32 (1 ulp): sqrtf() is expanded to the correctly rounded sequence (scale, refine, classify:
        // ~14 instructions) which tripled this epilogue's vector work -- and |cos| feeds a product, not a parity check
        const float ca = __builtin_amdgcn_sqrtf(fmaxf(c2, 0.f));
        // the cosine's sign: bit (sh + t) of the mask word moved to bit 31 and OR-ed in (v_lshlrev + v_and_or)
        const unsigned sgn = (mk[jb >> 1] << (31 - (16 * (jb & 1) + 4 * q + t))) & 0x80000000u;
        dh[t] = dh[t] * (fr[t] * __uint_as_float(__float_as_uint(ca) | sgn));
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
            if (half == 0) *Wk.at(SW_DRGB + c) = dpre[c];
        }
        if (half == 0) *Wk.at(SW_DSIG) = go.w;
    }
    const float dsig = go.w;

    f32x16 dzA[8], dzB[8];
    unsigned mk[4];
    // d h_c = W_rgb^T d pre;  dZ_c = d h_c * fr_8 * cos(arg_c)          (nerf.py:213-214)
    load_mask_row(S, SS_MASK + 8 * 8, mk);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const f32x16 sv = load_block(S, SS_HC + 32 * b);
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
            dh = film_grad<COND_LDS>(dh, sv, mk, fq, lfr, 8, b, q);
#pragma unroll
            for (int t = 0; t < 4; ++t) v[4 * q + t] = dh[t];
        }
        dzA[b] = v;
        store_block(Wk, SW_DZ + 8 * 256 + 32 * b, v);
    }
    __shared__ __attribute__((aligned(16))) float wlds[WLDS_FLOATS];
    const int wid = threadIdx.x >> 6;
    WeightStage ws;
    // d h_7 = W_c[:, 3:]^T dZ_c + w_sigma d sigma;  dZ_7 = d h_7 * fr_7 * cos(arg_7)      (nerf.py:212-213)
    load_mask_row(S, SS_MASK + 8 * 7, mk);
    layer_mfma_lds<8, 0, 8, 0, true>(packed + SOFF_TCOLOR, nullptr, dzA, nullptr, dzB,
                                     [&S](int jb) { return load_block(S, SS_H + 7 * 256 + 32 * jb); },
                                     [&](int jb, int q, f32x4 c, const f32x16 &sv) {
                                         const f32x4 w = ldg4(packed + SOFF_W_SIGMA + 32 * jb + 8 * q + 4 * half);
#pragma unroll
                                         for (int t = 0; t < 4; ++t) c[t] = __builtin_fmaf(w[t], dsig, c[t]);
                                         c = film_grad<COND_LDS>(c, sv, mk, fq, lfr, 7, jb, q);
                                         store_slice(Wk, SW_DZ + 7 * 256 + 32 * jb, q, c);
                                         return c;
                                     }, wlds, ws, wid, lane);
    // network.7 .. network.1: d h_{l-1} = W_l^T dZ_l;  dZ_{l-1} = d h_{l-1} * fr_{l-1} * cos(arg_{l-1})
    auto back = [&](int l, const f32x16 *in, f32x16 *out_dz) __attribute__((always_inline)) {
        load_mask_row(S, SS_MASK + 8 * (l - 1), mk);
        layer_mfma_lds<8, 0, 8, 0, false>(packed + SOFF_T7 + (7 - l) * SZ_HID, nullptr, in, nullptr, out_dz,
                                          [&S, l](int jb) { return load_block(S, SS_H + (l - 1) * 256 + 32 * jb); },
                                          [&, l](int jb, int q, f32x4 c, const f32x16 &sv) {
                                              c = film_grad<COND_LDS>(c, sv, mk, fq, lfr, l - 1, jb, q);
                                              store_slice(Wk, SW_DZ + (l - 1) * 256 + 32 * jb, q, c);
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
        case 0: dw_task<2, 8, 4, 1, SW_ROWS, SIREN_SAVED_ROWS>(T, chunk, work, saved, ld, partial, lds); break;   // 256 x 256
        case 1: dw_task<2, 1, 4, 1, SW_ROWS, SIREN_SAVED_ROWS>(T, chunk, work, saved, ld, partial, lds); break;   // 256 x 32 (3 real columns)
        default: dw_task<1, 2, 1, 4, SW_ROWS, SIREN_SAVED_ROWS>(T, chunk, work, saved, ld, partial, lds); break;  // 32 x 256
    }
#ifdef NERFMI_TIMING
    if (threadIdx.x == 0) nerfmi_dbg_siren_dw[blockIdx.x] = __builtin_readcyclecounter() - t_start;
#endif
}

static const int SKIND_JB[3] = {8, 8, 1};
static const int SKIND_KB[3] = {8, 1, 8};

static DwPlan siren_plan(int64_t ld) {
    DwPlan P;
    int n = 0;
    auto add = [&](int kind, int a_row0, int a_valid, int b_row0, int b_valid, int param, int col0, int in_f, int bias) {
        DwTask &t = P.t[n++];
        t.kind = kind; t.a_row0 = a_row0; t.a_valid = a_valid; t.b_row0 = b_row0; t.b_valid = b_valid;
        t.param = param; t.out_col0 = col0; t.in_f = in_f; t.bias_param = bias;
    };
    // B rows past b_valid are whatever follows in the tile (the GEMM's extra columns are dropped by the reduction)
    add(1, SW_DZ, 256, SS_X, 3, 0, 0, 3, 1);                                                    // network.0: X = warped xyz
    for (int l = 1; l < 8; ++l) add(0, SW_DZ + 256 * l, 256, SS_H + 256 * (l - 1), 256, 2 * l, 0, 256, 2 * l + 1);
    add(1, SW_DZ + 256 * 8, 256, SS_D, 3, 18, 0, 259, -1);                                      // colour layer, dir columns 0..2
    add(0, SW_DZ + 256 * 8, 256, SS_H + 256 * 7, 256, 18, 3, 259, 19);                          // colour layer, hidden columns
    add(2, SW_DRGB, 3, SS_HC, 256, 20, 0, 256, 21);                                             // color_layer_linear.0
    add(2, SW_DSIG, 1, SS_H + 256 * 7, 256, 16, 0, 256, 17);                                    // final_layer
    P.n_tasks = n;
    // 8 x 29 + 2 x 6 + 2 x 6 = 256 workgroups, one per CU (shares from per-task workgroup stamps, tools/exp_siren_dw_timing.py).
    // The two K = 3 tasks (network.0: X = xyz; colour layer: dir columns) use the narrowest B tile there is, 32 columns:
    // as 64-column tasks they kept 20 workgroups busy multiplying zero padding.
    static const int chunks_default[3] = {29, 6, 6};
    const int *chunks = chunks_default;
    int chunks_env[3];
    if (const char *e = getenv("NERFMI_SIREN_DW_CHUNKS")) {     // experiments (tools/exp_siren_dw_timing.py): "c0,c1,c2"
        if (sscanf(e, "%d,%d,%d", chunks_env, chunks_env + 1, chunks_env + 2) == 3) chunks = chunks_env;
    }
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
    return (size_t)SW_ROWS * (size_t)(ld + 32) + dw_partial_floats(siren_plan(ld));
}

int nerfmi_siren_forward_rays_train(const float *packed, const float *rays, const float *z, const float *frequencies,
                                    const float *phase_shifts, int n_rays, int n_per_ray, int64_t rays_per_cond,
                                    float *out, float *saved, nerfmi_stream_t stream) {
    NERFMI_REQUIRE(n_rays >= 0 && n_per_ray >= 1 && rays_per_cond >= 1, "siren_forward_rays_train: bad sizes");
    const int64_t n_points = (int64_t)n_rays * n_per_ray;
    if (n_points == 0) return NERFMI_OK;
    NERFMI_REQUIRE(packed && rays && z && frequencies && phase_shifts && out && saved, "siren_forward_rays_train: null pointer");
    const int64_t waves = (n_points + 31) / 32;
    const bool one_cond = rays_per_cond >= n_rays;
    SIREN_FORWARD_LAUNCH(true, false, true, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, packed, rays, z, nullptr, nullptr, frequencies, phase_shifts, n_points,
                       n_per_ray, rays_per_cond * n_per_ray, out, saved, siren_pad_points(n_points));
    return check_launch("siren_forward_rays_train");
}

int nerfmi_siren_forward_points_train(const float *packed, const float *points, const float *ray_directions,
                                      const float *frequencies, const float *phase_shifts, int64_t n_points,
                                      int64_t points_per_cond, float *out, float *saved, nerfmi_stream_t stream) {
    NERFMI_REQUIRE(n_points >= 0 && points_per_cond >= 1, "siren_forward_points_train: bad sizes");
    if (n_points == 0) return NERFMI_OK;
    NERFMI_REQUIRE(packed && points && ray_directions && frequencies && phase_shifts && out && saved,
                   "siren_forward_points_train: null pointer");
    const int64_t waves = (n_points + 31) / 32;
    const bool one_cond = points_per_cond >= n_points;
    SIREN_FORWARD_LAUNCH(false, false, true, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, packed, nullptr, nullptr, points, ray_directions, frequencies, phase_shifts,
                       n_points, 1, points_per_cond, out, saved, siren_pad_points(n_points));
    return check_launch("siren_forward_points_train");
}

int nerfmi_siren_backward(const float *packed, const float *saved, const float *grad_out, const float *frequencies,
                          int64_t n_points, int64_t points_per_cond, float *const *grad_params, float *workspace,
                          nerfmi_stream_t stream) {
    NERFMI_REQUIRE(n_points >= 1 && points_per_cond >= 1, "siren_backward: bad sizes");
    NERFMI_REQUIRE(packed && saved && grad_out && frequencies && grad_params && workspace, "siren_backward: null pointer");
    const int64_t ld = siren_pad_points(n_points);
    GradPtrs G;
    for (int i = 0; i < N_PARAMS; ++i) G.p[i] = nullptr;
    for (int i = 0; i < SIREN_N_PARAMS; ++i) {
        NERFMI_REQUIRE(grad_params[i], "siren_backward: grad_params[%d] is null", i);
        G.p[i] = grad_params[i];
    }
    hipStream_t st = (hipStream_t)stream;
    float *work = workspace;
    float *partial = workspace + (size_t)SW_ROWS * (ld + 32);
    const int64_t waves = (n_points + 31) / 32;
    const DwPlan P = siren_plan(ld);
    const size_t lds = sizeof(float) * 2 * 512 * LROW;  // two 73 728-B tile buffers (> the 64 KiB default dynamic-LDS limit)
    static PerDeviceOnce attr_set;
    int attr_dev;
    if (attr_set.needed(attr_dev)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(siren_dw_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            set_error("siren_backward: cannot raise the dynamic LDS limit");
            return NERFMI_E_LAUNCH;
        }
        attr_set.mark(attr_dev);
    }
    if (points_per_cond >= n_points)
        hipLaunchKernelGGL(siren_backward_chain_kernel<true>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, packed, saved,
                           grad_out, frequencies, n_points, points_per_cond, ld, work);
    else
        hipLaunchKernelGGL(siren_backward_chain_kernel<false>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, packed,
                           saved, grad_out, frequencies, n_points, points_per_cond, ld, work);
    hipLaunchKernelGGL(siren_dw_kernel, dim3(P.n_wg), dim3(256), lds, st, P, work, saved, ld, partial);
    hipLaunchKernelGGL(dw_reduce_kernel<1>, dim3(128, P.n_tasks), dim3(256), 0, st, P, partial, G);
    return check_launch("siren_backward");
}

}  // extern "C"
