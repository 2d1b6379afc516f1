// FiLM-SIREN field (models/nerf.py:142-151 FiLMLayer, :201-216 SemanticNeRF.forward_with_frequencies_phase_shifts):
// packed-parameter layout, saved-activation layout and the forward kernel template shared by the inference entry
// points (siren.hip) and the training path (siren_bwd.hip).
//
// Same register-resident scheme as the NeRF MLP (mlp_core.h): a wave owns 32 points, the 256-wide hidden state lives
// in accumulator registers through all nine FiLM layers, weights stream through the workgroup's LDS ring as packed
// 1 KiB MFMA fragments.  The FiLM epilogue  sin(freq * (W h + b) + phase)  runs on the VALU in four slices per
// 32-unit block, placed between the next block's MFMAs.
#pragma once
#include "mlp_core.h"

namespace nerfmi {

// packed SIREN image (floats)
constexpr int SOFF_L1 = 0;                          // 8 jb x 1 kb   (3 valid input columns)
constexpr int SSZ_L1 = 8 * 1 * 1024;
constexpr int SOFF_L2 = SOFF_L1 + SSZ_L1;           // 7 hidden layers, 8 x 8 each
constexpr int SOFF_COLOR = SOFF_L2 + 7 * SZ_HID;    // 8 jb x 9 kb   ([dir 3 | hidden 256], nerf.py:213)
constexpr int SSZ_COLOR = 8 * 9 * 1024;
constexpr int SOFF_BIAS = SOFF_COLOR + SSZ_COLOR;   // 9 x 256 (network.0..7, color_layer_sine)
constexpr int SOFF_W_SIGMA = SOFF_BIAS + 9 * 256;   // 256
constexpr int SOFF_B_SIGMA = SOFF_W_SIGMA + 256;    // 1 (+3)
constexpr int SOFF_W_RGB = SOFF_B_SIGMA + 4;        // 3 x 256
constexpr int SOFF_B_RGB = SOFF_W_RGB + 768;        // 3 (+1)
// transposed images for the backward dX chain, in the ORDER THE CHAIN WALKS THEM (its weight stream runs across layer
// ends like the forward one, mlp_core.h layer_mfma_lds): colour layer (hidden columns 3..258), network.7 .. network.1
//     T[((kbo*8 + jb)*4 + q)*256 + lane*4 + t] = W[32*jb + 8*q + 4*(lane>>5) + t][col0 + 32*kbo + (lane&31)]
constexpr int SOFF_TRANS = (SOFF_B_RGB + 4 + 255) / 256 * 256;
constexpr int SOFF_TCOLOR = SOFF_TRANS;             // 8 kbo x 8 jb
constexpr int SOFF_T7 = SOFF_TCOLOR + SZ_HID;       // network.l at SOFF_T7 + (7 - l) * SZ_HID, l = 7..1
// readable tail: the weight streams prefetch two stages past their last layer
// stage length of the weight ring in the fp32 FiLM-SIREN kernels (mlp_core.h layer_mfma_lds; every layer of this field is a
// whole number of 8-, 16- or 32-fragment stages).  Measured on MI355X, same box (round 3): 8 fragments (twice the workgroup
// barriers) forward-with-save +1.6 %, chain +0.0 %; 32 fragments (half the barriers) +9.5 % / +12.8 %, because the 32 staging
// registers no longer fit beside the two activation buffers (380 VGPR <-> AGPR moves): the barriers are not where the time goes
#ifndef NERFMI_SIREN_GS
#define NERFMI_SIREN_GS 16
#endif
constexpr int SIREN_GS = NERFMI_SIREN_GS;
#ifndef NERFMI_SIREN_DMA
#define NERFMI_SIREN_DMA 0
#endif
// the weight ring by LDS-DMA (mlp_core.h layer_mfma_lds<..., DMA>): measured slower (forward-with-save +0.9 %, chain +4.9 %,
// inference +1.4 %; with 32-fragment stages +3.0 % / +13.4 % / +4.7 %), kept as an experiment switch
constexpr bool SIREN_DMA = NERFMI_SIREN_DMA != 0;
constexpr int SIREN_PH = (32 / SIREN_GS) % NSLOT;      // ring phase behind network.0 (32 fragments); the hidden layers keep it
constexpr int SIREN_WLDS_BYTES = NSLOT * SIREN_GS * 1024;
constexpr int SIREN_STREAM_TAIL = 2 * SIREN_GS * 256;
constexpr int SIREN_PACKED_FLOATS = SOFF_T7 + 7 * SZ_HID + SIREN_STREAM_TAIL;
// split-bf16 image (bf16x3_core.h): a unit = 512 floats of `packed` = 3 x 1 KiB; the forward layers' units, then the eight
// transposed images of the backward chain in the order it walks them
constexpr int SIREN_FAST_FWD_UNITS = SOFF_BIAS / 512;
constexpr int SIREN_FAST_UNITS = SIREN_FAST_FWD_UNITS + 8 * (SZ_HID / 512);
constexpr int SIREN_N_PARAMS = 22;  // network.{0..7}.layer.{weight,bias}, final_layer.*, color_layer_sine.layer.*, color_layer_linear.0.*
static_assert(SOFF_TRANS - SOFF_BIAS + 8 * SZ_HID >= 2 * SIREN_GS * 256, "the forward stream's prefetch stays inside the image");

// Activations kept for training: tile-major images, same row map in two element orders:
//   fp32 path (siren_forward_kernel<save>, siren_backward_chain_kernel, siren_dw_kernel): the "x4" order (mlp_core.h at4 /
//     store_slice4: the four rows of a row group interleaved per point; dw_core.h dw_task4g); x, y, z / the direction sit in
//     rows 0, 4, 8 of their block (unit 0 of row groups 0..2);
//   split-bf16 path (siren_forward_bf16x3_kernel<save>, ..._chain_bf16x3, siren_dw_bf16x3_kernel): [row][32 points], x, y, z in
//     rows 0..2 (rows 3..31 = 0).
// An image is read back by the backward of the SAME math (include/nerfmi.h: nerfmi_siren_forward_*_train[_fast]).
constexpr int SS_X = 0;                   // 32 rows: box-warped xyz                            -> X of network.0
constexpr int SS_D = 32;                  // 32 rows: ray direction                             -> X of the colour layer, dir part
constexpr int SS_H = 64;                  // 8 x 256 rows: outputs of network.0..7 (the sines)
constexpr int SS_HC = SS_H + 8 * 256;     // 256 rows: output of color_layer_sine
constexpr int SS_RGB = SS_HC + 256;       // 3 rows (+1 pad): sigmoid output
// cos(freq * pre + phase) of every unit, 9 x 256 row-equivalents: the backward needs d sin = freq * cos(arg).  Rounds 1-2
// kept one SIGN BIT per unit and rebuilt |cos| = sqrt(1 - s^2) from the saved sine in the chain kernel: ~10 vector
// instructions per value there (a v_sqrt_f32, the bit extraction, four 4-byte loads of the sines per slice) next to fp32
// MFMAs that do not overlap with them.  Round 3: the forward has the argument in a register and v_cos_f32 costs two issue
// slots, so it writes the cosines themselves -- in a LANE-PRIVATE order, not as rows: slice (layer, jb, q) of a tile is the
// 1 KiB  [64 lanes][4 units]  at  tile + (SS_COS + 256 layer) * 32 + ((4 jb + q) * 64 + lane) * 4:  ONE 16-byte store per
// slice in the forward, ONE 16-byte load per slice in the chain kernel (same lane / register map on both sides; nothing else
// reads it), and the chain's epilogue is two packed multiplies per pair.  Price: 1 KB / point / layer more HBM written.
constexpr int SS_COS = SS_RGB + 4;
constexpr int SIREN_SAVED_ROWS = SS_COS + 9 * 256;

__device__ __forceinline__ float *cos_slice(const RowImage &im, int layer, int jb, int q) {
    return im.tile + (SS_COS + 256 * layer) * 32 + ((4 * jb + q) * 64 + im.lane) * 4;
}

struct SirenParamPtrs {
    const float *p[SIREN_N_PARAMS];
};

// The FiLM activation  sin(fr * pre + phase),  fr = 15 f + 30  (nerf.py:151, :202)  on gfx950 (round 3):
//   * v_sin_f32 takes its argument in REVOLUTIONS and -- measured on MI355X, tools/ubench/hw_sin.hip,
//     profiles/r03_ubench_hw_sin.txt -- is accurate to 1.25e-7 absolute (mean 2.8e-8) at every magnitude tried, 0.5 to
//     100 000 revolutions (the ISA manual's "valid for |t| <= 256" is not a limit of this part): as good as the degree-9
//     polynomial of rounds 1-2 (sin_pi: 1.4e-7) at TWO issue slots (8 cycles) instead of twelve instructions.  Round 2
//     assumed it could not hold 1e-4 through nine layers without measuring it.
//   * the FiLM constants are therefore kept pre-divided by 2 pi:  t = fma(fr', pre, phase'),  fr' = fr / 2pi,
//     phase' = phase / 2pi  -- one rounding of the argument where the reference makes two (freq * x, + phase); both differ
//     from the exact argument by ~|arg| * 6e-8.  Two values per instruction: v_pk_fma_f32 issues like v_fma_f32
//     (tools/ubench/pk_valu.hip, profiles/r03_ubench_pk_valu.txt).
//   * training: v_cos_f32 of the same t gives the cosine the backward needs (SS_COS below), two more issue slots.
// Forward epilogue: 2.5 issue slots per value at inference, 4.5 with the cosines (rounds 1-2: 14 / 15).  Measured
// (same box, round 3): forward-with-save 0.725 -> 0.783 of the fp32 MFMA peak, inference 0.77 -> 0.823, step 5.50 -> 5.31 ms.
constexpr float INV_2PI = 0.15915494309189535f;

__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

// fr' = (15 f + 30) / 2pi with the reference's two roundings of fr (nerf.py:202) and one for the division
__device__ __forceinline__ float film_scale(float f) { return __fmul_rn(__fadd_rn(__fmul_rn(f, 15.0f), 30.0f), INV_2PI); }

// FiLM constants of one conditioning row staged in LDS by the workgroup, computed ONCE per unit instead of once per
// (unit, point), and read with ds_read_b128 in the epilogues instead of L2-latency global loads inside the fenced epilogue
// clumps.  Used when the whole launch shares one conditioning row (COND_LDS; render_rays' SirenField always does);
// otherwise the rows are read from global memory per lane.
constexpr int FILM_FLOATS = 2 * 2304;       // [fr' 9 x 256][phase' 9 x 256]

__device__ __forceinline__ void stage_film(float *film, const float *__restrict__ freq, const float *__restrict__ phase) {
    for (int i = threadIdx.x; i < 2304; i += blockDim.x) {
        film[i] = film_scale(freq[i]);
        film[2304 + i] = __fmul_rn(phase[i], INV_2PI);
    }
    __syncthreads();
}

// four FiLM activations: c <- sin(2 pi (fr' c + ph')); SAVE: cs <- cos of the same arguments
template <bool SAVE>
__device__ __forceinline__ f32x4 film_sin4(f32x4 c, f32x4 fr, f32x4 ph, f32x4 &cs) {
    const f32x2 t0 = fma2(f32x2{fr[0], fr[1]}, f32x2{c[0], c[1]}, f32x2{ph[0], ph[1]});
    const f32x2 t1 = fma2(f32x2{fr[2], fr[3]}, f32x2{c[2], c[3]}, f32x2{ph[2], ph[3]});
    if (SAVE)
        cs = f32x4{__builtin_amdgcn_cosf(t0[0]), __builtin_amdgcn_cosf(t0[1]), __builtin_amdgcn_cosf(t1[0]),
                   __builtin_amdgcn_cosf(t1[1])};
    return f32x4{__builtin_amdgcn_sinf(t0[0]), __builtin_amdgcn_sinf(t0[1]), __builtin_amdgcn_sinf(t1[0]),
                 __builtin_amdgcn_sinf(t1[1])};
}

template <bool FROM_RAYS, bool SIGMA_ONLY, bool SAVE, bool COND_LDS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
siren_forward_kernel(const float *__restrict__ packed, const float *__restrict__ rays, const float *__restrict__ z,
                     const float *__restrict__ pts, const float *__restrict__ dirs, const float *__restrict__ freq,
                     const float *__restrict__ phase, int64_t n_points, int n_per_ray, int64_t points_per_cond,
                     float *__restrict__ out, float *__restrict__ saved, int64_t ld) {
    static_assert(!(SAVE && SIGMA_ONLY), "training saves the full (rgb, sigma) pass");
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t p0 = wave * 32;       // no early exit: the workgroup's waves share barriers (layer_mfma_lds)
    const int64_t praw = p0 + (lane & 31);
    const bool ok = praw < n_points;
    const int64_t p = ok ? praw : n_points - 1;
    RowImage S;
    S.init(saved, wave, ld / 32, SIREN_SAVED_ROWS, lane, ok, p0 < n_points);

    float x[3], d[3];
    if (FROM_RAYS) {
        const float *rr = rays + (p / n_per_ray) * 8;
        const float zz = z[p];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            x[c] = __fadd_rn(rr[c], __fmul_rn(rr[3 + c], zz));      // rendering.py:224-225
            d[c] = rr[3 + c];
        }
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) { x[c] = pts[p * 3 + c]; d[c] = dirs ? dirs[p * 3 + c] : 0.f; }
    }
    const float warp = 2.0f / 51.0f;                                 // UniformBoxWarp(51), nerf.py:134-140, :193
    f32x16 e[1], de[1];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int c = 8 * (r >> 2) + 4 * half + (r & 3);
        e[0][r] = (c < 3) ? __fmul_rn(x[c < 3 ? c : 0], warp) : 0.f;
        de[0][r] = (c < 3) ? d[c < 3 ? c : 0] : 0.f;
    }
    if (SAVE && half == 0) {
        // the two K = 3 operands of the dW GEMM: component c in unit 0 of row group c, where dw_task4g's narrow-B form reads them
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            *at4(S, SS_X + 4 * c) = e[0][c];
            *at4(S, SS_D + 4 * c) = de[0][c];
        }
    }
    // conditioning: one row for the launch staged in LDS (COND_LDS), else this lane's row (n_cond, 9*256) from memory
    __shared__ __attribute__((aligned(16))) float film[COND_LDS ? FILM_FLOATS : 4];
    if (COND_LDS) stage_film(film, freq, phase);
    const float *fq = freq + (COND_LDS ? 0 : (p / points_per_cond) * 2304) + 4 * half;
    const float *ph = phase + (COND_LDS ? 0 : (p / points_per_cond) * 2304) + 4 * half;
    const float *lfr = film + 4 * half;
    auto film_epi = [&](int layer) {
        return [fq, ph, lfr, layer, &S](int jb, int q, f32x4 c, int) {
            f32x4 fr, s;
            if (COND_LDS) {
                fr = *reinterpret_cast<const f32x4 *>(lfr + 256 * layer + 32 * jb + 8 * q);
                s = *reinterpret_cast<const f32x4 *>(lfr + 2304 + 256 * layer + 32 * jb + 8 * q);
            } else {
                const f32x4 f = ldg4(fq + 256 * layer + 32 * jb + 8 * q);
                s = ldg4(ph + 256 * layer + 32 * jb + 8 * q);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    fr[t] = film_scale(f[t]);                                                            // nerf.py:202
                    s[t] = __fmul_rn(s[t], INV_2PI);
                }
            }
            f32x4 cs;
            c = film_sin4<SAVE>(c, fr, s, cs);                                                          // nerf.py:151
            if (SAVE) {
                store_slice4(S, (layer < 8 ? SS_H + 256 * layer : SS_HC) + 32 * jb, q, c);
                __builtin_nontemporal_store(cs, reinterpret_cast<f32x4 *>(cos_slice(S, layer, jb, q)));
            }
            return c;
        };
    };
    auto no_pre = [](int) { return 0; };
    extern __shared__ __attribute__((aligned(16))) float wlds[];     // SIREN_WLDS_BYTES (dynamic: may exceed the 64 KiB static limit)
    const int wid = threadIdx.x >> 6;
    // the nine layers' biases staged in LDS: an L2-latency bias load sits right in front of each block's first MFMA
    // (measured here: training step 5.77 -> 5.71 ms; the same change made the NeRF kernels 1 % SLOWER and is not in mlp.hip)
    __shared__ __attribute__((aligned(16))) float lbias[9 * 256];
    for (int i = threadIdx.x; i < 9 * 256; i += blockDim.x) lbias[i] = packed[SOFF_BIAS + i];
    const float *bias = lbias + 4 * half;
    f32x16 hA[8], hB[8];                   // alternate: a layer reads one, its epilogue writes the other (no copies)
    WeightStageT<SIREN_GS> ws;
    // ring phases: network.0 is 32 fragments, every hidden layer a multiple of four stages: the hidden and color layers start at SIREN_PH
    layer_mfma_lds<1, 0, 8, 0, true, SIREN_GS, SIREN_DMA>(packed + SOFF_L1, bias, e, nullptr, hA, no_pre, film_epi(0), wlds, ws, wid, lane);
    auto hidden = [&](int l, const f32x16 *in, f32x16 *out_h) __attribute__((always_inline)) {
        layer_mfma_lds<8, 0, 8, SIREN_PH, false, SIREN_GS, SIREN_DMA>(packed + SOFF_L2 + (l - 1) * SZ_HID, bias + 256 * l, in, nullptr, out_h, no_pre,
                                          film_epi(l), wlds, ws, wid, lane);
    };
    hidden(1, hA, hB);
    hidden(2, hB, hA);
    hidden(3, hA, hB);
    hidden(4, hB, hA);
    hidden(5, hA, hB);
    hidden(6, hB, hA);
    hidden(7, hA, hB);
    const float sigma = dot_blocks<8>(hB, packed + SOFF_W_SIGMA + 4 * half) + packed[SOFF_B_SIGMA];   // nerf.py:212
    if (SIGMA_ONLY) {
        if (ok && half == 0) out[p] = sigma;
        if (SIREN_DMA) ring_drain();
        return;
    }
    layer_mfma_lds<1, 8, 8, SIREN_PH, false, SIREN_GS, SIREN_DMA>(packed + SOFF_COLOR, bias + 256 * 8, de, hB, hA, no_pre, film_epi(8), wlds, ws, wid, lane);            // nerf.py:213
    float rgb[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float pre = dot_blocks<8>(hA, packed + SOFF_W_RGB + 256 * c + 4 * half) + packed[SOFF_B_RGB + c];
        rgb[c] = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-pre)));                                         // nerf.py:214
    }
    if (ok && half == 0) {
        float4 o;
        o.x = rgb[0]; o.y = rgb[1]; o.z = rgb[2]; o.w = sigma;
        reinterpret_cast<float4 *>(out)[p] = o;
    }
    if (SAVE && half == 0 && S.live) {
        *S.at(SS_RGB + 0) = ok ? rgb[0] : 0.f;
        *S.at(SS_RGB + 1) = ok ? rgb[1] : 0.f;
        *S.at(SS_RGB + 2) = ok ? rgb[2] : 0.f;
    }
    if (SIREN_DMA) ring_drain();
}

static inline int64_t siren_pad_points(int64_t n) { return (n + 31) / 32 * 32; }

// the weight ring is dynamic LDS (SIREN_WLDS_BYTES may exceed the 64 KiB default limit): raise the limit of the two instances a
// launch may pick, once per device
template <bool FROM_RAYS, bool SIGMA_ONLY, bool SAVE>
static inline bool siren_forward_lds_ok() {
    static PerDeviceOnce once;
    int dev;
    if (!once.needed(dev)) return true;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(siren_forward_kernel<FROM_RAYS, SIGMA_ONLY, SAVE, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, SIREN_WLDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(siren_forward_kernel<FROM_RAYS, SIGMA_ONLY, SAVE, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, SIREN_WLDS_BYTES) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    once.mark(dev);
    return true;
}

// the launch shares ONE conditioning row (`one_cond` in scope): FiLM constants through LDS, else per-lane rows from memory
#define SIREN_FORWARD_LAUNCH(FROM_RAYS, SIGMA_ONLY, SAVE, grid, block, stream, ...)                                    \
    do {                                                                                                              \
        if (!siren_forward_lds_ok<FROM_RAYS, SIGMA_ONLY, SAVE>()) {                                                   \
            set_error("siren forward: cannot raise the dynamic LDS limit");                                           \
            return NERFMI_E_LAUNCH;                                                                                   \
        }                                                                                                             \
        if (one_cond)                                                                                                 \
            hipLaunchKernelGGL((siren_forward_kernel<FROM_RAYS, SIGMA_ONLY, SAVE, true>), grid, block, SIREN_WLDS_BYTES, stream, \
                               __VA_ARGS__);                                                                          \
        else                                                                                                          \
            hipLaunchKernelGGL((siren_forward_kernel<FROM_RAYS, SIGMA_ONLY, SAVE, false>), grid, block, SIREN_WLDS_BYTES, stream, \
                               __VA_ARGS__);                                                                          \
    } while (0)

}  // namespace nerfmi
