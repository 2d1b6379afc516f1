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
constexpr int SIREN_PACKED_FLOATS = SOFF_T7 + 7 * SZ_HID + STREAM_TAIL;
constexpr int SIREN_N_PARAMS = 22;  // network.{0..7}.layer.{weight,bias}, final_layer.*, color_layer_sine.layer.*, color_layer_linear.0.*
static_assert(SOFF_TRANS - SOFF_BIAS + 8 * SZ_HID >= 2 * GS * 256, "the forward stream's prefetch stays inside the image");

// Activations kept for training: tile-major [tile of 32 points][row][32] images (mlp_core.h RowImage).
constexpr int SS_X = 0;                   // 32 rows: box-warped xyz (rows 3..31 = 0)            -> X of network.0
constexpr int SS_D = 32;                  // 32 rows: ray direction (rows 3..31 = 0)             -> X of the colour layer, dir part
constexpr int SS_H = 64;                  // 8 x 256 rows: outputs of network.0..7 (the sines)
constexpr int SS_HC = SS_H + 8 * 256;     // 256 rows: output of color_layer_sine
constexpr int SS_RGB = SS_HC + 256;       // 3 rows (+1 pad): sigmoid output
// sign bits of cos(freq * pre + phase), 9 x 8 rows, same lane/register map as the NeRF ReLU masks (mlp_layout.h S_MASK):
// the backward needs d sin = freq * cos(arg); |cos| = sqrt((1 - s)(1 + s)) comes from the saved sine s, the sign from
// here -- so no second 1 KB/point/layer image of arguments or cosines is written.
constexpr int SS_MASK = SS_RGB + 4;
constexpr int SIREN_SAVED_ROWS = SS_MASK + 9 * 8;

struct SirenParamPtrs {
    const float *p[SIREN_N_PARAMS];
};

__device__ __forceinline__ void store_mask_row(const RowImage &im, int row, unsigned (&mk)[4]) {
    *reinterpret_cast<u32x4 *>(im.tile + row * 32 + 4 * im.lane) = u32x4{mk[0], mk[1], mk[2], mk[3]};
    mk[0] = mk[1] = mk[2] = mk[3] = 0u;
}
__device__ __forceinline__ void load_mask_row(const RowImage &im, int row, unsigned (&mk)[4]) {
    const u32x4 v = *reinterpret_cast<const u32x4 *>(im.tile + row * 32 + 4 * im.lane);
    mk[0] = v[0]; mk[1] = v[1]; mk[2] = v[2]; mk[3] = v[3];
}

// sin(x) and the SIGN BIT of cos(x) (1 = negative) by the Cody-Waite reduction of sincos_cw (mlp_core.h): with
// x = j*pi/2 + r, |r| <= pi/4, m = j mod 4:  cos x = cos r, -sin r, -cos r, sin r  for m = 0..3, so the sign is
// (m in {1,2}) flipped for odd m when r < 0.  (Where r == 0 and m is odd the cosine is 0 and the bit is irrelevant.)
__device__ __forceinline__ float sin_cossign_cw(float x, unsigned &cos_neg) {
    const float j = rintf(x * 0.63661977236758134308f);
    float r = __builtin_fmaf(-j, 1.57079637050628662109375f, x);
    r = __builtin_fmaf(-j, -4.37113900018624283e-8f, r);
    const float s2 = r * r;
    float ps = __builtin_fmaf(s2, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = __builtin_fmaf(s2, ps, -1.6666654611e-1f);
    const float sn = __builtin_fmaf(r * s2, ps, r);
    float pc = __builtin_fmaf(s2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = __builtin_fmaf(s2, pc, 4.166664568298827e-2f);
    const float cs = __builtin_fmaf(s2 * s2, pc, __builtin_fmaf(-0.5f, s2, 1.0f));
    const int q = (int)j;
    const float sv = (q & 1) ? cs : sn;
    const unsigned rneg = __float_as_uint(r) >> 31;
    cos_neg = ((unsigned)((q + 1) >> 1) ^ ((unsigned)q & rneg)) & 1u;
    return (q & 2) ? -sv : sv;
}

template <bool FROM_RAYS, bool SIGMA_ONLY, bool SAVE>
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
    if (SAVE) {
        store_block(S, SS_X, e[0]);
        store_block(S, SS_D, de[0]);
    }
    // this lane's conditioning row (frequencies, phase_shifts are (n_cond, 9*256))
    const float *fq = freq + (p / points_per_cond) * 2304 + 4 * half;
    const float *ph = phase + (p / points_per_cond) * 2304 + 4 * half;
    unsigned mk[4] = {0u, 0u, 0u, 0u};
    auto film_epi = [&](int layer) {
        return [fq, ph, layer, &S, &mk](int jb, int q, f32x4 c, int) {
            const f32x4 f = ldg4(fq + 256 * layer + 32 * jb + 8 * q);
            const f32x4 s = ldg4(ph + 256 * layer + 32 * jb + 8 * q);
            const int sh = 16 * (jb & 1) + 4 * q;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float fr = __fadd_rn(__fmul_rn(f[t], 15.0f), 30.0f);          // nerf.py:202
                const float arg = __fadd_rn(__fmul_rn(fr, c[t]), s[t]);              // nerf.py:151
                if (SAVE) {
                    unsigned neg;
                    c[t] = sin_cossign_cw(arg, neg);
                    asm("v_lshl_or_b32 %0, %1, %2, %0" : "+v"(mk[jb >> 1]) : "v"(neg), "s"(sh + t));
                } else {
                    c[t] = sin_cw(arg);
                }
            }
            if (SAVE) store_slice(S, (layer < 8 ? SS_H + 256 * layer : SS_HC) + 32 * jb, q, c);
            return c;
        };
    };
    auto no_pre = [](int) { return 0; };
    __shared__ __attribute__((aligned(16))) float wlds[WLDS_FLOATS];
    const int wid = threadIdx.x >> 6;
    const float *bias = packed + SOFF_BIAS + 4 * half;
    f32x16 hA[8], hB[8];                   // alternate: a layer reads one, its epilogue writes the other (no copies)
    WeightStage ws;
    // ring phases: network.0 is 2 stages, every hidden layer 16, so the hidden and color layers start at phase 2
    layer_mfma_lds<1, 0, 8, 0, true>(packed + SOFF_L1, bias, e, nullptr, hA, no_pre, film_epi(0), wlds, ws, wid, lane);
    if (SAVE) store_mask_row(S, SS_MASK, mk);
    auto hidden = [&](int l, const f32x16 *in, f32x16 *out_h) __attribute__((always_inline)) {
        layer_mfma_lds<8, 0, 8, 2, false>(packed + SOFF_L2 + (l - 1) * SZ_HID, bias + 256 * l, in, nullptr, out_h, no_pre,
                                          film_epi(l), wlds, ws, wid, lane);
        if (SAVE) store_mask_row(S, SS_MASK + 8 * l, mk);
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
        return;
    }
    layer_mfma_lds<1, 8, 8, 2, false>(packed + SOFF_COLOR, bias + 256 * 8, de, hB, hA, no_pre, film_epi(8), wlds, ws, wid, lane);            // nerf.py:213
    if (SAVE) store_mask_row(S, SS_MASK + 8 * 8, mk);
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
}

static inline int64_t siren_pad_points(int64_t n) { return (n + 31) / 32 * 32; }

}  // namespace nerfmi
