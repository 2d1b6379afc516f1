// FiLM-SIREN field forward for gfx950 (MI355X): FiLMLayer (models/nerf.py:142-151)
// and SemanticNeRF.forward_with_frequencies_phase_shifts (models/nerf.py:201-216).
//
// Same register-resident scheme as the NeRF MLP (mlp_core.h): a wave owns 32
// points, the 256-wide hidden state lives in accumulator registers through all
// nine FiLM layers, weights stream from L2 as packed 1 KiB MFMA fragments.  The
// FiLM epilogue  sin(freq * (W h + b) + phase)  runs on the VALU in four slices
// per 32-unit block, placed between the next block's MFMAs (layer_mfma), so the
// 2304 sines per point execute in the shadow of the matrix pipe.
#include "bf16x3_core.h"
#include "siren_core.h"

namespace nerfmi {

// state_dict tensors -> fragment-order images (siren_core.h; operand maps in mlp_layout.h)
__global__ void siren_pack_kernel(SirenParamPtrs P, float *__restrict__ packed) {
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < SIREN_PACKED_FLOATS; idx += gridDim.x * blockDim.x) {
        float v = 0.f;
        if (idx < SOFF_BIAS) {
            int layer, KB, in_f, rel;
            if (idx < SOFF_L2) { layer = 0; KB = 1; in_f = 3; rel = idx; }
            else if (idx < SOFF_COLOR) { layer = 1 + (idx - SOFF_L2) / SZ_HID; KB = 8; in_f = 256; rel = (idx - SOFF_L2) % SZ_HID; }
            else { layer = 8; KB = 9; in_f = 259; rel = idx - SOFF_COLOR; }
            const int t = rel & 3, lane = (rel >> 2) & 63, g = rel >> 8;
            const int q = g & 3, kb = (g >> 2) % KB, jb = (g >> 2) / KB;
            const int row = 32 * jb + (lane & 31);
            const int kc = 32 * kb + 8 * q + 4 * (lane >> 5) + t;
            int col = -1;
            if (layer == 0) { if (kc < 3) col = kc; }
            else if (layer < 8) col = kc;
            else { if (kc < 32) { if (kc < 3) col = kc; } else col = 3 + (kc - 32); }
            const int pi = (layer < 8) ? 2 * layer : 18;     // color_layer_sine.layer.weight = param 18
            if (col >= 0) v = P.p[pi][row * in_f + col];
        } else if (idx < SOFF_W_SIGMA) {
            const int s = idx - SOFF_BIAS, layer = s >> 8;
            v = P.p[layer < 8 ? 2 * layer + 1 : 19][s & 255];
        } else if (idx < SOFF_B_SIGMA) v = P.p[16][idx - SOFF_W_SIGMA];
        else if (idx < SOFF_W_RGB) v = (idx == SOFF_B_SIGMA) ? P.p[17][0] : 0.f;
        else if (idx < SOFF_B_RGB) v = P.p[20][idx - SOFF_W_RGB];
        else if (idx < SOFF_B_RGB + 4) v = (idx - SOFF_B_RGB < 3) ? P.p[21][idx - SOFF_B_RGB] : 0.f;
        else if (idx >= SOFF_TRANS && idx < SOFF_T7 + 7 * SZ_HID) {
            // transposed images, backward order: colour layer (hidden columns), network.7 .. network.1
            const int ti = (idx - SOFF_TRANS) / SZ_HID;                  // 0 = colour, 1 + k = network.(7 - k)
            const int rel = (idx - SOFF_TRANS) % SZ_HID;
            const int t = rel & 3, lane = (rel >> 2) & 63, g = rel >> 8;
            const int q = g & 3, jb = (g >> 2) % 8, kbo = (g >> 2) / 8;
            const int row = 32 * jb + 8 * q + 4 * (lane >> 5) + t;       // W row (output unit)
            const int col = 32 * kbo + (lane & 31);                      // hidden input unit
            v = (ti == 0) ? P.p[18][row * 259 + 3 + col] : P.p[2 * (8 - ti)][row * 256 + col];
        }
        packed[idx] = v;
    }
}

// ---------------------------------------------------------------------------------------------------
// OPT-IN split-bf16 FiLM-SIREN forward (bf16x3_core.h): the nine dense products as six bf16 MFMAs per fp32-equivalent
// product block, the FiLM activation of each layer by the hardware sine between the layers.  Same packed parameters, same
// outputs, same saved images (training), same tolerances.
// ---------------------------------------------------------------------------------------------------
static FastTable siren_fast_table() {
    FastTable T;
    int n = 0;
    auto add = [&](int off, int JB, int KB) { T.l[n++] = FastLayer{off, JB, KB, off / 512}; };
    add(SOFF_L1, 8, 1);
    for (int l = 1; l < 8; ++l) add(SOFF_L2 + (l - 1) * SZ_HID, 8, 8);
    add(SOFF_COLOR, 8, 9);
    // transposed images of the backward chain, in the order it walks them (colour layer, network.7 .. network.1):
    // (output blocks of dX, contraction blocks), units behind the forward ones
    for (int t = 0; t < 8; ++t) T.l[n++] = FastLayer{SOFF_TRANS + t * SZ_HID, 8, 8, SIREN_FAST_FWD_UNITS + t * (SZ_HID / 512)};
    T.n = n;
    T.n_units = SIREN_FAST_UNITS;
    return T;
}

template <bool SIGMA_ONLY, bool SAVE, bool COND_LDS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
siren_forward_bf16x3_kernel(const float *__restrict__ packed, const __bf16 *__restrict__ fast,
                            const float *__restrict__ rays, const float *__restrict__ z, const float *__restrict__ freq,
                            const float *__restrict__ phase, int64_t n_points, int n_per_ray, int64_t points_per_cond,
                            float *__restrict__ out, float *__restrict__ saved, int64_t ld) {
    static_assert(!(SAVE && SIGMA_ONLY), "training saves the full (rgb, sigma) pass");
    extern __shared__ __attribute__((aligned(16))) char wlds_fast[];
    const int lane = threadIdx.x & 63, half = lane >> 5, wid = threadIdx.x >> 6;
    const int64_t wave = (int64_t)blockIdx.x * 4 + wid;
    const int64_t praw = wave * 32 + (lane & 31);
    const bool ok = praw < n_points;
    const int64_t p = ok ? praw : n_points - 1;
    RowImage S;                                                       // training: same images as siren_forward_kernel<SAVE>
    S.init(saved, wave, ld / 32, SIREN_SAVED_ROWS, lane, ok, wave * 32 < n_points);
    const float *rr = rays + (p / n_per_ray) * 8;
    const float zz = z[p];
    float x[3], d[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        x[c] = __fadd_rn(rr[c], __fmul_rn(rr[3 + c], zz));          // rendering.py:224-225
        d[c] = rr[3 + c];
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
    // FiLM constants as in the fp32 kernels (siren_core.h): pre-divided by 2 pi, one row staged in LDS or this lane's row
    __shared__ __attribute__((aligned(16))) float film[COND_LDS ? FILM_FLOATS : 4];
    if (COND_LDS) stage_film(film, freq, phase);
    const float *fq = freq + (COND_LDS ? 0 : (p / points_per_cond) * 2304) + 4 * half;
    const float *ph = phase + (COND_LDS ? 0 : (p / points_per_cond) * 2304) + 4 * half;
    const float *lfr = film + 4 * half;
    // Round 3: the FiLM activation of a whole layer BETWEEN the layers, by the hardware sine (film_sin4: 2.5 issue slots per
    // value, 4.5 with the cosines) -- not in consuming-layer hooks as in rounds 1-2.  The hooks existed to hide a 13-instruction
    // polynomial sine behind the XDL MFMAs; with v_sin_f32 the whole activation of a layer is ~2.5 k cycles beside ~34 k of MFMAs,
    // the hook version paid more than that in scheduling constraints (inference fine pass 0.90 ms hooked).  Training: sines to the
    // saved rows, cosines to the lane-private cosine image, exactly as siren_forward_kernel<SAVE> writes them.
    auto film_all = [&](int layer, f32x16 *h) __attribute__((always_inline)) {
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 fr, s;
                if (COND_LDS) {
                    fr = *reinterpret_cast<const f32x4 *>(lfr + 256 * layer + 32 * b + 8 * q);
                    s = *reinterpret_cast<const f32x4 *>(lfr + 2304 + 256 * layer + 32 * b + 8 * q);
                } else {
                    const f32x4 f = ldg4(fq + 256 * layer + 32 * b + 8 * q);
                    s = ldg4(ph + 256 * layer + 32 * b + 8 * q);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        fr[t] = film_scale(f[t]);
                        s[t] = __fmul_rn(s[t], INV_2PI);
                    }
                }
                f32x4 c = {h[b][4 * q], h[b][4 * q + 1], h[b][4 * q + 2], h[b][4 * q + 3]}, cs;
                c = film_sin4<SAVE>(c, fr, s, cs);                                                     // nerf.py:151
                if (SAVE) {
                    store_slice(S, (layer < 8 ? SS_H + 256 * layer : SS_HC) + 32 * b, q, c);
                    __builtin_nontemporal_store(cs, reinterpret_cast<f32x4 *>(cos_slice(S, layer, b, q)));
                }
                h[b][4 * q] = c[0]; h[b][4 * q + 1] = c[1]; h[b][4 * q + 2] = c[2]; h[b][4 * q + 3] = c[3];
            }
    };
    const float *bias = packed + SOFF_BIAS + 4 * half;
    auto img = [&](int off) { return fast + fast_fwd_elems(off); };
    f32x16 hA[8], hB[8];
    FastStage fs;
    // ring phases: network.0 is 2 stages, every hidden layer 16, so the hidden and color layers start at phase 2
    layer_bf16x3<1, 0, 8, false, true, NoHook, NoHook, 0>(img(SOFF_L1), bias, e, nullptr, hA, wlds_fast, fs, wid, lane);
    film_all(0, hA);
    auto hidden = [&](int l, const f32x16 *in, f32x16 *out_h) __attribute__((always_inline)) {
        layer_bf16x3<0, 8, 8, false, false, NoHook, NoHook, 2>(img(SOFF_L2 + (l - 1) * SZ_HID), bias + 256 * l, nullptr, in, out_h,
                                                              wlds_fast, fs, wid, lane);
        film_all(l, out_h);
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
    layer_bf16x3<1, 8, 8, false, false, NoHook, NoHook, 2>(img(SOFF_COLOR), bias + 256 * 8, de, hB, hA, wlds_fast, fs, wid, lane);   // nerf.py:213
    film_all(8, hA);
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

}  // namespace nerfmi

using namespace nerfmi;

extern "C" {

size_t nerfmi_siren_packed_floats(void) { return (size_t)SIREN_PACKED_FLOATS; }

int nerfmi_siren_pack(const float *const *params, float *packed, nerfmi_stream_t stream) {
    NERFMI_ENTER();
    NERFMI_REQUIRE(params && packed, "siren_pack: null pointer");
    SirenParamPtrs P;
    for (int i = 0; i < SIREN_N_PARAMS; ++i) {
        NERFMI_REQUIRE(params[i], "siren_pack: params[%d] is null", i);
        P.p[i] = params[i];
    }
    hipLaunchKernelGGL(siren_pack_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, P, packed);
    return check_launch("siren_pack");
}

int nerfmi_siren_forward_points(const float *packed, const float *points, const float *ray_directions,
                                const float *frequencies, const float *phase_shifts, int64_t n_points,
                                int64_t points_per_cond, int sigma_only, float *out, nerfmi_stream_t stream) {
    NERFMI_ENTER();
    NERFMI_REQUIRE(n_points >= 0 && points_per_cond >= 1, "siren_forward_points: bad sizes");
    if (n_points == 0) return NERFMI_OK;
    NERFMI_REQUIRE(packed && points && frequencies && phase_shifts && out, "siren_forward_points: null pointer");
    NERFMI_REQUIRE(sigma_only || ray_directions, "siren_forward_points: ray_directions required for the colour branch");
    const int64_t waves = (n_points + 31) / 32;
    const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const bool one_cond = points_per_cond >= n_points;
    if (sigma_only)
        SIREN_FORWARD_LAUNCH(false, true, false, grid, block, st, packed, nullptr, nullptr, points,
                           ray_directions, frequencies, phase_shifts, n_points, 1, points_per_cond, out, nullptr, (int64_t)0);
    else
        SIREN_FORWARD_LAUNCH(false, false, false, grid, block, st, packed, nullptr, nullptr, points,
                           ray_directions, frequencies, phase_shifts, n_points, 1, points_per_cond, out, nullptr, (int64_t)0);
    return check_launch("siren_forward_points");
}

int nerfmi_siren_forward_rays(const float *packed, const float *rays, const float *z, const float *frequencies,
                              const float *phase_shifts, int n_rays, int n_per_ray, int64_t rays_per_cond,
                              int sigma_only, float *out, nerfmi_stream_t stream) {
    NERFMI_ENTER();
    NERFMI_REQUIRE(n_rays >= 0 && n_per_ray >= 1 && rays_per_cond >= 1, "siren_forward_rays: bad sizes");
    const int64_t n_points = (int64_t)n_rays * n_per_ray;
    if (n_points == 0) return NERFMI_OK;
    NERFMI_REQUIRE(packed && rays && z && frequencies && phase_shifts && out, "siren_forward_rays: null pointer");
    const int64_t waves = (n_points + 31) / 32;
    const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const bool one_cond = rays_per_cond >= n_rays;
    KernelSpan span(sigma_only ? "siren_forward_kernel<sigma_only>" : "siren_forward_kernel", n_points, st);
    if (sigma_only)
        SIREN_FORWARD_LAUNCH(true, true, false, grid, block, st, packed, rays, z, nullptr, nullptr,
                           frequencies, phase_shifts, n_points, n_per_ray, rays_per_cond * n_per_ray, out, nullptr, (int64_t)0);
    else
        SIREN_FORWARD_LAUNCH(true, false, false, grid, block, st, packed, rays, z, nullptr, nullptr,
                           frequencies, phase_shifts, n_points, n_per_ray, rays_per_cond * n_per_ray, out, nullptr, (int64_t)0);
    return check_launch("siren_forward_rays");
}

size_t nerfmi_siren_fast_bytes(void) { return (size_t)SIREN_FAST_UNITS * 3072 + FAST_TAIL_BYTES; }

int nerfmi_siren_pack_fast(const float *packed, void *fast, nerfmi_stream_t stream) {
    NERFMI_ENTER();
    NERFMI_REQUIRE(packed && fast, "siren_pack_fast: null pointer");
    hipLaunchKernelGGL(pack_bf16x3_table_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, siren_fast_table(), packed,
                       (__bf16 *)fast);
    return check_launch("siren_pack_fast");
}

static int siren_forward_fast_impl(const char *who, const float *packed, const void *fast, const float *rays, const float *z,
                                   const float *frequencies, const float *phase_shifts, int n_rays, int n_per_ray,
                                   int64_t rays_per_cond, int sigma_only, float *out, float *saved, nerfmi_stream_t stream) {
    NERFMI_ENTER();
    NERFMI_REQUIRE(n_rays >= 0 && n_per_ray >= 1 && rays_per_cond >= 1, "%s: bad sizes", who);
    const int64_t n_points = (int64_t)n_rays * n_per_ray;
    if (n_points == 0) return NERFMI_OK;
    NERFMI_REQUIRE(packed && fast && rays && z && frequencies && phase_shifts && out, "%s: null pointer", who);
    NERFMI_REQUIRE(!(saved && sigma_only), "%s: saved activations need the full (rgb, sigma) pass", who);
    static PerDeviceOnce attr_set;
    int attr_dev;
    if (attr_set.needed(attr_dev)) {
        bool okk = true;
#define NERFMI_RAISE(K) okk = okk && hipFuncSetAttribute(reinterpret_cast<const void *>(K), hipFuncAttributeMaxDynamicSharedMemorySize, FLDS_BYTES) == hipSuccess
        NERFMI_RAISE((siren_forward_bf16x3_kernel<false, false, true>));
        NERFMI_RAISE((siren_forward_bf16x3_kernel<false, false, false>));
        NERFMI_RAISE((siren_forward_bf16x3_kernel<true, false, true>));
        NERFMI_RAISE((siren_forward_bf16x3_kernel<true, false, false>));
        NERFMI_RAISE((siren_forward_bf16x3_kernel<false, true, true>));
        NERFMI_RAISE((siren_forward_bf16x3_kernel<false, true, false>));
#undef NERFMI_RAISE
        if (!okk) {
            (void)hipGetLastError();
            set_error("%s: cannot raise the dynamic LDS limit", who);
            return NERFMI_E_LAUNCH;
        }
        attr_set.mark(attr_dev);
    }
    const int64_t waves = (n_points + 31) / 32;
    const int64_t ld = siren_pad_points(n_points);
    const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const bool one_cond = rays_per_cond >= n_rays;
    KernelSpan span(sigma_only ? "siren_forward_bf16x3_kernel<sigma_only>"
                               : (saved ? "siren_forward_bf16x3_kernel<save>" : "siren_forward_bf16x3_kernel"), n_points, st);
#define NERFMI_LAUNCH_FAST(SO, SV, SP)                                                                                      \
    do {                                                                                                                    \
        if (one_cond)                                                                                                       \
            hipLaunchKernelGGL((siren_forward_bf16x3_kernel<SO, SV, true>), grid, block, FLDS_BYTES, st, packed,              \
                               (const __bf16 *)fast, rays, z, frequencies, phase_shifts, n_points, n_per_ray,                \
                               rays_per_cond * n_per_ray, out, SP, ld);                                                      \
        else                                                                                                                \
            hipLaunchKernelGGL((siren_forward_bf16x3_kernel<SO, SV, false>), grid, block, FLDS_BYTES, st, packed,             \
                               (const __bf16 *)fast, rays, z, frequencies, phase_shifts, n_points, n_per_ray,                \
                               rays_per_cond * n_per_ray, out, SP, ld);                                                      \
    } while (0)
    if (sigma_only) NERFMI_LAUNCH_FAST(true, false, nullptr);
    else if (saved) NERFMI_LAUNCH_FAST(false, true, saved);
    else NERFMI_LAUNCH_FAST(false, false, nullptr);
#undef NERFMI_LAUNCH_FAST
    return check_launch(who);
}

int nerfmi_siren_forward_rays_fast(const float *packed, const void *fast, const float *rays, const float *z,
                                   const float *frequencies, const float *phase_shifts, int n_rays, int n_per_ray,
                                   int64_t rays_per_cond, int sigma_only, float *out, nerfmi_stream_t stream) {
    return siren_forward_fast_impl("siren_forward_rays_fast", packed, fast, rays, z, frequencies, phase_shifts, n_rays, n_per_ray,
                                   rays_per_cond, sigma_only, out, nullptr, stream);
}

int nerfmi_siren_forward_rays_train_fast(const float *packed, const void *fast, const float *rays, const float *z,
                                         const float *frequencies, const float *phase_shifts, int n_rays, int n_per_ray,
                                         int64_t rays_per_cond, float *out, float *saved, nerfmi_stream_t stream) {
    NERFMI_REQUIRE(saved || (int64_t)n_rays * n_per_ray == 0, "siren_forward_rays_train_fast: null pointer");
    return siren_forward_fast_impl("siren_forward_rays_train_fast", packed, fast, rays, z, frequencies, phase_shifts, n_rays,
                                   n_per_ray, rays_per_cond, 0, out, saved, stream);
}

}  // extern "C"
