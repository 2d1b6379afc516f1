// Fused Embedding + NeRF MLP forward for gfx950 (MI355X), exact fp32 on the
// matrix cores (v_mfma_f32_32x32x2_f32).  models/nerf.py:21-38, :83-124 and the
// inference() head models/rendering.py:131-159.
//
// One wavefront owns 32 sample points and carries their whole 256-wide hidden
// state through all ten layers IN REGISTERS (8 blocks x 16 accumulator VGPRs);
// see mlp_layout.h for why a layer's accumulators are directly the next layer's
// B operand.  Weights stream from L2 (2.4 MB per model, L2-resident) as
// pre-packed, fully coalesced 1 KiB fragments through a small register ring; no
// LDS, no barrier: the four waves of a workgroup are independent and each SIMD
// runs one wave that issues MFMAs back to back.
#include "mlp_core.h"

namespace nerfmi {

__constant__ LayerDesc d_layers[NL_FWD] = {
    LAYERS[0], LAYERS[1], LAYERS[2], LAYERS[3], LAYERS[4], LAYERS[5], LAYERS[6], LAYERS[7], LAYERS[8], LAYERS[9]};

struct ParamPtrs {
    const float *p[N_PARAMS];
};
struct GradPtrs {
    float *p[N_PARAMS];
};

// ---------------------------------------------------------------------------
// pack: state_dict tensors -> fragment-order images (mlp_layout.h)
// ---------------------------------------------------------------------------
__global__ void pack_kernel(ParamPtrs P, float *__restrict__ packed) {
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < PACKED_FLOATS; idx += gridDim.x * blockDim.x) {
        float v = 0.f;
        if (idx < OFF_SMALL) {
            int li = 0;
#pragma unroll
            for (int l = 1; l < NL_FWD; ++l)
                if (idx >= d_layers[l].off) li = l;
            const LayerDesc L = d_layers[li];
            const int rel = idx - L.off;
            const int t = rel & 3, lane = (rel >> 2) & 63, g = rel >> 8;
            const int q = g & 3, kb = (g >> 2) % L.KB, jb = (g >> 2) / L.KB;
            const int row = 32 * jb + (lane & 31);
            const int kc = 32 * kb + 8 * q + 4 * (lane >> 5) + t;
            const int p0 = pad32(L.seg0);
            int col = -1;
            if (kc < p0) { if (kc < L.seg0) col = kc; }
            else if (kc - p0 < L.seg1) col = L.seg0 + (kc - p0);
            if (col >= 0 && row < L.out_f) v = P.p[L.param][row * L.in_f + col];
        } else if (idx < OFF_TRANS) {
            const int s = idx - OFF_SMALL;
            if (s < 8 * 256) v = P.p[2 * (s >> 8) + 1][s & 255];
            else if (idx < OFF_BIAS_DIR) v = P.p[17][idx - OFF_BIAS_FINAL];
            else if (idx < OFF_W_SIGMA) v = P.p[19][idx - OFF_BIAS_DIR];
            else if (idx < OFF_B_SIGMA) v = P.p[PARAM_SIGMA_W][idx - OFF_W_SIGMA];
            else if (idx < OFF_W_RGB) v = (idx == OFF_B_SIGMA) ? P.p[PARAM_SIGMA_B][0] : 0.f;
            else if (idx < OFF_B_RGB) v = P.p[PARAM_RGB_W][idx - OFF_W_RGB];
            else v = (idx - OFF_B_RGB < 3) ? P.p[PARAM_RGB_B][idx - OFF_B_RGB] : 0.f;
        } else if (idx < PACKED_FLOATS - STREAM_TAIL) {
            int li = 1;                                 // the transposed images are stored in backward order
#pragma unroll
            for (int l = 2; l < NL_FWD; ++l)
                if (idx >= d_layers[l].t_off && idx < d_layers[l].t_off + d_layers[l].t_KBO * d_layers[l].JB * 1024) li = l;
            const LayerDesc L = d_layers[li];
            const int JBc = L.JB;                       // contraction blocks = output blocks of W
            const int rel = idx - L.t_off;
            const int t = rel & 3, lane = (rel >> 2) & 63, g = rel >> 8;
            const int q = g & 3, jb = (g >> 2) % JBc, kbo = (g >> 2) / JBc;
            const int row = 32 * jb + 8 * q + 4 * (lane >> 5) + t;       // W row (output unit)
            const int col = L.t_col0 + 32 * kbo + (lane & 31);           // W col (input unit)
            if (row < L.out_f && col < L.in_f) v = P.p[L.param][row * L.in_f + col];
        }
        packed[idx] = v;
    }
}

// Embedding(3,10) of this lane's point, as the two B-operand blocks of layer 1
__device__ __forceinline__ void embed_xyz_blocks(float x, float y, float z, int half, f32x16 *e) {
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

// Embedding(3,4) of the ray direction, one block (channels 27..31 = 0)
__device__ __forceinline__ void embed_dir_block(float x, float y, float z, int half, f32x16 &e) {
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
__device__ unsigned long long nerfmi_dbg_ts[64 * 16];
#define NERFMI_TS(i)                                                                                        \
    do {                                                                                                    \
        if ((threadIdx.x == 0) && (blockIdx.x % 48 == 0) && (blockIdx.x / 48 < 32)) {                       \
            nerfmi_dbg_ts[(blockIdx.x / 48) * 16 + (i)] = __builtin_readcyclecounter();                     \
            if ((i) == 1) nerfmi_dbg_ts[(blockIdx.x / 48) * 16 + 13] = wall_clock64();                      \
            if ((i) == 12) nerfmi_dbg_ts[(blockIdx.x / 48) * 16 + 14] = wall_clock64();                     \
        }                                                                                                   \
    } while (0)
#else
#define NERFMI_TS(i) do { } while (0)
#endif

template <bool EMBEDDED, bool SIGMA_ONLY, bool SAVE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
nerf_forward_kernel(const float *__restrict__ packed, const float *__restrict__ rays, const float *__restrict__ z,
                    const float *__restrict__ xemb, int64_t n_points, int n_per_ray, float *__restrict__ out,
                    float *__restrict__ saved, int64_t ld) {
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    NERFMI_TS(0);
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t p0 = wave * 32;
    // no early exit: the four waves of a workgroup share the weight stream through LDS (barriers); a wave
    // past the end computes on a clamped point and stores nothing
    const int64_t praw = p0 + (lane & 31);
    const bool ok = praw < n_points;
    const int64_t p = ok ? praw : n_points - 1;
    RowImage S;
    S.init(saved, wave, ld / 32, SAVED_ROWS, lane, ok, p0 < n_points);

    f32x16 e[2], de[1];
    if (EMBEDDED) {
        const int ldx = SIGMA_ONLY ? 63 : 90;
        const float *xr = xemb + p * ldx;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = 32 * kb + 8 * (r >> 2) + 4 * half + (r & 3);
                e[kb][r] = (c < 63) ? xr[c] : 0.f;
            }
        if (!SIGMA_ONLY) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = 8 * (r >> 2) + 4 * half + (r & 3);
                de[0][r] = (c < 27) ? xr[63 + c] : 0.f;
            }
        }
    } else {
        const int64_t ray = p / n_per_ray;
        const float *rr = rays + ray * 8;
        const float zz = z[p];
        // rendering.py:224-225  xyz = o + d*z  (two roundings, as torch)
        const float x = __fadd_rn(rr[0], __fmul_rn(rr[3], zz));
        const float y = __fadd_rn(rr[1], __fmul_rn(rr[4], zz));
        const float w = __fadd_rn(rr[2], __fmul_rn(rr[5], zz));
        embed_xyz_blocks(x, y, w, half, e);
        if (!SIGMA_ONLY) embed_dir_block(rr[3], rr[4], rr[5], half, de[0]);
    }
    if (SAVE) {
        // x4 element order (mlp_core.h store_slice4): what the fp32 dW GEMM (dw_core.h dw_task4g) stages by LDS-DMA
        store_block4(S, S_EMB, e[0]);
        store_block4(S, S_EMB + 32, e[1]);
        store_block4(S, S_DEMB, de[0]);
    }

    __shared__ __attribute__((aligned(16))) float wlds[WLDS_FLOATS];
    const int wid = threadIdx.x >> 6;
    const float *bias = packed + OFF_BIAS + 4 * half;
    WeightStage ws;
    // two activation buffers used alternately: a layer reads one and its epilogue writes the other, so a
    // finished block goes accumulator -> ReLU -> next layer's operand with no copy in between (fp32 MFMAs
    // do not overlap with the wave's own vector instructions -- tools/ubench/mfma_valu.hip -- so every
    // register move between layers is paid in matrix-pipe time)
    f32x16 hA[8], hB[8];
    NERFMI_TS(1);

    auto no_pre = [](int) { return 0; };
    // ReLU epilogue; in training each finished block goes straight to the tile-major image
    unsigned mk[4] = {0u, 0u, 0u, 0u};
    auto relu_epi = [&](int row0) {
        return [&S, &mk, row0](int jb, int q, f32x4 c, int) {
#ifndef NERFMI_EXP_NOEPI
            c = relu4(c);
#endif
            if (SAVE) {
#ifndef NERFMI_EXP_NOMASK
                mask_or(mk, jb, q, c);
#endif
#ifndef NERFMI_EXP_NOSTORE
                store_slice4(S, row0 + 32 * jb, q, c);
#endif
            }
            return c;
        };
    };
    // xyz_encoding_l (l = 2..8 except 5): 256 -> 256 + ReLU  (nerf.py:62-70)
    auto hidden = [&](int l, const f32x16 *in, f32x16 *out_h) __attribute__((always_inline)) {
        layer_mfma_lds<8, 0, 8, 0, false>(packed + LAYERS[l - 1].off, bias + 256 * (l - 1), in, nullptr, out_h, no_pre,
                                          relu_epi(S_H + 256 * (l - 1)), wlds, ws, wid, lane);
        if (SAVE) store_mask(S, l - 1, mk);
        NERFMI_TS(1 + l);
    };

    layer_mfma_lds<2, 0, 8, 0, true>(packed + OFF_L1, bias, e, nullptr, hA, no_pre, relu_epi(S_H), wlds, ws, wid, lane);
    if (SAVE) store_mask(S, 0, mk);
    NERFMI_TS(2);
    hidden(2, hA, hB);
    hidden(3, hB, hA);
    hidden(4, hA, hB);
    // xyz_encoding_5 on [xyz embedding | h4]  (skip connection, nerf.py:108-109)
    layer_mfma_lds<2, 8, 8, 0, false>(packed + OFF_L5, bias + 256 * 4, e, hB, hA, no_pre, relu_epi(S_H + 256 * 4), wlds, ws, wid, lane);
    if (SAVE) store_mask(S, 4, mk);
    NERFMI_TS(6);
    hidden(6, hA, hB);
    hidden(7, hB, hA);
    hidden(8, hA, hB);
    const float sigma = dot_blocks<8>(hB, packed + OFF_W_SIGMA + 4 * half) + packed[OFF_B_SIGMA];   // nerf.py:112
    if (!SIGMA_ONLY) {
        // xyz_encoding_final: no activation (nerf.py:116)
        layer_mfma_lds<8, 0, 8, 0, false>(packed + OFF_FINAL, bias + 256 * 8, hB, nullptr, hA, no_pre,
                                          [&S](int jb, int q, f32x4 c, int) {
                                              if (SAVE) store_slice4(S, S_FINAL + 32 * jb, q, c);
                                              return c;
                                          }, wlds, ws, wid, lane);
        NERFMI_TS(10);
    }
    if (SIGMA_ONLY) {
        if (ok && half == 0) out[p] = sigma;
        return;
    }
    f32x16 dh[4];
    layer_mfma_lds<8, 1, 4, 0, false>(packed + OFF_DIR, packed + OFF_BIAS_DIR + 4 * half, hA, de, dh, no_pre, relu_epi(S_DIRH), wlds, ws, wid, lane);
    if (SAVE) store_mask(S, 8, mk);
    NERFMI_TS(11);
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
    if (SAVE && half == 0 && S.live) {
        *S.at(S_RGB + 0) = ok ? rgb[0] : 0.f;
        *S.at(S_RGB + 1) = ok ? rgb[1] : 0.f;
        *S.at(S_RGB + 2) = ok ? rgb[2] : 0.f;
    }
    NERFMI_TS(12);
}

static inline int64_t pad_points(int64_t n) { return (n + 31) / 32 * 32; }

}  // namespace nerfmi

using namespace nerfmi;

extern "C" {

#ifdef NERFMI_TIMING
int nerfmi_debug_timing(unsigned long long *host) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(nerfmi_dbg_ts), sizeof(unsigned long long) * 64 * 16) == hipSuccess ? 0 : 1;
}
#endif

size_t nerfmi_nerf_packed_floats(void) { return (size_t)PACKED_FLOATS; }

int nerfmi_nerf_pack(const float *const *params, float *packed, nerfmi_stream_t stream) {
    NERFMI_ENTER();
    NERFMI_REQUIRE(params && packed, "nerf_pack: null pointer");
    ParamPtrs P;
    for (int i = 0; i < N_PARAMS; ++i) {
        NERFMI_REQUIRE(params[i], "nerf_pack: params[%d] is null", i);
        P.p[i] = params[i];
    }
    hipLaunchKernelGGL(pack_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, P, packed);
    return check_launch("nerf_pack");
}

// + one dump tile for waves past the end (mlp_core.h RowImage)
size_t nerfmi_nerf_saved_floats(int64_t n_points) { return (size_t)SAVED_ROWS * (size_t)(pad_points(n_points) + 32); }

int nerfmi_nerf_forward_rays(const float *packed, const float *rays, const float *z, int n_rays, int n_per_ray,
                             int sigma_only, float *out, float *saved, nerfmi_stream_t stream) {
    NERFMI_ENTER();
    NERFMI_REQUIRE(n_rays >= 0 && n_per_ray >= 1, "nerf_forward_rays: bad sizes");
    const int64_t n_points = (int64_t)n_rays * n_per_ray;
    if (n_points == 0) return NERFMI_OK;
    NERFMI_REQUIRE(packed && rays && z && out, "nerf_forward_rays: null pointer");
    NERFMI_REQUIRE(!(saved && sigma_only), "nerf_forward_rays: saved activations need the full (rgb,sigma) pass");
    const int64_t waves = (n_points + 31) / 32;
    const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    const int64_t ld = pad_points(n_points);
    hipStream_t st = (hipStream_t)stream;
    KernelSpan span(sigma_only ? "nerf_forward_kernel<sigma_only>" : (saved ? "nerf_forward_kernel<save>" : "nerf_forward_kernel"),
                    n_points, st);
    if (sigma_only)
        hipLaunchKernelGGL((nerf_forward_kernel<false, true, false>), grid, block, 0, st, packed, rays, z, nullptr,
                           n_points, n_per_ray, out, nullptr, ld);
    else if (saved)
        hipLaunchKernelGGL((nerf_forward_kernel<false, false, true>), grid, block, 0, st, packed, rays, z, nullptr,
                           n_points, n_per_ray, out, saved, ld);
    else
        hipLaunchKernelGGL((nerf_forward_kernel<false, false, false>), grid, block, 0, st, packed, rays, z, nullptr,
                           n_points, n_per_ray, out, nullptr, ld);
    return check_launch("nerf_forward_rays");
}

int nerfmi_nerf_forward_embedded(const float *packed, const float *x, int64_t n, int sigma_only, float *out,
                                 nerfmi_stream_t stream) {
    NERFMI_ENTER();
    NERFMI_REQUIRE(n >= 0, "nerf_forward_embedded: bad size");
    if (n == 0) return NERFMI_OK;
    NERFMI_REQUIRE(packed && x && out, "nerf_forward_embedded: null pointer");
    const int64_t waves = (n + 31) / 32;
    const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (sigma_only)
        hipLaunchKernelGGL((nerf_forward_kernel<true, true, false>), grid, block, 0, st, packed, nullptr, nullptr, x, n,
                           1, out, nullptr, (int64_t)0);
    else
        hipLaunchKernelGGL((nerf_forward_kernel<true, false, false>), grid, block, 0, st, packed, nullptr, nullptr, x,
                           n, 1, out, nullptr, (int64_t)0);
    return check_launch("nerf_forward_embedded");
}

// NeRF.forward(x) on pre-embedded rows with the activations saved for nerfmi_nerf_backward_rays(n_rays = n, n_per_ray = 1)
int nerfmi_nerf_forward_embedded_train(const float *packed, const float *x, int64_t n, float *out, float *saved,
                                       nerfmi_stream_t stream) {
    NERFMI_ENTER();
    NERFMI_REQUIRE(n >= 0, "nerf_forward_embedded_train: bad size");
    if (n == 0) return NERFMI_OK;
    NERFMI_REQUIRE(packed && x && out && saved, "nerf_forward_embedded_train: null pointer");
    const int64_t waves = (n + 31) / 32;
    hipLaunchKernelGGL((nerf_forward_kernel<true, false, true>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, packed, nullptr, nullptr, x, n, 1, out, saved, pad_points(n));
    return check_launch("nerf_forward_embedded_train");
}

}  // extern "C"
