// OPT-IN fast math for the NeRF MLP forward on gfx950: fp32-equivalent products on the bf16 matrix cores.
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
#include "mlp_core.h"

namespace nerfmi {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// x = p0 + p1 + p2 exactly (each step round-to-nearest-even)
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8 (&sp)[3]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)x[j];
        const float r1 = x[j] - (float)h;
        const __bf16 m = (__bf16)r1;
        const float r2 = r1 - (float)m;
        sp[0][j] = h; sp[1][j] = m; sp[2][j] = (__bf16)r2;
    }
}
// registers 8s..8s+7 of a block are k-step s of the next layer (units 16s + 8(t>>2) + 4half + (t&3))
__device__ __forceinline__ void split_block(const f32x16 &v, bf16x8 (&out)[2][3]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        float x[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) x[t] = v[8 * s + t];
        split8(x, out[s]);
    }
}

// fast image: per layer, unit U = (kb*2 + s)*JB + jb (input-block-major: a layer walks its INPUT blocks in the
// outer loop, so each input block is split just before use and all JB accumulators advance together);
// unit -> 3 x 1 KiB: split i at bytes (unit_base + U)*3072 + i*1024 + lane*16 (8 bf16).  Built from the fp32
// fragment image: the lane's floats of groups 2s and 2s+1 of (jb,kb) ARE its 8 k-slots of k-step s.
struct FastLayer { int off, JB, KB; };
__constant__ FastLayer d_fast_layers[NL_FWD] = {
    {OFF_L1, 8, 2}, {OFF_L2, 8, 8}, {OFF_L3, 8, 8}, {OFF_L4, 8, 8}, {OFF_L5, 8, 10},
    {OFF_L6, 8, 8}, {OFF_L7, 8, 8}, {OFF_L8, 8, 8}, {OFF_FINAL, 8, 8}, {OFF_DIR, 4, 9}};

__global__ void pack_bf16x3_kernel(const float *__restrict__ packed, __bf16 *__restrict__ fast) {
    const int n_units = OFF_SMALL / 512;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n_units * 64; idx += gridDim.x * blockDim.x) {
        const int Usrc = idx >> 6, lane = idx & 63;          // source unit in (jb,kb,s) order
        int li = 0;
#pragma unroll
        for (int l = 1; l < NL_FWD; ++l)
            if (Usrc * 512 >= d_fast_layers[l].off) li = l;
        const FastLayer L = d_fast_layers[li];
        const int rel = Usrc - L.off / 512;
        const int s = rel & 1, kb = (rel >> 1) % L.KB, jb = (rel >> 1) / L.KB;
        const int Udst = L.off / 512 + (kb * 2 + s) * L.JB + jb;
        const f32x4 v0 = ldg4(packed + (int64_t)Usrc * 512 + lane * 4), v1 = ldg4(packed + (int64_t)Usrc * 512 + 256 + lane * 4);
        const float x[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        bf16x8 sp[3];
        split8(x, sp);
#pragma unroll
        for (int i = 0; i < 3; ++i)
            *reinterpret_cast<bf16x8 *>(fast + ((int64_t)Udst * 3 + i) * 512 + lane * 8) = sp[i];
    }
}

// LDS ring of stages of US units (3 KiB each)
#ifndef NERFMI_US
#define NERFMI_US 8
#endif
constexpr int US = NERFMI_US;                          // units per stage
constexpr int FSLOT = 3;
constexpr int FLDS_BYTES = FSLOT * US * 3072;
constexpr int PIECES = US * 3;                         // 1 KiB pieces per stage
constexpr int QP = PIECES / 4;                         // pieces per wave per stage

// acc[jb] = bias + W . [in0 ; act(in1)] with six bf16 MFMAs per (output block, k-step).
// in0 / in1 are fp32 blocks in accumulator layout; in1 is passed through ReLU when RELU1 (the previous layer's raw
// outputs).  use(kb, v) is called with each in1 block as it is consumed (training: store + sign mask).
template <int KB0, int KB1, int JB, bool RELU1, class Use>
__device__ __forceinline__ void layer_bf16x3(const __bf16 *__restrict__ wbase, const float *__restrict__ bias,
                                             const f32x16 *in0, const f32x16 *in1, f32x16 *acc, char *wlds, int wid,
                                             int lane, Use use) {
    constexpr int KBT = KB0 + KB1;
    constexpr int NU = JB * KBT * 2;                    // units in this layer
    constexpr int NP = NU * 3;                          // 1 KiB pieces
    constexpr int NST = (NU + US - 1) / US;
    const char *gsrc = reinterpret_cast<const char *>(wbase) + (QP * wid) * 1024 + lane * 16;
    char *ldst = wlds + (QP * wid) * 1024 + lane * 16;
    const char *lsrc = wlds + lane * 16;
    f32x4 st[QP];
    auto gload = [&](int stage) {
#pragma unroll
        for (int i = 0; i < QP; ++i)
            if ((stage + 1) * PIECES <= NP || stage * PIECES + QP * wid + i < NP)
                st[i] = *reinterpret_cast<const f32x4 *>(gsrc + (int64_t)(stage * PIECES + i) * 1024);
    };
    auto lwrite = [&](int stage) {
#pragma unroll
        for (int i = 0; i < QP; ++i)
            *reinterpret_cast<f32x4 *>(ldst + ((stage % FSLOT) * PIECES + i) * 1024) = st[i];
    };
    __syncthreads();
    gload(0);
    lwrite(0);
    if (NST > 1) gload(1);
    bf16x8 an[3];
    bf16x8 bs[2][3];
#pragma unroll
    for (int jb = 0; jb < JB; ++jb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b = ldg4(bias + 32 * jb + 8 * q);
            acc[jb][4 * q + 0] = b[0]; acc[jb][4 * q + 1] = b[1]; acc[jb][4 * q + 2] = b[2]; acc[jb][4 * q + 3] = b[3];
        }
#pragma unroll
    for (int kb = 0; kb < KBT; ++kb) {
        // split this input block just before use: its VALU work issues in the shadow of the previous block's MFMAs
        f32x16 v = (kb < KB0) ? in0[kb] : in1[kb - KB0];
        if (kb >= KB0) {
            if (RELU1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            use(kb - KB0, v);
        }
#ifdef FX_NOSPLIT
        if (kb == 0) split_block(v, bs);
#else
        split_block(v, bs);
#endif
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int jb = 0; jb < JB; ++jb) {
                const int U = (kb * 2 + s) * JB + jb;
                const int stage = U / US, ul = U % US;
                // Stage boundary: stage+1 goes to LDS, stage+2's loads are issued -- and NO barrier here.  The one
                // barrier per stage sits in the MIDDLE of the stage: it publishes the slot written at this
                // boundary (needed only at the next boundary) and protects the slot the next boundary will
                // overwrite (last read in stage-1, which every wave has left by then).  Crossing a boundary
                // therefore never drains the matrix pipe: the next stage's fragments are already visible and
                // can be read ahead.
                if (ul == 0) {
#ifndef FX_NOSTREAM
                    if (stage + 1 < NST) lwrite(stage + 1);
                    if (stage + 2 < NST) gload(stage + 2);
#endif
                    if (stage == 0) __syncthreads();            // first stage of the layer: written just above
                }
#ifndef FX_NOBAR
                if (ul == US / 2 && NST > 1) __syncthreads();
#endif
                // weight fragments are read one unit ahead (two register sets), so a unit's LDS latency hides
                // behind the previous unit's six MFMAs
                auto lread = [&](int uu, int t) {
                    return *reinterpret_cast<const bf16x8 *>(lsrc + (((uu / US) % FSLOT) * PIECES + (uu % US) * 3 + t) * 1024);
                };
                if (U == 0) { an[0] = lread(0, 0); an[1] = lread(0, 1); an[2] = lread(0, 2); }
                const bf16x8 a1 = an[0], a2 = an[1], a3 = an[2];
#ifndef FX_NOLDSR
                if (U + 1 < NU) { an[0] = lread(U + 1, 0); an[1] = lread(U + 1, 1); an[2] = lread(U + 1, 2); }
#endif
                f32x16 c = acc[jb];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, bs[s][0], c, 0, 0, 0);      // small terms first
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bs[s][2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, bs[s][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, bs[s][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bs[s][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bs[s][0], c, 0, 0, 0);
                acc[jb] = c;
            }
    }
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

template <bool SIGMA_ONLY>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
nerf_forward_bf16x3_kernel(const float *__restrict__ packed, const __bf16 *__restrict__ fast,
                           const float *__restrict__ rays, const float *__restrict__ z, int64_t n_points,
                           int n_per_ray, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char wlds[];
    const int lane = threadIdx.x & 63, half = lane >> 5, wid = threadIdx.x >> 6;
    NERFMI_TSF(0);
    const int64_t wave = (int64_t)blockIdx.x * 4 + wid;
    const int64_t praw = wave * 32 + (lane & 31);
    const bool ok = praw < n_points;
    const int64_t p = ok ? praw : n_points - 1;
    const float *rr = rays + (p / n_per_ray) * 8;
    const float zz = z[p];
    const float x = __fadd_rn(rr[0], __fmul_rn(rr[3], zz));
    const float y = __fadd_rn(rr[1], __fmul_rn(rr[4], zz));
    const float w = __fadd_rn(rr[2], __fmul_rn(rr[5], zz));
    f32x16 e[2];
    embed_xyz_blocks_f(x, y, w, half, e);
    const float *bias = packed + OFF_BIAS + 4 * half;
    auto img = [&](int off) { return fast + (int64_t)(off / 512) * 1536; };      // unit * 3 pieces * 512 bf16
    auto nouse = [](int, const f32x16 &) {};
    f32x16 h[8], acc[8];
    auto copy8 = [&]() {
#pragma unroll
        for (int b = 0; b < 8; ++b) h[b] = acc[b];
    };
    NERFMI_TSF(1);
    layer_bf16x3<2, 0, 8, false>(img(OFF_L1), bias, e, nullptr, acc, wlds, wid, lane, nouse);
    copy8();
    NERFMI_TSF(2);
    for (int l = 1; l <= 3; ++l) {
        layer_bf16x3<0, 8, 8, true>(img(OFF_L2 + (l - 1) * SZ_HID), bias + 256 * l, nullptr, h, acc, wlds, wid, lane, nouse);
        copy8();
        NERFMI_TSF(2 + l);
    }
    layer_bf16x3<2, 8, 8, true>(img(OFF_L5), bias + 256 * 4, e, h, acc, wlds, wid, lane, nouse);
    copy8();
    NERFMI_TSF(6);
    for (int l = 5; l <= 7; ++l) {                       // xyz_encoding_6..8
        layer_bf16x3<0, 8, 8, true>(img(OFF_L6 + (l - 5) * SZ_HID), bias + 256 * l, nullptr, h, acc, wlds, wid, lane, nouse);
        copy8();
        NERFMI_TSF(2 + l);
    }
    // h = raw outputs of xyz_encoding_8; sigma = w_sigma . relu(h) + b (nerf.py:112)
    float sigma;
    {
        f32x16 h8[8];
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) h8[b][r] = fmaxf(h[b][r], 0.f);
        sigma = dot_blocks<8>(h8, packed + OFF_W_SIGMA + 4 * half) + packed[OFF_B_SIGMA];
    }
    if (SIGMA_ONLY) {
        if (ok && half == 0) out[p] = sigma;
        return;
    }
    layer_bf16x3<0, 8, 8, true>(img(OFF_FINAL), bias + 256 * 8, nullptr, h, acc, wlds, wid, lane, nouse);
    copy8();                                             // xyz_encoding_final: no activation on its output
    NERFMI_TSF(10);
    f32x16 de[1], dh[4];
    embed_dir_block_f(rr[3], rr[4], rr[5], half, de[0]);
    // dir_encoding input = [final (no ReLU) | dir embedding]: the packed order is final first (mlp_layout.h)
    layer_bf16x3<8, 1, 4, false>(img(OFF_DIR), packed + OFF_BIAS_DIR + 4 * half, h, de, dh, wlds, wid, lane, nouse);
    NERFMI_TSF(11);
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) dh[b][r] = fmaxf(dh[b][r], 0.f);
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

size_t nerfmi_nerf_fast_bytes(void) { return (size_t)(OFF_SMALL / 512) * 3072; }

int nerfmi_nerf_pack_fast(const float *packed, void *fast, nerfmi_stream_t stream) {
    NERFMI_REQUIRE(packed && fast, "nerf_pack_fast: null pointer");
    hipLaunchKernelGGL(pack_bf16x3_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, packed, (__bf16 *)fast);
    return check_launch("nerf_pack_fast");
}

int nerfmi_nerf_forward_rays_fast(const float *packed, const void *fast, const float *rays, const float *z, int n_rays,
                                  int n_per_ray, int sigma_only, float *out, nerfmi_stream_t stream) {
    NERFMI_REQUIRE(n_rays >= 0 && n_per_ray >= 1, "nerf_forward_rays_fast: bad sizes");
    const int64_t n_points = (int64_t)n_rays * n_per_ray;
    if (n_points == 0) return NERFMI_OK;
    NERFMI_REQUIRE(packed && fast && rays && z && out, "nerf_forward_rays_fast: null pointer");
    static thread_local bool attr_set = false;
    if (!attr_set) {
        hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void *>(nerf_forward_bf16x3_kernel<false>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, FLDS_BYTES);
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void *>(nerf_forward_bf16x3_kernel<true>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, FLDS_BYTES);
        if (e1 != hipSuccess || e2 != hipSuccess) {
            (void)hipGetLastError();
            set_error("nerf_forward_rays_fast: cannot raise the dynamic LDS limit");
            return NERFMI_E_LAUNCH;
        }
        attr_set = true;
    }
    const int64_t waves = (n_points + 31) / 32;
    const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (sigma_only)
        hipLaunchKernelGGL((nerf_forward_bf16x3_kernel<true>), grid, block, FLDS_BYTES, st, packed, (const __bf16 *)fast,
                           rays, z, n_points, n_per_ray, out);
    else
        hipLaunchKernelGGL((nerf_forward_bf16x3_kernel<false>), grid, block, FLDS_BYTES, st, packed, (const __bf16 *)fast,
                           rays, z, n_points, n_per_ray, out);
    return check_launch("nerf_forward_rays_fast");
}

}  // extern "C"
