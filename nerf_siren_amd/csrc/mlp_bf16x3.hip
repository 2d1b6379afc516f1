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
constexpr int FSLOT = 4;
constexpr int FLDS_BYTES = FSLOT * US * 3072;          // 96 KiB
constexpr int PIECES = US * 3;                         // 1 KiB pieces per stage
constexpr int QP = PIECES / 4;                         // pieces per wave per stage
static_assert(QP <= US, "one staging piece per unit");
constexpr int FAST_TAIL_BYTES = 2 * US * 3072;         // the stream prefetches two stages past the last layer

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct FastStage {
    f32x4 st[QP];                                      // this wave's pieces of the stage in flight
};

// Exact three-way split of TWO fp32 values into packed bf16 words (low half = x0's term): w[i] = {bf16_i(x0), bf16_i(x1)}.
// Written with explicit instructions: the compiler otherwise vectorises the subtractions into v_pk_add_f32, and
// packed-fp32 instructions do NOT overlap with bf16 MFMAs (tools/ubench/mfma_valu.hip: 58 cycles for two) while
// plain VALU instructions do (about five per 32-cycle MFMA).
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned (&w)[3]) {
    float r0 = x0, r1 = x1;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        unsigned p;
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p) : "v"(r0), "v"(r1));
        w[i] = p;
        if (i < 2) {
            float f0, f1;
            asm("v_lshlrev_b32 %0, 16, %1" : "=v"(f0) : "v"(p));
            asm("v_and_b32 %0, 0xffff0000, %1" : "=v"(f1) : "v"(p));
            asm("v_sub_f32 %0, %1, %2" : "=v"(r0) : "v"(r0), "v"(f0));
            asm("v_sub_f32 %0, %1, %2" : "=v"(r1) : "v"(r1), "v"(f1));
        }
    }
}

// acc[jb] = bias + W . [in0 ; act(in1)] with six bf16 MFMAs per (output block, k-step).
// in0 / in1 are fp32 blocks in accumulator layout; in1 is passed through ReLU when RELU1 (the previous layer's raw
// outputs).  The layer walks its INPUT blocks in the outer loop, all JB accumulators advance together; `acc` must not
// alias the inputs (callers alternate two buffers).
//  * weights: the workgroup's shared stream (see mlp_core.h layer_mfma_lds): 4-slot LDS ring of 24 KiB stages, one
//    staging piece written + reloaded per unit, one barrier per stage, fragments read one unit ahead, and the stream
//    runs across layer ends (FIRST = false: stage 0 is already in LDS, stage 1 in `fs`);
//  * activations: block kb+1 is split into its bf16 terms pair by pair BETWEEN the units of block kb, so the ~13
//    vector instructions per pair issue in the shadow of the XDL MFMAs instead of as a 100-instruction clump.
//  * hooks: pre0(kb, p, x0, x1) / pre1(kb, p, x0, x1) see each PAIR of values (registers 2p, 2p+1) of an in0 / in1
//    block as it is consumed -- after the ReLU for in1 -- and may change them (backward: ReLU mask) and store them
//    (training: the saved activation images are written here, in the shadow of the MFMAs, not in an epilogue).
struct NoHook {
    __device__ __forceinline__ void operator()(int, int, float &, float &) const {}
};

template <int KB0, int KB1, int JB, bool RELU1, bool FIRST, class Pre0 = NoHook, class Pre1 = NoHook>
__device__ __forceinline__ void layer_bf16x3(const __bf16 *__restrict__ wbase, const float *__restrict__ bias,
                                             const f32x16 *in0, const f32x16 *in1, f32x16 *acc, char *wlds,
                                             FastStage &fs, int wid, int lane, Pre0 pre0 = Pre0(), Pre1 pre1 = Pre1()) {
    constexpr int KBT = KB0 + KB1;
    constexpr int UK = 2 * JB;                          // units per input block
    constexpr int NU = UK * KBT;                        // units in this layer
    static_assert(NU % US == 0, "a layer is a whole number of stages");
    const char *gsrc = reinterpret_cast<const char *>(wbase) + (QP * wid) * 1024 + lane * 16;
    char *ldst = wlds + (QP * wid) * 1024 + lane * 16;
    const char *lsrc = wlds + lane * 16;
    auto gload = [&](int stage, int i) {
        fs.st[i] = *reinterpret_cast<const f32x4 *>(gsrc + (int64_t)(stage * PIECES + i) * 1024);
    };
    auto lwrite = [&](int stage, int i) {
        *reinterpret_cast<f32x4 *>(ldst + ((stage % FSLOT) * PIECES + i) * 1024) = fs.st[i];
    };
    auto lread = [&](int uu, int t) {
        return *reinterpret_cast<const u32x4 *>(lsrc + (((uu / US) % FSLOT) * PIECES + (uu % US) * 3 + t) * 1024);
    };
    if (FIRST) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < QP; ++i) gload(0, i);
#pragma unroll
        for (int i = 0; i < QP; ++i) lwrite(0, i);
#pragma unroll
        for (int i = 0; i < QP; ++i) gload(1, i);
        __syncthreads();
    }
#pragma unroll
    for (int jb = 0; jb < JB; ++jb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b = bias ? ldg4(bias + 32 * jb + 8 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
            acc[jb][4 * q + 0] = b[0]; acc[jb][4 * q + 1] = b[1]; acc[jb][4 * q + 2] = b[2]; acc[jb][4 * q + 3] = b[3];
        }
    // bf16 terms of the current and the next input block: word [s][i][j] = values 8s+2j, 8s+2j+1 of the block, term i
    unsigned cur[2][3][4], nxt[2][3][4];
    auto block = [&](int kb) { return (kb < KB0) ? in0[kb] : in1[kb - KB0]; };
    auto split_pairs = [&](int kb, int p0, int p1, unsigned (&dst)[2][3][4]) {
        const f32x16 v = block(kb);
#pragma unroll
        for (int p = p0; p < p1; ++p) {
            float x0 = v[2 * p], x1 = v[2 * p + 1];
            if (RELU1 && kb >= KB0) { x0 = relu1(x0); x1 = relu1(x1); }
            if (kb < KB0) pre0(kb, p, x0, x1);
            else pre1(kb - KB0, p, x0, x1);
            unsigned w[3];
            split_pair(x0, x1, w);
            dst[p >> 2][0][p & 3] = w[0]; dst[p >> 2][1][p & 3] = w[1]; dst[p >> 2][2][p & 3] = w[2];
        }
    };
    split_pairs(0, 0, 8, cur);
    u32x4 an[3];
#pragma unroll
    for (int kb = 0; kb < KBT; ++kb) {
#pragma unroll
        for (int u = 0; u < UK; ++u) {
            const int s = u / JB, jb = u % JB;
            const int U = kb * UK + u;
            const int stage = U / US, ul = U % US;
            if (ul < QP) {                               // one staging piece per unit
                lwrite(stage + 1, ul);
                gload(stage + 2, ul);
            }
            if (ul == QP) __syncthreads();               // publishes stage+1, protects the slot stage+2 will take
            if (U == 0) { an[0] = lread(0, 0); an[1] = lread(0, 1); an[2] = lread(0, 2); }
            const bf16x8 a1 = __builtin_bit_cast(bf16x8, an[0]), a2 = __builtin_bit_cast(bf16x8, an[1]),
                         a3 = __builtin_bit_cast(bf16x8, an[2]);
            if (U + 1 < NU) { an[0] = lread(U + 1, 0); an[1] = lread(U + 1, 1); an[2] = lread(U + 1, 2); }
            const bf16x8 b1 = __builtin_bit_cast(bf16x8, u32x4{cur[s][0][0], cur[s][0][1], cur[s][0][2], cur[s][0][3]});
            const bf16x8 b2 = __builtin_bit_cast(bf16x8, u32x4{cur[s][1][0], cur[s][1][1], cur[s][1][2], cur[s][1][3]});
            const bf16x8 b3 = __builtin_bit_cast(bf16x8, u32x4{cur[s][2][0], cur[s][2][1], cur[s][2][2], cur[s][2][3]});
            f32x16 c = acc[jb];
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, c, 0, 0, 0);      // small terms first
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, c, 0, 0, 0);
            acc[jb] = c;
            // the next input block's split, 8 pairs spread over this block's UK units
            if (kb + 1 < KBT) split_pairs(kb + 1, (u * 8) / UK, ((u + 1) * 8) / UK, nxt);
            // keep every unit's staging piece, fragment reads and split pair WITH its six MFMAs: left alone the
            // scheduler bunches the staging of a whole stage and the split of a whole block together, and a clump
            // of more than ~5 vector instructions per MFMA is no longer hidden behind the matrix pipe
            // ... and inside the unit: next unit's fragment reads first (a whole unit of MFMAs to land), then the
            // staging piece, then MFMAs with the vector work dealt out between them
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);       // DS read
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);       // DS write
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);       // VMEM read
#pragma unroll
            for (int m = 0; m < 6; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // VALU
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (kb + 1 < KBT) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b)
#pragma unroll
                    for (int d = 0; d < 4; ++d) cur[a][b][d] = nxt[a][b][d];
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
    auto img = [&](int off) { return fast + (int64_t)(off / 512) * 1536; };      // unit * 3 pieces * 512 bf16
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
            unsigned one;
            asm("v_min_u32 %0, 1, %1" : "=v"(one) : "v"(x0));
            asm("v_lshl_or_b32 %0, %1, %2, %0" : "+v"(mk[kb >> 1]) : "v"(one), "s"(16 * (kb & 1) + r));
            asm("v_min_u32 %0, 1, %1" : "=v"(one) : "v"(x1));
            asm("v_lshl_or_b32 %0, %1, %2, %0" : "+v"(mk[kb >> 1]) : "v"(one), "s"(16 * (kb & 1) + r + 1));
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

size_t nerfmi_nerf_fast_bytes(void) { return (size_t)(OFF_SMALL / 512) * 3072 + FAST_TAIL_BYTES; }

int nerfmi_nerf_pack_fast(const float *packed, void *fast, nerfmi_stream_t stream) {
    NERFMI_REQUIRE(packed && fast, "nerf_pack_fast: null pointer");
    hipLaunchKernelGGL(pack_bf16x3_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, packed, (__bf16 *)fast);
    return check_launch("nerf_pack_fast");
}

int nerfmi_nerf_forward_rays_fast(const float *packed, const void *fast, const float *rays, const float *z, int n_rays,
                                  int n_per_ray, int sigma_only, float *out, float *saved, nerfmi_stream_t stream) {
    NERFMI_REQUIRE(n_rays >= 0 && n_per_ray >= 1, "nerf_forward_rays_fast: bad sizes");
    const int64_t n_points = (int64_t)n_rays * n_per_ray;
    if (n_points == 0) return NERFMI_OK;
    NERFMI_REQUIRE(packed && fast && rays && z && out, "nerf_forward_rays_fast: null pointer");
    NERFMI_REQUIRE(!(saved && sigma_only), "nerf_forward_rays_fast: saved activations need the full (rgb,sigma) pass");
    const void *kernels[3] = {reinterpret_cast<const void *>(nerf_forward_bf16x3_kernel<false, false>),
                              reinterpret_cast<const void *>(nerf_forward_bf16x3_kernel<true, false>),
                              reinterpret_cast<const void *>(nerf_forward_bf16x3_kernel<false, true>)};
    static thread_local bool attr_set = false;
    if (!attr_set) {
        for (const void *k : kernels)
            if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, FLDS_BYTES) != hipSuccess) {
                (void)hipGetLastError();
                set_error("nerf_forward_rays_fast: cannot raise the dynamic LDS limit");
                return NERFMI_E_LAUNCH;
            }
        attr_set = true;
    }
    const int64_t waves = (n_points + 31) / 32;
    const int64_t ld = waves * 32;
    const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
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
