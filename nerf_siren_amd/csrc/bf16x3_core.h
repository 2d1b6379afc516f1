// Split-bf16 ("bf16x3") layer shared by the forward (mlp_bf16x3.hip) and the backward dX chain (mlp_bwd.hip).
// See mlp_bf16x3.hip for the arithmetic (three exact bf16 terms per fp32 value, six bf16 MFMAs per product).
#pragma once
#include "mlp_core.h"

namespace nerfmi {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// Fast image: [forward layers' units][transposed (backward) layers' units][stream tail]; a unit = 3 x 1 KiB.
constexpr int FAST_FWD_UNITS = OFF_SMALL / 512;
constexpr int FAST_T_UNITS = (PACKED_FLOATS - STREAM_TAIL - OFF_TRANS) / 512;
// bf16 element offset of the fast image of the forward layer image at float offset `off` / of a transposed one
__device__ __host__ constexpr int64_t fast_fwd_elems(int off) { return (int64_t)(off / 512) * 1536; }
__device__ __host__ constexpr int64_t fast_t_elems(int off) { return (int64_t)(FAST_FWD_UNITS + (off - OFF_TRANS) / 512) * 1536; }

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
// Mostly written with explicit instructions: the compiler otherwise vectorises the subtractions into v_pk_add_f32, and
// packed-fp32 instructions do NOT overlap with bf16 MFMAs (tools/ubench/mfma_valu.hip: 58 cycles for two) while
// plain VALU instructions do (about five per 32-cycle MFMA).
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned (&w)[3]) {
    float r0 = x0, r1 = x1;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        // the conversion through the compiler, NOT inline asm: its result is an MFMA B operand, and gfx950 needs two wait
        // states between a VALU write and an MFMA read of the same register.  The hazard recogniser does not look inside
        // inline asm, so an asm v_cvt_pk_bf16_f32 scheduled right in front of its MFMA handed the matrix core the
        // register's STALE contents (round 2: scattered garbage / NaN in the FiLM-SIREN split-bf16 kernel, depending on
        // whether the s_waitcnt in between happened to stall).  The shift / mask / subtract stay asm (they feed only
        // vector instructions): left to the compiler they become v_pk_add_f32, which does not overlap with the MFMAs.
        const unsigned p = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{r0, r1}, bf16x2_t));
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

//  * the accumulators start from bias * bias_scale (bias_scale = 1: plain bias; the backward chain starts d h8 from
//    w_sigma * d sigma this way); bias == nullptr starts from zero.
//  * PH: ring phase of the layer's stage 0 (slot = (stage + PH) % 4); 0 when every earlier layer of the kernel is a
//    multiple of four stages.
template <int KB0, int KB1, int JB, bool RELU1, bool FIRST, class Pre0 = NoHook, class Pre1 = NoHook, int PH = 0>
__device__ __forceinline__ void layer_bf16x3(const __bf16 *__restrict__ wbase, const float *__restrict__ bias,
                                             const f32x16 *in0, const f32x16 *in1, f32x16 *acc, char *wlds,
                                             FastStage &fs, int wid, int lane, Pre0 pre0 = Pre0(), Pre1 pre1 = Pre1(),
                                             float bias_scale = 1.0f) {
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
        *reinterpret_cast<f32x4 *>(ldst + (((stage + PH) % FSLOT) * PIECES + i) * 1024) = fs.st[i];
    };
    auto lread = [&](int uu, int t) {
        return *reinterpret_cast<const u32x4 *>(lsrc + (((uu / US + PH) % FSLOT) * PIECES + (uu % US) * 3 + t) * 1024);
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
            f32x4 b = bias ? ldg4(bias + 32 * jb + 8 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
            if (bias_scale != 1.0f) b = b * bias_scale;
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
// unit -> 3 x 1 KiB: split i at bytes (unit0 + U)*3072 + i*1024 + lane*16 (8 bf16).  Built from the fp32
// fragment image: the lane's floats of groups 2s and 2s+1 of (jb,kb) ARE its 8 k-slots of k-step s.
struct FastLayer { int off, JB, KB, unit0; };   // float offset in `packed`, output blocks, input blocks, first fast unit
constexpr int MAX_FAST_LAYERS = 20;
struct FastTable {
    FastLayer l[MAX_FAST_LAYERS];
    int n, n_units;
};

static __global__ void pack_bf16x3_table_kernel(FastTable T, const float *__restrict__ packed, __bf16 *__restrict__ fast) {
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < T.n_units * 64; idx += gridDim.x * blockDim.x) {
        const int U = idx >> 6, lane = idx & 63;             // fast unit index; its layer, then its source unit (jb,kb,s)
        int li = 0;
        for (int l = 1; l < T.n; ++l)
            if (U >= T.l[l].unit0) li = l;
        const FastLayer L = T.l[li];
        const int rel = U - L.unit0;                         // destination order: (kb*2 + s)*JB + jb
        const int jb = rel % L.JB, s = (rel / L.JB) & 1, kb = rel / (2 * L.JB);
        const float *src = packed + L.off + (int64_t)((jb * L.KB + kb) * 2 + s) * 512;
        const f32x4 v0 = ldg4(src + lane * 4), v1 = ldg4(src + 256 + lane * 4);
        const float x[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        bf16x8 sp[3];
        split8(x, sp);
#pragma unroll
        for (int i = 0; i < 3; ++i)
            *reinterpret_cast<bf16x8 *>(fast + ((int64_t)U * 3 + i) * 512 + lane * 8) = sp[i];
    }
}

}  // namespace nerfmi
