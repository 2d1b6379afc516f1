// Register-resident dense layer on the gfx950 matrix cores, shared by the NeRF
// forward (mlp.hip) and the backward dX chain (mlp_bwd.hip).  See mlp_layout.h
// for the operand maps.
#pragma once
#include "common.h"
#include "mlp_layout.h"

namespace nerfmi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));


__device__ __forceinline__ f32x4 ldg4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }

// sin / cos for |x| up to ~1e5 with ~1 ulp error and no branches: Cody-Waite reduction by pi/2 in two fma
// steps (hi + lo split of pi/2), then the degree-7 sine / degree-8 cosine minimax polynomials on
// [-pi/4, pi/4].  ~20 VALU instructions, against ~70 plus a divergent huge-argument path for the libm
// call -- this matters where the sines ARE the VALU load: 2304 per point in the FiLM-SIREN field.
__device__ __forceinline__ void sincos_cw(float x, float &s_out, float &c_out) {
    const float j = rintf(x * 0.63661977236758134308f);              // x * 2/pi
    float r = __builtin_fmaf(-j, 1.57079637050628662109375f, x);     // pi/2 hi (fp32)
    r = __builtin_fmaf(-j, -4.37113900018624283e-8f, r);             // pi/2 lo
    const float s2 = r * r;
    float ps = __builtin_fmaf(s2, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = __builtin_fmaf(s2, ps, -1.6666654611e-1f);
    const float sn = __builtin_fmaf(r * s2, ps, r);
    float pc = __builtin_fmaf(s2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = __builtin_fmaf(s2, pc, 4.166664568298827e-2f);
    const float cs = __builtin_fmaf(s2 * s2, pc, __builtin_fmaf(-0.5f, s2, 1.0f));
    const int q = (int)j;
    const float sv = (q & 1) ? cs : sn, cv = (q & 1) ? sn : cs;
    s_out = (q & 2) ? -sv : sv;
    c_out = ((q + 1) & 2) ? -cv : cv;
}
__device__ __forceinline__ float sin_cw(float x) {
    float s, c;
    sincos_cw(x, s, c);
    return s;
}

// sin(x) by reduction modulo PI: x = j*pi + r, |r| <= pi/2, sin x = (-1)^j sin r, and cos x = (-1)^j cos r with
// cos r >= 0 -- so ONE odd polynomial (degree 9: max abs error 1.4e-7 = 2.3 ulp of 1 for |x| <= 300, mean 1.6e-8,
// fitted in tools/fit_sin.py; the degree-11 fit is no better in fp32 -- 2.0 ulp, the rounding of the last steps
// dominates -- and costs one more fma on each of the 2304 sines per point) serves every quadrant and the parity of j
// is at once the sign of the sine's flip and the SIGN OF THE COSINE, which is all the FiLM-SIREN backward needs
// beside the saved sine.  j is rounded with the 1.5 * 2^23 magic constant, so its parity is bit 0 of `jbits`.
// 12 VALU instructions against ~20 for sincos_cw -- the sines are the FiLM-SIREN field's whole vector load.
__device__ __forceinline__ float sin_pi(float x, unsigned &jbits) {
    const float MAGIC = 12582912.0f;
    const float jm = __builtin_fmaf(x, 0.31830988618379067f, MAGIC);
    const float j = jm - MAGIC;
    float r = __builtin_fmaf(-j, 3.14159274101257324f, x);            // pi hi (fp32)
    r = __builtin_fmaf(-j, -8.74227765734758577e-8f, r);               // pi lo
    const float r2 = r * r;
#ifdef NERFMI_EXP_SIN11
    float p = __builtin_fmaf(-2.3846691732387626e-08f, r2, 2.752261934801936e-06f);
    p = __builtin_fmaf(p, r2, -0.00019840804452542216f);
    p = __builtin_fmaf(p, r2, 0.008333330042660236f);
    p = __builtin_fmaf(p, r2, -0.1666666716337204f);
#else
    float p = __builtin_fmaf(2.600054585855105e-06f, r2, -0.00019806614727713168f);
    p = __builtin_fmaf(p, r2, 0.008333017118275166f);
    p = __builtin_fmaf(p, r2, -0.16666656732559204f);
#endif
    const float s = __builtin_fmaf(r * r2, p, r);
    jbits = __float_as_uint(jm);
    return __uint_as_float(__float_as_uint(s) ^ (jbits << 31));
}

// layer contract (layer_mfma_lds below):  out[jb] = epi(bias + W[jb-block rows] . [in0 ; in1])   (JB x (KB0+KB1) blocks)
//   bias : natural order + 4*half, or nullptr for a zero start (backward)
//   pre  : called when a block STARTS; whatever it loads is in flight under the block's MFMAs and handed to epi
//   epi  : per-block epilogue (ReLU / mask / stores) in four slices: epi(jb, q, v, pv) receives registers
//          4q..4q+3 of block jb (units 8q + 4*half + {0..3}) and returns their final values, which go to `out`
//          (`out` must not alias the inputs: callers alternate two activation buffers).
// The slices of block jb-1 are placed between the MFMA groups of block jb, so their stores and loads are spread
// over the block instead of queueing up at its end.  Their VECTOR instructions are not free there: on gfx950 a
// wave's fp32 MFMAs do not overlap with its own VALU work (tools/ubench/mfma_valu.hip: 64 + 13 + 4.5 cycles per
// instruction for an MFMA followed by V vector instructions, against 32-cycle bf16 MFMAs that hide ~5), so the
// epilogues are written for minimum instruction count.

// ---------------------------------------------------------------------------------------------------
// layer_mfma_lds: the same layer with the weight stream shared by the workgroup's four waves through LDS.
//
// Each wave streaming every fragment from L2 itself (layer_mfma) costs 4x the L1/L2 transactions and caps
// the prefetch distance at what a register ring can hold (6-12 fragments = 1.5-3k cycles) -- and because a
// wave's vector-memory operations retire IN ORDER, anything slow in that queue (an activation store, an HBM
// load) then stalls the next weight wait.  Here the four waves, which all walk the same packed image, fetch
// a quarter each: the stream is cut into stages of GS fragments (16 KiB); at a stage boundary a wave
// (1) writes the quarter it loaded ONE boundary ago into LDS slot (s+1)%4, (2) issues the global loads of
// stage s+2 into 16 VGPRs, then runs stage s from LDS (ds_read_b128 per four MFMAs, read one group ahead);
// the single barrier per stage sits in the MIDDLE of the stage: it publishes the slot written at this
// stage's boundary (read from the next boundary on) and protects the slot the next boundary overwrites.
//
// The stream does not stop at a layer's end: the images of consecutive layers are contiguous in `packed` in
// execution order (mlp_layout.h), so stages NST and NST+1 of a layer ARE stages 0 and 1 of the next one.  A
// layer that is not the first of its kernel (FIRST = false) finds its stage 0 already published in LDS and
// its stage 1 in the staging registers `st` -- no cold start (two serialized L2 round trips + a barrier)
// per layer, and the loads keep flying during the epilogue between layers.  PH is the ring phase of the
// layer's stage 0 (slot = (stage + PH) % 4); layers called from a runtime loop need NST % 4 == 0.  After
// the last layer the two prefetched stages are simply never used (they read valid bytes of `packed`).
// ---------------------------------------------------------------------------------------------------
#ifndef NERFMI_GS
#define NERFMI_GS 16
#endif
constexpr int GS = NERFMI_GS;              // fragments (1 KiB each) per stage
constexpr int NSLOT = 4;
constexpr int WLDS_FLOATS = NSLOT * GS * 256;   // 64 KiB
constexpr int QS = GS / 4;                 // fragments per wave per stage

typedef __attribute__((address_space(3))) void lds_void_t;    // operands of __builtin_amdgcn_global_load_lds
typedef __attribute__((address_space(1))) void gbl_void_t;

template <int GS_>
struct WeightStageT {
    f32x4 st[GS_ / 4];                     // this wave's quarter of the stage in flight
};
using WeightStage = WeightStageT<GS>;

template <int KB0, int KB1, int JB>
constexpr int layer_stages() { return JB * (KB0 + KB1) * 4 / GS; }

// GS_: fragments per stage of THIS kernel's ring (siren_core.h SIREN_GS).
// DMA (all layers of a kernel alike): the stream goes global -> LDS by LDS-DMA (global_load_lds_dwordx4: a wave's fragment is
// 64 lanes x 16 B = the 1 KiB the readers expect, no staging registers, no ds_write): during stage s a wave issues its quarter
// of stage s+2 straight into ring slot (s+2) % 4 -- free since the barrier in the middle of stage s-1, when every wave had left
// stage s-2 --, and the barrier in the middle of stage s+1 publishes it behind a COUNTED wait: everything but the newest QS
// vector-memory operations (stage s+3's pieces, issued a few groups earlier) must have landed, which covers stage s+2's
// pieces whatever stores the epilogues put into the queue since.  A raw s_barrier: __syncthreads() would drain the queue.
// A kernel that uses it ends with ring_drain(): an LDS-DMA still in flight when the workgroup's LDS is handed on would write
// into the next workgroup's.
__device__ __forceinline__ void ring_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int KB0, int KB1, int JB, int PH, bool FIRST, int GS_ = GS, bool DMA = false, class Pre, class Epi>
__device__ __forceinline__ void layer_mfma_lds(const float *__restrict__ wbase, const float *__restrict__ bias,
                                               const f32x16 *in0, const f32x16 *in1, f32x16 *out, Pre pre, Epi epi,
                                               float *wlds, WeightStageT<GS_> &ws, int wid, int lane) {
    constexpr int QS_ = GS_ / 4;
    constexpr int KBT = KB0 + KB1;
    constexpr int G = JB * KBT * 4;
    static_assert(G % GS_ == 0, "a layer is a whole number of stages");
    constexpr int NST = G / GS_;
    const float *gsrc = wbase + (QS_ * wid) * 256 + lane * 4;        // this wave's quarter of every stage
    float *ldst = wlds + (QS_ * wid) * 256 + lane * 4;
    const float *lsrc = wlds + lane * 4;
    auto gload = [&](int stage) {                                   // stage >= NST: the next layer's image
#pragma unroll
        for (int i = 0; i < QS_; ++i) ws.st[i] = ldg4(gsrc + (stage * GS_ + i) * 256);
    };
    auto lwrite = [&](int stage) {
#pragma unroll
        for (int i = 0; i < QS_; ++i)
            *reinterpret_cast<f32x4 *>(ldst + (((stage + PH) % NSLOT) * GS_ + i) * 256) = ws.st[i];
    };
    auto lread = [&](int g) {
        return *reinterpret_cast<const f32x4 *>(lsrc + (((g / GS_ + PH) % NSLOT) * GS_ + g % GS_) * 256);
    };
    const int swid = __builtin_amdgcn_readfirstlane(wid);           // (scalar: the DMA's LDS base goes through M0)
    auto dma = [&](int stage, int i) __attribute__((always_inline)) {   // fragment QS_ * wave + i of `stage` into its ring slot
        const float *src = wbase + (stage * GS_ + QS_ * swid + i) * 256 + lane * 4;
        float *dst = wlds + (((stage + PH) % NSLOT) * GS_ + QS_ * swid + i) * 256;
        __builtin_amdgcn_global_load_lds((gbl_void_t *)src, (lds_void_t *)dst, 16, 0, 0);
    };
    if (FIRST) {
        __syncthreads();
        if (DMA) {
#pragma unroll
            for (int i = 0; i < QS_; ++i) dma(0, i);
#pragma unroll
            for (int i = 0; i < QS_; ++i) dma(1, i);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        } else {
            gload(0);
            lwrite(0);
            gload(1);
            __syncthreads();
        }
    }
    f32x16 c_prev;
    f32x4 a_next;
    decltype(pre(0)) pv_prev = pre(0);
    auto run_slice = [&](int jb, int q, const f32x16 &c, decltype(pre(0)) pv) {
        const f32x4 o = epi(jb, q, f32x4{c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]}, pv);
        out[jb][4 * q] = o[0]; out[jb][4 * q + 1] = o[1]; out[jb][4 * q + 2] = o[2]; out[jb][4 * q + 3] = o[3];
    };
#pragma unroll
    for (int jb = 0; jb < JB; ++jb) {
        auto pv = (jb == 0) ? pv_prev : pre(jb);
        f32x16 c;
#ifdef NERFMI_EXP_NOBIAS
        if (false) {
#else
        if (bias) {
#endif
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b = ldg4(bias + 32 * jb + 8 * q);
                c[4 * q + 0] = b[0]; c[4 * q + 1] = b[1]; c[4 * q + 2] = b[2]; c[4 * q + 3] = b[3];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) c[r] = 0.f;
        }
#pragma unroll
        for (int kb = 0; kb < KBT; ++kb) {
            const f32x16 B = (kb < KB0) ? in0[kb] : in1[kb - KB0];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int g = (jb * KBT + kb) * 4 + q;
                const int stage = g / GS_, gl = g % GS_;
                // The stage's staging work is spread over its first groups, one piece per group: piece i of
                // stage+1 goes to LDS and its register is reloaded with piece i of stage+2.  Issued back to back
                // the four 1 KiB LDS writes (and loads) of all four waves queue on the LDS/TA pipes and the wave
                // waits; one per MFMA group is absorbed (tools/ubench/mfma_valu.hip).
                if (gl < QS_) {
                    if (DMA) {
                        dma(stage + 2, gl);
                    } else {
                        *reinterpret_cast<f32x4 *>(ldst + (((stage + 1 + PH) % NSLOT) * GS_ + gl) * 256) = ws.st[gl];
                        ws.st[gl] = ldg4(gsrc + ((stage + 2) * GS_ + gl) * 256);
                    }
                }
#ifndef NERFMI_EXP_NOBARRIER
                if (gl == GS_ / 2) {
                    if (DMA) {
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(QS_) : "memory");
                        __builtin_amdgcn_s_barrier();
                        asm volatile("" ::: "memory");
                    } else {
                        __syncthreads();
                    }
                }
#endif
                // fragments are read ONE group ahead (two register sets): the LDS latency of group g+1 hides
                // behind group g's four MFMAs instead of draining the pipe in front of every group
                const f32x4 a = (g == 0) ? lread(0) : a_next;
                if (g + 1 < G) a_next = lread(g + 1);
                // pin the read HERE, in front of this group's MFMAs: left alone the scheduler batches the reads of two
                // groups behind the previous group's last MFMA, and the first of them is then waited for with no MFMA in
                // between -- the LDS latency exposed once per eight MFMAs (measured: -2.4 % kernel time with the fence)
                __builtin_amdgcn_sched_barrier(0);
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], B[4 * q + 0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], B[4 * q + 1], c, 0, 0, 0);
                const int gq = kb * 4 + q;
                if (jb > 0 && gq % KBT == KBT / 2) {
                    // fenced: the scheduler would otherwise scatter the slice's vector instructions one by one
                    // between the MFMAs, and each interruption of the fp32 MFMA stream costs ~13 cycles on top of
                    // the instructions themselves (tools/ubench/mfma_valu.hip) -- one clump per slice is cheaper
                    __builtin_amdgcn_sched_barrier(0);
                    run_slice(jb - 1, gq / KBT, c_prev, pv_prev);
                    __builtin_amdgcn_sched_barrier(0);
                }
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], B[4 * q + 2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], B[4 * q + 3], c, 0, 0, 0);
            }
        }
        c_prev = c;
        pv_prev = pv;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) run_slice(JB - 1, q, c_prev, pv_prev);
}

// Activation images kept for training (mlp_layout.h S_* rows, mlp_bwd.hip W_* rows) are TILE-MAJOR:
// element (row, point p) lives at base[((p/32)*ROWS + row)*32 + p%32].  A wave owns exactly one 32-point
// tile, so everything it saves is one contiguous ROWS*128-byte region, and the dW GEMM later streams
// 32-point tiles of whole row ranges (1 KiB per wave-load).
// In the accumulator layout a lane holds, for ITS point, four consecutive units (registers 4q..4q+3): a slice
// is saved with four 4-byte stores, each covering two full 128-byte rows.
struct RowImage {
    float *tile;         // base + tile*ROWS*32
    int lane;
    bool ok;             // point < n_points
    bool live;           // the wave owns at least one real point
    // A wave past the end (it exists because the workgroup's waves share barriers) is pointed at the DUMP
    // tile, one spare tile behind the image's n_tiles real ones, so that the stores in the layers' epilogues
    // need no branch: a branch per store would cut the straight-line MFMA stream into basic blocks and pin the
    // epilogue's vector work between them instead of letting it issue in the MFMAs' shadow.
    __device__ __forceinline__ void init(float *base, int64_t tile_idx, int64_t n_tiles, int rows, int lane_, bool ok_,
                                         bool live_) {
#ifdef NERFMI_EXP_DUMMYSTORE
        tile = base + (live_ ? tile_idx % NERFMI_EXP_DUMMYSTORE : n_tiles) * (int64_t)(rows * 32);   // experiment: stores that stay in cache
#else
        tile = base + (live_ ? tile_idx : n_tiles) * (int64_t)(rows * 32);
#endif
        lane = lane_;
        ok = ok_;
        live = live_;
    }
    // scalar element (row, this lane's point)
    __device__ __forceinline__ float *at(int r) const { return tile + r * 32 + (lane & 31); }
};

// store registers 4q..4q+3 of a block (units row0 + 8q + 4*half + {0..3}, row0 = block's first row)
__device__ __forceinline__ void store_slice(const RowImage &im, int row0, int q, f32x4 v) {
    // columns past n_points of the last tile are stored as computed (from the clamped last point): the
    // backward chain's dZ is exactly 0 there (its grad_out is), which is what keeps them out of dW
    // Four 4-byte stores, each writing two full 128-byte rows (units 4*half + t of the slice, 32 points).  The
    // alternative -- a 4x4 cross-lane transpose (16 DPP/select instructions) feeding one 16-byte store -- is
    // slower here: fp32 MFMAs do not overlap with the wave's own vector instructions, so the transpose is paid
    // in matrix-pipe time (measured: forward+save -1.7 %, backward chain -2 % with the plain stores).
    // Non-temporal: streamed once and read back by a later kernel, so the activation stream should not displace
    // the L2-resident weights.
    float *dst = im.tile + (row0 + 8 * q + 4 * (im.lane >> 5)) * 32 + (im.lane & 31);
#pragma unroll
#ifdef NERFMI_EXP_TSTORE
    for (int t = 0; t < 4; ++t) dst[32 * t] = v[t];
#else
    for (int t = 0; t < 4; ++t) __builtin_nontemporal_store(v[t], dst + 32 * t);
#endif
}
// The "x4" images of the fp32 FiLM-SIREN training path (round 3; dw_core.h dw_task4g): element (row r, point p) of a tile at
// tile + ((r >> 2) * 32 + p) * 4 + (r & 3).  A lane's slice -- units row0 + 8q + 4*half + {0..3} of its point -- is ONE
// 16-byte word, a wave's slice two contiguous 512-byte runs: one global_store_dwordx4 per slice.
__device__ __forceinline__ float *at4(const RowImage &im, int r) {
    return im.tile + ((r >> 2) * 32 + (im.lane & 31)) * 4 + (r & 3);
}
__device__ __forceinline__ void store_slice4(const RowImage &im, int row0, int q, f32x4 v) {
    float *dst = im.tile + (((row0 + 8 * q) >> 2) + (im.lane >> 5)) * 128 + (im.lane & 31) * 4;
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(dst));       // streamed once, read back by a later kernel
}
__device__ __forceinline__ void store_block4(const RowImage &im, int row0, const f32x16 &v) {
#pragma unroll
    for (int q = 0; q < 4; ++q) store_slice4(im, row0, q, f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]});
}
__device__ __forceinline__ void store_block(const RowImage &im, int row0, const f32x16 &v) {
#pragma unroll
    for (int q = 0; q < 4; ++q) store_slice(im, row0, q, f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]});
}

// ReLU sign bits of slice q of block jb into the per-layer mask words (mlp_layout.h S_MASK)
__device__ __forceinline__ void mask_or(unsigned (&mk)[4], int jb, int q, const f32x4 &c) {
    // v >= +0 after the ReLU, so (v > 0) == (bits != 0) == min(bits, 1): two instructions per value, written
    // out because the compiler expands the C form into compare + select + or.  (Take the element into a
    // scalar first: __builtin_bit_cast applied to a vector element expression is miscompiled by this hipcc.)
    const int sh = 16 * (jb & 1) + 4 * q;
#ifdef NERFMI_EXP_MASK_CHAIN
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float v = c[t];
        unsigned one;
        asm("v_min_u32 %0, 1, %1" : "=v"(one) : "v"(v));
        asm("v_lshl_or_b32 %0, %1, %2, %0" : "+v"(mk[jb >> 1]) : "v"(one), "s"(sh + t));
    }
#else
    // the four bits are combined as a TREE and join the mask word once per slice: a vector instruction that consumes the
    // previous one's result issues after 8 cycles, an independent one after 4 (tools/ubench/mfma_valu.hip), so four
    // v_lshl_or_b32 chained through the mask word cost twice their issue slots
    unsigned b[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float v = c[t];
        asm("v_min_u32 %0, 1, %1" : "=v"(b[t]) : "v"(v));
    }
    const unsigned lo = (b[1] << 1) | b[0], hi = (b[3] << 1) | b[2];
    mk[jb >> 1] |= ((hi << 2) | lo) << sh;
#endif
}
__device__ __forceinline__ bool mask_bit(const unsigned (&mk)[4], int jb, int q, int t) {
    return (mk[jb >> 1] >> (16 * (jb & 1) + 4 * q + t)) & 1u;
}
// v where the bit is set, +0 where it is not: sign-extend the one-bit field and AND (two instructions)
__device__ __forceinline__ float mask_keep(const unsigned (&mk)[4], int jb, int q, int t, float v) {
    const int m = __builtin_amdgcn_sbfe((int)mk[jb >> 1], 16 * (jb & 1) + 4 * q + t, 1);
    return __int_as_float(__float_as_int(v) & m);
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_mask(const RowImage &im, int layer, unsigned (&mk)[4]) {
    *reinterpret_cast<u32x4 *>(im.tile + (S_MASK + 8 * layer) * 32 + 4 * im.lane) = u32x4{mk[0], mk[1], mk[2], mk[3]};
    mk[0] = mk[1] = mk[2] = mk[3] = 0u;
}
__device__ __forceinline__ void load_mask(const RowImage &im, int layer, unsigned (&mk)[4]) {
    const u32x4 v = *reinterpret_cast<const u32x4 *>(im.tile + (S_MASK + 8 * layer) * 32 + 4 * im.lane);
    mk[0] = v[0]; mk[1] = v[1]; mk[2] = v[2]; mk[3] = v[3];
}

// ReLU as a signed-integer max on the bit pattern (negative floats are negative integers, -0 -> +0): one
// v_max_i32 where fmaxf costs a canonicalisation plus a v_max_f32
__device__ __forceinline__ float relu1(float x) {
    return __builtin_bit_cast(float, max(__builtin_bit_cast(int, x), 0));
}
__device__ __forceinline__ f32x4 relu4(f32x4 c) {
#pragma unroll
    for (int r = 0; r < 4; ++r) c[r] = relu1(c[r]);
    return c;
}

// dot of the lane's share of NB blocks with a natural-order vector (w_half = w + 4*half),
// summed over both lane halves
template <int NB>
__device__ __forceinline__ float dot_blocks(const f32x16 *v, const float *__restrict__ w_half) {
    float s = 0.f;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 w = ldg4(w_half + 32 * b + 8 * q);
#pragma unroll
            for (int t = 0; t < 4; ++t) s = __builtin_fmaf(w[t], v[b][4 * q + t], s);
        }
    return s + __shfl_xor(s, 32, WAVE);
}

}  // namespace nerfmi
