// Register-resident dense layer on the gfx950 matrix cores, shared by the NeRF
// forward (mlp.hip) and the backward dX chain (mlp_bwd.hip).  See mlp_layout.h
// for the operand maps.
#pragma once
#include "common.h"
#include "mlp_layout.h"

namespace nerfmi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int PF = 6;  // weight-fragment groups (1 KiB each) in flight per wave

__device__ __forceinline__ f32x4 ldg4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }

// out[jb] = epi(jb, bias + W[jb-block rows] . [in0 ; in1])      (JB x (KB0+KB1) blocks)
//   wp   : this layer's packed image + lane*4 (mlp_layout.h), streamed through a PF-deep register ring
//   bias : natural order + 4*half, or nullptr for a zero start (backward)
//   epi  : per-block epilogue (ReLU / mask / stores), called as soon as a block is complete so that its
//          memory traffic is spread across the layer instead of piling up at its end
template <int KB0, int KB1, int JB, class Epi>
__device__ __forceinline__ void layer_mfma(const float *__restrict__ wp, const float *__restrict__ bias,
                                           const f32x16 *in0, const f32x16 *in1, f32x16 *out, Epi epi) {
    constexpr int KBT = KB0 + KB1;
    constexpr int G = JB * KBT * 4;
    f32x4 ring[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) ring[i] = ldg4(wp + i * 256);
#pragma unroll
    for (int jb = 0; jb < JB; ++jb) {
        f32x16 c;
        if (bias) {
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
                const f32x4 a = ring[g % PF];
                if (g + PF < G) ring[g % PF] = ldg4(wp + (g + PF) * 256);
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], B[4 * q + 0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], B[4 * q + 1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], B[4 * q + 2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], B[4 * q + 3], c, 0, 0, 0);
            }
        }
        out[jb] = epi(jb, c);
    }
}

// [row][point] activation images (mlp_layout.h S_* / W_*): element (row, p) at base[row*ld + p].
// A lane addresses them as  rowbase(row_uniform) + lane_off  with lane_off = 4*half*ld + p  (32-bit),
// so the row part stays scalar and stores/loads use the saddr + voffset form.
struct RowImage {
    float *base;
    int64_t ld;
    unsigned lane_off;   // 4*half*ld + point (the UNclamped point: columns [n_points, ld) are written as 0)
    bool ok;             // point < n_points
    __device__ __forceinline__ float *row(int r) const { return base + (int64_t)r * ld; }
};

// store one 32-unit block: unit row0 + 8*(r>>2) + 4*half + (r&3)
__device__ __forceinline__ void store_block(const RowImage &im, int row0, const f32x16 &v) {
#pragma unroll
    for (int r = 0; r < 16; ++r) im.row(row0 + 8 * (r >> 2) + (r & 3))[im.lane_off] = im.ok ? v[r] : 0.f;
}

__device__ __forceinline__ f32x16 load_block(const RowImage &im, int row0) {
    f32x16 v;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = im.row(row0 + 8 * (r >> 2) + (r & 3))[im.lane_off];
    return v;
}

__device__ __forceinline__ f32x16 relu16(f32x16 c) {
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = fmaxf(c[r], 0.f);
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
