// dW = dZ^T . X on the bf16 matrix cores (opt-in split-bf16 math), shared by the NeRF backward (mlp_bwd.hip) and the
// FiLM-SIREN backward (siren_bwd.hip).  A = rows of the backward workspace (AROWS rows per tile), B = rows of the saved
// activations (BROWS rows per tile), both tile-major [tile of 32 points][row][32] images (mlp_core.h RowImage).
#pragma once
#include "bf16x3_core.h"
#include "dw_core.h"

namespace nerfmi {

// ---------------------------------------------------------------------------
// 2b. the 256 x 256 tasks of the dW GEMM on the bf16 matrix cores (opt-in split-bf16 math).
//     dW = dZ^T . X needs BOTH operands split into their three bf16 terms.  Doing that per wave on the MFMA
//     fragments would cost ~4 vector instructions per MFMA; here the workgroup splits every value exactly ONCE while
//     staging it: a thread loads 4 consecutive points of a row (global -> registers, as in the fp32 kernel), splits
//     them in registers and writes the bf16 terms into an LDS image laid out for v_mfma_f32_32x32x16_bf16 fragments;
//     the MFMA loop then only reads ready fragments (one ds_read_b128 per term).  One 16-point k-step per barrier,
//     two LDS buffers of 48 KiB; the staging of k-step i+1 (8 float4 per thread: load, split, three 8-byte LDS writes)
//     is dealt out over the 16 accumulator units of k-step i, ~2 vector instructions per MFMA, hidden by the XDL pipe.
//     LDS image of a k-step: [term 3][block 16 (8 dZ + 8 X)][row 32][16 points] bf16, the two 16-byte halves of a row
//     swapped on rows where bit 2 and bit 3 differ: conflict-free ds_read_b128 whether the LDS serves 8 lanes x 32 banks or
//     16 lanes x 64 banks per pass (a swap on bit 2 alone left 31 % conflict cycles, SQ_LDS_BANK_CONFLICT).
// ---------------------------------------------------------------------------
constexpr int DWF_KSTEP_BYTES = 3 * 16 * 1024;      // largest task: 8 + 8 blocks

// JB x KB blocks per workgroup, JW x KW per wave, waves WJ x WK (JB = JW*WJ, KB = KW*WK, JB even, JB + KB even)
template <int JB, int KB, int JW, int KW, int WK, int AROWS, int BROWS>
__device__ __forceinline__ void dw_task_bf16x3(const DwTask &T, int chunk, const float *__restrict__ work,
                                               const float *__restrict__ saved, int64_t ld, float *__restrict__ partial,
                                               char *lds) {
    constexpr int NB = JB + KB;                         // blocks of the (dZ | X) operand
    constexpr int NS = NB / 2;                          // staging slots per thread (64 rows each); the first JB/2 are dZ
    constexpr int NU = JW * KW;                         // accumulator units per wave
    constexpr int KBYTES = 3 * NB * 1024;               // one k-step image
    static_assert(JB % 2 == 0 && NB % 2 == 0 && 2 * KBYTES <= 2 * DWF_KSTEP_BYTES, "task shape");
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, half = lane >> 5;
    const int wj = wid / WK, wk = wid % WK;
    const int64_t tiles = ld / 32;
    const int64_t t_lo = tiles * chunk / T.chunks, t_hi = tiles * (chunk + 1) / T.chunks;
    const int64_t k_lo = 2 * t_lo, k_hi = 2 * t_hi;     // k-steps of 16 points

    // staging: thread (srow = tid>>2, c = tid&3), slot j = rows 64j + srow of the (dZ | X) operand,
    // points 16s + 4c .. +3 of the k-step
    const int srow = tid >> 2, c = tid & 3;
    const float *abase = work + (int64_t)T.a_row0 * 32, *bbase = saved + (int64_t)T.b_row0 * 32;
    const unsigned voff = (unsigned)(srow * 32 + 4 * c);
    f32x4 st[NS];
    auto load_slot = [&](int j, int64_t ks) {
        const int64_t t = ks >> 1;
        const int s = (int)(ks & 1);
        const float *src = (j < JB / 2) ? abase + t * (int64_t)(AROWS * 32) + j * 2048 + 16 * s
                                        : bbase + t * (int64_t)(BROWS * 32) + (j - JB / 2) * 2048 + 16 * s;
        st[j] = ldg4(src + voff);
    };
    float bsum[JB / 2];                                 // row sums of dZ (bias gradient): rows 64j + srow, this thread's points
#pragma unroll
    for (int j = 0; j < JB / 2; ++j) bsum[j] = 0.f;
    // LDS byte offset of this thread's 8 bytes inside (term 0, block 2j + (srow>>5)): row r = srow & 31
    const int r = srow & 31;
    const unsigned woff = (unsigned)(((srow >> 5) * 1024) + r * 32 + (((c >> 1) ^ (((r >> 2) ^ (r >> 3)) & 1)) * 16) + (c & 1) * 8);
    auto split_write = [&](int j, char *buf, bool real) {      // real = false: the redundant restaging past the end
        const f32x4 v = st[j];
        if (j < JB / 2) bsum[j] += real ? (v[0] + v[1]) + (v[2] + v[3]) : 0.f;
        unsigned w0[3], w1[3];
        split_pair(v[0], v[1], w0);
        split_pair(v[2], v[3], w1);
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            *reinterpret_cast<u32x2 *>(buf + (t * NB + 2 * j) * 1024 + woff) = u32x2{w0[t], w1[t]};
        }
    };

    f32x16 acc[JW][KW];
#pragma unroll
    for (int a = 0; a < JW; ++a)
#pragma unroll
        for (int b = 0; b < KW; ++b)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;

    char *buf0 = lds, *buf1 = lds + KBYTES;
    if (k_lo < k_hi) {
#pragma unroll
        for (int j = 0; j < NS; ++j) load_slot(j, k_lo);
#pragma unroll
        for (int j = 0; j < NS; ++j) split_write(j, buf0, true);
        if (k_lo + 1 < k_hi) {
#pragma unroll
            for (int j = 0; j < NS; ++j) load_slot(j, k_lo + 1);
        }
    }
    __syncthreads();
    // fragment byte offset of this lane inside a (term, block) image
    const unsigned foff = (unsigned)((lane & 31) * 32 + ((half ^ ((((lane & 31) >> 2) ^ ((lane & 31) >> 3)) & 1)) * 16));
    auto kstep = [&](int64_t ks, const char *cur, char *nxt) __attribute__((always_inline)) {
        const int64_t k2 = (ks + 2 < k_hi) ? ks + 2 : k_hi - 1;     // past the end: restage the last one (branch-free)
        u32x4 a[JW][3], b[2][3];
#pragma unroll
        for (int x = 0; x < JW; ++x)
#pragma unroll
            for (int t = 0; t < 3; ++t) a[x][t] = *reinterpret_cast<const u32x4 *>(cur + (t * NB + wj * JW + x) * 1024 + foff);
#pragma unroll
        for (int t = 0; t < 3; ++t) b[0][t] = *reinterpret_cast<const u32x4 *>(cur + (t * NB + JB + wk * KW) * 1024 + foff);
#pragma unroll
        for (int y = 0; y < KW; ++y) {
            if (y + 1 < KW) {
#pragma unroll
                for (int t = 0; t < 3; ++t)
                    b[(y + 1) & 1][t] = *reinterpret_cast<const u32x4 *>(cur + (t * NB + JB + wk * KW + y + 1) * 1024 + foff);
            }
            const bf16x8 b1 = __builtin_bit_cast(bf16x8, b[y & 1][0]), b2 = __builtin_bit_cast(bf16x8, b[y & 1][1]),
                         b3 = __builtin_bit_cast(bf16x8, b[y & 1][2]);
#pragma unroll
            for (int x = 0; x < JW; ++x) {
                const bf16x8 a1 = __builtin_bit_cast(bf16x8, a[x][0]), a2 = __builtin_bit_cast(bf16x8, a[x][1]),
                             a3 = __builtin_bit_cast(bf16x8, a[x][2]);
                f32x16 cc = acc[x][y];
                cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, cc, 0, 0, 0);      // small terms first
                cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, cc, 0, 0, 0);
                cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, cc, 0, 0, 0);
                cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, cc, 0, 0, 0);
                cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, cc, 0, 0, 0);
                cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, cc, 0, 0, 0);
                acc[x][y] = cc;
                const int u = y * JW + x;                            // the staging slots are dealt out over the units
#pragma unroll
                for (int i = 0; i < NS; ++i)
                    if ((i * NU) / NS == u) {
                        split_write(i, nxt, ks + 1 < k_hi);
                        load_slot(i, k2);
                    }
            }
        }
        __syncthreads();
    };
    for (int64_t ks = k_lo; ks < k_hi; ks += 2) {
        kstep(ks, buf0, buf1);
        if (ks + 1 < k_hi) kstep(ks + 1, buf1, buf0);
    }
    // partial slab [chunk][32 JB][32 KB] then bias slab [chunk][32 JB] (same layout as dw_task<., ., ., .>)
    float *slab = partial + T.part_off + (int64_t)chunk * (JB * 32 * (KB * 32 + 1));
#pragma unroll
    for (int x = 0; x < JW; ++x)
#pragma unroll
        for (int y = 0; y < KW; ++y)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int j = 32 * (wj * JW + x) + 8 * (q >> 2) + 4 * half + (q & 3);
                const int k = 32 * (wk * KW + y) + (lane & 31);
                slab[j * (KB * 32) + k] = acc[x][y][q];
            }
    // bias: the four threads c = 0..3 of a row are neighbours in the wave
#pragma unroll
    for (int j = 0; j < JB / 2; ++j) {
        float sum = bsum[j];
        sum += __shfl_xor(sum, 1, WAVE);
        sum += __shfl_xor(sum, 2, WAVE);
        if (c == 0) slab[JB * 32 * KB * 32 + 64 * j + srow] = sum;
    }
}


}  // namespace nerfmi
