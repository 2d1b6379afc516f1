// dW = dZ^T . X over points (a GEMM whose contraction runs over POINTS) and its deterministic slab reduction, shared
// by the NeRF backward (mlp_bwd.hip) and the FiLM-SIREN backward (siren_bwd.hip).  Both operands are tile-major images
// (mlp_core.h RowImage): A = rows of the backward workspace (AROWS rows per tile), B = rows of the saved activations
// (BROWS rows per tile).
#pragma once
#include "mlp_core.h"

namespace nerfmi {

struct DwTask {
    int kind;       // template instance 0..5
    int a_row0;     // first dZ row in the workspace
    int a_valid;    // real rows (others read as 0)
    int b_row0;     // first X row in the saved image
    int b_valid;
    int param;      // weight tensor index
    int out_col0;   // first column of the weight this task covers
    int in_f;       // row stride of the weight tensor
    int bias_param; // bias tensor index or -1
    int chunks;     // split of the point range
    int wg0;        // first workgroup of this task
    int part_off;   // float offset of this task's slabs in the partial buffer
    int JB, KB;     // block counts (rows/cols of the slab = 32*JB x 32*KB)
};
constexpr int MAX_TASKS = 16;
struct DwPlan {
    DwTask t[MAX_TASKS];
    int n_tasks;
    int n_wg;
};

constexpr int LROW = 36;   // LDS row pitch in floats: 32 points + 4 pad (conflict-free ds_read_b128)

template <int JW, int KW, int WJ, int WK, int AROWS, int BROWS>
__device__ __forceinline__ void dw_task(const DwTask &T, int chunk, const float *__restrict__ work,
                                        const float *__restrict__ saved, int64_t ld, float *__restrict__ partial,
                                        float *lds) {
    constexpr int JB = JW * WJ, KB = KW * WK;
    constexpr int ROWS = (JB + KB) * 32;
    constexpr int NLD = JB + KB;              // staging slots per thread: slot i = rows 32i..32i+31 (A blocks, then B)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, half = lane >> 5;
    const int wj = wid / WK, wk = wid % WK;
    const int64_t tiles = ld / 32;
    const int64_t t_lo = tiles * chunk / T.chunks, t_hi = tiles * (chunk + 1) / T.chunks;

    // Inside a tile the rows of an image are contiguous (128 B each): thread (srow, c4) of slot i fetches float4
    // c4 of row 32i + srow, so one wave-load is 1 KiB contiguous.  The per-thread part of every address (global
    // and LDS) is the SAME for all slots and all tiles; the slot and tile parts are wave-uniform and live in
    // scalar registers / immediate offsets.  That matters because fp32 MFMAs do not overlap with the wave's own
    // vector instructions (tools/ubench/mfma_valu.hip): address arithmetic in the loop is paid in matrix-pipe time.
    // Rows past the real operand (the 3 + 1 rows of the heads' dZ) are read as whatever follows them in the image
    // -- at worst the dump tile behind the last real one (mlp_core.h RowImage): an MFMA output row depends on its own
    // A row only, and the reduce kernel never reads the slab rows >= a_valid.  The padded B rows (row 63 of the
    // xyz embedding, 27..31 of the direction embedding) are stored as zeros by the forward.
    const int srow = tid >> 3, c4 = tid & 7;
    const unsigned voff = (unsigned)(srow * 32 + 4 * c4);             // floats, global
    const unsigned loff = (unsigned)(srow * LROW + 4 * c4);           // floats, LDS
    const float *abase = work + (int64_t)T.a_row0 * 32;
    const float *bbase = saved + (int64_t)T.b_row0 * 32;

    f32x16 acc[JW][KW];
#pragma unroll
    for (int a = 0; a < JW; ++a)
#pragma unroll
        for (int b = 0; b < KW; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float bsum[JW];
#pragma unroll
    for (int a = 0; a < JW; ++a) bsum[a] = 0.f;

    f32x4 stage[NLD];
    auto load_slot = [&](int i, int64_t t) {
        const float *src = (i < JB) ? abase + t * (int64_t)(AROWS * 32) + i * 1024
                                    : bbase + t * (int64_t)(BROWS * 32) + (i - JB) * 1024;     // wave-uniform
#ifdef NERFMI_EXP_DW_TLOAD
        stage[i] = ldg4(src + voff);
#else
        // streamed exactly once by exactly one workgroup: non-temporal, so the 2.7 GB of dZ / X tiles do not sweep L2
        stage[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(src + voff));
#endif
    };
    auto write_slot = [&](int i, float *buf) {
        *reinterpret_cast<f32x4 *>(buf + i * (32 * LROW) + loff) = stage[i];
    };
    // Double-buffered LDS, ONE barrier per tile: at the top of iteration t the registers hold tile t+1
    // (loaded during iteration t-1); it is written into the other buffer (last read in iteration t-1, which
    // every wave left through the barrier), tile t+2's loads are issued, then tile t is consumed.
    float *buf0 = lds, *buf1 = lds + ROWS * LROW;
    if (t_lo < t_hi) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) load_slot(i, t_lo);
#pragma unroll
        for (int i = 0; i < NLD; ++i) write_slot(i, buf0);
        if (t_lo + 1 < t_hi) {
#pragma unroll
            for (int i = 0; i < NLD; ++i) load_slot(i, t_lo + 1);
        }
    }
    __syncthreads();
    // one tile: consume `cur`, stage tile t+1 into `nxt` and reload the registers with tile t+2 -- the staging of
    // slot i sits after MFMA group i*(groups/NLD), spread over the tile's MFMA stream instead of in front of it.
    // Past the end the (clamped) tile is staged redundantly, which keeps the body branch-free.
    auto tile = [&](int64_t t, const float *cur, float *nxt) __attribute__((always_inline)) {
        const int64_t t2 = (t + 2 < t_hi) ? t + 2 : t_hi - 1;
        const float *arow = cur + (32 * (wj * JW) + (lane & 31)) * LROW + 4 * half;
        const float *brow = cur + (32 * (JB + wk * KW) + (lane & 31)) * LROW + 4 * half;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 a[JW], b[KW];
#pragma unroll
            for (int x = 0; x < JW; ++x) a[x] = *reinterpret_cast<const f32x4 *>(arow + x * (32 * LROW) + 8 * q);
#pragma unroll
            for (int x = 0; x < KW; ++x) b[x] = *reinterpret_cast<const f32x4 *>(brow + x * (32 * LROW) + 8 * q);
#pragma unroll
            for (int x = 0; x < JW; ++x) bsum[x] += (a[x][0] + a[x][1]) + (a[x][2] + a[x][3]);
#pragma unroll
            for (int x = 0; x < JW; ++x)
#pragma unroll
                for (int y = 0; y < KW; ++y) {
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[x][s], b[y][s], acc[x][y], 0, 0, 0);
                    constexpr int NG = 4 * JW * KW;                       // MFMA groups per tile
                    const int gidx = (q * JW + x) * KW + y;
#pragma unroll
                    for (int i = 0; i < NLD; ++i)
                        if (gidx == (i * NG) / NLD) {
                            write_slot(i, nxt);
                            load_slot(i, t2);
                        }
                }
        }
        __syncthreads();
    };
    // two tiles per trip, so that which buffer is read and which is written is static inside the body
    for (int64_t t = t_lo; t < t_hi; t += 2) {
        tile(t, buf0, buf1);
        if (t + 1 < t_hi) tile(t + 1, buf1, buf0);
    }
    // partial slab [chunk][32*JB][32*KB] then bias slab [chunk][32*JB]
    float *slab = partial + T.part_off + (int64_t)chunk * (JB * 32 * (KB * 32 + 1));
#pragma unroll
    for (int x = 0; x < JW; ++x)
#pragma unroll
        for (int y = 0; y < KW; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = 32 * (wj * JW + x) + 8 * (r >> 2) + 4 * half + (r & 3);
                const int k = 32 * (wk * KW + y) + (lane & 31);
                slab[j * (KB * 32) + k] = acc[x][y][r];
            }
    if (wk == 0) {
#pragma unroll
        for (int x = 0; x < JW; ++x) {
            const float s = bsum[x] + __shfl_xor(bsum[x], 32, WAVE);
            if (half == 0) slab[JB * 32 * KB * 32 + 32 * (wj * JW + x) + lane] = s;
        }
    }
}

struct GradPtrs {
    float *p[N_PARAMS];            // NeRF: 24 tensors; the FiLM-SIREN field uses the first 22
};

// slab reduction: blockIdx.y = task.  (A template only so that every translation unit including this header may
// instantiate it.)
template <int TAG>
__global__ void dw_reduce_kernel(DwPlan plan, const float *__restrict__ partial, GradPtrs G) {
    const DwTask T = plan.t[blockIdx.y];
    const int rows = T.JB * 32, cols = T.KB * 32;
    const int slab = rows * (cols + 1);
    const int out_f_valid = T.a_valid, in_valid = T.b_valid;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < slab; idx += gridDim.x * blockDim.x) {
        const float *src = partial + T.part_off + idx;
        // fixed summation order (bit-reproducible); unrolled so the slab loads of a thread are all in flight at once
        // instead of one HBM round trip per chunk
        float s = 0.f;
        int c = 0;
        for (; c + 8 <= T.chunks; c += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(c + u) * slab];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; c < T.chunks; ++c) s += src[(int64_t)c * slab];
        if (idx < rows * cols) {
            const int j = idx / cols, k = idx % cols;
            if (j < out_f_valid && k < in_valid) G.p[T.param][j * T.in_f + T.out_col0 + k] = s;
        } else if (T.bias_param >= 0) {
            const int j = idx - rows * cols;
            if (j < out_f_valid) G.p[T.bias_param][j] = s;
        }
    }
}


// chunk split + slab offsets of a plan whose tasks are filled in: chunks[kind] workgroups per task, sized so that the
// grid is exactly 256 equal workgroups (one per CU)
static inline void dw_finish_plan(DwPlan &P, const int *kind_jb, const int *kind_kb, const int *chunks_by_kind, int64_t ld) {
    const int64_t tiles = ld / 32;
    int wg = 0, off = 0;
    for (int i = 0; i < P.n_tasks; ++i) {
        DwTask &t = P.t[i];
        t.JB = kind_jb[t.kind]; t.KB = kind_kb[t.kind];
        int c = chunks_by_kind[t.kind];
        if (c > tiles) c = (int)(tiles < 1 ? 1 : tiles);
        t.chunks = c; t.wg0 = wg; t.part_off = off;
        wg += c;
        off += c * (t.JB * 32 * (t.KB * 32 + 1));
    }
    P.n_wg = wg;
}

static inline size_t dw_partial_floats(const DwPlan &P) {
    const DwTask &t = P.t[P.n_tasks - 1];
    return (size_t)t.part_off + (size_t)t.chunks * (t.JB * 32 * (t.KB * 32 + 1));
}

}  // namespace nerfmi
