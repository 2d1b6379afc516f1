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
    int chunks;     // split of the point range = number of partial slabs
    int wp;         // dw_task4g: slabs per workgroup (the tile's points split over two wave pairs); 1 otherwise
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

// ---------------------------------------------------------------------------------------------------------------------
// dW on the "x4" images of the FiLM-SIREN training path (round 3), staged by LDS-DMA.
//
// x4 layout: element (row r, point p) of a 32-point tile lives at  tile + ((r >> 2) * 32 + p) * 4 + (r & 3):  the four rows of a
// ROW GROUP are interleaved per point.  In the MLP kernels a lane owns four consecutive units of one point (registers 4q..4q+3
// of an accumulator block), so a slice of a saved / dZ image is ONE 16-byte store (a wave: 1 KiB contiguous) where the
// [row][32 points] layout needs four 4-byte stores -- a vector-memory instruction costs the same issue slot and TA time
// whatever its width (tools/ubench/pk_valu.hip).
//
// The GEMM needs no transposing staging: the rows of an MFMA operand may be ANY 32 rows.  A lane reads one 16-byte LDS word
// = (row group g = its lane & 31, point 2s + (lane >> 5)) = the four units 4g..4g+3 of one point and feeds unit i of it to MFMA
// i: operand A_i is "unit i of 32 consecutive row groups" (rows 4g + i), B_j likewise, and the 4 x 4 products A_i (x) B_j
// fill a 128 x 128 tile of dW whose rows / columns are merely interleaved -- undone for free when the slab is written.  Per
// 32-point tile a wave issues 2 LDS reads per 16 MFMAs (the [row][point] kernel above: 10 per 16).
//
// Staging: `global_load_lds_dwordx4` (no staging registers, no ds_write): with 256 accumulator registers a register-staged
// version of this loop made hipcc spill 113-177 VGPRs into the MFMA loop (tools/experiments/README.md).  An LDS-DMA
// instruction writes 64 lanes x 16 B = 1 KiB CONTIGUOUSLY; which 16 bytes of global memory a lane fetches is free.  A PIECE
// = two row groups (1 KiB of the image) is stored with its two row groups interleaved per point -- LDS slot 2 p + (g & 1)
// holds (row group g, point p): DMA lane L fetches image word (L & 1) * 32 + (L >> 1) -- and pieces are 1056 bytes apart.
// The fragment address of a lane is then AFFINE in the k-step (base + 64 s: an immediate offset), and the 16 lanes of a
// ds_read_b128 pass (consecutive row groups, one point) start 16 B apart inside a piece and 1056 B = 8 banks + 16 rows
// apart across pieces: 16 different 4-bank groups.  All of the next tile's pieces are issued in the first half of the
// current tile's steps, so that the vmcnt(0) the barrier at the tile's end needs finds them landed.
// Forms (IA / JB4 = 4: all four units of 32 row groups per wave set, WA / WB sets of 128 rows; = 1: unit 0 only -- a narrow
// operand whose few rows sit in unit 0 of their own row groups; WP = point ranges of a tile handled by different waves, each
// writing its own slab: the reduction sums slabs anyway; NA_P / NB_P = 1 KiB pieces (8 rows) that really exist and are staged,
// the lanes past them multiply whatever the LDS holds into rows / columns the reduction drops):
//   in use: 256 x 256 <4,4,2,2,1,32,32> (both fields), NeRF 128 x 256 <4,4,1,2,2,16,32>; the tasks with a small operand run
//   the 16-lane form below (dw_task4g16), where their operand sets are full
// ---------------------------------------------------------------------------------------------------------------------
template <int IA, int JB4, int WA, int WB, int WP, int NA_P, int NB_P, int AROWS, int BROWS>
__device__ __forceinline__ void dw_task4g(const DwTask &T, int chunk, const float *__restrict__ work,
                                          const float *__restrict__ saved, int64_t ld, float *__restrict__ partial,
                                          float *lds) {
    static_assert(WA * WB * WP == 4, "four waves per workgroup");
    constexpr int PIECE = 1056;                           // LDS bytes between pieces (1 KiB + 32: see above)
    constexpr int A_AL = (IA == 4) ? WA * 16 : 16;        // a wave's 32 lanes read 32 row groups: keep the reads inside LDS
    constexpr int B_AL = (JB4 == 4) ? WB * 16 : 16;
    static_assert(NA_P <= A_AL && NB_P <= B_AL, "staged pieces fit their region");
    constexpr int BUF_BYTES = (A_AL + B_AL) * PIECE;
    constexpr bool A_FIRST = NA_P >= NB_P;                // piece order of a tile: the wide operand first
    constexpr int NP = NA_P + NB_P, NJ = (NP + 3) / 4;    // pieces per tile, per wave
    constexpr int STEPS = 16 / WP;                        // k-steps (point pairs) of a tile that this wave multiplies
    constexpr int ISSUE_STEPS = STEPS / 2;                // the next tile's pieces go out in the first half of the steps
    constexpr int PER_STEP = (NJ + ISSUE_STEPS - 1) / ISSUE_STEPS;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);       // (scalar: piece addresses stay in SGPRs)
    const int wa = wid % WA, wb = (wid / WA) % WB, wp = wid / (WA * WB);
    const int wg_chunks = T.chunks / WP;                  // workgroups of this task (T.chunks counts slabs)
    const int64_t tiles = ld / 32;
    const int64_t t_lo = tiles * chunk / wg_chunks, t_hi = tiles * (chunk + 1) / wg_chunks;
    const char *abase = reinterpret_cast<const char *>(work + (int64_t)T.a_row0 * 32);
    const char *bbase = reinterpret_cast<const char *>(saved + (int64_t)T.b_row0 * 32);
    char *lbase = reinterpret_cast<char *>(lds);
    const unsigned poff = (unsigned)((lane & 1) * 512 + (lane >> 1) * 16);   // this lane's word of a piece
    // piece 4 j + wave of a tile (rounds j < NF / 4: the wide operand, 4 | its piece count; then the narrow operand's pieces)
    // from the tile's operand bases `sa`, `sb` into `buf`
    constexpr int NF = A_FIRST ? NA_P : NB_P;
    static_assert(NF % 4 == 0, "the wide operand's pieces fill whole rounds");
    auto issue = [&](int j, const char *sa, const char *sb, char *buf) __attribute__((always_inline)) {
        const bool first = j < NF / 4;                    // compile-time after unrolling
        const int p = first ? 4 * j + wid : 4 * (j - NF / 4) + wid;
        if ((NP - NF) % 4 != 0 && !first && p >= NP - NF) return;   // (wave-uniform; only the narrow operand has a ragged round)
        const bool is_a = first == A_FIRST;
        const char *src = (is_a ? sa : sb) + p * 1024;
        char *dst = buf + (is_a ? 0 : A_AL * PIECE) + p * PIECE;
        __builtin_amdgcn_global_load_lds((gbl_void_t *)(src + poff), (lds_void_t *)dst, 16, 0, 0);
    };

    f32x16 acc[IA][JB4];
#pragma unroll
    for (int i = 0; i < IA; ++i)
#pragma unroll
        for (int j = 0; j < JB4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x2 bsum2[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};   // bias sums: sum over points of the A rows (units 0..3 of the lane's row group)

    char *buf0 = lbase;
    if (t_lo < t_hi) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) issue(j, abase + t_lo * (int64_t)(AROWS * 128), bbase + t_lo * (int64_t)(BROWS * 128), buf0);
    }
    __syncthreads();                                      // (hipcc drains the LDS-DMAs in flight with vmcnt(0) here)
    // fragment bases: (row group g, point 2 s + half) at piece (g >> 1), slot 2 (2 s + half) + (g & 1)
    const int rga = ((IA == 4) ? wa * 32 : 0) + (lane & 31), rgb = ((JB4 == 4) ? wb * 32 : 0) + (lane & 31);
    const unsigned a_off = (unsigned)((rga >> 1) * PIECE + (rga & 1) * 16 + half * 32 + STEPS * wp * 64);
    const unsigned b_off = (unsigned)(A_AL * PIECE + (rgb >> 1) * PIECE + (rgb & 1) * 16 + half * 32 + STEPS * wp * 64);
    // ONE tile per loop trip, the two buffers swapped by offset arithmetic: with all 256 accumulator registers in use, a loop
    // body of two unrolled tiles (the second one conditional) makes hipcc spill an accumulator block around the join
    int cur = 0;                                          // byte offset of the buffer being consumed
    for (int64_t t = t_lo; t < t_hi; ++t) {
        const int64_t t1 = (t + 1 < t_hi) ? t + 1 : t_hi - 1;   // past the end: restage the last tile (branch-free)
        const char *sa = abase + t1 * (int64_t)(AROWS * 128), *sb = bbase + t1 * (int64_t)(BROWS * 128);
        const char *ca = lbase + cur + a_off, *cb = lbase + cur + b_off;
        char *nxt = lbase + (BUF_BYTES - cur);
        // fragments are read ONE step ahead into a second register set, in the MIDDLE of the step's MFMAs (fenced, so that the
        // reads stay there): a read issued right in front of its first MFMA exposes its whole latency, and so does one issued
        // right behind the last MFMA of the step before (hipcc waits with lgkmcnt(0), which covers the newest read too)
        f32x4 a = *reinterpret_cast<const f32x4 *>(ca), b = *reinterpret_cast<const f32x4 *>(cb);
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            // (volatile asm: written in C, the compiler sinks the sixteen steps' adds to the end of the tile and keeps all
            // sixteen A fragments alive until then)
            if (IA == 4) {
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(bsum2[0]) : "v"(f32x2{a[0], a[1]}));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(bsum2[1]) : "v"(f32x2{a[2], a[3]}));
            } else {
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(bsum2[0][0]) : "v"(a[0]));
            }
#pragma unroll
            for (int m = 0; m < IA * JB4 / 2; ++m)
                acc[m / JB4][m % JB4] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m / JB4], b[m % JB4], acc[m / JB4][m % JB4], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 an = a, bn = b;
            if (s + 1 < STEPS) {
                an = *reinterpret_cast<const f32x4 *>(ca + 64 * (s + 1));
                bn = *reinterpret_cast<const f32x4 *>(cb + 64 * (s + 1));
            }
            if (s < ISSUE_STEPS) {
#pragma unroll
                for (int k = 0; k < PER_STEP; ++k)
                    if (s * PER_STEP + k < NJ) issue(s * PER_STEP + k, sa, sb, nxt);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = IA * JB4 / 2; m < IA * JB4; ++m)
                acc[m / JB4][m % JB4] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m / JB4], b[m % JB4], acc[m / JB4][m % JB4], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            a = an; b = bn;
        }
        __syncthreads();
        cur = BUF_BYTES - cur;
    }
    // slab [32*JB rows][32*KB cols] then the bias slab [32*JB]; slab index = chunk * WP + wp
    constexpr int ROWS = (IA == 4) ? WA * 128 : 32, COLS = (JB4 == 4) ? WB * 128 : 32;
    float *slab = partial + T.part_off + (int64_t)(chunk * WP + wp) * (ROWS * (COLS + 1));
#pragma unroll
    for (int i = 0; i < IA; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = 8 * (r >> 2) + 4 * half + (r & 3);               // accumulator row = A row group of this wave's set
            const int row = (IA == 4) ? 4 * (wa * 32 + m) + i : m;
            if (JB4 == 4) {
                const int col = 4 * (wb * 32 + (lane & 31));
                *reinterpret_cast<f32x4 *>(slab + row * COLS + col) = f32x4{acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
            } else {
                slab[row * COLS + (lane & 31)] = acc[i][0][r];
            }
        }
    if (wb == 0) {
#pragma unroll
        for (int i = 0; i < IA; ++i) {
            const float mine = bsum2[i >> 1][i & 1];
            const float sum = mine + __shfl_xor(mine, 32, WAVE);           // the two point parities of the row group
            const int row = (IA == 4) ? 4 * (wa * 32 + (lane & 31)) + i : (lane & 31);
            if (half == 0) slab[ROWS * COLS + row] = sum;
        }
    }
}

// The same GEMM on v_mfma_f32_16x16x4_f32 for tasks with a SMALL operand (the NeRF embeddings: 63 / 27 columns of dW; the
// K = 3 inputs and the heads of both fields): in the 32-lane form above an operand of <= 64 rows fills at most half of the 32
// lanes and every MFMA wastes that share of its work.  Here an operand set is 16 row groups: lane l reads the 16-byte word
// (row group l & 15, point 4 s + (l >> 4)), unit i of it is operand A_i / B_j of a 16 x 16 x 4 MFMA (same FLOP per cycle as
// 32 x 32 x 2), a step covers four points, a tile eight steps.  IA / JB4 = 4: all four units (64 rows per set, WA / WB sets);
// = 1: unit 0 only (a narrow operand: its rows in unit 0 of their own row groups).  Same staging, same slab format (slabs of
// narrow operands are padded to 32 rows / columns).
//   NeRF    256 x 63 <4,4,4,1,1,32,8,3>   128 x 27 <4,4,2,1,2,16,4,5>   rgb 3 x 128 <1,4,1,2,2,2,16,5>   sigma 1 x 256 <1,4,1,4,1,2,32,3>
//   SIREN   256 x 3 <4,1,4,1,1,32,2,3>    heads 3 x 256 / 1 x 256 <1,4,1,4,1,2,32,3>
template <int IA, int JB4, int WA, int WB, int WP, int NA_P, int NB_P, int NBUF, int AROWS, int BROWS>
__device__ __forceinline__ void dw_task4g16(const DwTask &T, int chunk, const float *__restrict__ work,
                                            const float *__restrict__ saved, int64_t ld, float *__restrict__ partial,
                                            float *lds) {
    static_assert(WA * WB * WP == 4, "four waves per workgroup");
    constexpr int PIECE = 1056;
    constexpr int A_AL = (IA == 4) ? WA * 8 : 8, B_AL = (JB4 == 4) ? WB * 8 : 8;   // pieces of LDS per operand: 16 row groups per set
    static_assert(NA_P <= A_AL && NB_P <= B_AL, "staged pieces fit their region");
    constexpr int BUF_BYTES = (A_AL + B_AL) * PIECE;
    // These tasks have 32-128 MFMAs (1-4 k cycles) per tile: with the next tile's pieces issued only ONE tile ahead, a tile took
    // the latency of its loads (2.8 k cycles per tile measured, whatever the MFMA count).  So NBUF LDS buffers, the pieces of
    // tile t + NBUF - 1 issued during tile t, and at the top of a tile a COUNTED wait -- the newest (NBUF - 2) tiles' pieces may
    // stay in flight -- in front of a raw s_barrier (__syncthreads() would drain every LDS-DMA in flight with vmcnt(0)).
    static_assert(NBUF >= 2 && NBUF * BUF_BYTES <= 147456, "the ring fits the workgroup's dynamic LDS");
    constexpr int AHEAD = NBUF - 1;
    constexpr bool A_FIRST = NA_P >= NB_P;                // piece order of a tile: the wide operand first
    constexpr int NF = A_FIRST ? NA_P : NB_P;
    static_assert(NF % 4 == 0, "the wide operand's pieces fill whole rounds");
    constexpr int NP = NA_P + NB_P, NJ = (NP + 3) / 4;    // pieces per tile, per wave
    constexpr int STEPS = 8 / WP;
    constexpr int ISSUE_STEPS = STEPS / 2;
    constexpr int PER_STEP = (NJ + ISSUE_STEPS - 1) / ISSUE_STEPS;
    constexpr int NM = IA * JB4;                          // MFMAs per step
    const int tid = threadIdx.x, lane = tid & 63, kq = lane >> 4;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = wid % WA, wb = (wid / WA) % WB, wp = wid / (WA * WB);
    const int wg_chunks = T.chunks / WP;
    const int64_t tiles = ld / 32;
    const int64_t t_lo = tiles * chunk / wg_chunks, t_hi = tiles * (chunk + 1) / wg_chunks;
    const char *abase = reinterpret_cast<const char *>(work + (int64_t)T.a_row0 * 32);
    const char *bbase = reinterpret_cast<const char *>(saved + (int64_t)T.b_row0 * 32);
    char *lbase = reinterpret_cast<char *>(lds);
    const unsigned poff = (unsigned)((lane & 1) * 512 + (lane >> 1) * 16);
    auto issue = [&](int j, const char *sa, const char *sb, char *buf) __attribute__((always_inline)) {
        const bool first = j < NF / 4;                    // compile-time after unrolling
        int p = first ? 4 * j + wid : 4 * (j - NF / 4) + wid;
        // a ragged last round re-fetches the operand's last piece (same bytes to the same place): every wave issues NJ pieces
        // per tile, which is what the counted wait below counts
        if ((NP - NF) % 4 != 0 && !first && p >= NP - NF) p = NP - NF - 1;
        const bool is_a = first == A_FIRST;
        const char *src = (is_a ? sa : sb) + p * 1024;
        char *dst = buf + (is_a ? 0 : A_AL * PIECE) + p * PIECE;
        __builtin_amdgcn_global_load_lds((gbl_void_t *)(src + poff), (lds_void_t *)dst, 16, 0, 0);
    };
    f32x4 acc[IA][JB4];
#pragma unroll
    for (int i = 0; i < IA; ++i)
#pragma unroll
        for (int j = 0; j < JB4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x2 bsum2[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
    static_assert((AHEAD - 1) * NJ <= 63, "vmcnt is a 6-bit counter");
    if (t_lo < t_hi) {
#pragma unroll
        for (int d = 0; d < AHEAD; ++d) {
            const int64_t td = (t_lo + d < t_hi) ? t_lo + d : t_hi - 1;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                issue(j, abase + td * (int64_t)(AROWS * 128), bbase + td * (int64_t)(BROWS * 128), lbase + d * BUF_BYTES);
        }
    }
    // (row group g, point 4 s + kq) at piece (g >> 1), slot 2 (4 s + kq) + (g & 1): 128 bytes per step
    const int rga = ((IA == 4) ? wa * 16 : 0) + (lane & 15), rgb = ((JB4 == 4) ? wb * 16 : 0) + (lane & 15);
    const unsigned a_off = (unsigned)((rga >> 1) * PIECE + (rga & 1) * 16 + kq * 32 + STEPS * wp * 128);
    const unsigned b_off = (unsigned)(A_AL * PIECE + (rgb >> 1) * PIECE + (rgb & 1) * 16 + kq * 32 + STEPS * wp * 128);
    int cur = 0;                                          // byte offset of the buffer being consumed
    for (int64_t t = t_lo; t < t_hi; ++t) {
        // tile t's pieces (issued AHEAD tiles ago, by every wave) have landed; everybody is done with the buffer of tile t - 1,
        // which is the one tile t + AHEAD goes to
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * NJ) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int64_t t1 = (t + AHEAD < t_hi) ? t + AHEAD : t_hi - 1;   // past the end: restage the last tile (branch-free)
        const char *sa = abase + t1 * (int64_t)(AROWS * 128), *sb = bbase + t1 * (int64_t)(BROWS * 128);
        const char *ca = lbase + cur + a_off, *cb = lbase + cur + b_off;
        char *nxt = lbase + (cur == 0 ? (NBUF - 1) * BUF_BYTES : cur - BUF_BYTES);
        f32x4 a = *reinterpret_cast<const f32x4 *>(ca), b = *reinterpret_cast<const f32x4 *>(cb);
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            if (IA == 4) {
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(bsum2[0]) : "v"(f32x2{a[0], a[1]}));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(bsum2[1]) : "v"(f32x2{a[2], a[3]}));
            } else {
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(bsum2[0][0]) : "v"(a[0]));
            }
#pragma unroll
            for (int m = 0; m < NM / 2; ++m)
                acc[m / JB4][m % JB4] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m / JB4], b[m % JB4], acc[m / JB4][m % JB4], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 an = a, bn = b;
            if (s + 1 < STEPS) {
                an = *reinterpret_cast<const f32x4 *>(ca + 128 * (s + 1));
                bn = *reinterpret_cast<const f32x4 *>(cb + 128 * (s + 1));
            }
            if (s < ISSUE_STEPS) {
#pragma unroll
                for (int k = 0; k < PER_STEP; ++k)
                    if (s * PER_STEP + k < NJ) issue(s * PER_STEP + k, sa, sb, nxt);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = NM / 2; m < NM; ++m)
                acc[m / JB4][m % JB4] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m / JB4], b[m % JB4], acc[m / JB4][m % JB4], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            a = an; b = bn;
        }
        cur = (cur == (NBUF - 1) * BUF_BYTES) ? 0 : cur + BUF_BYTES;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the restaged pieces past the end: nothing in flight when the wave ends)
    // accumulator (i, j) register r of lane l = dW row 4 (A row group 4 (l >> 4) + r of the set) + i, column 4 (l & 15) + j
    constexpr int ROWS = (IA == 4) ? WA * 64 : 32, COLS = (JB4 == 4) ? WB * 64 : 32;   // (narrow: 16 real, padded to a block)
    float *slab = partial + T.part_off + (int64_t)(chunk * WP + wp) * (ROWS * (COLS + 1));
#pragma unroll
    for (int i = 0; i < IA; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = 4 * kq + r;
            const int row = (IA == 4) ? 4 * (wa * 16 + m) + i : m;
            if (JB4 == 4) {
                const int col = 4 * (wb * 16 + (lane & 15));
                *reinterpret_cast<f32x4 *>(slab + row * COLS + col) = f32x4{acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
            } else {
                slab[row * COLS + (lane & 15)] = acc[i][0][r];
            }
        }
    if (wb == 0) {
#pragma unroll
        for (int i = 0; i < IA; ++i) {
            float sum = bsum2[i >> 1][i & 1];
            sum += __shfl_xor(sum, 16, WAVE);                              // the four point residues of the row group
            sum += __shfl_xor(sum, 32, WAVE);
            const int row = (IA == 4) ? 4 * (wa * 16 + (lane & 15)) + i : (lane & 15);
            if (kq == 0) slab[ROWS * COLS + row] = sum;
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
        const int wp = t.wp > 1 ? t.wp : 1;
        t.wp = wp;
        t.chunks = c * wp; t.wg0 = wg; t.part_off = off;
        wg += c;
        off += c * wp * (t.JB * 32 * (t.KB * 32 + 1));
    }
    P.n_wg = wg;
}

static inline size_t dw_partial_floats(const DwPlan &P) {
    const DwTask &t = P.t[P.n_tasks - 1];
    return (size_t)t.part_off + (size_t)t.chunks * (t.JB * 32 * (t.KB * 32 + 1));
}

}  // namespace nerfmi
