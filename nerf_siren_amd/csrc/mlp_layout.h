// Packed-parameter layout of one NeRF() (models/nerf.py:41-124) for gfx950.
//
// v_mfma_f32_32x32x2_f32 computes D[32x32] += A[32x2] * B[2x32]; lane l holds
// A[row l&31][k = l>>5] and B[k = l>>5][col l&31]; D register r of lane l is
// D[row (r&3) + 8*(r>>2) + 4*(l>>5)][col l&31].
//
// The MLP runs TRANSPOSED: D = H_out^T (rows = output units, cols = 32 points of
// a wave), A = a 32x2 sliver of W, B = H_in^T.  Because D's row map
// (r, half) -> unit 8*(r>>2) + 4*half + (r&3) is exactly the order in which the
// next layer consumes B registers, a layer's accumulators feed the next layer
// with no transpose, no LDS and no barrier: the K index of MFMA step r of input
// block kb is k = 32*kb + 8*(r>>2) + 4*half + (r&3).
//
// So the weights are re-packed once per update into "fragment order": for a
// layer with JB output blocks (of 32 units) and KB input blocks (of 32 inputs)
//     P[((jb*KB + kb)*4 + q)*256 + lane*4 + t] = W[32*jb + (lane&31)][32*kb + 8*q + 4*(lane>>5) + t]
// i.e. one fully coalesced 1 KiB global_load_dwordx4 per four MFMAs.  Inputs
// that are concatenations (skip layer 5: [xyz-emb 63 | hidden 256], dir layer:
// [final 256 | dir-emb 27], nerf.py:109,118) are two segments each zero-padded
// to a multiple of 32 columns.
//
// The backward dX chain needs W^T as the A operand:
//     T[((kbo*JBc + jb)*4 + q)*256 + lane*4 + t] = W[32*jb + 8*q + 4*(lane>>5) + t][col0 + 32*kbo + (lane&31)]
#pragma once

namespace nerfmi {

constexpr int NL_FWD = 10;   // xyz_encoding_1..8, xyz_encoding_final, dir_encoding
constexpr int N_PARAMS = 24; // state_dict tensors, nerf.py:61-81 order

struct LayerDesc {
    int param;     // index of the weight tensor in params[] (bias = param+1)
    int out_f;     // rows of W
    int in_f;      // cols of W
    int seg0;      // first input segment length (cols [0,seg0))
    int seg1;      // second segment length (cols [seg0, seg0+seg1)), 0 if none
    int JB;        // output blocks
    int KB;        // padded input blocks = ceil32(seg0)/32 + ceil32(seg1)/32
    int off;       // float offset of the forward image inside `packed`
    int t_col0;    // transposed image: first source column (skip layer: 63, else 0)
    int t_KBO;     // transposed image: number of 32-column blocks (0 = none)
    int t_off;     // float offset of the transposed image
};

constexpr int pad32(int n) { return (n + 31) / 32 * 32; }

// forward image offsets
constexpr int OFF_L1 = 0;                                  // 8 x 2 blocks
constexpr int SZ_L1 = 8 * 2 * 1024;
constexpr int SZ_HID = 8 * 8 * 1024;
constexpr int OFF_L2 = OFF_L1 + SZ_L1;
constexpr int OFF_L3 = OFF_L2 + SZ_HID;
constexpr int OFF_L4 = OFF_L3 + SZ_HID;
constexpr int OFF_L5 = OFF_L4 + SZ_HID;                    // 8 x 10 blocks
constexpr int SZ_L5 = 8 * 10 * 1024;
constexpr int OFF_L6 = OFF_L5 + SZ_L5;
constexpr int OFF_L7 = OFF_L6 + SZ_HID;
constexpr int OFF_L8 = OFF_L7 + SZ_HID;
constexpr int OFF_FINAL = OFF_L8 + SZ_HID;
constexpr int OFF_DIR = OFF_FINAL + SZ_HID;                // 4 x 9 blocks
constexpr int SZ_DIR = 4 * 9 * 1024;
constexpr int OFF_SMALL = OFF_DIR + SZ_DIR;
// small section (natural unit order; the D-row map makes fragment order == natural order)
constexpr int OFF_BIAS = OFF_SMALL;                        // 8 x 256 (xyz_encoding_1..8)
constexpr int OFF_BIAS_FINAL = OFF_BIAS + 8 * 256;
constexpr int OFF_BIAS_DIR = OFF_BIAS_FINAL + 256;         // 128
constexpr int OFF_W_SIGMA = OFF_BIAS_DIR + 128;            // 256
constexpr int OFF_B_SIGMA = OFF_W_SIGMA + 256;             // 1 (+3 pad)
constexpr int OFF_W_RGB = OFF_B_SIGMA + 4;                 // 3 x 128
constexpr int OFF_B_RGB = OFF_W_RGB + 384;                 // 3 (+1 pad)
constexpr int OFF_TRANS = OFF_B_RGB + 4;
// transposed images (backward): dir(final part), final, layers 8..2 hidden part -- stored in the ORDER THE
// BACKWARD CHAIN WALKS THEM, so that its weight stream runs across layer ends exactly like the forward one
// (mlp_core.h layer_mfma_lds)
constexpr int OFF_TDIR = OFF_TRANS;                        // 8 kbo x 4 jb
constexpr int SZ_TDIR = 8 * 4 * 1024;
constexpr int OFF_TFINAL = OFF_TDIR + SZ_TDIR;
constexpr int OFF_T8 = OFF_TFINAL + SZ_HID;
constexpr int OFF_T7 = OFF_T8 + SZ_HID;
constexpr int OFF_T6 = OFF_T7 + SZ_HID;
constexpr int OFF_T5 = OFF_T6 + SZ_HID;
constexpr int OFF_T4 = OFF_T5 + SZ_HID;
constexpr int OFF_T3 = OFF_T4 + SZ_HID;
constexpr int OFF_T2 = OFF_T3 + SZ_HID;
// the stream prefetches two 16 KiB stages past the layer it is in: readable tail behind the last image
constexpr int STREAM_TAIL = 2 * 16 * 256;
constexpr int PACKED_FLOATS = OFF_T2 + SZ_HID + STREAM_TAIL;

// {param, out, in, seg0, seg1, JB, KB, off, t_col0, t_KBO, t_off}
constexpr LayerDesc LAYERS[NL_FWD] = {
    {0, 256, 63, 63, 0, 8, 2, OFF_L1, 0, 0, 0},
    {2, 256, 256, 256, 0, 8, 8, OFF_L2, 0, 8, OFF_T2},
    {4, 256, 256, 256, 0, 8, 8, OFF_L3, 0, 8, OFF_T3},
    {6, 256, 256, 256, 0, 8, 8, OFF_L4, 0, 8, OFF_T4},
    {8, 256, 319, 63, 256, 8, 10, OFF_L5, 63, 8, OFF_T5},
    {10, 256, 256, 256, 0, 8, 8, OFF_L6, 0, 8, OFF_T6},
    {12, 256, 256, 256, 0, 8, 8, OFF_L7, 0, 8, OFF_T7},
    {14, 256, 256, 256, 0, 8, 8, OFF_L8, 0, 8, OFF_T8},
    {16, 256, 256, 256, 0, 8, 8, OFF_FINAL, 0, 8, OFF_TFINAL},
    {18, 128, 283, 256, 27, 4, 9, OFF_DIR, 0, 8, OFF_TDIR},
};
constexpr int PARAM_SIGMA_W = 20, PARAM_SIGMA_B = 21, PARAM_RGB_W = 22, PARAM_RGB_B = 23;

// Activations kept for training: tile-major [tile of 32 points][row][32] images
// (mlp_core.h RowImage), n_points padded to 32: rows of S_* below.  Post-ReLU
// values (nn.ReLU(True) saves its output, nerf.py:68).
constexpr int S_EMB = 0;        // 64 rows: xyz embedding (row 63 = 0)
constexpr int S_H = 64;         // 8 x 256 rows: h1..h8
constexpr int S_FINAL = S_H + 8 * 256;   // 256 rows: xyz_encoding_final output
constexpr int S_DEMB = S_FINAL + 256;    // 32 rows: dir embedding (rows 27..31 = 0)
constexpr int S_DIRH = S_DEMB + 32;      // 128 rows: dir_encoding output
constexpr int S_RGB = S_DIRH + 128;      // 3 rows (+1 pad): sigmoid output
// ReLU sign bits, 9 x 8 rows (= 9 x 1 KiB per tile): for layer m (0..7 = xyz_encoding_1..8, 8 =
// dir_encoding) lane l of the owning wave keeps 4 dwords at row S_MASK + 8*m, float offset 4*l:
// bit 16*(jb&1) + r of dword jb>>1 is (output register r of block jb > 0).  The backward chain uses
// the same lane/register map, so it masks with one 16-byte load per layer instead of re-reading the
// fp32 activations (32x less traffic and no HBM-latency load in its in-order vmcnt queue).
constexpr int S_MASK = S_RGB + 4;
constexpr int SAVED_ROWS = S_MASK + 9 * 8;

}  // namespace nerfmi
