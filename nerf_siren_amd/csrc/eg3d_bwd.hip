// Backward of the EG3D tri-plane importance renderer (the autograd graph the reference
// builds in volumetric_rendering/renderer.py:88-142): gradients w.r.t. the feature planes and the
// OSGDecoder parameters.  Depths carry no gradient (stratified draws; sample_importance runs under
// no_grad, renderer.py:201), so the chain is
//   marcher(fine) -> unify (a permutation) -> [+ marcher(coarse)] -> decoder -> bilinear scatter.
//
//  * mip_march_backward_kernel : one wave per ray, suffix scan in fp64 (as composite_backward)
//  * unify_backward_kernel     : inverse permutation (no atomics: each source sample appears once)
//  * triplane_backward_kernel  : one point per lane recomputes gather + decoder, back-propagates to the
//    32 mean features; the scatter into the channels-last plane gradient is re-shaped through LDS so that
//    every atomic wave-instruction adds two contiguous 128-byte texels (the full-rate shape of
//    global_atomic_add_f32) instead of 64 scattered dwords; per-point decoder intermediates go to a
//    [field][point] scratch image
//  * decoder_wgrad_kernel / _reduce : the 2 372 decoder-parameter gradients as a small LDS-tiled reduction
//    over points, chunk slabs + a deterministic final sum (gains of FullyConnectedLayer applied there)
#include "common.h"

namespace nerfmi {

constexpr int EC = 32, DEC_H = 64;
constexpr int DEC_FLOATS = DEC_H * EC + DEC_H + 4 * DEC_H + 4;
constexpr int AUX_F = DEC_H + EC + DEC_H + 4;     // [d_pre(64) | m(32) | h(64) | d_x(4)] per point

__device__ __forceinline__ float sigmoidf_(float x) { return __fdiv_rn(1.f, __fadd_rn(1.f, expf(-x))); }

// ---------------------------------------------------------------------------
// MipRayMarcher2 backward
// ---------------------------------------------------------------------------
template <int SPL>
__global__ void __launch_bounds__(64)
mip_march_backward_kernel(const float *__restrict__ colors, const float *__restrict__ dens,
                          const float *__restrict__ depths, const float *__restrict__ minmax,
                          const float *__restrict__ g_rgb, const float *__restrict__ g_depth,
                          const float *__restrict__ g_wsum, int64_t R, int S, int white_back, int accumulate,
                          float *__restrict__ d_colors, float *__restrict__ d_dens) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x, NI = S - 1;
    float *wl = lds, *gdl = lds + S;       // per-interval weight and d(density_mid)
    for (int64_t r = blockIdx.x; r < R; r += gridDim.x) {
        const float *cr = colors + r * S * 3, *dr = dens + r * S, *zr = depths + r * S;
        float alpha[SPL], aa[SPL], Tt[SPL], w[SPL], cm[SPL][3], zm[SPL], delta[SPL], ee[SPL], dmm[SPL];
        double pl[SPL], lp = 1.0;
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            const int i = lane * SPL + j;
            const bool ok = i < NI;
            const int ic = ok ? i : (NI > 0 ? NI - 1 : 0);
            const float z0 = zr[ic], z1 = zr[ic + 1];
            delta[j] = __fsub_rn(z1, z0);
#pragma unroll
            for (int k = 0; k < 3; ++k) cm[j][k] = __fdiv_rn(__fadd_rn(cr[ic * 3 + k], cr[(ic + 1) * 3 + k]), 2.f);
            dmm[j] = __fsub_rn(__fdiv_rn(__fadd_rn(dr[ic], dr[ic + 1]), 2.f), 1.f);
            zm[j] = __fdiv_rn(__fadd_rn(z0, z1), 2.f);
            const float sp = dmm[j] > 20.f ? dmm[j] : (float)log1p(exp((double)dmm[j]));
            ee[j] = (float)exp(-(double)__fmul_rn(sp, delta[j]));
            alpha[j] = ok ? __fsub_rn(1.f, ee[j]) : 0.f;
            aa[j] = ok ? __fadd_rn(__fsub_rn(1.f, alpha[j]), 1e-10f) : 1.f;
            pl[j] = lp;
            lp *= (double)aa[j];
        }
        const double incl = wave_incl_prod_d(lp, lane);
        double excl = shfl_up_d(incl, 1);
        if (lane == 0) excl = 1.0;
        double sw = 0, sz = 0;
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            const int i = lane * SPL + j;
            Tt[j] = (float)(excl * pl[j]);
            w[j] = (i < NI) ? __fmul_rn(alpha[j], Tt[j]) : 0.f;
            sw += (double)w[j];
            sz += (double)__fmul_rn(w[j], zm[j]);
        }
        sw = wave_sum_d(sw);
        sz = wave_sum_d(sz);
        const double W = sw, D = sz / sw;
        // composite_depth = clamp(nan_to_num(D, inf), min, max): gradient flows only strictly inside the clamp
        const bool pass = (D == D) && D >= (double)minmax[0] && D <= (double)minmax[1];
        const double gr = g_rgb ? (double)g_rgb[r * 3] : 0.0, gg = g_rgb ? (double)g_rgb[r * 3 + 1] : 0.0,
                     gb = g_rgb ? (double)g_rgb[r * 3 + 2] : 0.0;
        const double gd = (g_depth && pass) ? (double)g_depth[r] : 0.0, gw = g_wsum ? (double)g_wsum[r] : 0.0;
        const double wbt = white_back ? (gr + gg + gb) : 0.0;
        double v[SPL], wv_incl[SPL], lsum = 0;
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            v[j] = (double)cm[j][0] * gr + (double)cm[j][1] * gg + (double)cm[j][2] * gb +
                   gd * ((double)zm[j] - D) / W + gw - wbt;
            lsum += (double)w[j] * v[j];
            wv_incl[j] = lsum;
        }
        const double inc2 = wave_incl_sum_d(lsum, lane);
        const double total = __shfl(inc2, WAVE - 1, WAVE);
        const double exc2 = inc2 - lsum;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            const int i = lane * SPL + j;
            if (i < NI) {
                const double suffix = total - (exc2 + wv_incl[j]);
                const double d_alpha = (double)Tt[j] * v[j] - suffix / (double)aa[j];
                const double d_sp = d_alpha * (double)delta[j] * (double)ee[j];
                const double sg = dmm[j] > 20.f ? 1.0 : 1.0 / (1.0 + exp(-(double)dmm[j]));      // softplus'
                wl[i] = w[j];
                gdl[i] = (float)(d_sp * sg);
            }
        }
        __syncthreads();
        for (int s = lane; s < S; s += WAVE) {
            const float wa = s > 0 ? wl[s - 1] : 0.f, wb = s < NI ? wl[s] : 0.f;
            const float ga = s > 0 ? gdl[s - 1] : 0.f, gbb = s < NI ? gdl[s] : 0.f;
            const float ws2 = 0.5f * (wa + wb);
            float o0 = (float)(ws2 * gr), o1 = (float)(ws2 * gg), o2 = (float)(ws2 * gb), od = 0.5f * (ga + gbb);
            float *dc = d_colors + (r * S + s) * 3;
            if (accumulate) { o0 += dc[0]; o1 += dc[1]; o2 += dc[2]; od += d_dens[r * S + s]; }
            dc[0] = o0; dc[1] = o1; dc[2] = o2;
            d_dens[r * S + s] = od;
        }
    }
}

__global__ void unify_backward_kernel(const int *__restrict__ idx, const float *__restrict__ g_c,
                                      const float *__restrict__ g_s, int64_t R, int n1, int n2,
                                      float *__restrict__ d_c1, float *__restrict__ d_s1, float *__restrict__ d_c2,
                                      float *__restrict__ d_s2) {
    const int n = n1 + n2;
    const int64_t total = R * n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / n;
        const int src = idx[i];
        float *dc = src < n1 ? d_c1 + (r * n1 + src) * 3 : d_c2 + (r * n2 + (src - n1)) * 3;
        dc[0] = g_c[i * 3]; dc[1] = g_c[i * 3 + 1]; dc[2] = g_c[i * 3 + 2];
        if (src < n1) d_s1[r * n1 + src] = g_s[i]; else d_s2[r * n2 + (src - n1)] = g_s[i];
    }
}

// ---------------------------------------------------------------------------
// tri-plane gather + decoder backward
// ---------------------------------------------------------------------------
struct Tap {
    int off[4];      // texel index (y*W + x) or -1
    float w[4];
};

__device__ __forceinline__ void plane_taps(int H, int W, float gx, float gy, Tap &t) {
    const float ix = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(gx, 1.f), (float)W), 1.f), 2.f);
    const float iy = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(gy, 1.f), (float)H), 1.f), 2.f);
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float x1f = x0f + 1.f, y1f = y0f + 1.f;
    const float wx1 = __fsub_rn(ix, x0f), wx0 = __fsub_rn(x1f, ix);
    const float wy1 = __fsub_rn(iy, y0f), wy0 = __fsub_rn(y1f, iy);
    const float wgt[4] = {__fmul_rn(wx0, wy0), __fmul_rn(wx1, wy0), __fmul_rn(wx0, wy1), __fmul_rn(wx1, wy1)};
    const float xs[4] = {x0f, x1f, x0f, x1f}, ys[4] = {y0f, y0f, y1f, y1f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool ok = xs[k] >= 0.f && xs[k] < (float)W && ys[k] >= 0.f && ys[k] < (float)H;
        t.off[k] = ok ? (int)ys[k] * W + (int)xs[k] : -1;
        t.w[k] = wgt[k];
    }
}

__global__ void __launch_bounds__(256)
triplane_backward_kernel(const float *__restrict__ planes, int N, int H, int W, const float *__restrict__ ray_o,
                         const float *__restrict__ ray_d, const float *__restrict__ depths, int S, int64_t P,
                         float coord_scale, const float *__restrict__ dec, const float *__restrict__ d_rgb,
                         const float *__restrict__ d_sigma, float *__restrict__ gplanes, float *__restrict__ aux,
                         int64_t aux_ld) {
    // per wave: dm[64][33] | tap offsets [64][12] | tap weights [64][12]
    __shared__ float s_dm[4][64 * 33];
    __shared__ int s_off[4][64 * 12];
    __shared__ float s_w[4][64 * 12];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t total = (int64_t)N * P;
    // the loop bound is uniform across the workgroup (barriers inside); a wave past the end works on a clamped
    // point with ok == false
    for (int64_t base_wg = (int64_t)blockIdx.x * 256; base_wg < total; base_wg += (int64_t)gridDim.x * 256) {
        const int64_t base = base_wg + wid * 64;
        const int64_t idx = base + lane;
        const bool ok = idx < total;
        const int64_t ic = ok ? idx : total - 1;
        const int64_t n = ic / P, p = ic % P;
        const int64_t ray = n * (P / S) + p / S;
        const float dz = depths[ic];
        float c[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) c[k] = __fmul_rn(coord_scale, __fadd_rn(ray_o[ray * 3 + k], __fmul_rn(dz, ray_d[ray * 3 + k])));
        const int sa[3] = {0, 0, 2}, sb[3] = {1, 2, 0};
        Tap taps[3];
        float m[EC];
#pragma unroll
        for (int k = 0; k < EC; ++k) m[k] = 0.f;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            plane_taps(H, W, c[sa[pl]], c[sb[pl]], taps[pl]);
            const float *plane = planes + ((n * 3 + pl) * (int64_t)H * W) * EC;
            float f[EC];
#pragma unroll
            for (int k = 0; k < EC; ++k) f[k] = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (taps[pl].off[t] >= 0) {
                    const float4 *tex = reinterpret_cast<const float4 *>(plane + (int64_t)taps[pl].off[t] * EC);
#pragma unroll
                    for (int v = 0; v < EC / 4; ++v) {
                        const float4 q = tex[v];
                        f[4 * v + 0] = __fadd_rn(f[4 * v + 0], __fmul_rn(q.x, taps[pl].w[t]));
                        f[4 * v + 1] = __fadd_rn(f[4 * v + 1], __fmul_rn(q.y, taps[pl].w[t]));
                        f[4 * v + 2] = __fadd_rn(f[4 * v + 2], __fmul_rn(q.z, taps[pl].w[t]));
                        f[4 * v + 3] = __fadd_rn(f[4 * v + 3], __fmul_rn(q.w, taps[pl].w[t]));
                    }
                }
#pragma unroll
            for (int k = 0; k < EC; ++k) m[k] = (pl == 0) ? f[k] : __fadd_rn(m[k], f[k]);
        }
#pragma unroll
        for (int k = 0; k < EC; ++k) m[k] = div3_rn(m[k]);
        const float *w0 = dec, *b0 = dec + DEC_H * EC, *w1 = b0 + DEC_H, *b1 = w1 + 4 * DEC_H;
        // pass 1: outputs
        float o4[4] = {b1[0], b1[1], b1[2], b1[3]};
        for (int j = 0; j < DEC_H; ++j) {
            float h = b0[j];
#pragma unroll
            for (int k = 0; k < EC; ++k) h = __builtin_fmaf(w0[j * EC + k], m[k], h);
            h = softplus_hw(h);
#pragma unroll
            for (int k = 0; k < 4; ++k) o4[k] = __builtin_fmaf(w1[k * DEC_H + j], h, o4[k]);
        }
        float dx[4];
        dx[0] = ok ? d_sigma[ic] : 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float sg = sigmoidf_(o4[1 + k]);
            dx[1 + k] = ok ? d_rgb[ic * 3 + k] * 1.002f * sg * (1.f - sg) : 0.f;     // triplane.py:165
        }
        // pass 2: hidden layer backward
        float dm[EC];
#pragma unroll
        for (int k = 0; k < EC; ++k) dm[k] = 0.f;
        for (int j = 0; j < DEC_H; ++j) {
            float pre = b0[j];
#pragma unroll
            for (int k = 0; k < EC; ++k) pre = __builtin_fmaf(w0[j * EC + k], m[k], pre);
            const float h = softplus_hw(pre);
            float dh = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) dh = __builtin_fmaf(w1[k * DEC_H + j], dx[k], dh);
            const float dpre = dh * (pre > 20.f ? 1.f : sigmoid_hw(pre));
            if (ok) {
                aux[(int64_t)j * aux_ld + idx] = dpre;
                aux[(int64_t)(DEC_H + EC + j) * aux_ld + idx] = h;
            }
#pragma unroll
            for (int k = 0; k < EC; ++k) dm[k] = __builtin_fmaf(w0[j * EC + k], dpre, dm[k]);
        }
        if (ok) {
#pragma unroll
            for (int k = 0; k < EC; ++k) aux[(int64_t)(DEC_H + k) * aux_ld + idx] = m[k];
#pragma unroll
            for (int k = 0; k < 4; ++k) aux[(int64_t)(2 * DEC_H + EC + k) * aux_ld + idx] = dx[k];
        }
        // re-shape the scatter: lane = channel, two points per atomic wave-instruction
#pragma unroll
        for (int k = 0; k < EC; ++k) s_dm[wid][lane * 33 + k] = dm[k] * (1.f / 3.f);     // d(mean over planes)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                s_off[wid][lane * 12 + pl * 4 + t] = ok ? taps[pl].off[t] : -1;
                s_w[wid][lane * 12 + pl * 4 + t] = taps[pl].w[t];
            }
        __syncthreads();
        const int ch = lane & 31, hp = lane >> 5;
        for (int pp = 0; pp < 64; pp += 2) {
            const int q = pp + hp;                                  // the point this half-wave scatters
            const int64_t qi = base + q;
            const int64_t qn = (qi < total ? qi : total - 1) / P;
            const float g = s_dm[wid][q * 33 + ch];
#pragma unroll
            for (int tt = 0; tt < 12; ++tt) {
                const int off = s_off[wid][q * 12 + tt];
                if (off >= 0 && qi < total)
                    atomicAdd(gplanes + ((qn * 3 + tt / 4) * (int64_t)H * W + off) * EC + ch, g * s_w[wid][q * 12 + tt]);
            }
        }
        __syncthreads();
    }
}

// decoder parameter gradients: chunk of points -> partial slab [DEC_FLOATS]
// A 64-point tile of the per-point intermediates ([field][point] in memory) is transposed into LDS as [point][field];
// the four waves take every fourth point, and a lane owns a 4 x 8 block of dW0 (rows 4jb.., columns 8kb..), one row of
// dW1^T and one entry of db0: per point it reads 3 + 2 sixteen-byte fragments for 38 FMAs.  (The first version read one
// LDS dword per FMA at a 64-float row pitch -- a 16-way bank conflict -- and was 29 % of the renderer's training step.)
__global__ void __launch_bounds__(256)
decoder_wgrad_kernel(const float *__restrict__ aux, int64_t aux_ld, int64_t total, int n_chunks,
                     float *__restrict__ partial) {
    constexpr int PITCH = AUX_F;                       // 164 floats: 16-byte aligned rows
    __shared__ __attribute__((aligned(16))) float tile[64 * PITCH];
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int t = threadIdx.x, g = t >> 6, l = t & 63;
    const int64_t per = (total + n_chunks - 1) / n_chunks;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = (lo + per < total) ? lo + per : total;
    const int jb = l >> 2, kb = l & 3;
    float a0[4][8], a1[4] = {0.f, 0.f, 0.f, 0.f}, ab0 = 0.f, ab1 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 8; ++k) a0[i][k] = 0.f;
    for (int64_t p0 = lo; p0 < hi; p0 += 64) {
        __syncthreads();
        for (int u = t; u < AUX_F * 64; u += 256) {
            const int f = u >> 6, pp = u & 63;
            tile[pp * PITCH + f] = (p0 + pp < hi) ? aux[(int64_t)f * aux_ld + p0 + pp] : 0.f;
        }
        __syncthreads();
#pragma unroll 4
        for (int pp = g; pp < 64; pp += 4) {
            const float *row = tile + pp * PITCH;         // [d_pre(64) | m(32) | h(64) | d_x(4)]
            const f4 dp = *reinterpret_cast<const f4 *>(row + 4 * jb);
            const f4 m0 = *reinterpret_cast<const f4 *>(row + DEC_H + 8 * kb), m1 = *reinterpret_cast<const f4 *>(row + DEC_H + 8 * kb + 4);
            const f4 dx = *reinterpret_cast<const f4 *>(row + 2 * DEC_H + EC);
            const float hh = row[DEC_H + EC + l];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    a0[i][k] = __builtin_fmaf(dp[i], m0[k], a0[i][k]);
                    a0[i][4 + k] = __builtin_fmaf(dp[i], m1[k], a0[i][4 + k]);
                }
                a1[i] = __builtin_fmaf(dx[i], hh, a1[i]);
            }
            ab0 += row[l];
            if (l < 4) ab1 += row[2 * DEC_H + EC + l];
        }
    }
    // sum the four waves' partial blocks through LDS, then one slab per chunk
    __syncthreads();
    float *red = tile;                                    // [g][DEC_FLOATS]
    float *mine = red + g * DEC_FLOATS;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) mine[(4 * jb + i) * EC + 8 * kb + k] = a0[i][k];
        mine[DEC_H * EC + DEC_H + i * DEC_H + l] = a1[i];
    }
    mine[DEC_H * EC + l] = ab0;
    if (l < 4) mine[DEC_H * EC + DEC_H + 4 * DEC_H + l] = ab1;
    __syncthreads();
    float *out = partial + (int64_t)blockIdx.x * DEC_FLOATS;
    for (int i = t; i < DEC_FLOATS; i += 256)
        out[i] = (red[i] + red[DEC_FLOATS + i]) + (red[2 * DEC_FLOATS + i] + red[3 * DEC_FLOATS + i]);
}

// sum the chunk slabs, apply FullyConnectedLayer's gains (w_eff = w*gain, b_eff = b*lr_mul)
__global__ void decoder_wgrad_reduce_kernel(const float *__restrict__ partial, int n_chunks, float lr_mul, int accumulate,
                                            float *__restrict__ g_w0, float *__restrict__ g_b0,
                                            float *__restrict__ g_w1, float *__restrict__ g_b1) {
    const float gain0 = (float)((double)lr_mul / sqrt((double)EC)), gain1 = (float)((double)lr_mul / sqrt((double)DEC_H));
    // one workgroup per 64 entries: wave w sums chunks [w*q, (w+1)*q) with eight slab loads in flight, then the four
    // partial sums meet in LDS in a fixed order (bit-reproducible run to run)
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    const int q = (n_chunks + 3) / 4, c0 = w * q, c1 = min(n_chunks, c0 + q);
    float s = 0.f;
    if (i < DEC_FLOATS) {
        int c = c0;
        for (; c + 8 <= c1; c += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(int64_t)(c + u) * DEC_FLOATS + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; c < c1; ++c) s += partial[(int64_t)c * DEC_FLOATS + i];
    }
    part[w][lane] = s;
    __syncthreads();
    if (w == 0 && i < DEC_FLOATS) {
        s = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
        float *dst;
        float g;
        if (i < DEC_H * EC) { dst = g_w0 + i; g = gain0; }
        else if (i < DEC_H * EC + DEC_H) { dst = g_b0 + (i - DEC_H * EC); g = lr_mul; }
        else if (i < DEC_H * EC + DEC_H + 4 * DEC_H) { dst = g_w1 + (i - DEC_H * EC - DEC_H); g = gain1; }
        else { dst = g_b1 + (i - DEC_H * EC - DEC_H - 4 * DEC_H); g = lr_mul; }
        *dst = (accumulate ? *dst : 0.f) + s * g;
    }
}

__global__ void unpack_planes_kernel(const float *__restrict__ src, int64_t n_img, int C, int H, int W,
                                     float *__restrict__ dst) {
    const int64_t total = n_img * C * H * W;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(idx % W);                  // dst index order (img, c, y, x)
        const int y = (int)((idx / W) % H);
        const int c = (int)((idx / ((int64_t)W * H)) % C);
        const int64_t img = idx / ((int64_t)W * H * C);
        dst[idx] = src[((img * H + y) * W + x) * C + c];
    }
}

static inline int grid_for(int64_t total, int block) {
    int64_t g = (total + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}
constexpr int WGRAD_CHUNKS = 768;   // three workgroups per CU (42 KB of LDS each): one loads its tile while another computes

}  // namespace nerfmi

using namespace nerfmi;

extern "C" {

int nerfmi_eg3d_march_backward(const float *colors, const float *densities, const float *depths, const float *minmax,
                               const float *g_rgb, const float *g_depth, const float *g_weight_sum, int64_t n_rays,
                               int n_samples, int white_back, int accumulate, float *d_colors, float *d_densities,
                               nerfmi_stream_t stream) {
    NERFMI_ENTER();
    NERFMI_REQUIRE(n_rays >= 0 && n_samples >= 2 && n_samples <= 1025, "eg3d_march_backward: n_samples=%d out of [2,1025]", n_samples);
    if (n_rays == 0) return NERFMI_OK;
    NERFMI_REQUIRE(colors && densities && depths && minmax && d_colors && d_densities, "eg3d_march_backward: null pointer");
    const dim3 grid((unsigned)(n_rays < 65536 ? n_rays : 65536)), block(64);
    const size_t lds = sizeof(float) * 2 * (size_t)n_samples;
    hipStream_t st = (hipStream_t)stream;
    const int spl = (n_samples - 1 + 63) / 64;
#define CALL(SPL) hipLaunchKernelGGL((mip_march_backward_kernel<SPL>), grid, block, lds, st, colors, densities, depths, \
                                     minmax, g_rgb, g_depth, g_weight_sum, n_rays, n_samples, white_back, accumulate,   \
                                     d_colors, d_densities)
    if (spl <= 1) CALL(1); else if (spl <= 2) CALL(2); else if (spl <= 4) CALL(4); else if (spl <= 8) CALL(8); else CALL(16);
#undef CALL
    return check_launch("eg3d_march_backward");
}

int nerfmi_eg3d_unify_backward(const int32_t *idx, const float *g_colors, const float *g_densities, int64_t n_rays, int n1,
                               int n2, float *d_c1, float *d_s1, float *d_c2, float *d_s2, nerfmi_stream_t stream) {
    NERFMI_ENTER();
    NERFMI_REQUIRE(n_rays >= 0 && n1 >= 1 && n2 >= 0, "eg3d_unify_backward: bad sizes");
    if (n_rays == 0) return NERFMI_OK;
    NERFMI_REQUIRE(idx && g_colors && g_densities && d_c1 && d_s1 && (n2 == 0 || (d_c2 && d_s2)), "eg3d_unify_backward: null pointer");
    hipLaunchKernelGGL(unify_backward_kernel, dim3(grid_for(n_rays * (n1 + n2), 256)), dim3(256), 0, (hipStream_t)stream,
                       (const int *)idx, g_colors, g_densities, n_rays, n1, n2, d_c1, d_s1, d_c2, d_s2);
    return check_launch("eg3d_unify_backward");
}

size_t nerfmi_eg3d_backward_aux_floats(int64_t n_points) { return (size_t)AUX_F * (size_t)((n_points + 63) / 64 * 64); }
size_t nerfmi_eg3d_wgrad_partial_floats(void) { return (size_t)WGRAD_CHUNKS * DEC_FLOATS; }

int nerfmi_eg3d_run_model_rays_backward(const float *planes_hwc, int n, int h, int w, const float *decoder_packed,
                                        const float *ray_origins, const float *ray_directions, const float *depths,
                                        int64_t n_rays_per_batch, int n_samples, float box_warp, const float *d_rgb,
                                        const float *d_sigma, float *gplanes_hwc, float *aux, nerfmi_stream_t stream) {
    NERFMI_ENTER();
    NERFMI_REQUIRE(n >= 1 && h >= 1 && w >= 1 && n_rays_per_batch >= 0 && n_samples >= 1 && box_warp != 0.f,
                   "eg3d_run_model_rays_backward: bad sizes");
    const int64_t P = n_rays_per_batch * n_samples;
    if (P == 0) return NERFMI_OK;
    NERFMI_REQUIRE(planes_hwc && decoder_packed && ray_origins && ray_directions && depths && d_rgb && d_sigma &&
                   gplanes_hwc && aux, "eg3d_run_model_rays_backward: null pointer");
    const int64_t total = (int64_t)n * P;
    const int64_t aux_ld = (total + 63) / 64 * 64;
    const float scale = (float)(2.0 / (double)box_warp);
    const int64_t waves = (total + 63) / 64;
    int64_t grid = (waves + 3) / 4;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(triplane_backward_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, planes_hwc, n, h, w,
                       ray_origins, ray_directions, depths, n_samples, P, scale, decoder_packed, d_rgb, d_sigma,
                       gplanes_hwc, aux, aux_ld);
    return check_launch("eg3d_run_model_rays_backward");
}

int nerfmi_eg3d_decoder_wgrad(const float *aux, int64_t n_points, float lr_multiplier, int accumulate, float *partial,
                              float *g_w0, float *g_b0, float *g_w1, float *g_b1, nerfmi_stream_t stream) {
    NERFMI_ENTER();
    NERFMI_REQUIRE(n_points >= 1 && aux && partial && g_w0 && g_b0 && g_w1 && g_b1, "eg3d_decoder_wgrad: bad arguments");
    const int64_t aux_ld = (n_points + 63) / 64 * 64;
    hipLaunchKernelGGL(decoder_wgrad_kernel, dim3(WGRAD_CHUNKS), dim3(256), 0, (hipStream_t)stream, aux, aux_ld, n_points,
                       WGRAD_CHUNKS, partial);
    hipLaunchKernelGGL(decoder_wgrad_reduce_kernel, dim3((DEC_FLOATS + 63) / 64), dim3(256), 0, (hipStream_t)stream, partial, WGRAD_CHUNKS,
                       lr_multiplier, accumulate, g_w0, g_b0, g_w1, g_b1);
    return check_launch("eg3d_decoder_wgrad");
}

int nerfmi_eg3d_unpack_planes(const float *planes_hwc, int n_planes, int channels, int h, int w, float *planes_nchw,
                              nerfmi_stream_t stream) {
    NERFMI_ENTER();
    NERFMI_REQUIRE(n_planes >= 1 && channels >= 1 && h >= 1 && w >= 1 && planes_hwc && planes_nchw, "eg3d_unpack_planes: bad arguments");
    const int64_t total = (int64_t)n_planes * channels * h * w;
    hipLaunchKernelGGL(unpack_planes_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, planes_hwc,
                       (int64_t)n_planes, channels, h, w, planes_nchw);
    return check_launch("eg3d_unpack_planes");
}

}  // extern "C"
