// Per-ray kernels of the render_rays hot path for gfx950 (MI355X):
//   stratified sampler (a2), compositing fwd/bwd (a8), sample_pdf (a3),
//   searchsorted, merge (a4), importance_resample (a3+a4 fused), Embedding (a5).
// One 64-lane wavefront owns one ray; per-ray state lives in registers/LDS;
// transmittance and cdf are wave-level scans with fp64 accumulation (the
// arithmetic the reference's CPU cumsum/cumprod use -- oracle/nerf_oracle.py).
// All are HBM-bound: loads/stores are coalesced along the sample axis.
#include <stdarg.h>

#include "philox.h"

namespace nerfmi {

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---------------------------------------------------------------------------
// a2  stratified sampler (models/rendering.py:207-222)
// ---------------------------------------------------------------------------
__device__ __forceinline__ float z_at(int i, int S, float near, float far, bool disp) {
    const float t = linspace01(i, S);
    const float omt = __fsub_rn(1.0f, t);
    if (!disp) return __fadd_rn(__fmul_rn(near, omt), __fmul_rn(far, t));
    const float a = __fmul_rn(__fdiv_rn(1.0f, near), omt);
    const float b = __fmul_rn(__fdiv_rn(1.0f, far), t);
    return __fdiv_rn(1.0f, __fadd_rn(a, b));
}

__global__ void sample_stratified_kernel(const float *__restrict__ rays, const float *__restrict__ prand, DrawKey key,
                                         int n_rays, int S, int use_disp, float perturb,
                                         float *__restrict__ z_out) {
    const int64_t total = (int64_t)n_rays * S;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(idx / S), i = (int)(idx % S);
        const float near = rays[(int64_t)r * 8 + 6], far = rays[(int64_t)r * 8 + 7];
        float z = z_at(i, S, near, far, use_disp);
        if (perturb > 0.f) {
            // rendering.py:215-222
            float lower = z, upper = z;
            if (i > 0) lower = __fmul_rn(0.5f, __fadd_rn(z_at(i - 1, S, near, far, use_disp), z));
            if (i < S - 1) upper = __fmul_rn(0.5f, __fadd_rn(z, z_at(i + 1, S, near, far, use_disp)));
            const float pr = __fmul_rn(perturb, prand ? prand[idx] : draw_one(key, idx));      // :221 torch.rand
            z = __fadd_rn(lower, __fmul_rn(__fsub_rn(upper, lower), pr));
        }
        z_out[idx] = z;
    }
}

// ---------------------------------------------------------------------------
// nerfmi_render_draws: the draws written to memory (one launch for all four segments)
// ---------------------------------------------------------------------------
struct DrawSegs {
    float *out[4];
    long long n[4];          // floats in the segment (0 = skipped); segments 0, 2 uniform, 1, 3 normal
    long long quad0[5];      // first quad of each segment, total at [4]
};

__global__ void render_draws_kernel(DrawSegs G, unsigned long long seed, unsigned long long offset) {
    for (long long qd = (long long)blockIdx.x * blockDim.x + threadIdx.x; qd < G.quad0[4];
         qd += (long long)gridDim.x * blockDim.x) {
        int seg = 0;
#pragma unroll
        for (int k = 1; k < 4; ++k)
            if (qd >= G.quad0[k]) seg = k;
        const long long i = qd - G.quad0[seg];
        const DrawKey key = {seed, offset, seg, 1};
        float v[4];
        draw_quad(key, i, v);
        float *dst = G.out[seg] + 4 * i;
        const long long left = G.n[seg] - 4 * i;
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (t < left) dst[t] = v[t];
    }
}

// ---------------------------------------------------------------------------
// a5  Embedding.forward (models/nerf.py:21-38)
// ---------------------------------------------------------------------------
__global__ void embed_kernel(const float *__restrict__ x, int64_t n, int n_freqs, float *__restrict__ out) {
    const int C = 3 * (2 * n_freqs + 1);
    const int64_t total = n * C;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = idx / C;
        const int c = (int)(idx % C);
        float v;
        if (c < 3) {
            v = x[row * 3 + c];
        } else {
            const int m = c - 3, f = m / 6, s = (m % 6) / 3, d = m % 3;
            const float arg = __fmul_rn(x[row * 3 + d], (float)(1 << f));  // exact (power of two)
            v = s ? cosf(arg) : sinf(arg);
        }
        out[idx] = v;
    }
}

// ---------------------------------------------------------------------------
// a8  compositing (models/rendering.py:162-190): one wave per ray,
//     lane l owns samples [l*SPL, l*SPL+SPL).
// ---------------------------------------------------------------------------
template <int SPL>
struct RaySamples {
    float alpha[SPL], a[SPL], T[SPL], w[SPL], z[SPL], e[SPL], delta[SPL], sg[SPL];
    float c[SPL][3];
};

template <int SPL, bool SIGMA_ONLY>
__device__ __forceinline__ void composite_forward_ray(RaySamples<SPL> &R, const float *__restrict__ field,
                                                      const float *__restrict__ zrow,
                                                      const float *__restrict__ ray,
                                                      const float *__restrict__ noise_row, const DrawKey &key,
                                                      int64_t base, float noise_std, int P, int lane) {
    const float dn = ray_norm(ray[3], ray[4], ray[5]);
    double lp = 1.0;  // local running product
    double pl[SPL];
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
        const int s = lane * SPL + j;
        const bool ok = s < P;
        const int sc = ok ? s : P - 1;
        float sigma;
        if (SIGMA_ONLY) {
            sigma = field[sc];
            R.c[j][0] = R.c[j][1] = R.c[j][2] = 0.f;
        } else {
            const float4 f = reinterpret_cast<const float4 *>(field)[sc];
            R.c[j][0] = f.x; R.c[j][1] = f.y; R.c[j][2] = f.z;
            sigma = f.w;
        }
        const float z0 = zrow[sc];
        float delta = (sc < P - 1) ? __fsub_rn(zrow[sc + 1], z0) : 1e10f;   // :162-164
        delta = __fmul_rn(delta, dn);                                        // :168
        float sg = sigma;
        if (noise_row) sg = __fadd_rn(sigma, __fmul_rn(noise_row[sc], noise_std));   // :170,173
        else if (key.on) sg = __fadd_rn(sigma, __fmul_rn(draw_one(key, base + sc), noise_std));   // torch.randn in place
        // exp evaluated in fp64 and rounded once: the correctly rounded fp32 value, which is what
        // the oracle specifies (torch's SLEEF expf is within 1 ulp of it)
        const float e = (float)exp((double)(-__fmul_rn(delta, fmaxf(sg, 0.f))));
        const float alpha = ok ? __fsub_rn(1.0f, e) : 0.f;                   // :173
        const float a = ok ? __fadd_rn(__fsub_rn(1.0f, alpha), 1e-10f) : 1.0f;  // :175
        R.alpha[j] = alpha; R.a[j] = a; R.z[j] = z0; R.e[j] = e; R.delta[j] = delta; R.sg[j] = sg;
        pl[j] = lp;
        lp *= (double)a;
    }
    // exclusive product over lanes, fp64 (torch.cumprod accumulates in fp64)
    const double incl = wave_incl_prod_d(lp, lane);
    double excl = shfl_up_d(incl, 1);
    if (lane == 0) excl = 1.0;
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
        R.T[j] = (float)(excl * pl[j]);
        R.w[j] = __fmul_rn(R.alpha[j], R.T[j]);                              // :176-177
    }
}

template <int SPL, bool SIGMA_ONLY>
__global__ void __launch_bounds__(64)
composite_kernel(const float *__restrict__ field, const float *__restrict__ z, const float *__restrict__ rays,
                 const float *__restrict__ noise, DrawKey key, float noise_std, int n_rays, int P, int white_back,
                 float *__restrict__ weights_out, float *__restrict__ rgb_out, float *__restrict__ depth_out,
                 float *__restrict__ opacity_out) {
    const int lane = threadIdx.x;
    for (int r = blockIdx.x; r < n_rays; r += gridDim.x) {
        RaySamples<SPL> R;
        const int64_t base = (int64_t)r * P;
        composite_forward_ray<SPL, SIGMA_ONLY>(R, field + base * (SIGMA_ONLY ? 1 : 4), z + base, rays + (int64_t)r * 8,
                                               noise ? noise + base : nullptr, key, base, noise_std, P, lane);
        double so = 0, sr = 0, sgc = 0, sb = 0, sd = 0;
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            const int s = lane * SPL + j;
            if (s < P) {
                if (weights_out) weights_out[base + s] = R.w[j];
                so += (double)R.w[j];
                if (!SIGMA_ONLY) {
                    sr += (double)__fmul_rn(R.w[j], R.c[j][0]);
                    sgc += (double)__fmul_rn(R.w[j], R.c[j][1]);
                    sb += (double)__fmul_rn(R.w[j], R.c[j][2]);
                    sd += (double)__fmul_rn(R.w[j], R.z[j]);
                }
            }
        }
        so = wave_sum_d(so);
        if (!SIGMA_ONLY) { sr = wave_sum_d(sr); sgc = wave_sum_d(sgc); sb = wave_sum_d(sb); sd = wave_sum_d(sd); }
        if (lane == 0) {
            const float op = (float)so;
            if (opacity_out) opacity_out[r] = op;
            if (!SIGMA_ONLY) {
                float c0 = (float)sr, c1 = (float)sgc, c2 = (float)sb;
                if (white_back) {                                            // :187-188
                    c0 = __fsub_rn(__fadd_rn(c0, 1.0f), op);
                    c1 = __fsub_rn(__fadd_rn(c1, 1.0f), op);
                    c2 = __fsub_rn(__fadd_rn(c2, 1.0f), op);
                }
                if (rgb_out) { rgb_out[r * 3 + 0] = c0; rgb_out[r * 3 + 1] = c1; rgb_out[r * 3 + 2] = c2; }
                if (depth_out) depth_out[r] = (float)sd;
            }
        }
    }
}

// backward (SURVEY section 8a contract; oracle composite_backward)
template <int SPL>
__global__ void __launch_bounds__(64)
composite_backward_kernel(const float *__restrict__ field, const float *__restrict__ z,
                          const float *__restrict__ rays, const float *__restrict__ noise, DrawKey key, float noise_std,
                          int n_rays, int P, int white_back, const float *__restrict__ g_rgb,
                          const float *__restrict__ g_depth, const float *__restrict__ g_opacity,
                          float *__restrict__ grad_field) {
    const int lane = threadIdx.x;
    for (int r = blockIdx.x; r < n_rays; r += gridDim.x) {
        RaySamples<SPL> R;
        const int64_t base = (int64_t)r * P;
        composite_forward_ray<SPL, false>(R, field + base * 4, z + base, rays + (int64_t)r * 8,
                                          noise ? noise + base : nullptr, key, base, noise_std, P, lane);
        const double gr = g_rgb ? (double)g_rgb[r * 3 + 0] : 0.0, gg = g_rgb ? (double)g_rgb[r * 3 + 1] : 0.0,
                     gb = g_rgb ? (double)g_rgb[r * 3 + 2] : 0.0;
        const double gd = g_depth ? (double)g_depth[r] : 0.0, go = g_opacity ? (double)g_opacity[r] : 0.0;
        const double wbterm = white_back ? (gr + gg + gb) : 0.0;
        double v[SPL], wv_incl[SPL];
        double lsum = 0;
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            const int s = lane * SPL + j;
            v[j] = (double)R.c[j][0] * gr + (double)R.c[j][1] * gg + (double)R.c[j][2] * gb + (double)R.z[j] * gd + go - wbterm;
            const double wv = (s < P) ? (double)R.w[j] * v[j] : 0.0;
            lsum += wv;
            wv_incl[j] = lsum;  // local inclusive
        }
        const double incl = wave_incl_sum_d(lsum, lane);
        const double total = __shfl(incl, WAVE - 1, WAVE);
        const double excl = incl - lsum;
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            const int s = lane * SPL + j;
            if (s < P) {
                const double suffix = total - (excl + wv_incl[j]);           // sum_{k>s} w_k v_k
                const double d_alpha = (double)R.T[j] * v[j] - suffix / (double)R.a[j];
                const double d_s = (R.sg[j] > 0.f) ? d_alpha * (double)R.delta[j] * (double)R.e[j] : 0.0;
                float4 g;
                g.x = (float)((double)R.w[j] * gr);
                g.y = (float)((double)R.w[j] * gg);
                g.z = (float)((double)R.w[j] * gb);
                g.w = (float)d_s;
                reinterpret_cast<float4 *>(grad_field)[base + s] = g;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// a3  sample_pdf (models/rendering.py:22-67).  One wave (= one workgroup) per
// ray; cdf and bins staged in LDS; per-lane binary search.
// LDS layout per workgroup: cdf[nb] | bins[nb] | sort[npad]
// ---------------------------------------------------------------------------
__device__ __forceinline__ void build_cdf_lds(float *cdf, const float *__restrict__ wrow, int nw, int wstride,
                                              int lane) {
    // rendering.py:36-40.  cdf[1+k] temporarily holds w+eps, then pdf, then the prefix.
    double ls = 0;
    for (int k = lane; k < nw; k += WAVE) {
        const float w = __fadd_rn(wrow[(int64_t)k * wstride], 1e-5f);
        cdf[1 + k] = w;
        ls += (double)w;
    }
    const float tot = (float)wave_sum_d(ls);
    __syncthreads();
    for (int k = lane; k < nw; k += WAVE) cdf[1 + k] = __fdiv_rn(cdf[1 + k], tot);
    __syncthreads();
    const int chunk = (nw + WAVE - 1) / WAVE;
    const int k0 = lane * chunk, k1 = min(k0 + chunk, nw);
    double loc = 0;
    for (int k = k0; k < k1; ++k) loc += (double)cdf[1 + k];
    const double incl = wave_incl_sum_d(loc, lane);
    double run = incl - loc;
    __syncthreads();
    for (int k = k0; k < k1; ++k) {
        run += (double)cdf[1 + k];
        cdf[1 + k] = (float)run;
    }
    if (lane == 0) cdf[0] = 0.f;
    __syncthreads();
}

__device__ __forceinline__ float search_lerp_one(const float *cdf, const float *bins, int nw, float u, int &inds) {
    // torch.searchsorted(cdf, u, right=True) = #{k: cdf[k] <= u} over nb = nw+1 entries
    int lo = 0, hi = nw + 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
    }
    inds = lo;
    const int below = max(lo - 1, 0), above = min(lo, nw);                  // :55-56
    const float cb = cdf[below], ca = cdf[above], bb = bins[below], ba = bins[above];
    float denom = __fsub_rn(ca, cb);
    if (denom < 1e-5f) denom = 1.0f;                                         // :63
    const float t = __fdiv_rn(__fsub_rn(u, cb), denom);
    return __fadd_rn(bb, __fmul_rn(t, __fsub_rn(ba, bb)));                   // :66
}

__global__ void __launch_bounds__(64)
sample_pdf_kernel(const float *__restrict__ bins_g, const float *__restrict__ weights, const float *__restrict__ cdf_in,
                  const float *__restrict__ u_g, int n_rays, int nw, int F, float *__restrict__ cdf_out,
                  int64_t *__restrict__ inds_out, float *__restrict__ samples_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int nb = nw + 1;
    float *cdf = lds, *bins = lds + nb;
    const int lane = threadIdx.x;
    for (int r = blockIdx.x; r < n_rays; r += gridDim.x) {
        __syncthreads();
        for (int k = lane; k < nb; k += WAVE) bins[k] = bins_g[(int64_t)r * nb + k];
        if (cdf_in) {
            for (int k = lane; k < nb; k += WAVE) cdf[k] = cdf_in[(int64_t)r * nb + k];
            __syncthreads();
        } else {
            build_cdf_lds(cdf, weights + (int64_t)r * nw, nw, 1, lane);
        }
        if (cdf_out)
            for (int k = lane; k < nb; k += WAVE) cdf_out[(int64_t)r * nb + k] = cdf[k];
        for (int f = lane; f < F; f += WAVE) {
            const float u = u_g ? u_g[(int64_t)r * F + f] : linspace01(f, F);
            int inds;
            const float s = search_lerp_one(cdf, bins, nw, u, inds);
            samples_out[(int64_t)r * F + f] = s;
            if (inds_out) inds_out[(int64_t)r * F + f] = inds;
        }
    }
}

// in-LDS bitonic sort of npad (power of two) floats by one wave
__device__ __forceinline__ void bitonic_sort_lds(float *buf, int npad, int lane) {
    for (int k = 2; k <= npad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = lane; t < npad / 2; t += WAVE) {
                const int i = ((t / j) * 2 * j) + (t % j);     // lower index of the pair
                const int p = i + j;
                const bool up = ((i & k) == 0);
                const float a = buf[i], b = buf[p];
                if ((a > b) == up) { buf[i] = b; buf[p] = a; }
            }
            __syncthreads();
        }
    }
}

__global__ void __launch_bounds__(64)
merge_sorted_kernel(const float *__restrict__ za, const float *__restrict__ zb, int n_rays, int na, int nb_,
                    int npad, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x, n = na + nb_;
    for (int r = blockIdx.x; r < n_rays; r += gridDim.x) {
        __syncthreads();
        for (int k = lane; k < npad; k += WAVE)
            lds[k] = k < na ? za[(int64_t)r * na + k] : (k < n ? zb[(int64_t)r * nb_ + (k - na)] : INFINITY);
        __syncthreads();
        bitonic_sort_lds(lds, npad, lane);
        for (int k = lane; k < n; k += WAVE) out[(int64_t)r * n + k] = lds[k];
    }
}

// rendering.py:242-247 fused: z_mid -> sample_pdf(z_mid, w[:,1:-1]) -> sort(cat[z, z_new])
__global__ void __launch_bounds__(64)
importance_resample_kernel(const float *__restrict__ zc, const float *__restrict__ wc, const float *__restrict__ u_g,
                           DrawKey key, int n_rays, int S, int F, int npad, float *__restrict__ z_new_out,
                           float *__restrict__ z_fine_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int nw = S - 2, nb = S - 1, n = S + F;
    float *cdf = lds, *bins = lds + nb, *srt = lds + 2 * nb;
    const int lane = threadIdx.x;
    for (int r = blockIdx.x; r < n_rays; r += gridDim.x) {
        __syncthreads();
        const float *zr = zc + (int64_t)r * S;
        for (int k = lane; k < nb; k += WAVE) bins[k] = __fmul_rn(0.5f, __fadd_rn(zr[k], zr[k + 1]));   // :242
        for (int k = lane; k < S; k += WAVE) srt[k] = zr[k];
        for (int k = S + F + lane; k < npad; k += WAVE) srt[k] = INFINITY;
        build_cdf_lds(cdf, wc + (int64_t)r * S + 1, nw, 1, lane);
        for (int f = lane; f < F; f += WAVE) {
            const float u = u_g ? u_g[(int64_t)r * F + f] : (key.on ? draw_one(key, (int64_t)r * F + f) : linspace01(f, F));
            int inds;
            const float s = search_lerp_one(cdf, bins, nw, u, inds);
            srt[S + f] = s;
            if (z_new_out) z_new_out[(int64_t)r * F + f] = s;
        }
        __syncthreads();
        // the coarse run is sorted when it comes from the stratified sampler (always, in render_rays); checked, because
        // the rank merge below relies on it and the reference sorts the concatenation whatever its input
        bool sorted_run = true;
        for (int k = lane; k + 1 < S; k += WAVE) sorted_run = sorted_run && (srt[k] <= srt[k + 1]);
        if (F <= WAVE && __all(sorted_run)) {
            // sort(cat[z_coarse (already sorted), z_new]) without sorting 128 keys through LDS: the F <= 64 new depths
            // are sorted in registers, one per lane (21 compare-exchange steps by cross-lane shuffles), then every
            // depth goes straight to its rank = own index + number of depths of the OTHER run in front of it
            // (binary searches; "<=" on one side and "<" on the other make the ranks of equal values distinct).
            float v = lane < F ? srt[S + lane] : INFINITY;
            for (int k = 2; k <= WAVE; k <<= 1)
                for (int j = k >> 1; j > 0; j >>= 1) {
                    const float o = __shfl_xor(v, j, WAVE);
                    const bool keep_min = ((lane & j) == 0) == ((lane & k) == 0);
                    v = keep_min ? fminf(v, o) : fmaxf(v, o);
                }
            __syncthreads();
            if (lane < F) srt[S + lane] = v;                     // the sorted new depths replace the unsorted ones
            __syncthreads();
            auto new_at = [&](int i) { return srt[S + i]; };
            if (lane < F) {                                      // rank of this new depth: coarse depths <= it
                int lo = 0, hi = S;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (srt[mid] <= v) lo = mid + 1; else hi = mid;
                }
                z_fine_out[(int64_t)r * n + lane + lo] = v;
            }
            for (int k = lane; k < S; k += WAVE) {               // rank of a coarse depth: new depths < it
                const float zk = srt[k];
                int lo = 0, hi = F;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (new_at(mid) < zk) lo = mid + 1; else hi = mid;
                }
                z_fine_out[(int64_t)r * n + k + lo] = zk;
            }
        } else {
            bitonic_sort_lds(srt, npad, lane);
            for (int k = lane; k < n; k += WAVE) z_fine_out[(int64_t)r * n + k] = srt[k];
        }
    }
}

// torchsearchsorted/src/cuda/searchsorted_cuda_kernel.cu:84-107 semantics
// (== numpy.searchsorted): one thread per (row, col) of v.
__global__ void searchsorted_kernel(const float *__restrict__ a, const float *__restrict__ v, int nrow_a, int nrow_v,
                                    int ncol_a, int ncol_v, int side_left, int64_t *__restrict__ out) {
    const int nrow = max(nrow_a, nrow_v);
    const int64_t total = (int64_t)nrow * ncol_v;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int row = (int)(idx / ncol_v), col = (int)(idx % ncol_v);
        const float *ar = a + (int64_t)(nrow_a == 1 ? 0 : row) * ncol_a;
        const float val = v[(int64_t)(nrow_v == 1 ? 0 : row) * ncol_v + col];
        int lo = 0, hi = ncol_a;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            const float am = ar[mid];
            const bool go_right = side_left ? (am < val) : (am <= val);
            if (go_right) lo = mid + 1; else hi = mid;
        }
        out[idx] = lo;
    }
}

static inline int next_pow2(int n) { int p = 1; while (p < n) p <<= 1; return p; }
static inline int grid_for(int64_t total, int block) {
    int64_t g = (total + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace nerfmi

using namespace nerfmi;

extern "C" {

int nerfmi_version(void) { return 100; }
const char *nerfmi_last_error(void) { return g_err; }

static int sample_stratified_impl(const char *who, const float *rays, const float *perturb_rand, DrawKey key, int n_rays,
                                  int n_samples, int use_disp, float perturb, float *z_out, nerfmi_stream_t stream) {
    NERFMI_ENTER();
    NERFMI_REQUIRE(n_rays >= 0 && n_samples >= 1, "%s: bad sizes n_rays=%d n_samples=%d", who, n_rays, n_samples);
    if (n_rays == 0) return NERFMI_OK;
    NERFMI_REQUIRE(rays && z_out, "%s: null pointer", who);
    NERFMI_REQUIRE(!(perturb > 0.f) || perturb_rand || key.on, "%s: perturb>0 needs perturb_rand", who);
    const int64_t total = (int64_t)n_rays * n_samples;
    hipLaunchKernelGGL(sample_stratified_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, rays,
                       perturb_rand, key, n_rays, n_samples, use_disp, perturb, z_out);
    return check_launch(who);
}

int nerfmi_sample_stratified(const float *rays, const float *perturb_rand, int n_rays, int n_samples, int use_disp,
                             float perturb, float *z_out, nerfmi_stream_t stream) {
    return sample_stratified_impl("sample_stratified", rays, perturb_rand, DrawKey{0, 0, 0, 0}, n_rays, n_samples, use_disp,
                                  perturb, z_out, stream);
}

int nerfmi_sample_stratified_philox(const float *rays, uint64_t seed, uint64_t offset, int n_rays, int n_samples,
                                    int use_disp, float perturb, float *z_out, nerfmi_stream_t stream) {
    return sample_stratified_impl("sample_stratified_philox", rays, nullptr, DrawKey{seed, offset, 0, 1}, n_rays, n_samples,
                                  use_disp, perturb, z_out, stream);
}

int nerfmi_render_draws(uint64_t seed, uint64_t offset, int64_t n_perturb, float *perturb_rand, int64_t n_noise_coarse,
                        float *noise_coarse, int64_t n_u, float *u, int64_t n_noise_fine, float *noise_fine,
                        nerfmi_stream_t stream) {
    NERFMI_ENTER();
    DrawSegs G;
    const int64_t n[4] = {n_perturb, n_noise_coarse, n_u, n_noise_fine};
    float *const o[4] = {perturb_rand, noise_coarse, u, noise_fine};
    long long q = 0;
    for (int k = 0; k < 4; ++k) {
        NERFMI_REQUIRE(n[k] >= 0 && (n[k] == 0 || o[k]), "render_draws: segment %d has a size but no buffer", k);
        G.out[k] = o[k];
        G.n[k] = n[k];
        G.quad0[k] = q;
        q += (n[k] + 3) / 4;
    }
    G.quad0[4] = q;
    if (q == 0) return NERFMI_OK;
    const long long blocks = (q + 255) / 256;
    hipLaunchKernelGGL(render_draws_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, (hipStream_t)stream,
                       G, (unsigned long long)seed, (unsigned long long)offset);
    return check_launch("render_draws");
}

int nerfmi_embed(const float *x, int64_t n, int n_freqs, float *out, nerfmi_stream_t stream) {
    NERFMI_ENTER();
    NERFMI_REQUIRE(n >= 0 && n_freqs >= 0 && n_freqs <= 24, "embed: bad sizes");
    if (n == 0) return NERFMI_OK;
    NERFMI_REQUIRE(x && out, "embed: null pointer");
    const int64_t total = n * 3 * (2 * n_freqs + 1);
    hipLaunchKernelGGL(embed_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, n, n_freqs, out);
    return check_launch("embed");
}

#define NERFMI_SPL_DISPATCH(P, CALL)                                    \
    do {                                                                \
        const int spl_ = ((P) + 63) / 64;                               \
        if (spl_ <= 1) { CALL(1); }                                     \
        else if (spl_ <= 2) { CALL(2); }                                \
        else if (spl_ <= 4) { CALL(4); }                                \
        else if (spl_ <= 8) { CALL(8); }                                \
        else { CALL(16); }                                              \
    } while (0)

static int composite_impl(const char *who, const float *field, int sigma_only, const float *z, const float *rays,
                          const float *noise, DrawKey key, float noise_std, int n_rays, int n_per_ray, int white_back,
                          float *weights_out, float *rgb_out, float *depth_out, float *opacity_out, nerfmi_stream_t stream) {
    NERFMI_ENTER();
    NERFMI_REQUIRE(n_rays >= 0 && n_per_ray >= 1 && n_per_ray <= 1024, "%s: n_per_ray=%d out of [1,1024]", who, n_per_ray);
    if (n_rays == 0) return NERFMI_OK;
    NERFMI_REQUIRE(field && z && rays, "%s: null input", who);
    const dim3 grid(n_rays < 65536 ? n_rays : 65536), block(64);
    hipStream_t st = (hipStream_t)stream;
    if (noise_std == 0.f) { noise = nullptr; key.on = 0; }
#define CALL(SPL)                                                                                                   \
    if (sigma_only)                                                                                                 \
        hipLaunchKernelGGL((composite_kernel<SPL, true>), grid, block, 0, st, field, z, rays, noise, key, noise_std, \
                           n_rays, n_per_ray, white_back, weights_out, rgb_out, depth_out, opacity_out);            \
    else                                                                                                            \
        hipLaunchKernelGGL((composite_kernel<SPL, false>), grid, block, 0, st, field, z, rays, noise, key, noise_std, \
                           n_rays, n_per_ray, white_back, weights_out, rgb_out, depth_out, opacity_out)
    NERFMI_SPL_DISPATCH(n_per_ray, CALL);
#undef CALL
    return check_launch(who);
}

int nerfmi_composite(const float *field, int sigma_only, const float *z, const float *rays, const float *noise,
                     float noise_std, int n_rays, int n_per_ray, int white_back, float *weights_out, float *rgb_out,
                     float *depth_out, float *opacity_out, nerfmi_stream_t stream) {
    return composite_impl("composite", field, sigma_only, z, rays, noise, DrawKey{0, 0, 0, 0}, noise_std, n_rays, n_per_ray,
                          white_back, weights_out, rgb_out, depth_out, opacity_out, stream);
}

int nerfmi_composite_philox(const float *field, int sigma_only, const float *z, const float *rays, uint64_t seed,
                            uint64_t offset, int segment, float noise_std, int n_rays, int n_per_ray, int white_back,
                            float *weights_out, float *rgb_out, float *depth_out, float *opacity_out,
                            nerfmi_stream_t stream) {
    NERFMI_REQUIRE(segment == 1 || segment == 3, "composite_philox: segment must be 1 (coarse) or 3 (fine)");
    return composite_impl("composite_philox", field, sigma_only, z, rays, nullptr, DrawKey{seed, offset, segment, 1},
                          noise_std, n_rays, n_per_ray, white_back, weights_out, rgb_out, depth_out, opacity_out, stream);
}

static int composite_backward_impl(const char *who, const float *field, const float *z, const float *rays,
                                   const float *noise, DrawKey key, float noise_std, int n_rays, int n_per_ray,
                                   int white_back, const float *g_rgb, const float *g_depth, const float *g_opacity,
                                   float *grad_field, nerfmi_stream_t stream) {
    NERFMI_ENTER();
    NERFMI_REQUIRE(n_rays >= 0 && n_per_ray >= 1 && n_per_ray <= 1024, "%s: n_per_ray=%d out of [1,1024]", who, n_per_ray);
    if (n_rays == 0) return NERFMI_OK;
    NERFMI_REQUIRE(field && z && rays && grad_field, "%s: null pointer", who);
    const dim3 grid(n_rays < 65536 ? n_rays : 65536), block(64);
    hipStream_t st = (hipStream_t)stream;
    if (noise_std == 0.f) { noise = nullptr; key.on = 0; }
#define CALL(SPL)                                                                                                 \
    hipLaunchKernelGGL((composite_backward_kernel<SPL>), grid, block, 0, st, field, z, rays, noise, key, noise_std, \
                       n_rays, n_per_ray, white_back, g_rgb, g_depth, g_opacity, grad_field)
    NERFMI_SPL_DISPATCH(n_per_ray, CALL);
#undef CALL
    return check_launch(who);
}

int nerfmi_composite_backward(const float *field, const float *z, const float *rays, const float *noise,
                              float noise_std, int n_rays, int n_per_ray, int white_back, const float *g_rgb,
                              const float *g_depth, const float *g_opacity, float *grad_field,
                              nerfmi_stream_t stream) {
    return composite_backward_impl("composite_backward", field, z, rays, noise, DrawKey{0, 0, 0, 0}, noise_std, n_rays,
                                   n_per_ray, white_back, g_rgb, g_depth, g_opacity, grad_field, stream);
}

int nerfmi_composite_backward_philox(const float *field, const float *z, const float *rays, uint64_t seed, uint64_t offset,
                                     int segment, float noise_std, int n_rays, int n_per_ray, int white_back,
                                     const float *g_rgb, const float *g_depth, const float *g_opacity, float *grad_field,
                                     nerfmi_stream_t stream) {
    NERFMI_REQUIRE(segment == 1 || segment == 3, "composite_backward_philox: segment must be 1 (coarse) or 3 (fine)");
    return composite_backward_impl("composite_backward_philox", field, z, rays, nullptr, DrawKey{seed, offset, segment, 1},
                                   noise_std, n_rays, n_per_ray, white_back, g_rgb, g_depth, g_opacity, grad_field, stream);
}

static int sample_pdf_common(const float *bins, const float *weights, const float *cdf_in, const float *u, int n_rays,
                             int n_weights, int n_importance, float *cdf_out, int64_t *inds_out, float *samples_out,
                             nerfmi_stream_t stream, const char *who) {
    NERFMI_REQUIRE(n_rays >= 0 && n_weights >= 1 && n_weights <= 8190 && n_importance >= 0, "%s: bad sizes", who);
    if (n_rays == 0 || n_importance == 0) return NERFMI_OK;
    NERFMI_REQUIRE(bins && (weights || cdf_in) && samples_out, "%s: null pointer", who);
    const size_t lds = sizeof(float) * 2 * (n_weights + 1);
    hipLaunchKernelGGL(sample_pdf_kernel, dim3(n_rays < 65536 ? n_rays : 65536), dim3(64), lds, (hipStream_t)stream,
                       bins, weights, cdf_in, u, n_rays, n_weights, n_importance, cdf_out, inds_out, samples_out);
    return check_launch(who);
}

int nerfmi_sample_pdf(const float *bins, const float *weights, const float *u, int n_rays, int n_weights,
                      int n_importance, float *cdf_out, int64_t *inds_out, float *samples_out,
                      nerfmi_stream_t stream) {
    return sample_pdf_common(bins, weights, nullptr, u, n_rays, n_weights, n_importance, cdf_out, inds_out,
                             samples_out, stream, "sample_pdf");
}

int nerfmi_search_lerp(const float *bins, const float *cdf, const float *u, int n_rays, int n_weights,
                       int n_importance, int64_t *inds_out, float *samples_out, nerfmi_stream_t stream) {
    NERFMI_REQUIRE(n_rays == 0 || n_importance == 0 || u, "search_lerp: u is required");
    return sample_pdf_common(bins, nullptr, cdf, u, n_rays, n_weights, n_importance, nullptr, inds_out, samples_out,
                             stream, "search_lerp");
}

int nerfmi_searchsorted(const float *a, const float *v, int nrow_a, int nrow_v, int ncol_a, int ncol_v, int side_left,
                        int64_t *out, nerfmi_stream_t stream) {
    NERFMI_ENTER();
    NERFMI_REQUIRE(nrow_a >= 1 && nrow_v >= 1 && ncol_a >= 0 && ncol_v >= 0, "searchsorted: bad sizes");
    NERFMI_REQUIRE(nrow_a == nrow_v || nrow_a == 1 || nrow_v == 1,
                   "searchsorted: row counts %d vs %d do not broadcast", nrow_a, nrow_v);
    if (ncol_v == 0) return NERFMI_OK;
    NERFMI_REQUIRE(a && v && out, "searchsorted: null pointer");
    const int64_t total = (int64_t)(nrow_a > nrow_v ? nrow_a : nrow_v) * ncol_v;
    hipLaunchKernelGGL(searchsorted_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, a, v, nrow_a,
                       nrow_v, ncol_a, ncol_v, side_left, out);
    return check_launch("searchsorted");
}

int nerfmi_merge_sorted(const float *za, const float *zb, int n_rays, int na, int nb, float *out,
                        nerfmi_stream_t stream) {
    NERFMI_ENTER();
    NERFMI_REQUIRE(n_rays >= 0 && na >= 0 && nb >= 0 && na + nb <= 16384, "merge_sorted: bad sizes");
    if (n_rays == 0 || na + nb == 0) return NERFMI_OK;
    NERFMI_REQUIRE((za || na == 0) && (zb || nb == 0) && out, "merge_sorted: null pointer");
    const int npad = next_pow2(na + nb < 2 ? 2 : na + nb);
    hipLaunchKernelGGL(merge_sorted_kernel, dim3(n_rays < 65536 ? n_rays : 65536), dim3(64), sizeof(float) * npad,
                       (hipStream_t)stream, za, zb, n_rays, na, nb, npad, out);
    return check_launch("merge_sorted");
}

static int importance_resample_impl(const char *who, const float *z_coarse, const float *weights_coarse, const float *u,
                                    DrawKey key, int n_rays, int n_samples, int n_importance, float *z_new_out,
                                    float *z_fine_out, nerfmi_stream_t stream) {
    NERFMI_ENTER();
    NERFMI_REQUIRE(n_rays >= 0 && n_samples >= 3 && n_importance >= 1 && n_samples + n_importance <= 8192,
                   "%s: bad sizes S=%d F=%d (need S>=3, F>=1, S+F<=8192)", who, n_samples, n_importance);
    if (n_rays == 0) return NERFMI_OK;
    NERFMI_REQUIRE(z_coarse && weights_coarse && z_fine_out, "%s: null pointer", who);
    const int npad = next_pow2(n_samples + n_importance);
    const size_t lds = sizeof(float) * (2 * (n_samples - 1) + npad);
    hipLaunchKernelGGL(importance_resample_kernel, dim3(n_rays < 65536 ? n_rays : 65536), dim3(64), lds,
                       (hipStream_t)stream, z_coarse, weights_coarse, u, key, n_rays, n_samples, n_importance, npad,
                       z_new_out, z_fine_out);
    return check_launch(who);
}

int nerfmi_importance_resample(const float *z_coarse, const float *weights_coarse, const float *u, int n_rays,
                               int n_samples, int n_importance, float *z_new_out, float *z_fine_out,
                               nerfmi_stream_t stream) {
    return importance_resample_impl("importance_resample", z_coarse, weights_coarse, u, DrawKey{0, 0, 0, 0}, n_rays, n_samples,
                                    n_importance, z_new_out, z_fine_out, stream);
}

int nerfmi_importance_resample_philox(const float *z_coarse, const float *weights_coarse, uint64_t seed, uint64_t offset,
                                      int n_rays, int n_samples, int n_importance, float *z_new_out, float *z_fine_out,
                                      nerfmi_stream_t stream) {
    return importance_resample_impl("importance_resample_philox", z_coarse, weights_coarse, nullptr, DrawKey{seed, offset, 2, 1},
                                    n_rays, n_samples, n_importance, z_new_out, z_fine_out, stream);
}

}  // extern "C"
