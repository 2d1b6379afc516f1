// Counter-based random draws shared by the render_rays kernels (rays.hip) and the EG3D sampling kernels (eg3d.hip).
#pragma once
#include "common.h"

namespace nerfmi {

// ---------------------------------------------------------------------------
// The four random draws of one render_rays call (rand(N,S) [rendering.py:221], randn(N,S) [:170],
// rand(N,F) [:47], randn(N,S+F)) from a counter-based generator instead of four aten distribution launches:
// Philox4x32-10 (Salmon et al., SC'11; Random123 known-answer vectors in tests/), key = seed, counter =
// (quad index, segment, offset lo, offset hi); one 128-bit block = four floats of one segment.
// uniform: (x >> 8) * 2^-24 in [0,1) (24 bits, like torch.rand); normal: Box-Muller on two such pairs.
// Perf mode draws IN the consuming kernels (the *_philox entry points: no draw ever touches memory; the compositor's
// backward regenerates its forward's noise from the same key); nerfmi_render_draws materialises the same streams.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

// (seed, offset, segment) address one stream of draws; `on` = 0 means "no generator" (draws injected or not needed).
// Segments: 0 perturb_rand, 1 noise_coarse, 2 u, 3 noise_fine.  The SAME functions serve nerfmi_render_draws (draws
// written to memory) and the kernels that draw in place (sample_stratified / composite / composite_backward /
// importance_resample with a key): element e of a segment is the same float either way, bit for bit.
struct DrawKey {
    unsigned long long seed, offset;
    int seg, on;
};

// the four floats of quad `i` of the key's segment (uniform for segments 0, 2; Box-Muller normals for 1, 3)
__device__ __forceinline__ void draw_quad(const DrawKey &k, long long i, float (&v)[4]) {
    unsigned c[4] = {(unsigned)i, (unsigned)k.seg | ((unsigned)(i >> 32) << 2), (unsigned)k.offset, (unsigned)(k.offset >> 32)};
    philox4x32_10(c, (unsigned)k.seed, (unsigned)(k.seed >> 32));
    if (k.seg & 1) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float u1 = (float)((c[2 * h] >> 8) + 1u) * 5.9604644775390625e-8f;      // (0, 1]
            const float u2 = (float)(c[2 * h + 1] >> 8) * 5.9604644775390625e-8f;         // [0, 1)
            const float r = sqrtf(-2.0f * logf(u1));
            float sn, cs;
            sincosf(6.283185307179586f * u2, &sn, &cs);
            v[2 * h] = r * cs;
            v[2 * h + 1] = r * sn;
        }
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = (float)(c[t] >> 8) * 5.9604644775390625e-8f;
    }
}
__device__ __forceinline__ float draw_one(const DrawKey &k, long long e) {
    float v[4];
    draw_quad(k, e >> 2, v);
    const int t = (int)(e & 3);
    return t == 0 ? v[0] : (t == 1 ? v[1] : (t == 2 ? v[2] : v[3]));
}

}  // namespace nerfmi
