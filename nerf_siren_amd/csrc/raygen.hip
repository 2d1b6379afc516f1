// The step right before render_rays: ray generation on the device.
//   datasets/ray_utils.py:5-24   get_ray_directions   (kornia.create_meshgrid(H, W, False) = pixel coords i in [0,W), j in [0,H))
//   datasets/ray_utils.py:27-50  get_rays             (rotate by c2w[:, :3], normalise, origin = c2w[:, 3])
//   datasets/ray_utils.py:53-93  get_ndc_rays
//   datasets/blender.py:60-69, datasets/llff.py:234-250   the (N, 8) [o, d, near, far] packing
// One thread per ray; pure streaming (R 8 B index / W 32 B per ray), so the 512 MB host ray buffer of a 100-view
// 400x400 scene and its per-step H2D copy are replaced by (c2w, focal, pixel index) -> rays in HBM.
// Arithmetic: fp32, one rounding per operation in the reference's order (oracle/nerf_oracle.py generate_rays);
// Python-float constants are formed in double and rounded once, as torch does for scalar operands.
#include "common.h"

namespace nerfmi {

struct NdcConst {
    float cw, ch;        // -1/(W/(2 focal)), -1/(H/(2 focal))
    float near;          // near plane of get_ndc_rays (1.0 in llff.py:237)
    float two_near;      // 2*near
};

__device__ __forceinline__ void camera_dir(int64_t pix, int H, int W, float half_w, float half_h, float focal,
                                           float &dx, float &dy, float &dz) {
    const float i = (float)(pix % W), j = (float)(pix / W);
    dx = __fdiv_rn(__fsub_rn(i, half_w), focal);                       // (i - W/2) / focal
    dy = __fdiv_rn(-__fsub_rn(j, half_h), focal);                      // -(j - H/2) / focal
    dz = -1.0f;
}

// rays_d = directions @ c2w[:, :3].T ; rays_d /= ||rays_d|| ; rays_o = c2w[:, 3]
__device__ __forceinline__ void world_ray(const float *__restrict__ c, float dx, float dy, float dz, float (&o)[3],
                                          float (&d)[3]) {
    float r[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
        r[k] = __fadd_rn(__fadd_rn(__fmul_rn(dx, c[4 * k + 0]), __fmul_rn(dy, c[4 * k + 1])), __fmul_rn(dz, c[4 * k + 2]));
    const float nrm = ray_norm(r[0], r[1], r[2]);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        d[k] = __fdiv_rn(r[k], nrm);
        o[k] = c[4 * k + 3];
    }
}

__device__ __forceinline__ void ndc_ray(const NdcConst &K, float (&o)[3], float (&d)[3]) {
    const float t = __fdiv_rn(-__fadd_rn(K.near, o[2]), d[2]);         // -(near + o_z) / d_z
#pragma unroll
    for (int k = 0; k < 3; ++k) o[k] = __fadd_rn(o[k], __fmul_rn(t, d[k]));
    const float ox_oz = __fdiv_rn(o[0], o[2]), oy_oz = __fdiv_rn(o[1], o[2]);
    const float o0 = __fmul_rn(K.cw, ox_oz), o1 = __fmul_rn(K.ch, oy_oz);
    // 1. + 2.*near / o_z: a Python scalar divided by a tensor is Tensor.__rtruediv__ = reciprocal(o_z) * scalar
    const float o2 = __fadd_rn(1.0f, __fmul_rn(__fdiv_rn(1.0f, o[2]), K.two_near));
    const float d0 = __fmul_rn(K.cw, __fsub_rn(__fdiv_rn(d[0], d[2]), ox_oz));
    const float d1 = __fmul_rn(K.ch, __fsub_rn(__fdiv_rn(d[1], d[2]), oy_oz));
    const float d2 = __fsub_rn(1.0f, o2);
    o[0] = o0; o[1] = o1; o[2] = o2;
    d[0] = d0; d[1] = d1; d[2] = d2;
}

__global__ void __launch_bounds__(256)
generate_rays_kernel(const float *__restrict__ c2w, int n_images, int H, int W, float half_w, float half_h, float focal,
                     const int64_t *__restrict__ pixel_index, int64_t n_rays, int ndc, NdcConst K, float near, float far,
                     float *__restrict__ rays) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    const int64_t hw = (int64_t)H * W;
    int64_t idx = pixel_index ? pixel_index[r] : r;
    const int64_t total = hw * n_images;
    idx = idx < 0 ? 0 : (idx >= total ? total - 1 : idx);              // indices are validated on the host when given from there
    const int64_t img = idx / hw, pix = idx - img * hw;
    float dx, dy, dz, o[3], d[3];
    camera_dir(pix, H, W, half_w, half_h, focal, dx, dy, dz);
    world_ray(c2w + img * 12, dx, dy, dz, o, d);
    if (ndc) ndc_ray(K, o, d);
    float4 a, b;
    a.x = o[0]; a.y = o[1]; a.z = o[2]; a.w = d[0];
    b.x = d[1]; b.y = d[2]; b.z = near; b.w = far;
    reinterpret_cast<float4 *>(rays)[2 * r] = a;
    reinterpret_cast<float4 *>(rays)[2 * r + 1] = b;
}

__global__ void __launch_bounds__(256)
ray_directions_kernel(int H, int W, float half_w, float half_h, float focal, float *__restrict__ dirs) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (int64_t)H * W) return;
    float dx, dy, dz;
    camera_dir(p, H, W, half_w, half_h, focal, dx, dy, dz);
    dirs[3 * p] = dx; dirs[3 * p + 1] = dy; dirs[3 * p + 2] = dz;
}

__global__ void __launch_bounds__(256)
get_rays_kernel(const float *__restrict__ dirs, const float *__restrict__ c2w, int64_t n, float *__restrict__ o_out,
                float *__restrict__ d_out) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    float o[3], d[3];
    world_ray(c2w, dirs[3 * p], dirs[3 * p + 1], dirs[3 * p + 2], o, d);
#pragma unroll
    for (int k = 0; k < 3; ++k) { o_out[3 * p + k] = o[k]; d_out[3 * p + k] = d[k]; }
}

__global__ void __launch_bounds__(256)
ndc_rays_kernel(NdcConst K, const float *__restrict__ o_in, const float *__restrict__ d_in, int64_t n,
                float *__restrict__ o_out, float *__restrict__ d_out) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    float o[3] = {o_in[3 * p], o_in[3 * p + 1], o_in[3 * p + 2]}, d[3] = {d_in[3 * p], d_in[3 * p + 1], d_in[3 * p + 2]};
    ndc_ray(K, o, d);
#pragma unroll
    for (int k = 0; k < 3; ++k) { o_out[3 * p + k] = o[k]; d_out[3 * p + k] = d[k]; }
}

static NdcConst ndc_const(int H, int W, double focal, double near) {
    NdcConst K;
    K.cw = (float)(-1.0 / (W / (2.0 * focal)));
    K.ch = (float)(-1.0 / (H / (2.0 * focal)));
    K.near = (float)near;
    K.two_near = (float)(2.0 * near);
    return K;
}

// create_samples of extract_color_mesh_eg3d.py:72-94 (the N^3 query grid of the EG3D "neural volume"), op by op in
// fp32 as the torch CPU ops run it: column 2 = idx % N; column 1 = (float(idx) / N) % N; column 0 =
// ((float(idx) / N) / N) % N -- FLOAT divisions, so columns 1 and 0 are not integer voxel indices (restated as is) --
// then  s * voxel_size + origin  per column (origin index reversed, :88-90).  Correctly rounded divisions: torch's
// device division by a scalar multiplies by the reciprocal, which moves these values by up to 10 ulp.
__global__ void create_samples_kernel(int N, int64_t n_points, float voxel_size, float o0, float o1, float o2,
                                      float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_points) return;
    const float fN = (float)N;
    const float q = __fdiv_rn((float)i, fN);
    const float s2 = (float)(i % N);
    const float s1 = fmodf(q, fN);
    const float s0 = fmodf(__fdiv_rn(q, fN), fN);
    out[3 * i + 0] = __fadd_rn(__fmul_rn(s0, voxel_size), o2);
    out[3 * i + 1] = __fadd_rn(__fmul_rn(s1, voxel_size), o1);
    out[3 * i + 2] = __fadd_rn(__fmul_rn(s2, voxel_size), o0);
}

}  // namespace nerfmi

using namespace nerfmi;

extern "C" {

int nerfmi_ray_directions(int H, int W, double focal, float *dirs_out, nerfmi_stream_t stream) {
    NERFMI_ENTER();
    NERFMI_REQUIRE(H >= 1 && W >= 1 && focal > 0, "ray_directions: H, W >= 1 and focal > 0 required");
    NERFMI_REQUIRE(dirs_out, "ray_directions: null pointer");
    const int64_t n = (int64_t)H * W;
    hipLaunchKernelGGL(ray_directions_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, H, W,
                       (float)(W / 2.0), (float)(H / 2.0), (float)focal, dirs_out);
    return check_launch("ray_directions");
}

int nerfmi_get_rays(const float *directions, const float *c2w, int64_t n, float *rays_o_out, float *rays_d_out,
                    nerfmi_stream_t stream) {
    NERFMI_ENTER();
    NERFMI_REQUIRE(n >= 0, "get_rays: n must be >= 0");
    if (n == 0) return NERFMI_OK;
    NERFMI_REQUIRE(directions && c2w && rays_o_out && rays_d_out, "get_rays: null pointer");
    hipLaunchKernelGGL(get_rays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, directions, c2w,
                       n, rays_o_out, rays_d_out);
    return check_launch("get_rays");
}

int nerfmi_ndc_rays(int H, int W, double focal, double near, const float *rays_o, const float *rays_d, int64_t n,
                    float *rays_o_out, float *rays_d_out, nerfmi_stream_t stream) {
    NERFMI_ENTER();
    NERFMI_REQUIRE(H >= 1 && W >= 1 && focal > 0 && n >= 0, "ndc_rays: bad sizes");
    if (n == 0) return NERFMI_OK;
    NERFMI_REQUIRE(rays_o && rays_d && rays_o_out && rays_d_out, "ndc_rays: null pointer");
    hipLaunchKernelGGL(ndc_rays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       ndc_const(H, W, focal, near), rays_o, rays_d, n, rays_o_out, rays_d_out);
    return check_launch("ndc_rays");
}

int nerfmi_generate_rays(const float *c2w, int n_images, int H, int W, double focal, const int64_t *pixel_index,
                         int64_t n_rays, int ndc, double near, double far, float *rays_out, nerfmi_stream_t stream) {
    NERFMI_ENTER();
    NERFMI_REQUIRE(n_images >= 1 && H >= 1 && W >= 1 && focal > 0 && n_rays >= 0, "generate_rays: bad sizes");
    NERFMI_REQUIRE(pixel_index || n_rays == (int64_t)n_images * H * W,
                   "generate_rays: without pixel_index n_rays must be n_images*H*W");
    if (n_rays == 0) return NERFMI_OK;
    NERFMI_REQUIRE(c2w && rays_out, "generate_rays: null pointer");
    // llff.py:235-241: NDC rays use the near plane 1.0 and are bounded by near 0 / far 1
    const NdcConst K = ndc_const(H, W, focal, 1.0);
    hipLaunchKernelGGL(generate_rays_kernel, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, (hipStream_t)stream, c2w,
                       n_images, H, W, (float)(W / 2.0), (float)(H / 2.0), (float)focal, pixel_index, n_rays, ndc, K,
                       ndc ? 0.0f : (float)near, ndc ? 1.0f : (float)far, rays_out);
    return check_launch("generate_rays");
}

int nerfmi_create_samples(int N, double origin_x, double origin_y, double origin_z, double voxel_size, float *samples_out,
                          nerfmi_stream_t stream) {
    NERFMI_ENTER();
    NERFMI_REQUIRE(N >= 2 && N <= 1024, "create_samples: 2 <= N <= 1024 required");
    NERFMI_REQUIRE(samples_out, "create_samples: null pointer");
    const int64_t n = (int64_t)N * N * N;
    hipLaunchKernelGGL(create_samples_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, N, n,
                       (float)voxel_size, (float)origin_x, (float)origin_y, (float)origin_z, samples_out);
    return check_launch("create_samples");
}

}  // extern "C"
