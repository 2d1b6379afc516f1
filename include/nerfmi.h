/*
 * nerfmi.h -- C ABI of the MI355X-native volumetric-rendering hot path.
 *
 * Drop-in boundary for Freedomcls/nerf-siren's render_rays() path.  The
 * reference has no FFI of its own here (the path is torch tensor code behind
 * the Python signature models/rendering.py:70-83); its native-plugin convention
 * (torch_utils/custom_ops.py:61, torchsearchsorted/src/cuda/
 * searchsorted_cuda_wrapper.cpp:9-20) is caller-allocated tensors + contiguity
 * checks + launch on the current stream.  This header keeps that shape without
 * torch types: plain device pointers, explicit sizes, an explicit hipStream_t.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous row-major fp32 unless
 *     stated; outputs are caller-allocated; nothing is allocated or freed here;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - launches are asynchronous; the functions never synchronise;
 *   - return 0 on success, <0 on error (NERFMI_E_*); the message of the last
 *     error on the calling thread is nerfmi_last_error();
 *   - no global mutable state besides that thread-local message: re-entrant
 *     across streams and threads.
 */
#ifndef NERFMI_H
#define NERFMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NERFMI_OK 0
#define NERFMI_E_INVALID (-1) /* bad argument (null pointer, size out of range) */
#define NERFMI_E_LAUNCH (-2)  /* HIP reported a launch/runtime error            */
#define NERFMI_E_UNSUPPORTED (-3)

typedef void *nerfmi_stream_t;

int nerfmi_version(void);
const char *nerfmi_last_error(void);

/* ---- a2: stratified sampler -- models/rendering.py:207-222 ---------------
 * rays (n_rays,8) = [o(3) d(3) near far]; perturb_rand (n_rays,n_samples) is the
 * torch.rand draw of :221 (may be NULL when perturb == 0);  z_out (n_rays,n_samples). */
int nerfmi_sample_stratified(const float *rays, const float *perturb_rand, int n_rays, int n_samples,
                             int use_disp, float perturb, float *z_out, nerfmi_stream_t stream);

/* Perf-mode variants that draw IN the kernel from Philox4x32-10 streams addressed by (seed, offset, segment) instead of
 * reading a tensor of draws (segments: 0 perturb_rand, 1 noise_coarse, 2 u, 3 noise_fine).  Element e of a segment is
 * bit-identical to what nerfmi_render_draws(seed, offset, ...) writes at index e, so the fused and the materialised paths
 * render the same image; the compositor's backward regenerates its forward's noise from the same key. */
int nerfmi_sample_stratified_philox(const float *rays, uint64_t seed, uint64_t offset, int n_rays, int n_samples,
                                    int use_disp, float perturb, float *z_out, nerfmi_stream_t stream);
int nerfmi_composite_philox(const float *field, int sigma_only, const float *z, const float *rays, uint64_t seed,
                            uint64_t offset, int segment, float noise_std, int n_rays, int n_per_ray, int white_back,
                            float *weights_out, float *rgb_out, float *depth_out, float *opacity_out,
                            nerfmi_stream_t stream);
int nerfmi_composite_backward_philox(const float *field, const float *z, const float *rays, uint64_t seed, uint64_t offset,
                                     int segment, float noise_std, int n_rays, int n_per_ray, int white_back,
                                     const float *g_rgb, const float *g_depth, const float *g_opacity, float *grad_field,
                                     nerfmi_stream_t stream);
int nerfmi_importance_resample_philox(const float *z_coarse, const float *weights_coarse, uint64_t seed, uint64_t offset,
                                      int n_rays, int n_samples, int n_importance, float *z_new_out, float *z_fine_out,
                                      nerfmi_stream_t stream);

/* The same streams written to memory by ONE launch (tests; callers that want the tensors): perturb_rand (n_perturb floats, U[0,1)) [rendering.py:221], noise_coarse (N(0,1)) [:170], u (U[0,1))
 * [:47], noise_fine (N(0,1)).  Philox4x32-10 keyed by `seed`, counter = (quad, segment, offset): the same
 * (seed, offset) always gives the same draws, independent of the sizes of the other segments.  A segment with size 0
 * is skipped (its pointer may be NULL). */
int nerfmi_render_draws(uint64_t seed, uint64_t offset, int64_t n_perturb, float *perturb_rand, int64_t n_noise_coarse,
                        float *noise_coarse, int64_t n_u, float *u, int64_t n_noise_fine, float *noise_fine,
                        nerfmi_stream_t stream);

/* ---- a5: Embedding.forward -- models/nerf.py:21-38 -----------------------
 * x (n,3) -> out (n, 3*(2*n_freqs+1)) = [x, sin(2^k x), cos(2^k x)]_k. */
int nerfmi_embed(const float *x, int64_t n, int n_freqs, float *out, nerfmi_stream_t stream);

/* ---- a6: NeRF MLP -- models/nerf.py:41-124 -------------------------------
 * Parameters are handed over once per weight update as the 24 state_dict
 * tensors in nerf.py:61-81 order (weight (out,in) row-major, then bias):
 *   xyz_encoding_{1..8}.0.{weight,bias}, xyz_encoding_final.{weight,bias},
 *   dir_encoding.0.{weight,bias}, sigma.{weight,bias}, rgb.0.{weight,bias}
 * and repacked into MFMA-fragment order (`packed`, nerfmi_nerf_packed_floats()
 * floats) -- forward and transposed (for the backward dX chain) images.
 * `params` is a HOST array of 24 DEVICE pointers. */
size_t nerfmi_nerf_packed_floats(void);
int nerfmi_nerf_pack(const float *const *params, float *packed, nerfmi_stream_t stream);

/* Fused inference() head (rendering.py:131-159): for every sample p of every
 * ray: xyz = o + d*z, Embedding(3,10), Embedding(3,4)(d), NeRF.forward.
 * z (n_rays,n_per_ray).  out: (n_rays*n_per_ray,4) [rgb,sigma] or, when
 * sigma_only, (n_rays*n_per_ray,1).
 * saved: NULL for inference, else nerfmi_nerf_saved_floats(n_points) floats of
 * activations kept for nerfmi_nerf_backward_rays (training).  */
size_t nerfmi_nerf_saved_floats(int64_t n_points);
int nerfmi_nerf_forward_rays(const float *packed, const float *rays, const float *z, int n_rays, int n_per_ray,
                             int sigma_only, float *out, float *saved, nerfmi_stream_t stream);

/* OPT-IN fast math for the same forward (inference): every fp32 product is formed on the bf16 matrix cores
 * from exact three-way bf16 splits of both operands (six v_mfma_f32_32x32x16_bf16 per fp32-equivalent
 * product block, fp32 accumulation; dropped terms <= 2^-24 relative), 2.7x the fp32-MFMA rate at fp32-level
 * accuracy.  `fast` (nerfmi_nerf_fast_bytes() bytes) is derived from `packed` by nerfmi_nerf_pack_fast.
 * saved: as in nerfmi_nerf_forward_rays (NULL for inference; same images, consumed by nerfmi_nerf_backward_rays). */
size_t nerfmi_nerf_fast_bytes(void);
int nerfmi_nerf_pack_fast(const float *packed, void *fast, nerfmi_stream_t stream);
int nerfmi_nerf_forward_rays_fast(const float *packed, const void *fast, const float *rays, const float *z,
                                  int n_rays, int n_per_ray, int sigma_only, float *out, float *saved,
                                  nerfmi_stream_t stream);
/* nerfmi_nerf_backward_rays with the dX chain and the 256 x 256 dW tasks on the split-bf16 path; same saved / workspace sizes.
 * A `saved` image is OPAQUE and belongs to the math that wrote it (the two paths order the elements of a 32-point tile
 * differently): nerfmi_nerf_forward_rays[/_embedded_train] -> nerfmi_nerf_backward_rays, _forward_rays_fast -> _backward_rays_fast. */
int nerfmi_nerf_backward_rays_fast(const float *packed, const void *fast, int n_rays, int n_per_ray, const float *saved,
                                   const float *grad_out, float *const *grad_params, float *workspace,
                                   nerfmi_stream_t stream);

/* NeRF.forward(x, sigma_only) on pre-embedded rows x (n, 90) / (n, 63)
 * (nerf.py:83-124) -- the module-level API (dense grid queries,
 * extract_color_mesh.py:117-143). */
int nerfmi_nerf_forward_embedded(const float *packed, const float *x, int64_t n, int sigma_only, float *out,
                                 nerfmi_stream_t stream);

/* Backward of nerfmi_nerf_forward_rays w.r.t. the 24 parameters.
 * grad_out (n_points,4) = dL/d[rgb,sigma].  grad_params: HOST array of 24
 * DEVICE pointers, same order/shapes as `params`; gradients are WRITTEN (not
 * accumulated).  workspace: nerfmi_nerf_backward_workspace_floats(n_points). */
size_t nerfmi_nerf_backward_workspace_floats(int64_t n_points);
/* NeRF.forward(x) (nerf.py:83-124) on pre-embedded rows x (n, 90) with the activations saved, for training through the
 * module-level API: back-propagate with nerfmi_nerf_backward_rays(packed, NULL, NULL, n, 1, saved, grad_out, ...). */
int nerfmi_nerf_forward_embedded_train(const float *packed, const float *x, int64_t n, float *out, float *saved,
                                       nerfmi_stream_t stream);
int nerfmi_nerf_backward_rays(const float *packed, const float *rays, const float *z, int n_rays, int n_per_ray,
                              const float *saved, const float *grad_out, float *const *grad_params,
                              float *workspace, nerfmi_stream_t stream);

/* ---- a7: FiLM-SIREN field -- models/nerf.py:142-151 (FiLMLayer), :159-216 (SemanticNeRF) -----------
 * params: HOST array of 22 DEVICE pointers in state_dict order
 *   network.{0..7}.layer.{weight,bias}, final_layer.{weight,bias},
 *   color_layer_sine.layer.{weight,bias}, color_layer_linear.0.{weight,bias}
 * forward_with_frequencies_phase_shifts(input, frequencies, phase_shifts, ray_directions) (:201-216):
 *   points (n_points,3), ray_directions (n_points,3), frequencies / phase_shifts (n_cond, 9*256) where
 *   consecutive groups of points_per_cond points share one conditioning row (the reference's (Bz,Np,3)
 *   input flattened: points_per_cond = Np).  out (n_points,4) [rgb,sigma] or (n_points,1) sigma_only. */
size_t nerfmi_siren_packed_floats(void);
int nerfmi_siren_pack(const float *const *params, float *packed, nerfmi_stream_t stream);
int nerfmi_siren_forward_points(const float *packed, const float *points, const float *ray_directions,
                                const float *frequencies, const float *phase_shifts, int64_t n_points,
                                int64_t points_per_cond, int sigma_only, float *out, nerfmi_stream_t stream);
/* The same field fused behind the ray sampler (inference() head, rendering.py:131-159, with raw xyz / d
 * instead of embeddings): xyz = o + d*z per sample; rays_per_cond consecutive rays share a conditioning row. */
int nerfmi_siren_forward_rays(const float *packed, const float *rays, const float *z, const float *frequencies,
                              const float *phase_shifts, int n_rays, int n_per_ray, int64_t rays_per_cond,
                              int sigma_only, float *out, nerfmi_stream_t stream);

/* Training path of the FiLM-SIREN field (autograd of models/nerf.py:142-151, :201-216 w.r.t. the 22 parameters; the
 * inputs carry no gradient; the conditioning rows: nerfmi_siren_backward_cond).  The *_train forwards also write `saved`
 * (nerfmi_siren_saved_floats(n_points) floats: per 32-point tile the layer inputs (the sines) and cos(arg) of every unit);
 * nerfmi_siren_backward turns grad_out (n_points,4) = [d rgb, d sigma] into the 22 gradients (written, not
 * accumulated; bit-reproducible: fixed-order slab reduction, no float atomics).  grad_params: HOST array of 22 DEVICE
 * pointers in state_dict order; workspace: nerfmi_siren_backward_workspace_floats(n_points) floats. */
size_t nerfmi_siren_saved_floats(int64_t n_points);
size_t nerfmi_siren_backward_workspace_floats(int64_t n_points);
int nerfmi_siren_forward_rays_train(const float *packed, const float *rays, const float *z, const float *frequencies,
                                    const float *phase_shifts, int n_rays, int n_per_ray, int64_t rays_per_cond,
                                    float *out, float *saved, nerfmi_stream_t stream);
int nerfmi_siren_forward_points_train(const float *packed, const float *points, const float *ray_directions,
                                      const float *frequencies, const float *phase_shifts, int64_t n_points,
                                      int64_t points_per_cond, float *out, float *saved, nerfmi_stream_t stream);
int nerfmi_siren_backward(const float *packed, const float *saved, const float *grad_out, const float *frequencies,
                          int64_t n_points, int64_t points_per_cond, float *const *grad_params, float *workspace,
                          nerfmi_stream_t stream);
/* The same backward for a launch that shares ONE conditioning row, with the gradients of that row as well
 * (models/nerf.py:147-151, :201-216 are differentiable in `frequencies` / `phase_shifts`, which the reference's mapping
 * network -- nerf.py:185 -- would produce): grad_frequencies, grad_phase_shifts (9*256 floats each, written).  They come out
 * of the dW slab reduction (one dot product per unit), not out of a second pass over the activations.  Several
 * conditioning rows: one forward_*_train + one call per row. */
int nerfmi_siren_backward_cond(const float *packed, const float *saved, const float *grad_out, const float *frequencies,
                               int64_t n_points, float *const *grad_params, float *grad_frequencies,
                               float *grad_phase_shifts, float *workspace, nerfmi_stream_t stream);

/* OPT-IN split-bf16 math for the FiLM-SIREN field (see nerfmi_nerf_forward_rays_fast): `fast`
 * (nerfmi_siren_fast_bytes() bytes) is derived from the SIREN `packed` blob by nerfmi_siren_pack_fast. */
size_t nerfmi_siren_fast_bytes(void);
int nerfmi_siren_pack_fast(const float *packed, void *fast, nerfmi_stream_t stream);
/* Training on the same math: the forward that also writes `saved` (same size and row map as nerfmi_siren_forward_rays_train's,
 * but a different element order inside a 32-point tile: a `saved` image is OPAQUE and must be handed to the backward of the
 * math that wrote it -- _train -> nerfmi_siren_backward[_cond], _train_fast -> nerfmi_siren_backward_fast), and the
 * backward whose dX chain and 256 x 256 dW tasks run on the bf16 matrix cores (same workspace; grad_frequencies /
 * grad_phase_shifts both NULL, or both set for a launch that shares one conditioning row). */
int nerfmi_siren_forward_rays_train_fast(const float *packed, const void *fast, const float *rays, const float *z,
                                         const float *frequencies, const float *phase_shifts, int n_rays, int n_per_ray,
                                         int64_t rays_per_cond, float *out, float *saved, nerfmi_stream_t stream);
int nerfmi_siren_backward_fast(const float *packed, const void *fast, const float *saved, const float *grad_out,
                               const float *frequencies, int64_t n_points, int64_t points_per_cond, float *const *grad_params,
                               float *grad_frequencies, float *grad_phase_shifts, float *workspace, nerfmi_stream_t stream);
int nerfmi_siren_forward_rays_fast(const float *packed, const void *fast, const float *rays, const float *z,
                                   const float *frequencies, const float *phase_shifts, int n_rays, int n_per_ray,
                                   int64_t rays_per_cond, int sigma_only, float *out, nerfmi_stream_t stream);

/* ---- a8: compositing -- models/rendering.py:162-190 ----------------------
 * field: (n_rays,n_per_ray,4) [rgb,sigma] or, when sigma_only, (n_rays,n_per_ray)
 * sigma (the weights_only branch :179-180: only weights/opacity are produced).
 * noise (n_rays,n_per_ray) = the torch.randn draw of :170 or NULL (noise_std 0).
 * Outputs (each may be NULL except weights_out/opacity_out when sigma_only):
 *   weights_out (n_rays,n_per_ray), rgb_out (n_rays,3), depth_out, opacity_out (n_rays). */
int nerfmi_composite(const float *field, int sigma_only, const float *z, const float *rays, const float *noise,
                     float noise_std, int n_rays, int n_per_ray, int white_back, float *weights_out,
                     float *rgb_out, float *depth_out, float *opacity_out, nerfmi_stream_t stream);

/* Backward of nerfmi_composite: d loss/d field given d loss/d (rgb,depth,opacity)
 * (any of g_* may be NULL = zero).  grad_field (n_rays,n_per_ray,4). */
int nerfmi_composite_backward(const float *field, const float *z, const float *rays, const float *noise,
                              float noise_std, int n_rays, int n_per_ray, int white_back, const float *g_rgb,
                              const float *g_depth, const float *g_opacity, float *grad_field,
                              nerfmi_stream_t stream);

/* ---- a3: sample_pdf -- models/rendering.py:22-67 -------------------------
 * bins (n_rays,n_weights+1), weights (n_rays,n_weights); u (n_rays,n_importance)
 * = the torch.rand draw of :47, or NULL for det=True (u = linspace(0,1,F)).
 * Optional outputs (NULL to skip): cdf_out (n_rays,n_weights+1),
 * inds_out (n_rays,n_importance) int64 (torch.searchsorted(right=True)).  */
int nerfmi_sample_pdf(const float *bins, const float *weights, const float *u, int n_rays, int n_weights,
                      int n_importance, float *cdf_out, int64_t *inds_out, float *samples_out,
                      nerfmi_stream_t stream);

/* Stage 2 alone (rendering.py:54-66) on a caller-supplied cdf. */
int nerfmi_search_lerp(const float *bins, const float *cdf, const float *u, int n_rays, int n_weights,
                       int n_importance, int64_t *inds_out, float *samples_out, nerfmi_stream_t stream);

/* Row-wise searchsorted with row broadcasting -- torchsearchsorted/src/cuda/
 * searchsorted_cuda_kernel.cu:84-142 (searchsorted_cuda_wrapper(a,v,res,side_left)).
 * a (nrow_a,ncol_a) sorted rows, v (nrow_v,ncol_v); nrow_a==1 or nrow_v==1 broadcast.
 * out (max(nrow_a,nrow_v), ncol_v) int64 = numpy.searchsorted(side). */
int nerfmi_searchsorted(const float *a, const float *v, int nrow_a, int nrow_v, int ncol_a, int ncol_v,
                        int side_left, int64_t *out, nerfmi_stream_t stream);

/* ---- a4: merge -- models/rendering.py:247 --------------------------------
 * out (n_rays, na+nb) = sort(cat[za, zb], -1) values. */
int nerfmi_merge_sorted(const float *za, const float *zb, int n_rays, int na, int nb, float *out,
                        nerfmi_stream_t stream);

/* rendering.py:242-247 fused: z_mid -> sample_pdf(z_mid, w[:,1:-1]) -> sort(cat).
 * z_coarse (n_rays,S) (any order; ascending input -- what the stratified sampler produces -- takes the fast
 * rank-merge path), weights_coarse (n_rays,S), u NULL => det.
 * z_fine_out (n_rays,S+F).  z_new_out optional (n_rays,F). */
int nerfmi_importance_resample(const float *z_coarse, const float *weights_coarse, const float *u, int n_rays,
                               int n_samples, int n_importance, float *z_new_out, float *z_fine_out,
                               nerfmi_stream_t stream);

/* ---- a1: one call for a whole render_rays() pass under no_grad -- models/rendering.py:70-262 as eval.py:85-96 and the
 * validation loop system.py:243-256 call it.  Enqueues on `stream`, out of ONE caller-provided workspace
 * (nerfmi_render_rays_workspace_floats floats): sampler -> field(coarse; sigma-only when test_time) -> compositor ->
 * importance resampling -> field(fine) -> compositor.  field_kind 0 = NeRF (packed_* from nerfmi_nerf_pack), 1 = FiLM-SIREN
 * (packed_* from nerfmi_siren_pack; cond_* = one conditioning row each, [frequencies 2304 | phase_shifts 2304]).  Random
 * draws (perturb > 0: jitter and u; noise_std != 0: density noise) come from the Philox streams (seed, offset) exactly as in
 * the *_philox entry points, so the result is bit-identical to the call-by-call sequence with the same key.
 * Outputs as in the reference's result dict: rgb_* (n_rays,3), depth_*, opacity_* (n_rays); with test_time the coarse
 * rgb/depth are not produced (rendering.py:227-231) and may be NULL; n_importance == 0 skips the fine pass. */
size_t nerfmi_render_rays_workspace_floats(int n_rays, int n_samples, int n_importance, int test_time);
int nerfmi_render_rays_fused(int field_kind, const float *packed_coarse, const float *packed_fine,
                             const float *cond_coarse, const float *cond_fine, const float *rays, int n_rays,
                             int n_samples, int n_importance, int use_disp, float perturb, float noise_std,
                             int white_back, int test_time, uint64_t seed, uint64_t offset, float *workspace,
                             float *rgb_coarse, float *depth_coarse, float *opacity_coarse, float *rgb_fine,
                             float *depth_fine, float *opacity_fine, nerfmi_stream_t stream);

/* ---- measurement aid: HIP events recorded on the launch stream around the field-MLP kernels (forward, dX chain, dW
 * GEMM, slab reduction) between nerfmi_profile_start() and nerfmi_profile_stop().  nerfmi_profile_report waits for the
 * recorded spans and writes one line per (kernel tag, units per launch): "<tag>\t<points>\t<launches>\t<total ms>\n";
 * it returns the length of the full report (call with cap 0 to size the buffer).  Process-wide, mutex-guarded; off by
 * default (the only global state of the library besides the thread-local error string). */
int nerfmi_profile_start(void);
int nerfmi_profile_stop(void);
int64_t nerfmi_profile_report(char *buf, size_t cap);

/* ==== the step before the path: ray generation on the device ========================
 * datasets/ray_utils.py:5-24 get_ray_directions(H, W, focal) -> dirs_out (H, W, 3): ((i-W/2)/focal, -(j-H/2)/focal, -1),
 * i = column, j = row (kornia.create_meshgrid(H, W, normalized_coordinates=False)). */
int nerfmi_ray_directions(int H, int W, double focal, float *dirs_out, nerfmi_stream_t stream);

/* datasets/ray_utils.py:27-50 get_rays(directions (n,3), c2w (3,4) row-major) -> rays_o (n,3) = c2w[:,3],
 * rays_d (n,3) = normalised directions @ c2w[:, :3].T */
int nerfmi_get_rays(const float *directions, const float *c2w, int64_t n, float *rays_o_out, float *rays_d_out,
                    nerfmi_stream_t stream);

/* datasets/ray_utils.py:53-93 get_ndc_rays(H, W, focal, near, rays_o, rays_d) */
int nerfmi_ndc_rays(int H, int W, double focal, double near, const float *rays_o, const float *rays_d, int64_t n,
                    float *rays_o_out, float *rays_d_out, nerfmi_stream_t stream);

/* The three above fused with the (N,8) [o, d, near, far] packing of datasets/blender.py:60-69 (ndc = 0, near/far as
 * given: 2 / 6) and datasets/llff.py:234-250 (ndc != 0: get_ndc_rays with near plane 1.0, bounds 0 / 1).
 * c2w (n_images,3,4); pixel_index (n_rays) int64 = image*H*W + row*W + column, or NULL for every pixel of every
 * image in order (then n_rays = n_images*H*W).  rays_out (n_rays, 8). */
int nerfmi_generate_rays(const float *c2w, int n_images, int H, int W, double focal, const int64_t *pixel_index,
                         int64_t n_rays, int ndc, double near, double far, float *rays_out, nerfmi_stream_t stream);

/* The N^3 sample grid of the EG3D dense query ("neural volume", BASELINE configs[4]):
 * extract_color_mesh_eg3d.py:72-94 create_samples, op by op in fp32.  origin_* = voxel_origin - cube_length/2
 * (the reference's `voxel_origin` after :74), voxel_size = cube_length/(N-1); samples_out (N^3, 3). */
int nerfmi_create_samples(int N, double origin_x, double origin_y, double origin_z, double voxel_size, float *samples_out,
                          nerfmi_stream_t stream);

/* ==== the step after the path in training ===========================================
 * losses.py:10-20 MSELoss = nn.MSELoss(mean)(rgb_coarse, t) [+ nn.MSELoss(mean)(rgb_fine, t)], its autograd
 * (d x = (2/n_elems) * (x - t) * grad_out) and metrics.py:4-13 psnr = -10 log10(mse), in one launch.
 * rgb_coarse / rgb_fine / targets: n_elems floats ((n_rays,3) flattened); either prediction may be NULL.
 * out4 = {loss, mse_coarse, mse_fine, psnr of the fine (else coarse) prediction}; grad_* optional (NULL = skip). */
int nerfmi_mse_loss(const float *rgb_coarse, const float *rgb_fine, const float *targets, int64_t n_elems,
                    float grad_out, float *out4, float *grad_coarse, float *grad_fine, nerfmi_stream_t stream);

/* utils/__init__.py:20 -> torch.optim.Adam(params, lr, eps=1e-8, weight_decay) (torch/optim/adam.py
 * _single_tensor_adam, amsgrad=False): one launch over a FLAT parameter buffer and its flat gradient/state.
 * step = 1 for the first update.  grad is multiplied by grad_scale first (1/world_size after a summed
 * all-reduce; 1 = off).  beta1 must be > 0.5 (torch's lerp formula switches below). */
int nerfmi_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, double lr,
                     double beta1, double beta2, double eps, double weight_decay, int64_t step, double grad_scale,
                     nerfmi_stream_t stream);

/* ==== EG3D tri-plane importance renderer (forward) ========================================================
 * volumetric_rendering/renderer.py, ray_marcher.py, ray_sampler.py, math_utils.py; eg3d_training/triplane.py.
 * Planes are consumed channels-last: planes_hwc (n*3, H, W, 32), produced once per plane update from the
 * reference's (n,3,32,H,W) layout (triplane.py:65).  R below = number of rays of the call (N*M). */
int nerfmi_eg3d_pack_planes(const float *planes_nchw, int n_planes /* = n*3 */, int channels, int h, int w,
                            float *planes_hwc, nerfmi_stream_t stream);
/* OSGDecoder (triplane.py:144-167) = FullyConnectedLayer(32,64) -> Softplus -> FullyConnectedLayer(64,4)
 * (networks_stylegan2.py:96-127: w*lr_mul/sqrt(in), b*lr_mul).  packed: nerfmi_eg3d_decoder_floats() floats. */
size_t nerfmi_eg3d_decoder_floats(void);
int nerfmi_eg3d_pack_decoder(const float *w0, const float *b0, const float *w1, const float *b1, float lr_multiplier,
                             float *packed, nerfmi_stream_t stream);
/* sample_from_planes (renderer.py:55-65): coords (n, n_points, 3) -> feats_out (n, 3, n_points, 32). */
int nerfmi_eg3d_sample_planes(const float *planes_hwc, int n, int h, int w, const float *coords, int64_t n_points,
                              float box_warp, float *feats_out, nerfmi_stream_t stream);
/* run_model (renderer.py:144-151) fused with the decoder: rgb_out (n, n_points, 3), sigma_out (n, n_points). */
int nerfmi_eg3d_run_model(const float *planes_hwc, int n, int h, int w, const float *decoder_packed,
                          const float *coords, int64_t n_points, float box_warp, float *rgb_out, float *sigma_out,
                          nerfmi_stream_t stream);
/* the same with coords = ray_origins + depths * ray_directions (renderer.py:105, :125);
 * ray_* (n, M, 3), depths (n, M, n_samples), outputs (n, M*n_samples, .). */
int nerfmi_eg3d_run_model_rays(const float *planes_hwc, int n, int h, int w, const float *decoder_packed,
                               const float *ray_origins, const float *ray_directions, const float *depths,
                               int64_t n_rays_per_batch, int n_samples, float box_warp, float *rgb_out,
                               float *sigma_out, nerfmi_stream_t stream);
/* sample_stratified (renderer.py:172-195); rand = the rand_like draw (R, n_samples).  ray_start_t/ray_end_t (R)
 * select the per-ray ('auto') branch, else the scalar ray_start/ray_end (+ optional disparity sampling). */
int nerfmi_eg3d_sample_stratified(const float *ray_start_t, const float *ray_end_t, float ray_start, float ray_end,
                                  const float *rand, int64_t n_rays, int n_samples, int disparity, float *depths_out,
                                  nerfmi_stream_t stream);
/* The same with the draw made IN the kernel from the Philox stream (seed, offset), segment 0 -- element e of the draw is what
 * nerfmi_render_draws(seed, offset, ...) writes at perturb_rand[e]. */
int nerfmi_eg3d_sample_stratified_philox(const float *ray_start_t, const float *ray_end_t, float ray_start, float ray_end,
                                         uint64_t seed, uint64_t offset, int64_t n_rays, int n_samples, int disparity,
                                         float *depths_out, nerfmi_stream_t stream);
/* torch.min / torch.max over all depths of the call (ray_marcher.py:50) -> minmax_out[2] on the device. */
int nerfmi_eg3d_minmax(const float *x, int64_t n, float *minmax_out, nerfmi_stream_t stream);
/* MipRayMarcher2.run_forward (ray_marcher.py:25-57): colors (R,S,3), densities (R,S), depths (R,S) ->
 * rgb_out (R,3), depth_out (R), weights_out (R,S-1) [optional], weight_sum_out (R) [optional]. */
int nerfmi_eg3d_march(const float *colors, const float *densities, const float *depths, const float *minmax,
                      int64_t n_rays, int n_samples, int white_back, float *rgb_out, float *depth_out,
                      float *weights_out, float *weight_sum_out, nerfmi_stream_t stream);
/* sample_importance (renderer.py:197-256): depths (R,S), weights (R,S-1), u (R,F) = the torch.rand draw -> z_out (R,F). */
int nerfmi_eg3d_sample_importance(const float *depths, const float *weights, const float *u, int64_t n_rays,
                                  int n_samples, int n_importance, float *z_out, nerfmi_stream_t stream);
/* ... with u drawn in the kernel: segment 2 of the Philox stream (seed, offset) (= nerfmi_render_draws' `u`). */
int nerfmi_eg3d_sample_importance_philox(const float *depths, const float *weights, uint64_t seed, uint64_t offset,
                                         int64_t n_rays, int n_samples, int n_importance, float *z_out,
                                         nerfmi_stream_t stream);
/* unify_samples (renderer.py:160-170): sort by depth and gather colours (.,3) and densities.
 * idx_out (R, n1+n2) int32, optional: source position of every sorted sample (for the backward). */
int nerfmi_eg3d_unify(const float *d1, const float *c1, const float *s1, const float *d2, const float *c2,
                      const float *s2, int64_t n_rays, int n1, int n2, float *d_out, float *c_out, float *s_out,
                      int32_t *idx_out, nerfmi_stream_t stream);
/* RaySampler.forward (ray_sampler.py:24-63): cam2world (n,4,4), intrinsics (n,3,3) -> origins/dirs (n, res*res, 3). */
int nerfmi_eg3d_ray_sampler(const float *cam2world, const float *intrinsics, int n, int resolution, float *origins_out,
                            float *dirs_out, nerfmi_stream_t stream);
/* get_ray_limits_box (math_utils.py:46-98): (n,3),(n,3) -> tmin (n), tmax (n); misses = (-1,-2). */
int nerfmi_eg3d_ray_limits_box(const float *rays_o, const float *rays_d, int64_t n, float box_side_length,
                               float *tmin_out, float *tmax_out, nerfmi_stream_t stream);

/* ---- EG3D backward: the autograd graph of renderer.py:88-142 w.r.t. planes and OSGDecoder parameters ----
 * (depths carry no gradient: stratified draws, and sample_importance runs under no_grad, renderer.py:201) */
/* d(colors, densities) of MipRayMarcher2.run_forward given d(rgb, depth, weights.sum); accumulate != 0 adds
 * into d_colors / d_densities (coarse samples also receive gradient through the fine march). */
int nerfmi_eg3d_march_backward(const float *colors, const float *densities, const float *depths, const float *minmax,
                               const float *g_rgb, const float *g_depth, const float *g_weight_sum, int64_t n_rays,
                               int n_samples, int white_back, int accumulate, float *d_colors, float *d_densities,
                               nerfmi_stream_t stream);
/* inverse of nerfmi_eg3d_unify's permutation. */
int nerfmi_eg3d_unify_backward(const int32_t *idx, const float *g_colors, const float *g_densities, int64_t n_rays,
                               int n1, int n2, float *d_c1, float *d_s1, float *d_c2, float *d_s2,
                               nerfmi_stream_t stream);
/* backward of nerfmi_eg3d_run_model_rays: d_rgb (n, M*S, 3), d_sigma (n, M*S) -> float-atomic scatter-add into
 * gplanes_hwc (n*3, H, W, 32; caller zero-fills) and a per-point scratch image `aux`
 * (nerfmi_eg3d_backward_aux_floats(n*M*S) floats) consumed by nerfmi_eg3d_decoder_wgrad. */
size_t nerfmi_eg3d_backward_aux_floats(int64_t n_points);
int nerfmi_eg3d_run_model_rays_backward(const float *planes_hwc, int n, int h, int w, const float *decoder_packed,
                                        const float *ray_origins, const float *ray_directions, const float *depths,
                                        int64_t n_rays_per_batch, int n_samples, float box_warp, const float *d_rgb,
                                        const float *d_sigma, float *gplanes_hwc, float *aux, nerfmi_stream_t stream);
/* decoder parameter gradients from `aux` (deterministic chunk slabs; partial: nerfmi_eg3d_wgrad_partial_floats()). */
size_t nerfmi_eg3d_wgrad_partial_floats(void);
int nerfmi_eg3d_decoder_wgrad(const float *aux, int64_t n_points, float lr_multiplier, int accumulate, float *partial,
                              float *g_w0, float *g_b0, float *g_w1, float *g_b1, nerfmi_stream_t stream);
/* channels-last (n_planes,H,W,C) -> (n_planes,C,H,W): the plane gradient in the reference's layout. */
int nerfmi_eg3d_unpack_planes(const float *planes_hwc, int n_planes, int channels, int h, int w, float *planes_nchw,
                              nerfmi_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NERFMI_H */
