// The step right after render_rays in training: loss + PSNR and the optimizer update,
// each as ONE launch over flat buffers instead of ~10 elementwise launches and 7 multi-tensor launches.
//
//   mse_loss_kernel : losses.py:10-20 (MSELoss: nn.MSELoss(mean) on rgb_coarse [+ rgb_fine]) with its autograd
//                     (torch mse_loss_backward: (2/numel) * (x - t) * grad) and metrics.py:4-13 (psnr = -10 log10 mse)
//   adam_kernel     : utils/__init__.py:20 -> torch.optim.Adam(lr, eps=1e-8, weight_decay), restated from torch's
//                     single-tensor path (torch/optim/adam.py _single_tensor_adam, amsgrad=False, maximize=False)
#include "common.h"

namespace nerfmi {

// One workgroup, fp64 partial sums (the batch is a few thousand values; a deterministic tree keeps the loss
// bit-reproducible run to run).  out[0] = loss, out[1] = mse(coarse), out[2] = mse(fine), out[3] = psnr(fine or coarse).
__global__ void __launch_bounds__(1024)
mse_loss_kernel(const float *__restrict__ coarse, const float *__restrict__ fine, const float *__restrict__ target,
                int64_t n, float grad_out, float *__restrict__ out, float *__restrict__ g_coarse,
                float *__restrict__ g_fine) {
    __shared__ double red[2][16];
    const float norm = (float)(2.0 / (double)n);         // torch: norm = 2. / input.numel()
    double sc = 0.0, sf = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const float t = target[i];
        if (coarse) {
            const float d = __fsub_rn(coarse[i], t);
            sc += (double)__fmul_rn(d, d);
            if (g_coarse) g_coarse[i] = __fmul_rn(__fmul_rn(norm, d), grad_out);
        }
        if (fine) {
            const float d = __fsub_rn(fine[i], t);
            sf += (double)__fmul_rn(d, d);
            if (g_fine) g_fine[i] = __fmul_rn(__fmul_rn(norm, d), grad_out);
        }
    }
    sc = wave_sum_d(sc);
    sf = wave_sum_d(sf);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { red[0][wid] = sc; red[1][wid] = sf; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { a += red[0][w]; b += red[1][w]; }
        const float mc = coarse ? (float)(a / (double)n) : 0.f;
        const float mf = fine ? (float)(b / (double)n) : 0.f;
        out[0] = (coarse && fine) ? __fadd_rn(mc, mf) : (coarse ? mc : mf);
        out[1] = mc;
        out[2] = mf;
        out[3] = -10.0f * log10f(fine ? mf : mc);
    }
}

struct AdamScalars {
    float beta1_w;        // 1 - beta1  (lerp weight)
    float beta2;
    float one_m_beta2;
    float neg_step_size;  // -lr / (1 - beta1^step)
    float bc2_sqrt;       // sqrt(1 - beta2^step)
    float eps;
    float weight_decay;
    float grad_scale;     // gradients are multiplied by this first (1/world for a summed all-reduce); 1 = off
};

__global__ void __launch_bounds__(256)
adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v,
            int64_t n, AdamScalars S) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float grad = g[i];
        if (S.grad_scale != 1.0f) grad = __fmul_rn(grad, S.grad_scale);
        const float param = p[i];
        if (S.weight_decay != 0.f) grad = __builtin_fmaf(S.weight_decay, param, grad);   // grad.add(param, alpha=wd)
        // exp_avg.lerp_(grad, 1 - beta1): weight < 0.5 -> fmadd(weight, end - start, start)
        const float mm = __builtin_fmaf(S.beta1_w, __fsub_rn(grad, m[i]), m[i]);
        // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
        const float vv = __fadd_rn(__fmul_rn(v[i], S.beta2), __fmul_rn(__fmul_rn(S.one_m_beta2, grad), grad));
        // denom = (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps);  param.addcdiv_(exp_avg, denom, value = -step_size)
        const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(vv), S.bc2_sqrt), S.eps);
        p[i] = __fadd_rn(param, __fdiv_rn(__fmul_rn(S.neg_step_size, mm), denom));
        m[i] = mm;
        v[i] = vv;
    }
}

}  // namespace nerfmi

using namespace nerfmi;

extern "C" {

int nerfmi_mse_loss(const float *rgb_coarse, const float *rgb_fine, const float *targets, int64_t n_elems,
                    float grad_out, float *out4, float *grad_coarse, float *grad_fine, nerfmi_stream_t stream) {
    NERFMI_ENTER();
    NERFMI_REQUIRE(n_elems >= 1, "mse_loss: n_elems must be >= 1");
    NERFMI_REQUIRE((rgb_coarse || rgb_fine) && targets && out4, "mse_loss: null pointer");
    NERFMI_REQUIRE(!(grad_coarse && !rgb_coarse) && !(grad_fine && !rgb_fine), "mse_loss: gradient without its input");
    hipLaunchKernelGGL(mse_loss_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, rgb_coarse, rgb_fine, targets,
                       n_elems, grad_out, out4, grad_coarse, grad_fine);
    return check_launch("mse_loss");
}

int nerfmi_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, double lr,
                     double beta1, double beta2, double eps, double weight_decay, int64_t step, double grad_scale,
                     nerfmi_stream_t stream) {
    NERFMI_ENTER();
    NERFMI_REQUIRE(n >= 0 && step >= 1, "adam_step: n >= 0 and step >= 1 required");
    if (n == 0) return NERFMI_OK;
    NERFMI_REQUIRE(param && grad && exp_avg && exp_avg_sq, "adam_step: null pointer");
    NERFMI_REQUIRE(lr >= 0 && eps >= 0 && beta1 >= 0 && beta1 < 1 && beta2 >= 0 && beta2 < 1 && weight_decay >= 0,
                   "adam_step: bad hyper-parameter");
    NERFMI_REQUIRE(1.0 - beta1 < 0.5, "adam_step: beta1 <= 0.5 is not supported (torch's lerp switches formula)");
    // the scalars are formed in double, like the Python floats of torch/optim/adam.py, and rounded once
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    AdamScalars S;
    S.beta1_w = (float)(1.0 - beta1);
    S.beta2 = (float)beta2;
    S.one_m_beta2 = (float)(1.0 - beta2);
    S.neg_step_size = (float)(-(lr / bc1));
    S.bc2_sqrt = (float)sqrt(bc2);
    S.eps = (float)eps;
    S.weight_decay = (float)weight_decay;
    S.grad_scale = (float)grad_scale;
    const int64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, (hipStream_t)stream,
                       param, grad, exp_avg, exp_avg_sq, n, S);
    return check_launch("adam_step");
}

}  // extern "C"
