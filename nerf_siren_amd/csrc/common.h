// Shared helpers for the gfx950 kernels behind include/nerfmi.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/nerfmi.h"

namespace nerfmi {

void set_error(const char *fmt, ...);

#define NERFMI_REQUIRE(cond, ...)                 \
    do {                                          \
        if (!(cond)) {                            \
            nerfmi::set_error(__VA_ARGS__);       \
            return NERFMI_E_INVALID;              \
        }                                         \
    } while (0)

// First statement of every entry point that launches: forget an error some OTHER user of the (process-wide, shared) HIP
// runtime left in this thread's last-error slot -- PyTorch probes devices and peers through the same runtime, and
// check_launch() below would otherwise report its stale hipErrorNoDevice as this call's failure (seen in round 3: the first
// nerfmi call of a fresh process raised "no ROCm-capable device is detected" right after torch had put tensors on that
// device).
#define NERFMI_ENTER() (void)hipGetLastError()

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return NERFMI_E_LAUNCH;
    }
    return NERFMI_OK;
}

// Kernel-span profiler (render.hip): HIP events on the launch stream around a kernel, collected while
// nerfmi_profile_start() ... nerfmi_profile_stop() is active (bench.py's per-kernel rooflines); one branch when off.
extern bool g_profile_on;
int profile_begin(const char *tag, int64_t units, hipStream_t st);
void profile_end(int idx, hipStream_t st);
struct KernelSpan {
    int idx;
    hipStream_t st;
    KernelSpan(const char *tag, int64_t units, hipStream_t s) : idx(g_profile_on ? profile_begin(tag, units, s) : -1), st(s) {}
    ~KernelSpan() { if (idx >= 0) profile_end(idx, st); }
};

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device's instance of the kernel, so "done" is
// tracked per device (a per-thread flag would skip the call on a second GPU driven from the same thread).  Racing
// threads at worst both set the same value.
struct PerDeviceOnce {
    bool done[64] = {};
    // true if the caller must (re)do the per-device setup now
    bool needed(int &dev) {
        dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = -1; return true; }
        return dev < 0 || dev >= 64 || !done[dev];
    }
    void mark(int dev) { if (dev >= 0 && dev < 64) done[dev] = true; }
};

// torch.nn.Softplus (beta 1, threshold 20: triplane.py:150) as max(x, 0) + log1p(exp(-|x|)) on the hardware exp2 / log2
// (v_exp_f32, v_log_f32, ~1 ulp each): 7 instructions against ~50 for log1pf(expf(x)).  The OSGDecoder evaluates 64 of
// them per sample, which made the fused tri-plane kernel VALU-bound rather than gather-bound.  Absolute error <= 2e-7
// (x > 20 returns x exactly, as torch's threshold branch does); its derivative is sigmoid_hw.
__device__ __forceinline__ float softplus_hw(float x) {
    const float t = __builtin_amdgcn_exp2f(-fabsf(x) * 1.44269504088896341f);
    return fmaxf(x, 0.f) + __builtin_amdgcn_logf(1.0f + t) * 0.693147180559945309f;
}
__device__ __forceinline__ float sigmoid_hw(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-x * 1.44269504088896341f));
}

// x / 3 correctly rounded (== __fdiv_rn(x, 3.f) for finite x; checked on 2e7 values over 40 decades) in three
// instructions instead of the ~10 of the division expansion: q = x * fl(1/3), exact residual by fma, one correction
// (Markstein).  torch's mean over the three planes divides by 3; 32 of these per sample sat in the tri-plane kernel.
__device__ __forceinline__ float div3_rn(float x) {
    const float y = 0.3333333432674407958984375f;
    const float q = x * y;
    return __builtin_fmaf(__builtin_fmaf(-q, 3.0f, x), y, q);
}

constexpr int WAVE = 64;

// torch.linspace(0,1,n)[i] on CPU fp32: symmetric fill with one rounding
// (fma form) -- oracle/nerf_oracle.py linspace01.
__device__ __forceinline__ float linspace01(int i, int n) {
    if (n <= 1) return 0.f;
    const float step = __fdiv_rn(1.0f, (float)(n - 1));
    return (i < n / 2) ? __fmul_rn(step, (float)i) : __builtin_fmaf(-step, (float)(n - 1 - i), 1.0f);
}

// ---- wave-level (64-lane) scans / reductions in fp64 -----------------------
__device__ __forceinline__ double shfl_up_d(double v, int d) { return __shfl_up(v, d, WAVE); }
__device__ __forceinline__ double shfl_xor_d(double v, int m) { return __shfl_xor(v, m, WAVE); }

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_d(v, m);
    return v;
}

// inclusive scans over the 64 lanes (Kogge-Stone)
__device__ __forceinline__ double wave_incl_prod_d(double v, int lane) {
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        double o = shfl_up_d(v, d);
        if (lane >= d) v *= o;
    }
    return v;
}
__device__ __forceinline__ double wave_incl_sum_d(double v, int lane) {
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        double o = shfl_up_d(v, d);
        if (lane >= d) v += o;
    }
    return v;
}

// ||d|| as restated in the oracle (ray_norm): fp32, x,y,z order, no fma.
__device__ __forceinline__ float ray_norm(float dx, float dy, float dz) {
    float s = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
    s = __fadd_rn(s, __fmul_rn(dz, dz));
    return (float)sqrt((double)s);   // correctly rounded (fp64 sqrt of an fp32 value rounds innocuously)
}

}  // namespace nerfmi
