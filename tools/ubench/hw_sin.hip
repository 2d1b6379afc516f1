// Accuracy of the hardware v_sin_f32 / v_cos_f32 (input in revolutions) against double precision, for the FiLM-SIREN
// argument range.  Build: hipcc --offload-arch=gfx950 -O3 hw_sin.hip -o hw_sin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float *t, float *s, float *c, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        float a, b;
        asm volatile("v_sin_f32 %0, %1" : "=v"(a) : "v"(t[i]));
        asm volatile("v_cos_f32 %0, %1" : "=v"(b) : "v"(t[i]));
        s[i] = a; c[i] = b;
    }
}
int main() {
    const int n = 1 << 22;
    std::vector<float> t(n), s(n), c(n);
    const double ranges[7] = {0.5, 4.0, 32.0, 200.0, 255.9, 1000.0, 100000.0};
    float *dt, *ds, *dc;
    (void)hipMalloc(&dt, n * 4); (void)hipMalloc(&ds, n * 4); (void)hipMalloc(&dc, n * 4);
    for (int r = 0; r < 7; ++r) {
        unsigned long long st = 88172645463325252ull;
        for (int i = 0; i < n; ++i) {
            st ^= st << 13; st ^= st >> 7; st ^= st << 17;
            t[i] = (float)(((double)(st >> 11) / 9007199254740992.0 * 2 - 1) * ranges[r]);
        }
        (void)hipMemcpy(dt, t.data(), n * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dt, ds, dc, n);
        (void)hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
        double es = 0, ec = 0, ms = 0, mc = 0, rel = 0;
        for (int i = 0; i < n; ++i) {
            const double x = 6.283185307179586476925 * (double)t[i];
            const double e1 = fabs((double)s[i] - sin(x)), e2 = fabs((double)c[i] - cos(x));
            if (e1 > es) es = e1;
            if (e2 > ec) ec = e2;
            ms += e1; mc += e2;
            if (fabs(sin(x)) > 1e-3 && e1 / fabs(sin(x)) > rel) rel = e1 / fabs(sin(x));
        }
        printf("|t| <= %6.1f rev: sin max abs err %.3e mean %.3e (max rel err where |sin|>1e-3: %.3e) | cos max abs err %.3e mean %.3e\n",
               ranges[r], es, ms / n, rel, ec, mc / n);
    }
    return 0;
}
