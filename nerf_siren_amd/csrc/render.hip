// One C-ABI call for a whole render_rays() inference pass (models/rendering.py:70-262 under no_grad: eval.py:85-96, the
// validation loop system.py:243-256) -- SURVEY section 8(b) `nerfmi_render_rays_fused` -- and the kernel-span profiler
// that bench.py reads its per-kernel rooflines from.
//
// "Fused" at this boundary means ONE host call that enqueues the whole pass on the caller's stream out of one caller-
// provided workspace: sampler -> field(coarse) -> compositor -> importance resampling -> field(fine) -> compositor, six
// launches with no host work in between (the field MLP, which is >99 % of the time, is already one kernel per pass with
// point generation, embedding and all layers inside; joining the per-ray kernels to it would put wave-per-ray code into a
// kernel tuned to 512 registers per wave).  What the call removes is the per-launch host cost of the Python layer.
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

#include "common.h"

using namespace nerfmi;

// ---------------------------------------------------------------------------------------------------------------------
// kernel-span profiler: HIP events recorded on the launch stream around the field-MLP kernels (forward, dX chain, dW
// GEMM, slab reduction).  Off by default (one predictable branch per launch); bench.py turns it on for its timed region.
// The only global state of the library besides the thread-local error string; guarded by a mutex.
// ---------------------------------------------------------------------------------------------------------------------
namespace nerfmi {

bool g_profile_on = false;

namespace {
struct Span {
    const char *tag;
    int64_t units;
    hipEvent_t a, b;
};
std::mutex g_prof_mu;
std::vector<Span> g_spans;            // spans of the current collection
std::vector<hipEvent_t> g_pool;       // events to reuse
constexpr size_t MAX_SPANS = 1 << 16;

hipEvent_t take_event() {
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return e;
}
}  // namespace

int profile_begin(const char *tag, int64_t units, hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_profile_on || g_spans.size() >= MAX_SPANS) return -1;
    Span s{tag, units, take_event(), take_event()};
    if (!s.a || !s.b) return -1;
    if (hipEventRecord(s.a, st) != hipSuccess) { (void)hipGetLastError(); g_pool.push_back(s.a); g_pool.push_back(s.b); return -1; }
    g_spans.push_back(s);
    return (int)g_spans.size() - 1;
}

void profile_end(int idx, hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (idx < 0 || idx >= (int)g_spans.size()) return;
    if (hipEventRecord(g_spans[idx].b, st) != hipSuccess) (void)hipGetLastError();
}

}  // namespace nerfmi

extern "C" {

int nerfmi_profile_start(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (const Span &s : g_spans) { g_pool.push_back(s.a); g_pool.push_back(s.b); }
    g_spans.clear();
    g_profile_on = true;
    return NERFMI_OK;
}

int nerfmi_profile_stop(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_profile_on = false;
    return NERFMI_OK;
}

// Waits for the recorded spans and writes one line per (kernel tag, units per launch):
//     "<tag>\t<units>\t<launches>\t<total ms>\n"
// Returns the number of bytes the full report needs (excluding the terminating 0); writes at most cap-1 of them.
int64_t nerfmi_profile_report(char *buf, size_t cap) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    struct Agg { const char *tag; int64_t units; int64_t n; double ms; };
    std::vector<Agg> agg;
    for (const Span &s : g_spans) {
        if (hipEventSynchronize(s.b) != hipSuccess) { (void)hipGetLastError(); continue; }
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.a, s.b) != hipSuccess) { (void)hipGetLastError(); continue; }
        bool found = false;
        for (Agg &a : agg)
            if (a.units == s.units && std::string(a.tag) == s.tag) { a.n++; a.ms += ms; found = true; break; }
        if (!found) agg.push_back(Agg{s.tag, s.units, 1, ms});
    }
    std::string out;
    char line[256];
    for (const Agg &a : agg) {
        snprintf(line, sizeof line, "%s\t%lld\t%lld\t%.6f\n", a.tag, (long long)a.units, (long long)a.n, a.ms);
        out += line;
    }
    if (buf && cap) {
        const size_t n = out.size() < cap - 1 ? out.size() : cap - 1;
        memcpy(buf, out.data(), n);
        buf[n] = 0;
    }
    return (int64_t)out.size();
}

// ---------------------------------------------------------------------------------------------------------------------
// nerfmi_render_rays_fused
// ---------------------------------------------------------------------------------------------------------------------
static inline size_t up4(size_t n) { return (n + 3) & ~(size_t)3; }     // 16-byte aligned sub-buffers

size_t nerfmi_render_rays_workspace_floats(int n_rays, int n_samples, int n_importance, int test_time) {
    const size_t N = n_rays < 0 ? 0 : (size_t)n_rays, S = (size_t)(n_samples < 1 ? 1 : n_samples);
    const size_t F = (size_t)(n_importance < 0 ? 0 : n_importance);
    size_t n = up4(N * S) /* z */ + up4(N * S) /* weights_coarse */ + up4(N * S * (test_time ? 1 : 4)) /* coarse field */;
    if (F) n += up4(N * (S + F)) /* z_fine */ + up4(N * (S + F) * 4) /* fine field */;
    return n;
}

int nerfmi_render_rays_fused(int field_kind, const float *packed_coarse, const float *packed_fine,
                             const float *cond_coarse, const float *cond_fine, const float *rays, int n_rays,
                             int n_samples, int n_importance, int use_disp, float perturb, float noise_std,
                             int white_back, int test_time, uint64_t seed, uint64_t offset, float *workspace,
                             float *rgb_coarse, float *depth_coarse, float *opacity_coarse, float *rgb_fine,
                             float *depth_fine, float *opacity_fine, nerfmi_stream_t stream) {
    NERFMI_REQUIRE(field_kind == 0 || field_kind == 1, "render_rays_fused: field_kind must be 0 (NeRF) or 1 (FiLM-SIREN)");
    NERFMI_REQUIRE(n_rays >= 0 && n_samples >= 1 && n_importance >= 0, "render_rays_fused: bad sizes");
    if (n_rays == 0) return NERFMI_OK;
    NERFMI_REQUIRE(packed_coarse && rays && workspace && opacity_coarse, "render_rays_fused: null pointer");
    NERFMI_REQUIRE(test_time || (rgb_coarse && depth_coarse), "render_rays_fused: rgb_coarse / depth_coarse required unless test_time");
    NERFMI_REQUIRE(n_importance == 0 || (packed_fine && rgb_fine && depth_fine && opacity_fine),
                   "render_rays_fused: the fine pass needs packed_fine and the three fine outputs");
    NERFMI_REQUIRE(field_kind == 0 || (cond_coarse && (n_importance == 0 || cond_fine)),
                   "render_rays_fused: the FiLM-SIREN field needs its conditioning rows [frequencies 2304 | phase_shifts 2304]");
    const int N = n_rays, S = n_samples, F = n_importance;
    float *z = workspace;
    float *w_coarse = z + up4((size_t)N * S);
    float *field_c = w_coarse + up4((size_t)N * S);
    float *z_fine = field_c + up4((size_t)N * S * (test_time ? 1 : 4));
    float *field_f = z_fine + up4((size_t)N * (S + F));
    int rc;
    // rendering.py:207-222
    if (perturb > 0) rc = nerfmi_sample_stratified_philox(rays, seed, offset, N, S, use_disp, perturb, z, stream);
    else rc = nerfmi_sample_stratified(rays, nullptr, N, S, use_disp, 0.f, z, stream);
    if (rc) return rc;
    auto field = [&](const float *packed, const float *cond, const float *zz, int P, int sigma_only, float *out) {
        if (field_kind == 0) return nerfmi_nerf_forward_rays(packed, rays, zz, N, P, sigma_only, out, nullptr, stream);
        return nerfmi_siren_forward_rays(packed, rays, zz, cond, cond + 2304, N, P, (int64_t)N, sigma_only, out, stream);
    };
    auto composite = [&](const float *fld, int sigma_only, const float *zz, int P, int segment, float *w, float *rgb,
                         float *depth, float *op) {
        if (noise_std != 0.f)
            return nerfmi_composite_philox(fld, sigma_only, zz, rays, seed, offset, segment, noise_std, N, P, white_back, w, rgb,
                                           depth, op, stream);
        return nerfmi_composite(fld, sigma_only, zz, rays, nullptr, 0.f, N, P, white_back, w, rgb, depth, op, stream);
    };
    // rendering.py:227-241: test_time -> sigma-only coarse pass (weights_only), else the full coarse pass
    if ((rc = field(packed_coarse, cond_coarse, z, S, test_time ? 1 : 0, field_c))) return rc;
    if ((rc = composite(field_c, test_time ? 1 : 0, z, S, 1, w_coarse, test_time ? nullptr : rgb_coarse,
                        test_time ? nullptr : depth_coarse, opacity_coarse)))
        return rc;
    if (F == 0) return NERFMI_OK;
    // rendering.py:242-247 (det = perturb == 0), :249-256
    if (perturb != 0) rc = nerfmi_importance_resample_philox(z, w_coarse, seed, offset, N, S, F, nullptr, z_fine, stream);
    else rc = nerfmi_importance_resample(z, w_coarse, nullptr, N, S, F, nullptr, z_fine, stream);
    if (rc) return rc;
    if ((rc = field(packed_fine, cond_fine, z_fine, S + F, 0, field_f))) return rc;
    return composite(field_f, 0, z_fine, S + F, 3, nullptr, rgb_fine, depth_fine, opacity_fine);
}

}  // extern "C"
