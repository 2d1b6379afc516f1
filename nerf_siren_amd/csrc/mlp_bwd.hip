// NeRF MLP backward (placeholder until the dX-chain / dW kernels land).
#include "common.h"
#include "mlp_layout.h"

using namespace nerfmi;

extern "C" {

size_t nerfmi_nerf_backward_workspace_floats(int64_t n_points) { (void)n_points; return 4; }

int nerfmi_nerf_backward_rays(const float *, const float *, const float *, int, int, const float *, const float *,
                              float *const *, float *, nerfmi_stream_t) {
    set_error("nerf_backward_rays: not implemented yet");
    return NERFMI_E_UNSUPPORTED;
}

}  // extern "C"
