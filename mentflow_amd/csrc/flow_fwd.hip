// Forward kernels of the flow layers (instances + launchers); see flow_launch.h.
#include "flow_launch.h"

namespace mf {

int launch_rqs_fwd(int bins, int L, int block, int grid, size_t smem, void* stream, const float* image, int d, const float* x,
                   int64_t n, float* y, const float* logp_in, float* logp_out, int init_logp, const Sparsity& sp, float* act,
                   int act_level) {
#define XB(KK, LL, BB)                                                                                                \
    if (block == BB) {                                                                                                \
        MF_ALLOW_DYN_SMEM((rqs_layer_fwd_kernel<KK, LL, BB>), smem);                                                  \
        MF_LAUNCH((rqs_layer_fwd_kernel<KK, LL, BB>), grid, BB, smem, stream, image, d, x, n, y, logp_in, logp_out,   \
                  init_logp, sp, bins, act, act_level);                                                               \
        return 0;                                                                                                     \
    }
#define X(KK, LL)                                                                                                     \
    if (rqs_case_matches(KK, bins) && L == LL) {                                                                      \
        XB(KK, LL, 256) XB(KK, LL, 512) XB(KK, LL, 1024)                                                              \
        return LAUNCH_NO_INSTANCE;                                                                                    \
    }
    MF_RQS_CASES(X)
#undef X
#undef XB
    return LAUNCH_NO_INSTANCE;
}

int launch_affine_fwd(int L, int grid, size_t smem, void* stream, const float* image, int d, const float* x, int64_t n, float* y,
                      const float* logp_in, float* logp_out, int init_logp, const Sparsity& sp) {
#define X(LL)                                                                                                         \
    if (L == LL) {                                                                                                    \
        MF_ALLOW_DYN_SMEM((affine_layer_fwd_kernel<LL, 1024>), smem);                                                 \
        MF_LAUNCH((affine_layer_fwd_kernel<LL, 1024>), grid, 1024, smem, stream, image, d, x, n, y, logp_in, logp_out, \
                  init_logp, sp);                                                                                     \
        return 0;                                                                                                     \
    }
    MF_AFFINE_CASES(X)
#undef X
    return LAUNCH_NO_INSTANCE;
}

}  // namespace mf
