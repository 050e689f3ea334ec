// Inverse kernels of the flow layers (instances + launchers); see flow_launch.h.
#include "flow_launch.h"

namespace mf {

int launch_rqs_inv(int bins, int L, int grid, size_t smem, void* stream, const float* image, int d, const float* y, int64_t n,
                   float* x, const Sparsity& sp, const InvOrder& io) {
#define X(KK, LL)                                                                                                     \
    if (rqs_case_matches(KK, bins) && L == LL) {                                                                      \
        MF_ALLOW_DYN_SMEM((layer_inv_kernel<KK, LL>), smem);                                                          \
        MF_LAUNCH((layer_inv_kernel<KK, LL>), grid, INV_BLOCK, smem, stream, image, d, y, n, x, sp, io, bins);         \
        return 0;                                                                                                     \
    }
    MF_RQS_CASES(X)
#undef X
    return LAUNCH_NO_INSTANCE;
}

int launch_affine_inv(int L, int grid, size_t smem, void* stream, const float* image, int d, const float* y, int64_t n, float* x,
                      const Sparsity& sp, const InvOrder& io) {
#define X(LL)                                                                                                         \
    if (L == LL) {                                                                                                    \
        MF_ALLOW_DYN_SMEM((layer_inv_kernel<0, LL>), smem);                                                           \
        MF_LAUNCH((layer_inv_kernel<0, LL>), grid, INV_BLOCK, smem, stream, image, d, y, n, x, sp, io, 0);             \
        return 0;                                                                                                     \
    }
    MF_AFFINE_CASES(X)
#undef X
    return LAUNCH_NO_INSTANCE;
}

}  // namespace mf
