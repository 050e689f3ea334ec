// Two-kernel backward of the flow layers + the parameter-gradient contraction (instances + launchers); see flow_launch.h.
#define MF_TU_OUTER_ACCUM 1
#include "flow_launch.h"

namespace mf {

int launch_rqs_bwd2(int bins, int L, int block, int grid, size_t smem, void* stream, const float* image, int d, const float* x,
                    int64_t n, const float* gy, const float* glogp, float* gx, float* scratch, const Sparsity& sp) {
#define XB(KK, LL, BB)                                                                                                \
    if (block == BB) {                                                                                                \
        MF_ALLOW_DYN_SMEM((rqs_layer_bwd_kernel<KK, LL, BB>), smem);                                                  \
        MF_LAUNCH((rqs_layer_bwd_kernel<KK, LL, BB>), grid, BB, smem, stream, image, d, x, n, gy, glogp, gx, scratch,  \
                  sp, bins);                                                                                          \
        return 0;                                                                                                     \
    }
#define X(KK, LL)                                                                                                     \
    if (rqs_case_matches(KK, bins) && L == LL) {                                                                      \
        XB(KK, LL, 256) XB(KK, LL, 512)                                                                               \
        return LAUNCH_NO_INSTANCE;                                                                                    \
    }
    MF_RQS_CASES(X)
#undef X
#undef XB
    return LAUNCH_NO_INSTANCE;
}

int launch_affine_bwd2(int L, int grid, size_t smem, void* stream, const float* image, int d, const float* x, int64_t n,
                       const float* gy, const float* glogp, float* gx, float* scratch, const Sparsity& sp) {
#define X(LL)                                                                                                         \
    if (L == LL) {                                                                                                    \
        MF_ALLOW_DYN_SMEM((affine_layer_bwd_kernel<LL>), smem);                                                       \
        MF_LAUNCH((affine_layer_bwd_kernel<LL>), grid, FLOW_BLOCK, smem, stream, image, d, x, n, gy, glogp, gx,        \
                  scratch, sp);                                                                                       \
        return 0;                                                                                                     \
    }
    MF_AFFINE_CASES(X)
#undef X
    return LAUNCH_NO_INSTANCE;
}

int launch_outer_accum(int grid_x, int nwaves, void* stream, const float* scratch, const float* x, int64_t n, int d, int L,
                       int nblk, float* gslab, int accumulate, const Sparsity& sp) {
    MF_LAUNCH(outer_accum_kernel, dim3((unsigned)grid_x, 2), 64 * nwaves, 0, stream, scratch, x, n, d, L, nblk, gslab,
              accumulate, sp);
    return 0;
}

}  // namespace mf
