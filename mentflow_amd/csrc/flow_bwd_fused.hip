// Fused backward kernels of the flow layers (instances + launchers); see flow_launch.h.
// Compiled once per MF_FUSED_SAVED: 0 = the backward recomputes the conditioner from the layer input; 1 = the forward handed
// over the hidden tiles; 2 = also the conditioner outputs (rqs_layer_bwd_fused_kernel's SAVED parameter).
#include "flow_launch.h"

#ifndef MF_FUSED_SAVED
#define MF_FUSED_SAVED 0
#endif

namespace mf {

#define MF_PASTE2(a, b) a##b
#define MF_PASTE(a, b) MF_PASTE2(a, b)
int MF_PASTE(launch_rqs_bwd_fused_s, MF_FUSED_SAVED)(int bins, int L, int grid, size_t smem, void* stream, const float* image,
                                                     int d, const float* x, int64_t n, const float* gy, const float* glogp,
                                                     float* gx, float* gslab, int accumulate, const Sparsity& sp,
                                                     const float* act) {
#define XF(KK, LL)                                                                                                    \
    if (rqs_case_matches(KK, bins) && L == LL) {                                                                      \
        MF_ALLOW_DYN_SMEM((rqs_layer_bwd_fused_kernel<KK, LL, MF_FUSED_SAVED>), smem);                                \
        MF_LAUNCH((rqs_layer_bwd_fused_kernel<KK, LL, MF_FUSED_SAVED>), grid, FB_BLOCK, smem, stream, image, d, x, n, gy, \
                  glogp, gx, gslab, accumulate, sp, bins, act);                                                       \
        return 0;                                                                                                     \
    }
#if MF_FUSED_SAVED == 0
    MF_RQS_CASES(XF)
#else
    MF_RQS_SAVED_CASES(XF)
#endif
#undef XF
    return LAUNCH_NO_INSTANCE;
}

#if MF_FUSED_SAVED == 0

int launch_affine_bwd_fused(int L, int grid, size_t smem, void* stream, const float* image, int d, const float* x, int64_t n,
                            const float* gy, const float* glogp, float* gx, float* gslab, int accumulate) {
#define XF(LL)                                                                                                        \
    if (L == LL) {                                                                                                    \
        MF_ALLOW_DYN_SMEM((affine_layer_bwd_fused_kernel<LL>), smem);                                                 \
        MF_LAUNCH((affine_layer_bwd_fused_kernel<LL>), grid, FB_BLOCK, smem, stream, image, d, x, n, gy, glogp, gx,    \
                  gslab, accumulate);                                                                                 \
        return 0;                                                                                                     \
    }
    MF_AFFINE_CASES(XF)
#undef XF
    return LAUNCH_NO_INSTANCE;
}
#endif

}  // namespace mf

// diagnostic build (tools/fb_diag.py): the stamps of the hand-off level named by -DMF_WS_DIAG_LEVEL (default 0)
#ifndef MF_WS_DIAG_LEVEL
#define MF_WS_DIAG_LEVEL 0
#endif
#if defined(MF_WS_DIAG) && !defined(MF_EMU) && MF_FUSED_SAVED == MF_WS_DIAG_LEVEL
extern "C" int mf_debug_ws_read(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(mf::g_ws_diag), sizeof(unsigned long long) * mf::NUM_CU * 4 * 16) == hipSuccess ? 0 : 1;
}
#endif
