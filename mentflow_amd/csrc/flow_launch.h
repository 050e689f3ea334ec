// Launchers of the flow-layer kernels, one translation unit per kernel family so that the (long) device compiles of the
// template instances run in parallel and a change to one family rebuilds one object:
//   flow.hip            C ABI of the flow layers (argument checks, variant choice, gradient reduce)  -> calls the launchers
//   flow_fwd.hip        forward kernels (RQS + affine)
//   flow_bwd_fused.hip  fused backward (+ parameter gradients), compiled once per hand-off level MF_FUSED_SAVED = 0, 1, 2
//   flow_bwd2.hip       two-kernel backward + the parameter-gradient contraction (outer_accum)
//   flow_inv.hip        inverse (density of a point)
// Every launcher returns 0 once the kernel is enqueued (the caller runs check_launch) and 2 when no instance is built for
// (bins, hidden_layers); `smem` / `grid` are computed by the caller.
#pragma once
#include "flow_kernels.inc"

namespace mf {

// Built spline instances: bins 20 (the reference's value, experiments/setup.py:119-121) and 8 (zuko's default) at compile
// time; every other 2 <= bins <= 21 through the run-time instance (RQS_ANY: slots laid out for 21 bins, slower).
// hidden_layers: 3 (the reference's default, config/gen/flow.yaml:2) and 2 are the tuned instances; 1 and 4 are built as well
// (mentflow/generate/build.py:36-38 takes the depth from the config) — 4 layers fit the fused backward's LDS budget only for small d
// and otherwise take the two-kernel backward, neither has a hand-off level.
#define MF_RQS_CASES(X) X(20, 3) X(20, 2) X(8, 3) X(8, 2) X(RQS_ANY, 3) X(RQS_ANY, 2) X(20, 4) X(20, 1) X(8, 4) X(8, 1) X(RQS_ANY, 4) X(RQS_ANY, 1)
#define MF_AFFINE_CASES(X) X(3) X(2) X(4) X(1)
inline bool rqs_case_matches(int KK, int bins) {
    return KK == RQS_ANY ? (bins != 20 && bins != 8 && bins >= 2 && bins <= RQS_KMAX) : bins == KK;
}

constexpr int LAUNCH_NO_INSTANCE = 2;

int launch_rqs_fwd(int bins, int L, int block, int grid, size_t smem, void* stream, const float* image, int d, const float* x,
                   int64_t n, float* y, const float* logp_in, float* logp_out, int init_logp, const Sparsity& sp, float* act,
                   int act_level);
int launch_affine_fwd(int L, int grid, size_t smem, void* stream, const float* image, int d, const float* x, int64_t n, float* y,
                      const float* logp_in, float* logp_out, int init_logp, const Sparsity& sp);

// one function per hand-off level (MF_FUSED_SAVED = 0, 1, 2: separate objects); levels 1 and 2 are built for the compile-time
// spline instances only (bins 20 and 8)
#define MF_DECL_FUSED(S)                                                                                              \
    int launch_rqs_bwd_fused_s##S(int bins, int L, int grid, size_t smem, void* stream, const float* image, int d,    \
                                  const float* x, int64_t n, const float* gy, const float* glogp, float* gx, float* gslab, \
                                  int accumulate, const Sparsity& sp, const float* act);
MF_DECL_FUSED(0) MF_DECL_FUSED(1) MF_DECL_FUSED(2)
#undef MF_DECL_FUSED
#define MF_RQS_SAVED_CASES(X) X(20, 3) X(20, 2) X(8, 3) X(8, 2)
inline bool rqs_saved_instance(int bins, int L) { return (bins == 20 || bins == 8) && (L == 2 || L == 3); }
int launch_affine_bwd_fused(int L, int grid, size_t smem, void* stream, const float* image, int d, const float* x, int64_t n,
                            const float* gy, const float* glogp, float* gx, float* gslab, int accumulate);

int launch_rqs_bwd2(int bins, int L, int block, int grid, size_t smem, void* stream, const float* image, int d, const float* x,
                    int64_t n, const float* gy, const float* glogp, float* gx, float* scratch, const Sparsity& sp);
int launch_affine_bwd2(int L, int grid, size_t smem, void* stream, const float* image, int d, const float* x, int64_t n,
                       const float* gy, const float* glogp, float* gx, float* scratch, const Sparsity& sp);
int launch_outer_accum(int grid_x, int nwaves, void* stream, const float* scratch, const float* x, int64_t n, int d, int L,
                       int nblk, float* gslab, int accumulate, const Sparsity& sp);

int launch_rqs_inv(int bins, int L, int grid, size_t smem, void* stream, const float* image, int d, const float* y, int64_t n,
                   float* x, const Sparsity& sp, const InvOrder& io);
int launch_affine_inv(int L, int grid, size_t smem, void* stream, const float* image, int d, const float* y, int64_t n, float* x,
                      const Sparsity& sp, const InvOrder& io);

}  // namespace mf
