/* mentflow_hip.h — C ABI of libmentflow_hip.so, the MI355X (gfx950) implementation of the MENT-Flow hot path.
 *
 * The reference (austin-hoover/ment-flow) is pure Python: it has no FFI of its own.  Each entry point below
 * replaces the chain of eager PyTorch ops cited next to it (paths relative to the reference repository);
 * INTEGRATION.md shows the ctypes stub a maintainer would add on the reference side.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless stated otherwise; float = IEEE binary32; row-major.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises.
 *   - return value: 0 on success, non-zero on error (mf_last_error() gives the message, thread-local).
 *   - no entry point allocates: scratch is passed in by the caller (sizes from the *_floats helpers).
 */
#ifndef MENTFLOW_HIP_H
#define MENTFLOW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MF_ABI_VERSION 5
#define MF_ENTROPY_SCRATCH_DOUBLES 2048

int mf_abi_version(void);
const char* mf_last_error(void);
/* 0 for the gfx950 library.  (1 only for the host-emulated kernel build under tests/emu, which is test
 * infrastructure and is never loaded by the product package.)                                               */
int mf_is_emulation(void);

/* Per-kernel timing for bench.py: while enabled, every launch of the kernels listed below is bracketed by HIP
 * events recorded on its own stream.  mf_prof_report(id) synchronises those events and returns the summed
 * duration (ms) and the number of launches.  ids: 0 flow layer fwd, 1 flow layer bwd, 2 parameter-gradient
 * contraction (outer_accum), 3 kde1d fwd, 4 kde1d bwd, 5 kde2d fwd, 6 kde2d bwd.                              */
int mf_prof_enable(int enable);
int mf_prof_report(int kernel_id, double* total_ms, int64_t* launches);

/* ------------------------------------------------------------------------------------------------------------
 * Generic index gather (weight packing / gradient unpacking for the flow kernels).
 *   dst[j] = idx[j] >= 0 ? src[idx[j]] : 0        (accumulate != 0:  dst[j] += ...)
 * Replaces `mask * weight` in zuko.nn.MaskedLinear.forward (masked-out entries have idx = -1) and the
 * reshape/permute of conditioner outputs (zuko MaskedAutoregressiveTransform.meta).                          */
int mf_gather_f32(const float* src, const int32_t* idx, float* dst, int64_t n, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Autoregressive rational-quadratic-spline flow layer  (zuko NSF layer as inverted by
 * mentflow/generate/build.py:42-43; sampling direction of mentflow/generate/flows/zuko.py:24-29).
 *
 * `image` is the packed per-layer weight image (mf_flow_image_floats floats), laid out exactly as it sits in
 * LDS: [W0 64 x S0][b0 64]{[W_l 64 x 65][b_l 64]}(hidden_layers-1)[W_out d x 64 x 65][b_out d x 64];
 * see mentflow_amd/generate/packing.py for the row permutation of W_out.  hidden width is 64.
 *
 * fwd:  y[n,d] = RQS(x; MLP(x)),   logp_out[n] = (init_logp ? logN(x) : logp_in[n]) - sum_i ladj_i
 *       (logN(x) = -1/2 |x|^2 - d/2 log 2pi: zuko DiagNormal.log_prob of the base draw, first layer only).
 * order: HOST pointer to the d autoregressive orders of this layer (zuko `order` buffer), or NULL.  With it the kernels
 *       skip the MFMA k-steps that only multiply masked-out (zero) weights — the image must then hold the hidden
 *       units sorted by dependency class as packing.py lays them out; NULL = dense products (any image).
 * bwd:  given gy[n,d] = dL/dy and glogp[n] = dL/dlogp, writes gx[n,d] = dL/dx (NULL for the first layer: the
 *       base draw needs no gradient) and the parameter gradients dL/d(image) as PARTIAL SUMS, one per workgroup, into
 *       the rows of gslab[slab_rows][mf_flow_image_floats] (image layout per row) with plain stores — no float
 *       atomics, so the result is bitwise reproducible.  slab_rows must equal mf_flow_bwd_slab_rows(n, ...) (the
 *       number of workgroup columns this call launches).  accumulate = 0: every row is overwritten (first chunk of a
 *       backward pass: no zeroing needed); accumulate != 0: each workgroup adds to its own row (later chunks of the
 *       same size class).  mf_flow_grad_reduce sums the rows in a fixed order into the flat parameter gradient.
 *       `scratch` needs mf_flow_bwd_scratch_floats(n,...,order) floats: 0 when the call takes the fused kernel
 *       (parameter gradients inside the backward kernel: large batches with `order`), else (2 hidden_layers + d) * 64
 *       floats per particle for the hand-off to the parameter-gradient kernel (callers then process the batch in
 *       chunks to bound it).                                                                                    */
int64_t mf_flow_image_floats(int d, int hidden_layers);
/* Spline bins: 20 (the reference's value, experiments/setup.py:119-121) and 8 (zuko's default) are compile-time instances;
 * every other 2 <= bins <= 21 runs through a run-time instance whose last-layer block is laid out for 21 bins.  Returns the
 * slot (0..31) of a lane half's first derivative logit in the packed block — `bins` for the compile-time instances, 21
 * for the run-time one — or -1 if there is no kernel for that number of bins (the packer needs it: packing.py).        */
int mf_flow_rqs_deriv_slot(int bins);
/* Backward variant of the flow layers: -1 = default (environment variable MENTFLOW_BWD_FUSED, read once; unset = fused),
 * 0 = two-kernel path (backward + parameter-gradient contraction through an HBM scratch), 1 = fused kernel.  Process-wide;
 * tests use it to run both variants in one process.                                                                */
int mf_flow_set_bwd_variant(int variant);
int64_t mf_flow_bwd_scratch_floats(int64_t n, int d, int hidden_layers, const int32_t* order);
int mf_flow_rqs_layer_fwd(const float* image, int d, int hidden_layers, int bins, const int32_t* order,
                          const float* x, int64_t n, float* y, const float* logp_in, float* logp_out, int init_logp,
                          void* stream);
int mf_flow_bwd_slab_rows(int64_t n, int d, int hidden_layers, const int32_t* order);
int mf_flow_rqs_layer_bwd(const float* image, int d, int hidden_layers, int bins, const int32_t* order,
                          const float* x, int64_t n, const float* gy, const float* glogp, float* gx, float* gslab,
                          int slab_rows, int accumulate, float* scratch, int64_t scratch_floats, void* stream);

/* Activation hand-off from the training forward to the fused backward (ABI 4).  The eager reference keeps every activation
 * of zuko's conditioner for autograd (reached from mentflow/generate/flows/zuko.py:24-26).  mf_flow_rqs_layer_fwd saves nothing
 * (evaluation / no-grad; its backward mf_flow_rqs_layer_bwd recomputes the conditioner); mf_flow_rqs_layer_fwd_save is the same
 * forward that ALSO writes, per 32-particle tile, conditioner activations into `act` in the register layout of the fused
 * backward kernel:
 *   level 1: the post-ReLU hidden tiles of levels 1 .. hidden_layers-1 (level 0 is 8 MFMAs from the layer input: recomputed)
 *            64 floats per particle and level:                                                  512 B at hidden_layers = 3
 *   level 2: level 1 + the conditioner outputs (spline logits) of the d-1 features that have a conditioner, the 3 bins / 2
 *            slots per lane half the spline reads:                            + 240 (d-1) B at 20 bins: 1 712 B at d = 6
 * and mf_flow_rqs_layer_bwd_saved is the fused backward that loads them instead of re-running the conditioner's forward
 * chains (a quarter of that MFMA-bound kernel).  `act` holds mf_flow_rqs_act_floats(n, d, hidden_layers, bins, level) floats
 * per layer and belongs to ONE (layer, batch) pair: forward writes it, the backward of the same layer and particles reads it.
 * mf_flow_rqs_act_level(...) = highest level the built kernels take for the configuration under the current backward variant:
 * 2 with `order`, d <= 6 and bins in {20, 8}; 0 otherwise (two-kernel backward, run-time-bins instance) — callers then use
 * mf_flow_rqs_layer_fwd / _bwd.  Gradients are bitwise identical at every level (same arithmetic on the same values).      */
int mf_flow_rqs_act_level(int d, int hidden_layers, int bins, const int32_t* order);
int64_t mf_flow_rqs_act_floats(int64_t n, int d, int hidden_layers, int bins, int level);
int mf_flow_rqs_layer_fwd_save(const float* image, int d, int hidden_layers, int bins, const int32_t* order,
                               const float* x, int64_t n, float* y, const float* logp_in, float* logp_out, int init_logp,
                               float* act, int64_t act_floats, int level, void* stream);
int mf_flow_rqs_layer_bwd_saved(const float* image, int d, int hidden_layers, int bins, const int32_t* order,
                                const float* x, int64_t n, const float* gy, const float* glogp, float* gx, float* gslab,
                                int slab_rows, int accumulate, const float* act, int64_t act_floats, int level, void* stream);

/* Sum of the slab rows, all layers in one launch: gslab[layers][rows][image_floats] ->
 * gflat[j] = sum_{r < rows} gslab[t][r][pos] with grad_index[j] = t * image_floats + pos (-1: parameter without an
 * image slot -> 0).  Fixed summation order, fp64 accumulator: replaces the autograd accumulation of
 * d(mask * W)/dW of zuko's MaskedLinear (reached from mentflow/generate/flows/zuko.py:24-26).                       */
int mf_flow_grad_reduce(const float* gslab, int layers, int rows, int64_t image_floats, const int32_t* grad_index,
                        float* gflat, int64_t numel, void* stream);

/* Inverse of one layer, x = T^-1(y): the d autoregressive passes of zuko AutoregressiveTransform._inverse
 * (mentflow/generate/flows/zuko.py:21-22,31-32 -> log_prob / inverse of an arbitrary point).  `order` is required. */
int mf_flow_rqs_layer_inv(const float* image, int d, int hidden_layers, int bins, const int32_t* order,
                          const float* y, int64_t n, float* x, void* stream);

/* Affine (MAF) variant: zuko MonotonicAffineTransform, y = x*exp(s~)+t, s~ = s/(1+|s/log(1e-3)|), ladj = s~
 * (mentflow/generate/build.py:28 "maf"; BASELINE config C1).  Same image layout with ONE output block whose slot i of
 * lane half 0 is shift_i and of half 1 is scale_i.                                                             */
int64_t mf_flow_affine_image_floats(int d, int hidden_layers);
int64_t mf_flow_affine_bwd_scratch_floats(int64_t n, int hidden_layers);
int mf_flow_affine_layer_fwd(const float* image, int d, int hidden_layers, const int32_t* order, const float* x,
                             int64_t n, float* y, const float* logp_in, float* logp_out, int init_logp, void* stream);
int mf_flow_affine_bwd_slab_rows(int64_t n);
int mf_flow_affine_layer_bwd(const float* image, int d, int hidden_layers, const int32_t* order, const float* x,
                             int64_t n, const float* gy, const float* glogp, float* gx, float* gslab, int slab_rows,
                             int accumulate, float* scratch, int64_t scratch_floats, void* stream);
int mf_flow_affine_layer_inv(const float* image, int d, int hidden_layers, const int32_t* order, const float* y,
                             int64_t n, float* x, void* stream);

/* Wide conditioners (ABI 5): the same layers for hidden widths up to 128 units and / or up to 16 features — shapes whose weights
 * do not fit the 160 KiB of LDS the 64-wide entry points above keep them in (mentflow/generate/build.py:36-38 takes hidden_units
 * and hidden_layers from the config; zuko accepts any).  One family for both transforms: bins = 0 selects the affine (MAF)
 * transform, 2 <= bins <= 21 the rational-quadratic spline.  Differences to the entry points above:
 *   - `image` (mf_flow_wide_image_floats(hidden_layers, nblk) floats, nblk = d for the spline, 1 for affine) holds every weight
 *     matrix twice, as forward and as transposed 32 x 32 MFMA FRAGMENT blocks, and stays in global memory (L2-resident); its layout
 *     is documented in mentflow_amd/csrc/flow_wide.hip and produced by mentflow_amd/generate/packing.py (wide_image_index);
 *   - gslab rows have mf_flow_wide_grad_floats(hidden_layers, nblk) floats in NATURAL order (padded physical rows / columns);
 *     pass that number as `image_floats` to mf_flow_grad_reduce;
 *   - `hidden` = hidden units (1 .. 128; all hidden layers share it), hidden_layers 1 .. 4, d 1 .. 16 (mf_flow_wide_limits);
 *   - the backward is always the two-kernel form: `scratch` needs mf_flow_wide_bwd_scratch_floats floats.                        */
int mf_flow_wide_limits(int* max_features, int* max_hidden, int* max_hidden_layers);
int64_t mf_flow_wide_image_floats(int hidden_layers, int nblk);
int64_t mf_flow_wide_grad_floats(int hidden_layers, int nblk);
int mf_flow_wide_layer_fwd(const float* image, int d, int hidden, int hidden_layers, int bins, const int32_t* order,
                           const float* x, int64_t n, float* y, const float* logp_in, float* logp_out, int init_logp,
                           void* stream);
int64_t mf_flow_wide_bwd_scratch_floats(int64_t n, int d, int hidden_layers, int bins);
int mf_flow_wide_bwd_slab_rows(int64_t n);
int mf_flow_wide_layer_bwd(const float* image, int d, int hidden, int hidden_layers, int bins, const int32_t* order,
                           const float* x, int64_t n, const float* gy, const float* glogp, float* gx, float* gslab,
                           int slab_rows, int accumulate, float* scratch, int64_t scratch_floats, void* stream);
int mf_flow_wide_layer_inv(const float* image, int d, int hidden, int hidden_layers, int bins, const int32_t* order,
                           const float* y, int64_t n, float* x, void* stream);
/* Activation hand-off of the wide family (the counterpart of mf_flow_rqs_layer_fwd_save / _bwd_saved above): a training forward
 * stores every hidden level (in the layout of the backward's scratch tiles, so that the parameter-gradient contraction reads them in
 * place) and every output block's conditioner outputs into `act` (mf_flow_wide_act_floats(n, d, hidden_layers, bins) floats per layer
 * and batch: 3 KB per particle at 128 units, d = 6, three hidden layers); mf_flow_wide_layer_bwd_saved then neither recomputes the
 * conditioner nor re-writes its activations (half of the backward's MFMAs and a third of its scratch traffic).  The whole batch in one
 * call: `x`, `act` and the scratch must cover the same n particles.  Same results as mf_flow_wide_layer_fwd / _bwd.                 */
int64_t mf_flow_wide_act_floats(int64_t n, int d, int hidden_layers, int bins);
int mf_flow_wide_layer_fwd_save(const float* image, int d, int hidden, int hidden_layers, int bins, const int32_t* order,
                                const float* x, int64_t n, float* y, const float* logp_in, float* logp_out, int init_logp,
                                float* act, int64_t act_floats, void* stream);
int mf_flow_wide_layer_bwd_saved(const float* image, int d, int hidden, int hidden_layers, int bins, const int32_t* order,
                                 const float* x, int64_t n, const float* gy, const float* glogp, float* gx, float* gslab,
                                 int slab_rows, int accumulate, float* scratch, int64_t scratch_floats, const float* act,
                                 int64_t act_floats, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Fused linear projection + 1-D Gaussian-KDE histogram over P projections.
 * Replaces, for all P transforms at once: `x.clone() @ M.T` (mentflow/simulate/transform.py:67-68, only the
 * consumed output row V_p of each matrix), Histogram1D.project (mentflow/diagnostics/diagnostics.py:116-122) and the
 * residuals/exp/mean of marginal_pdf (mentflow/diagnostics/histogram.py:37-39).
 *   S[p,k] = sum_n exp(-1/2 ((x_n . V_p - coords[k]) / sigma)^2)        (raw sums: the 1/N and the
 *   normalisation of histogram.py:39-43 happen in mf_hist_norm_discrepancy_fwd, after any cross-GPU sum)
 * Only the 2*radius+1 bins around each projected particle are visited (dropped kernel values are below
 * exp(-(radius+1/2)^2 delta^2 / (2 sigma^2)), 3e-18 for the reference's sigma = delta/2 and radius 4); pass
 * radius >= B for the dense sum.  radius and sigma are independent arguments: every pair is evaluated correctly.
 * (radius = 4 with delta / sigma in [2, 2.6] — the reference's bandwidths 0.39 .. 0.5 bins — takes a specialised window
 * with factorised tail weights and, in 2-D, without the 12 corner cells whose weight is below the 2^-50 quantum for
 * every position of the particle; the kernels test the ratio themselves and run the plain loop otherwise.)       */
/* ws: mf_proj_kde_ws_bytes(P, bins) bytes of scratch: one 64-bit INTEGER accumulator per bin.
 * Weights are summed as fixed-point integers at every level (2^-50 units with 64-bit LDS atomics inside a workgroup,
 * one 64-bit integer global atomic per bin and workgroup in 2^(s-50) units, s = max(0, ceil(log2 n) - 13): exact up to
 * 8192 particles per call): order independent, bitwise reproducible.                                            */
int64_t mf_proj_kde_ws_bytes(int P, int bins);
int mf_proj_kde1d_fwd(const float* x, int64_t n, int d, const float* V, int P, const float* coords, int B,
                      float sigma, int radius, float* S, void* ws, void* stream);
/* gx[n,d] (+)= sum_p V_p * sum_k gS[p,k] * K_npk * (-(u_np - c_k)/sigma^2)             */
int mf_proj_kde1d_bwd(const float* x, int64_t n, int d, const float* V, int P, const float* coords, int B,
                      float sigma, int radius, const float* gS, float* gx, int accumulate, void* stream);

/* 2-D variant: two projection vectors per transform (rows `axis[0]`, `axis[1]` of the matrix); replaces
 * Histogram2D.project + marginal_pdf x2 + joint_pdf's Kx^T Ky (histogram.py:47-74,89-101).
 *   S[p,a,b] = sum_n Kx_na Ky_nb   (no 1/N, as the reference)                                                */
int mf_proj_kde2d_fwd(const float* x, int64_t n, int d, const float* V0, const float* V1, int P,
                      const float* coords_x, int Bx, float sigma_x, int radius_x, const float* coords_y, int By,
                      float sigma_y, int radius_y, float* S, void* ws, void* stream);
int mf_proj_kde2d_bwd(const float* x, int64_t n, int d, const float* V0, const float* V1, int P,
                      const float* coords_x, int Bx, float sigma_x, int radius_x, const float* coords_y, int By,
                      float sigma_y, int radius_y, const float* gS, float* gx, int accumulate, void* stream);

/* Thin multipole kick ahead of a linear map (the non-linear transport of experiments/rec_2d/nonlinear:
 * mentflow/simulate/transform.py:78-146, orders 3..5, k = strength / (order-1)!): u = kick(x), and its adjoint.   */
int mf_multipole_kick_fwd(const float* x, int64_t n, int d, int order, float k, int skew, float* u, void* stream);
int mf_multipole_kick_bwd(const float* x, int64_t n, int d, int order, float k, int skew, const float* gu, float* gx,
                          void* stream);

/* Hard-binned projection histograms (measurement generation / eval): counts[p,k] of u = x.V_p in uniform bins
 * [lo, lo + B*delta], out-of-range ignored, last bin right-inclusive: torch.histogram semantics
 * (mentflow/diagnostics/diagnostics.py:128-131); density/renormalisation is done by the caller.                */
int mf_proj_hist1d_counts(const float* x, int64_t n, int d, const float* V, int P, const float* edges, int B,
                          int32_t* counts, void* stream);
int mf_proj_hist2d_counts(const float* x, int64_t n, int d, const float* V0, const float* V1, int P,
                          const float* edges_x, int Bx, const float* edges_y, int By, int32_t* counts,
                          void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Histogram normalisation + discrepancy for P projections (tiny: P*bins elements; one workgroup per projection).
 *   normalize != 0:  prob = S * pre_scale   (1-D: pre_scale = 1/N_total, histogram.py:39;  2-D: 1, histogram.py:69)
 *                    ghat = prob / (sum(prob) * cell + eps)                        (histogram.py:40-43,70-73)
 *   normalize == 0:  ghat = S  (S already is a prediction; the standalone discrepancy functions)
 *   meas != NULL:    kind 0 (kld): D_p = sum[xlogy(m,m) - m*log(ghat+pad)] / batch_div     (mentflow/loss.py:15-17;
 *                                  batch_div = pred.shape[0] = B for 1-D, Bx for 2-D predictions)
 *                    kind 1 (mae): D_p = sum|ghat-m| / batch_div  (loss.py:7-8, batch_div = number of bins)
 *                    kind 2 (mse): D_p = sum(ghat-m)^2 / batch_div (loss.py:11-12)
 *   ghat may be NULL (not wanted), meas/D may be NULL (normalisation only).
 * bwd: gS[p,k] = dL/dS from gD[p] = dL/dD_p (may be NULL) and/or gghat[p,k] = dL/dghat (may be NULL);
 *      closed form of SURVEY.md Appendix B.                                                                  */
int mf_hist_norm_discrepancy_fwd(const float* S, int P, int bins, int normalize, float pre_scale, float cell,
                                 float eps, const float* meas, int kind, float pad, float batch_div, float* ghat,
                                 float* D, void* stream);
int mf_hist_norm_discrepancy_bwd(const float* S, int P, int bins, int normalize, float pre_scale, float cell,
                                 float eps, const float* meas, int kind, float pad, float batch_div, const float* gD,
                                 const float* gghat, float* gS, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Monte-Carlo entropy sums  (mentflow/entropy.py:58-62 + mentflow/prior.py:25-26):
 *   out[0] = sum_n logp[n],  out[1] = sum_n |x_n|^2      (means / prior constants applied by the caller after
 *   any cross-GPU sum; scratch2 = MF_ENTROPY_SCRATCH_DOUBLES doubles of device scratch: per-workgroup fp64 partials,
 *   summed in a fixed order — no atomics, bitwise reproducible).
 *   mf_scale_rows: gx[n,:] (+)= coef[0] * cscale * x[n,:]  — the adjoint of the |x|^2 term; coef is a DEVICE
 *   scalar (the upstream gradient) so that no host synchronisation is needed.                                 */
int mf_mc_entropy_sums(const float* x, const float* logp, int64_t n, int d, float* out2, double* scratch2,
                       void* stream);
int mf_scale_rows(const float* x, int64_t n, int d, const float* coef, float cscale, float* gx, int accumulate,
                  void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MENTFLOW_HIP_H */
