// C ABI of the autoregressive flow layers (include/mentflow_hip.h): argument checks, launch geometry, choice of the
// backward variant, gradient reduce.  The kernels live in flow_kernels.inc; their instances are compiled in the
// translation units listed in flow_launch.h.
#include "flow_launch.h"
#include <stdlib.h>

namespace mf {

static int flow_check(int d, int L, int64_t n) {
    if (d < 1 || d > FLOW_DMAX) return fail("flow kernels support 1 <= d <= %d (got %d)", FLOW_DMAX, d);
    if (L < 1) return fail("hidden_layers must be >= 1");
    if (n < 0) return fail("negative particle count");
    return 0;
}

static int flow_grid(int64_t n, int waves = FLOW_WAVES) {
    const int64_t ntiles = (n + 31) / 32;
    int64_t g = (ntiles + waves - 1) / waves;
    if (g > NUM_CU) g = NUM_CU;
    if (g < 1) g = 1;
    return (int)g;
}

static int no_rqs_instance(int bins, int hidden_layers) {
    return fail("no RQS kernel instance for bins=%d hidden_layers=%d (built: 2 <= bins <= 21, hidden_layers in {1,2,3,4})", bins,
                hidden_layers);
}
static int no_affine_instance(int hidden_layers) {
    return fail("no affine kernel instance for hidden_layers=%d (built: 1 .. 4)", hidden_layers);
}

}  // namespace mf

using namespace mf;

extern "C" int64_t mf_flow_image_floats(int d, int hidden_layers) { return image_layout(d, hidden_layers, d).total; }

// Does mf_flow_rqs_layer_bwd take the fused kernel (parameter gradients inside the backward kernel, no scratch) for
// this call?  It needs the mask structure (order) for the compact last-layer image, d <= FB_DMAX accumulator blocks and
// an LDS budget that fits.  Since the deterministic slab flush replaced the contended float atomics the fused kernel also
// wins at small batches (25 000 particles, C4 step: 1.21 ms against 1.68 ms for the two-kernel path, profiles/r02), so
// it is the default for every batch size; MENTFLOW_BWD_FUSED=0 (read once) or mf_flow_set_bwd_variant(0) force the
// two-kernel path (tests).
// backward variant: -1 = default (the environment variable MENTFLOW_BWD_FUSED, read ONCE; unset = fused), 0 = two-kernel
// path, 1 = fused kernel.  mf_flow_set_bwd_variant overrides it (tests switch variants inside one process).
static int g_bwd_variant = -1;
static bool bwd_fused_wanted() {
    static const int env_default = [] { const char* e = getenv("MENTFLOW_BWD_FUSED"); return (e && atoi(e) == 0) ? 0 : 1; }();
    return (g_bwd_variant < 0 ? env_default : g_bwd_variant) != 0;
}
extern "C" int mf_flow_set_bwd_variant(int variant) {
    if (variant < -1 || variant > 1) return fail("mf_flow_set_bwd_variant: -1 (default), 0 (two-kernel) or 1 (fused)");
    g_bwd_variant = variant;
    return 0;
}

static bool rqs_bwd_fused(int64_t n, int d, int hidden_layers, const int32_t* order, const Sparsity& sp, size_t* smem) {
    if (!bwd_fused_wanted() || order == nullptr || d > FB_DMAX) return false;
    (void)n;
    size_t fl = image_layout(d, hidden_layers, d).offW3;
    for (int i = 0; i < d; ++i) fl += (size_t)WS * (2 * ((sp.kend3[i] + 3) & ~3));
    fl += (size_t)d * HID + HID;
    fl = (fl + 3) & ~(size_t)3;
    *smem = sizeof(float) * (fl + 8 * (size_t)FB_TILE);
    return *smem <= 160 * 1024;
}

extern "C" int64_t mf_flow_bwd_scratch_floats(int64_t n, int d, int hidden_layers, const int32_t* order) {
    if (order != nullptr && d >= 1 && d <= FLOW_DMAX) {
        size_t smem;
        if (rqs_bwd_fused(n, d, hidden_layers, order, make_sparsity(d, order, d), &smem)) return 0;
    }
    const int64_t npad = ((n + 31) / 32) * 32;
    return (2 * (int64_t)hidden_layers + d) * npad * 64;
}

// slot of this lane half's first derivative logit in the packed last-layer block (mentflow_amd/generate/packing.py): the
// number of bins for the compile-time instances, 21 for the run-time one; -1: no kernel for this number of bins
extern "C" int mf_flow_rqs_deriv_slot(int bins) {
    if (bins == 20 || bins == 8) return bins;
    return (bins >= 2 && bins <= RQS_KMAX) ? RQS_KMAX : -1;
}

// ---- activation hand-off (forward -> fused backward through HBM; act_store in flow_kernels.inc) --------------------------
// highest level the built kernels can take for this configuration under the CURRENT backward variant: 2 when the call takes the
// fused backward with a compile-time spline instance, else 0 (two-kernel path, d = 7, no order, run-time bins)
extern "C" int mf_flow_rqs_act_level(int d, int hidden_layers, int bins, const int32_t* order) {
    if (order == nullptr || d < 1 || d > FLOW_DMAX || !rqs_saved_instance(bins, hidden_layers)) return 0;
    size_t smem;
    return rqs_bwd_fused(1, d, hidden_layers, order, make_sparsity(d, order, d), &smem) ? 2 : 0;
}
extern "C" int64_t mf_flow_rqs_act_floats(int64_t n, int d, int hidden_layers, int bins, int level) {
    if (n <= 0 || d < 1 || hidden_layers < 1 || level < 1 || level > 2) return 0;
    const int phif = bins == 20 ? phi_floats<20>() : (bins == 8 ? phi_floats<8>() : phi_floats<RQS_ANY>());
    return ((n + 31) / 32) * (int64_t)act_tile_floats(hidden_layers, level, d - 1, phif);     // d - 1 features have a conditioner
}

static int rqs_layer_fwd_impl(const float* image, int d, int hidden_layers, int bins, const int32_t* order, const float* x,
                              int64_t n, float* y, const float* logp_in, float* logp_out, int init_logp, float* act, int level,
                              void* stream);

extern "C" int mf_flow_rqs_layer_fwd(const float* image, int d, int hidden_layers, int bins, const int32_t* order,
                                      const float* x, int64_t n, float* y, const float* logp_in, float* logp_out,
                                      int init_logp, void* stream) {
    return rqs_layer_fwd_impl(image, d, hidden_layers, bins, order, x, n, y, logp_in, logp_out, init_logp, nullptr, 0, stream);
}

extern "C" int mf_flow_rqs_layer_fwd_save(const float* image, int d, int hidden_layers, int bins, const int32_t* order,
                                           const float* x, int64_t n, float* y, const float* logp_in, float* logp_out,
                                           int init_logp, float* act, int64_t act_floats, int level, void* stream) {
    if (level < 1 || level > 2) return fail("mf_flow_rqs_layer_fwd_save: level must be 1 or 2 (got %d)", level);
    if (level > mf_flow_rqs_act_level(d, hidden_layers, bins, order))
        return fail("no backward kernel consumes saved activations for d=%d hidden_layers=%d bins=%d (mf_flow_rqs_act_level)", d,
                    hidden_layers, bins);
    if (n > 0 && (act == nullptr || act_floats < mf_flow_rqs_act_floats(n, d, hidden_layers, bins, level)))
        return fail("act buffer too small: %lld floats, need %lld (mf_flow_rqs_act_floats)", (long long)act_floats,
                    (long long)mf_flow_rqs_act_floats(n, d, hidden_layers, bins, level));
    return rqs_layer_fwd_impl(image, d, hidden_layers, bins, order, x, n, y, logp_in, logp_out, init_logp, act, level, stream);
}

static int rqs_layer_fwd_impl(const float* image, int d, int hidden_layers, int bins, const int32_t* order, const float* x,
                              int64_t n, float* y, const float* logp_in, float* logp_out, int init_logp, float* act, int level,
                              void* stream) {
    if (flow_check(d, hidden_layers, n)) return 1;
    if (n == 0) return 0;
    const Sparsity sp = make_sparsity(d, order, d);
    const size_t smem = sizeof(float) * (size_t)image_layout(d, hidden_layers, d).total;
    // Workgroup size: 1024 threads (4 waves per SIMD at <= 128 VGPRs) for big batches; small batches (the reference's
    // 25 000 particles = 782 tiles) use fewer waves per workgroup so that the tiles spread over all 256 CUs.
    const int64_t nt = (n + 31) / 32;
    static const int fwd_block_env = [] { const char* e = getenv("MENTFLOW_FWD_BLOCK"); return e ? atoi(e) : 0; }();
    const int fwd_block = fwd_block_env ? fwd_block_env : (nt <= 4 * NUM_CU ? 256 : (nt <= 8 * NUM_CU ? 512 : 1024));
    if (fwd_block != 256 && fwd_block != 512 && fwd_block != 1024) return fail("MENTFLOW_FWD_BLOCK must be 256, 512 or 1024");
    ProfScope prof(PK_FLOW_FWD, stream);
    if (launch_rqs_fwd(bins, hidden_layers, fwd_block, flow_grid(n, fwd_block / 64), smem, stream, image, d, x, n, y, logp_in,
                       logp_out, init_logp, sp, act, level))
        return no_rqs_instance(bins, hidden_layers);
    return check_launch("mf_flow_rqs_layer_fwd");
}

// grid sizes of the backward kernels: the number of slab rows a call writes (one per workgroup column)
static int fused_grid(int64_t n) {
    const int64_t ngroups = ((n + 31) / 32 + 3) / 4;
    int64_t cap = NUM_CU;
#ifdef MF_EMU
    // emulator build only (tests): a small cap makes a workgroup walk several groups at test sizes, which exercises the
    // cross-group prefetches (particle rows, handed-over activations) that a 256-workgroup grid only reaches past 32 768 particles
    if (const char* e = getenv("MENTFLOW_EMU_FUSED_GRID")) cap = atoi(e) > 0 ? atoi(e) : cap;
#endif
    return (int)(ngroups > cap ? cap : (ngroups < 1 ? 1 : ngroups));
}
static int outer_accum_grid(int64_t n) {
    static const int oa_mult = [] { const char* e = getenv("MENTFLOW_OA_MULT"); return e ? atoi(e) : 2; }();
    const int64_t ntiles = (n + 31) / 32;
    int64_t G = (ntiles + 7) / 8;                    // at least 8 tiles of work per workgroup
    if (G > oa_mult * NUM_CU) G = oa_mult * NUM_CU;
    if (G < 1) G = 1;
    return (int)G;
}

extern "C" int mf_flow_bwd_slab_rows(int64_t n, int d, int hidden_layers, const int32_t* order) {
    if (n <= 0) return 0;
    if (order != nullptr && d >= 1 && d <= FLOW_DMAX) {
        size_t smem;
        if (rqs_bwd_fused(n, d, hidden_layers, order, make_sparsity(d, order, d), &smem)) return fused_grid(n);
    }
    return outer_accum_grid(n);
}

extern "C" int mf_flow_rqs_layer_bwd_saved(const float* image, int d, int hidden_layers, int bins, const int32_t* order,
                                            const float* x, int64_t n, const float* gy, const float* glogp, float* gx,
                                            float* gslab, int slab_rows, int accumulate, const float* act, int64_t act_floats,
                                            int level, void* stream) {
    if (flow_check(d, hidden_layers, n)) return 1;
    if (n == 0) return 0;
    if (level < 1 || level > 2) return fail("mf_flow_rqs_layer_bwd_saved: level must be 1 or 2 (got %d)", level);
    if (level > mf_flow_rqs_act_level(d, hidden_layers, bins, order))
        return fail("no backward kernel consumes saved activations for d=%d hidden_layers=%d bins=%d (mf_flow_rqs_act_level)", d,
                    hidden_layers, bins);
    if (act == nullptr || act_floats < mf_flow_rqs_act_floats(n, d, hidden_layers, bins, level)) return fail("act buffer too small");
    const Sparsity sp = make_sparsity(d, order, d);
    size_t smem_f = 0;
    if (!rqs_bwd_fused(n, d, hidden_layers, order, sp, &smem_f)) return fail("the saved-activation backward is the fused kernel");
    if (slab_rows != fused_grid(n)) return fail("gslab has %d rows, this call writes %d (mf_flow_bwd_slab_rows)", slab_rows, fused_grid(n));
    ProfScope prof(PK_FLOW_BWD, stream);
    const int rc = level == 1 ? launch_rqs_bwd_fused_s1(bins, hidden_layers, fused_grid(n), smem_f, stream, image, d, x, n, gy, glogp,
                                                        gx, gslab, accumulate, sp, act)
                              : launch_rqs_bwd_fused_s2(bins, hidden_layers, fused_grid(n), smem_f, stream, image, d, x, n, gy, glogp,
                                                        gx, gslab, accumulate, sp, act);
    if (rc) return no_rqs_instance(bins, hidden_layers);
    return check_launch("mf_flow_rqs_layer_bwd_saved");
}

extern "C" int mf_flow_rqs_layer_bwd(const float* image, int d, int hidden_layers, int bins, const int32_t* order,
                                      const float* x, int64_t n, const float* gy, const float* glogp, float* gx,
                                      float* gslab, int slab_rows, int accumulate, float* scratch, int64_t scratch_floats,
                                      void* stream) {
    if (flow_check(d, hidden_layers, n)) return 1;
    if (n == 0) return 0;
    const Sparsity sp = make_sparsity(d, order, d);
    if (scratch_floats < mf_flow_bwd_scratch_floats(n, d, hidden_layers, order)) return fail("scratch too small");
    if (slab_rows != mf_flow_bwd_slab_rows(n, d, hidden_layers, order))
        return fail("gslab has %d rows, this call writes %d (mf_flow_bwd_slab_rows)", slab_rows,
                    mf_flow_bwd_slab_rows(n, d, hidden_layers, order));
    // fused backward + parameter gradients (no scratch traffic): 19.2 ms against 11.1 + 9.3 ms at 2 M particles (C4)
    size_t smem_f = 0;
    if (rqs_bwd_fused(n, d, hidden_layers, order, sp, &smem_f)) {
        ProfScope prof(PK_FLOW_BWD, stream);
        if (launch_rqs_bwd_fused_s0(bins, hidden_layers, fused_grid(n), smem_f, stream, image, d, x, n, gy, glogp, gx, gslab,
                                    accumulate, sp, nullptr))
            return no_rqs_instance(bins, hidden_layers);
        return check_launch("mf_flow_rqs_layer_bwd(fused)");
    }
    const size_t smem = sizeof(float) * (size_t)image_layout(d, hidden_layers, d).total;
    const int64_t ntb = (n + 31) / 32;
    static const int bwd_block_env = [] { const char* e = getenv("MENTFLOW_BWD_BLOCK"); return e ? atoi(e) : 0; }();
    // small batches: one tile per SIMD on as many CUs as possible
    const int bwd_block = bwd_block_env == 256 || bwd_block_env == 512 ? bwd_block_env : (ntb <= 4 * NUM_CU ? 256 : 512);
    {
        ProfScope prof(PK_FLOW_BWD, stream);
        if (launch_rqs_bwd2(bins, hidden_layers, bwd_block, flow_grid(n, bwd_block / 64), smem, stream, image, d, x, n, gy, glogp,
                            gx, scratch, sp))
            return no_rqs_instance(bins, hidden_layers);
    }
    if (check_launch("mf_flow_rqs_layer_bwd")) return 1;
    const int nwaves = d > hidden_layers ? d : hidden_layers;
    if (nwaves > OA_MAX_WAVES) return fail("too many linear blocks for the gradient kernel");
    ProfScope prof(PK_OUTER_ACCUM, stream);
    launch_outer_accum(outer_accum_grid(n), nwaves, stream, scratch, x, n, d, hidden_layers, d, gslab, accumulate, sp);
    return check_launch("mf_flow_rqs_layer_bwd(outer_accum)");
}

// ------------------------------------------------------------------------------------------------ affine C ABI
extern "C" int64_t mf_flow_affine_image_floats(int d, int hidden_layers) { return image_layout(d, hidden_layers, 1).total; }

// the fused affine backward needs no mask structure (dense image, 117 KB of LDS for every d <= 7); default for every
// batch size, MENTFLOW_BWD_FUSED=0 forces the two-kernel path
static bool affine_bwd_fused(int64_t n) {
    (void)n;
    return bwd_fused_wanted();
}

extern "C" int64_t mf_flow_affine_bwd_scratch_floats(int64_t n, int hidden_layers) {
    if (affine_bwd_fused(n)) return 0;
    const int64_t npad = ((n + 31) / 32) * 32;
    return (2 * (int64_t)hidden_layers + 1) * npad * 64;
}

extern "C" int mf_flow_affine_layer_fwd(const float* image, int d, int hidden_layers, const int32_t* order, const float* x,
                                         int64_t n, float* y, const float* logp_in, float* logp_out, int init_logp,
                                         void* stream) {
    if (flow_check(d, hidden_layers, n)) return 1;
    if (n == 0) return 0;
    const Sparsity sp = make_sparsity(d, order, 1);
    const size_t smem = sizeof(float) * (size_t)image_layout(d, hidden_layers, 1).total;
    ProfScope prof(PK_FLOW_FWD, stream);
    if (launch_affine_fwd(hidden_layers, flow_grid(n, 16), smem, stream, image, d, x, n, y, logp_in, logp_out, init_logp, sp))
        return no_affine_instance(hidden_layers);
    return check_launch("mf_flow_affine_layer_fwd");
}

extern "C" int mf_flow_affine_bwd_slab_rows(int64_t n) {
    if (n <= 0) return 0;
    return affine_bwd_fused(n) ? fused_grid(n) : outer_accum_grid(n);
}

extern "C" int mf_flow_affine_layer_bwd(const float* image, int d, int hidden_layers, const int32_t* order, const float* x,
                                         int64_t n, const float* gy, const float* glogp, float* gx, float* gslab,
                                         int slab_rows, int accumulate, float* scratch, int64_t scratch_floats, void* stream) {
    if (flow_check(d, hidden_layers, n)) return 1;
    if (n == 0) return 0;
    if (scratch_floats < mf_flow_affine_bwd_scratch_floats(n, hidden_layers)) return fail("scratch too small");
    if (slab_rows != mf_flow_affine_bwd_slab_rows(n))
        return fail("gslab has %d rows, this call writes %d (mf_flow_affine_bwd_slab_rows)", slab_rows,
                    mf_flow_affine_bwd_slab_rows(n));
    const Sparsity sp = make_sparsity(d, order, 1);
    if (affine_bwd_fused(n)) {
        const size_t smem_f = sizeof(float) * ((((size_t)image_layout(d, hidden_layers, 1).total + 3) & ~(size_t)3) + 8 * (size_t)FB_TILE);
        ProfScope prof(PK_FLOW_BWD, stream);
        if (launch_affine_bwd_fused(hidden_layers, fused_grid(n), smem_f, stream, image, d, x, n, gy, glogp, gx, gslab, accumulate))
            return no_affine_instance(hidden_layers);
        return check_launch("mf_flow_affine_layer_bwd(fused)");
    }
    const size_t smem = sizeof(float) * (size_t)image_layout(d, hidden_layers, 1).total;
    {
        ProfScope prof(PK_FLOW_BWD, stream);
        if (launch_affine_bwd2(hidden_layers, flow_grid(n), smem, stream, image, d, x, n, gy, glogp, gx, scratch, sp))
            return no_affine_instance(hidden_layers);
    }
    if (check_launch("mf_flow_affine_layer_bwd")) return 1;
    ProfScope prof(PK_OUTER_ACCUM, stream);
    launch_outer_accum(outer_accum_grid(n), hidden_layers, stream, scratch, x, n, d, hidden_layers, 1, gslab, accumulate, sp);
    return check_launch("mf_flow_affine_layer_bwd(outer_accum)");
}

// ------------------------------------------------------------------------------------------------ gradient reduce
// gflat[j] = sum over the slab rows r = 0 .. rows-1 (fixed order, fp64 accumulation) of gslab[t][r][pos], where
// grad_index[j] = t * image_floats + pos (or -1: masked-out / padding parameter -> 0).  One launch for all T layers.
namespace mf {
__global__ __launch_bounds__(256) void grad_reduce_kernel(const float* __restrict__ gslab, int rows, int64_t image_floats,
                                                          const int32_t* __restrict__ grad_index, float* __restrict__ gflat,
                                                          int64_t numel) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= numel) return;
    const int32_t idx = grad_index[j];
    if (idx < 0) {
        gflat[j] = 0.0f;
        return;
    }
    const int64_t t = idx / image_floats, pos = idx - t * image_floats;
    const float* p = gslab + (t * rows) * image_floats + pos;
    double acc = 0.0;
    int r = 0;
    for (; r + 8 <= rows; r += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(r + u) * image_floats];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += (double)v[u];
    }
    for (; r < rows; ++r) acc += (double)p[(int64_t)r * image_floats];
    gflat[j] = (float)acc;
}
}  // namespace mf

extern "C" int mf_flow_grad_reduce(const float* gslab, int layers, int rows, int64_t image_floats, const int32_t* grad_index,
                                    float* gflat, int64_t numel, void* stream) {
    if (layers < 1 || rows < 1 || image_floats < 1 || numel < 0) return fail("bad arguments to mf_flow_grad_reduce");
    if (numel == 0) return 0;
    MF_LAUNCH(grad_reduce_kernel, (unsigned)((numel + 255) / 256), 256, 0, stream, gslab, rows, image_floats, grad_index, gflat,
              numel);
    return check_launch("mf_flow_grad_reduce");
}

// ------------------------------------------------------------------------------------------------ inverse C ABI
static int inv_order(int d, const int32_t* order, InvOrder* io) {
    if (order == nullptr) return fail("the inverse needs the autoregressive order of the layer");
    for (int t = 0; t <= FLOW_DMAX; ++t) io->feat[t] = 0;
    for (int i = 0; i < d; ++i) {
        if (order[i] < 0 || order[i] >= d) return fail("order[%d] = %d out of range", i, order[i]);
        io->feat[order[i]] = i;
    }
    return 0;
}

extern "C" int mf_flow_rqs_layer_inv(const float* image, int d, int hidden_layers, int bins, const int32_t* order,
                                      const float* y, int64_t n, float* x, void* stream) {
    if (flow_check(d, hidden_layers, n)) return 1;
    InvOrder io;
    if (inv_order(d, order, &io)) return 1;
    if (n == 0) return 0;
    const Sparsity sp = make_sparsity(d, order, d);
    const size_t smem = sizeof(float) * ((size_t)image_layout(d, hidden_layers, d).total + (INV_BLOCK / 64) * 32 * 8);
    if (launch_rqs_inv(bins, hidden_layers, flow_grid(n, INV_BLOCK / 64), smem, stream, image, d, y, n, x, sp, io))
        return no_rqs_instance(bins, hidden_layers);
    return check_launch("mf_flow_rqs_layer_inv");
}

extern "C" int mf_flow_affine_layer_inv(const float* image, int d, int hidden_layers, const int32_t* order, const float* y,
                                         int64_t n, float* x, void* stream) {
    if (flow_check(d, hidden_layers, n)) return 1;
    InvOrder io;
    if (inv_order(d, order, &io)) return 1;
    if (n == 0) return 0;
    const Sparsity sp = make_sparsity(d, order, 1);
    const size_t smem = sizeof(float) * ((size_t)image_layout(d, hidden_layers, 1).total + (INV_BLOCK / 64) * 32 * 8);
    if (launch_affine_inv(hidden_layers, flow_grid(n, INV_BLOCK / 64), smem, stream, image, d, y, n, x, sp, io))
        return no_affine_instance(hidden_layers);
    return check_launch("mf_flow_affine_layer_inv");
}
