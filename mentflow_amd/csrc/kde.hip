// Fused linear projection + Gaussian-KDE histogram kernels (1-D and 2-D), hard-binned counts, the
// normalisation + discrepancy tail and the Monte-Carlo entropy sums, for gfx950.
//
// Reference chain replaced (paths relative to austin-hoover/ment-flow):
//   simulate.forward loop            mentflow/simulate/simulate.py:30-33
//   LinearTransform.forward          mentflow/simulate/transform.py:67-68
//   Histogram1D/2D.project/.bin      mentflow/diagnostics/diagnostics.py:116-131,179-201
//   marginal_pdf / joint_pdf         mentflow/diagnostics/histogram.py:11-74
//   kl_divergence / mae / mse        mentflow/loss.py:7-17
//   MonteCarloEntropyEstimator       mentflow/entropy.py:58-62, prior.Gaussian mentflow/prior.py:25-26
//
// Design (no MFMA here; the forward kernels are bound by LDS-atomic issue, the backward ones by LDS gathers / HBM rows):
// one lane per particle, particle rows read once (coalesced, d floats per lane), projection vectors / bin centres /
// the histogram image of a group of projections live in LDS; every particle touches only the 2R+1 bins whose Gaussian
// weight is above fp32 resolution (sigma = bw*delta, R = ceil(9 bw - 1/2): dropped weights < 3e-18).
// Accumulation is FIXED POINT and therefore exact and order independent at every level: each weight (<= 1) is
// converted to a 2^-50 integer and added with 64-bit integer LDS atomics (8.9e-16 quantum: far-tail bins, whose log
// enters the KL discrepancy, keep their relative accuracy); a workgroup (<= 8192 particles: 2^13 * 2^50 < 2^64) flushes
// every non-zero bin with ONE 64-bit INTEGER global atomic (2^(shift-50) units, exact up to 8192 particles per call,
// see fix_flush), and a finishing kernel scales and rounds to fp32 once.  Integer sums do not depend on their order: the histograms are bitwise reproducible.
// Measured on MI355X (tools/ubench_lds_atomics2.hip, profiles/r02_ubench_lds_atomics2.txt): ds_add_f32 sustains 0.33
// lane-ops/clk/CU, ds_add_u64 3.4-3.9 with enough waves in flight WHATEVER the address pattern (consecutive, random, one hot
// row, private copies per lane: all within 10 %; 2.6-2.75 with only 4 waves per CU) — the float LDS atomic is 10x slower,
// and bank conflicts do not matter: the atomic unit is the bottleneck either way, so the image is kept once (an earlier
// version replicated it per lane group; that bought nothing and was removed), occupancy is what counts.
#include "common.h"

namespace mf {

constexpr int KDE_DMAX = 8;          // phase-space dimension limit of the projection kernels
constexpr int KDE_BLOCK = 256;
constexpr int KDE_LDS_FLOATS = 24576;   // 96 KiB histogram image per workgroup
constexpr int KDE_RMAX2D = 5;

__device__ __forceinline__ void load_row(const float* __restrict__ x, int64_t p, int d, float (&xv)[KDE_DMAX]) {
#pragma unroll
    for (int j = 0; j < KDE_DMAX; ++j) xv[j] = (j < d) ? x[p * d + j] : 0.0f;
}

__device__ __forceinline__ float project(const float (&xv)[KDE_DMAX], const float* __restrict__ v, int d) {
    float u = 0.0f;
#pragma unroll
    for (int j = 0; j < KDE_DMAX; ++j)
        if (j < d) u = fmaf(xv[j], v[j], u);
    return u;
}

// centre bin of u on the uniform grid of bin centres (c0, delta); clamped so the window loop is empty when far out
__device__ __forceinline__ int centre_bin(float u, float c0, float inv_delta, int B, int R) {
    float t = (u - c0) * inv_delta;
    t = fminf(fmaxf(t, -(float)(R + 2)), (float)(B + R + 1));   // NaN -> lower clamp -> empty window
    return (int)rintf(t);
}

// ------------------------------------------------------------------------------------------------ 1-D forward
typedef unsigned long long u64;
constexpr double KDE_FIX_INV = 1.0 / 1125899906842624.0; // 2^-50: fixed-point quantum of the LDS histogram image
constexpr float KDE_FIX_MIN = 8.8817841970012523e-16f;   // 2^-50: smaller weights round down to zero
constexpr int KDE_MAX_PER_WG = 8192;                     // particles per workgroup: 2^13 * 2^50 < 2^64, no overflow
constexpr float KDE_EXP2_SCALE = 0.7213475204444817f;    // log2(e) / 2:  exp(-r^2/2) = exp2(-KDE_EXP2_SCALE r^2)
constexpr int KDE_VS = 8;                                // LDS row stride of the projection vectors (two 16-byte reads)

__device__ __forceinline__ float gauss_weight(float r) { return __builtin_amdgcn_exp2f(-KDE_EXP2_SCALE * r * r); }

// w in [0, 1] -> round-down 2^-50 fixed point (exact split: high 18 bits and low 32 bits of w * 2^50)
__device__ __forceinline__ u64 to_fix(float w) {
    const float t = w * 262144.0f;
    const unsigned hi = (unsigned)t;
    const unsigned lo = (unsigned)((t - (float)hi) * 4294967296.0f);
    return ((u64)hi << 32) | (u64)lo;
}

// flush of one workgroup's bin total v (2^-50 units, < 2^64) into the bin's 64-bit global accumulator: ONE integer atomic
// per non-zero bin and workgroup.  The global sum of n particles needs log2(n) more bits than a weight, so the
// accumulator is kept in 2^(shift-50) units with shift = max(0, ceil(log2 n) - 13): EXACT (shift 0) up to 8192 particles
// per call, 2^-43 units at 2 M, 2^-40 at 16 M.  The dropped bits cost at most one unit per flush: relative to a bin
// total that is n * 2^-76 / (density * bin width), 2e-6 of a 1e-10 density at 2 M particles — far below the reference's
// own 1e-12 padding — and integer addition keeps the result independent of the order of the workgroups.
__device__ __forceinline__ void fix_flush(u64* __restrict__ acc, u64 v, int shift) {
    v >>= shift;
    if (v != 0) atomicAdd(acc, v);
}
static int kde_global_shift(int64_t n) {
    int k = 0;
    while (((int64_t)1 << k) < n) ++k;
    return k > 13 ? k - 13 : 0;
}

// Gaussian window, partly factorised.  With kc the centre bin of u, r0 = (u - c_kc) / sigma and s = delta / sigma, bin
// kc + j of a UNIFORM grid sits at r_j = r0 - j s and
//     w_j = exp(-r_j^2 / 2) = A rho^j gam_j,   A = exp(-r0^2/2),  rho = exp(s r0),  gam_j = exp(-j^2 s^2 / 2).
// The table of bin centres is not exactly uniform (every centre is rounded separately: |dr| <= 4.4e-6), and the loss
// gradient dD/dS = 1 - m / ghat cancels to ~1e-2 near convergence, so a 1e-6 relative error in a bin total shows up as
// 1e-4 in the parameter gradients.  The three bins that carry 99.97 % of a particle's weight — j = 0, +1, -1 — are
// therefore evaluated from the table exactly as the reference does ((u - c_k) / sigma, one v_exp each); only |j| >= 2
// (weights <= 3.4e-4 at s = 2) use the factorised form: 5 v_exp (quarter rate) per particle and projection instead of
// 2 RT + 1 = 9, residual error ~1e-9 of a bin total.  |s r0| <= s^2 / 2, and RT = 4 is only used for bandwidths in
// (0.389, 0.5] bin widths (s in [2, 2.57]): rho^4 <= 5.4e5, nothing overflows.
template <int RT>
struct GaussGamma {
    float g[RT + 1];
};
template <int RT>
__device__ __forceinline__ GaussGamma<RT> gauss_gamma(float s) {
    GaussGamma<RT> gm;
#pragma unroll
    for (int j = 0; j <= RT; ++j) gm.g[j] = __builtin_amdgcn_exp2f(-KDE_EXP2_SCALE * (float)(j * j) * s * s);
    return gm;
}
// cl: table of bin centres; kc may lie up to RT + 2 bins outside [0, B) (then every bin that is read through a clamped
// index is out of range and never used by the caller)
template <int RT>
__device__ __forceinline__ void gauss_window(float u, const float* cl, int kc, int B, float inv_sigma, float s,
                                             const GaussGamma<RT>& gm, float (&w)[2 * RT + 1]) {
    static_assert(RT >= 2, "factorised tail needs at least two bins per side");
    const int k0 = min(max(kc, 0), B - 1), kp = min(max(kc + 1, 0), B - 1), km = min(max(kc - 1, 0), B - 1);
    const float r0 = fmaf((float)(k0 - kc), s, (u - cl[k0]) * inv_sigma);     // virtual centre when kc is outside
    const float A = __builtin_amdgcn_exp2f(-KDE_EXP2_SCALE * r0 * r0);
    w[RT] = A;
    w[RT + 1] = gauss_weight((u - cl[kp]) * inv_sigma);
    w[RT - 1] = gauss_weight((u - cl[km]) * inv_sigma);
    const float e = 2.0f * KDE_EXP2_SCALE * s * r0;
    const float rp = __builtin_amdgcn_exp2f(e), rm = __builtin_amdgcn_exp2f(-e);
    float pp = A * rp, pm = A * rm;
#pragma unroll
    for (int j = 2; j <= RT; ++j) {
        pp *= rp;
        pm *= rm;
        w[RT + j] = pp * gm.g[j];
        w[RT - j] = pm * gm.g[j];
    }
}

__device__ __forceinline__ void load_vrow(const float* Vl, int q, float (&v)[KDE_DMAX]) {
    const float4 a = *reinterpret_cast<const float4*>(Vl + q * KDE_VS);
    const float4 b = *reinterpret_cast<const float4*>(Vl + q * KDE_VS + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ float project8(const float (&xv)[KDE_DMAX], const float (&v)[KDE_DMAX]) {
    float u = 0.0f;
#pragma unroll
    for (int j = 0; j < KDE_DMAX; ++j) u = fmaf(xv[j], v[j], u);     // columns >= d: xv = 0, v = 0
    return u;
}
template <int BLOCK>
__device__ __forceinline__ void stage_vrows(float* Vl, const float* __restrict__ V, int p_begin, int np, int d) {
    for (int i = threadIdx.x; i < np * KDE_VS; i += BLOCK) {
        const int q = i / KDE_VS, j = i - q * KDE_VS;
        Vl[i] = (j < d) ? V[(p_begin + q) * d + j] : 0.0f;
    }
}

// grid (G, ngroups); block BLOCK; workgroup (bx, by) takes particles [bx * per_wg, (bx + 1) * per_wg) and the
// projections of group by.  LDS: [Pg * B] u64 image | [Pg * 8] V | [B] coords.  All lanes of a wave work on the SAME
// projection at a time (its row of V is a broadcast LDS read): LDS atomics sustain the same ~3.5-3.9 lane-ops/clk/CU
// whether the lanes' addresses are spread, random or piled on one row (tools/ubench_lds_atomics2.hip), so nothing is
// gained by staggering the lanes over projections, and bank conflicts do not matter either.
template <int RT, int BLOCK>   // RT > 0: compile-time window radius (factorised weights);  RT == 0: runtime radius
__global__ __launch_bounds__(BLOCK) void proj_kde1d_fwd_kernel(
    const float* __restrict__ x, int64_t n, int d, const float* __restrict__ V, int P, int Pg,
    const float* __restrict__ coords, int B, float inv_sigma, int Rrt, u64* __restrict__ Sacc, int per_wg, int gshift) {
    MF_DYN_SMEM(u64, lds);
    u64* img = lds;
    float* Vl = reinterpret_cast<float*>(img + (size_t)Pg * B);
    float* cl = Vl + Pg * KDE_VS;
    const int p_begin = blockIdx.y * Pg;
    const int np = min(Pg, P - p_begin);
    const int R = RT > 0 ? RT : Rrt;
    for (int i = threadIdx.x; i < np * B; i += BLOCK) img[i] = 0;
    stage_vrows<BLOCK>(Vl, V, p_begin, np, d);
    for (int i = threadIdx.x; i < B; i += BLOCK) cl[i] = coords[i];
    __syncthreads();
    const float c0 = cl[0];
    const float delta = cl[1] - cl[0];
    const float inv_delta = 1.0f / delta;
    const float s = delta * inv_sigma;
    // The factorised window (and, in 2-D, the dead corner cells) assumes s = delta / sigma in [2, 2.6]: below 2 the radius-4
    // window would drop weights above the quantum, far above it rho^j gam_j overflows to inf * 0.  The C ABI takes radius and
    // sigma independently, so the specialisation is taken only inside that range (wave-uniform test); any other pair runs
    // the generic loop with the radius given.
    const bool fact_ok = RT > 0 && s >= 2.0f && s <= 2.6f;
    const GaussGamma<(RT > 0 ? RT : 2)> gm = gauss_gamma<(RT > 0 ? RT : 2)>(fact_ok ? s : 2.0f);
    const int64_t p_lo = (int64_t)blockIdx.x * per_wg;
    const int64_t p_hi = min(n, p_lo + per_wg);
    for (int64_t p = p_lo + threadIdx.x; p < p_hi; p += BLOCK) {
        float xv[KDE_DMAX];
        load_row(x, p, d, xv);
        for (int q = 0; q < np; ++q) {
            float vq[KDE_DMAX];
            load_vrow(Vl, q, vq);
            const float u = project8(xv, vq);
            const int kc = centre_bin(u, c0, inv_delta, B, R);
            u64* row = img + (size_t)q * B;
            if (fact_ok) {
                float w[2 * (RT > 0 ? RT : 2) + 1];
                gauss_window<(RT > 0 ? RT : 2)>(u, cl, kc, B, inv_sigma, s, gm, w);
#pragma unroll
                for (int j = -RT; j <= RT; ++j) {
                    const int k = kc + j;
                    if (k >= 0 && k < B) {
                        const u64 f = to_fix(w[j + RT]);
                        if (f != 0) atomicAdd(&row[k], f);
                    }
                }
            } else {
                for (int j = -R; j <= R; ++j) {
                    const int k = kc + j;
                    if (k >= 0 && k < B) {
                        const u64 f = to_fix(gauss_weight((u - cl[k]) * inv_sigma));
                        if (f != 0) atomicAdd(&row[k], f);
                    }
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < np * B; i += BLOCK) fix_flush(Sacc + (int64_t)p_begin * B + i, img[i], gshift);
}

// S[i] = fp32(acc[i] * 2^(shift-50))
__global__ __launch_bounds__(KDE_BLOCK) void acc_to_float_kernel(const u64* __restrict__ Sacc, float* __restrict__ S,
                                                                  int64_t total, double unit) {
    for (int64_t i = (int64_t)blockIdx.x * KDE_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * KDE_BLOCK)
        S[i] = (float)((double)Sacc[i] * unit);
}

// ------------------------------------------------------------------------------------------------ 1-D backward
// grid (G); loops over projection groups; LDS: [Pg*B] gS | [Pg*8] V | [B] coords.
// CH (1, 2, 4 or 8) adjacent lanes share a particle and take every CH-th projection; their partial gradient rows are
// summed with a fixed butterfly (deterministic, no atomics).  Small batches use CH > 1: one lane per particle walks
// all P projections serially, which leaves most of the chip idle at the reference's 25 000-particle batch.
template <int RT, int BLOCK>   // RT > 0: compile-time window radius (factorised, branch-free);  RT == 0: runtime radius
__global__ __launch_bounds__(BLOCK) void proj_kde1d_bwd_kernel(
    const float* __restrict__ x, int64_t n, int d, const float* __restrict__ V, int P, int Pg,
    const float* __restrict__ coords, int B, float inv_sigma, int R, const float* __restrict__ gS,
    float* __restrict__ gx, int accumulate, int CH) {
    MF_DYN_SMEM(float, lds);
    float* img = lds;
    float* Vl = img + Pg * B;
    float* cl = Vl + Pg * KDE_VS;
    for (int i = threadIdx.x; i < B; i += BLOCK) cl[i] = coords[i];
    __syncthreads();
    const float c0 = cl[0];
    const float delta = cl[1] - cl[0];
    const float inv_delta = 1.0f / delta;
    const int per_wg = BLOCK / CH;
    const int chunk = threadIdx.x % CH;
    const int64_t p = (int64_t)blockIdx.x * per_wg + threadIdx.x / CH;
    const bool valid = p < n;
    float xv[KDE_DMAX], gv[KDE_DMAX];
    load_row(x, valid ? p : 0, d, xv);
#pragma unroll
    for (int j = 0; j < KDE_DMAX; ++j) gv[j] = 0.0f;
    for (int p_begin = 0; p_begin < P; p_begin += Pg) {
        const int np = min(Pg, P - p_begin);
        __syncthreads();
        for (int i = threadIdx.x; i < np * B; i += BLOCK) img[i] = gS[(int64_t)p_begin * B + i];
        stage_vrows<BLOCK>(Vl, V, p_begin, np, d);
        __syncthreads();
        for (int q = chunk; q < np; q += CH) {
            float vq[KDE_DMAX];
            load_vrow(Vl, q, vq);
            const float u = project8(xv, vq);
            const int kc = centre_bin(u, c0, inv_delta, B, R);
            float du = 0.0f;
            if (RT > 0 && kc >= RT && kc < B - RT) {
                // INTERIOR particle (all but the outermost RT bins of a distribution): the whole window lies inside the grid —
                // no clamps, no masks, and the 2 RT + 1 bin centres / gS values are consecutive LDS words read at immediate
                // offsets of ONE address each.  The kernel is bound by vector issue (168 full-rate instructions per particle
                // and projection in the general path, profiles/r04_kde_bwd_instruction_mix.txt): this path issues a third
                // fewer.  Same residuals from the table of bin centres, same terms, summed in the same order.
                const float* cp = cl + (kc - RT);
                const float* gp = img + q * B + (kc - RT);
#pragma unroll
                for (int j = 0; j <= 2 * RT; ++j) {
                    const float r = (u - cp[j]) * inv_sigma;
                    du += gp[j] * gauss_weight(r) * (-r * inv_sigma);
                }
            } else if (RT > 0) {
                // window bins outside [0, B) read a clamped bin and are masked: no divergent branch in the unrolled loop
                // (NaN / inf rows: every bin is out of range, every term is masked, the gradient row is exactly 0)
                // residuals from the table of bin centres, exactly as the reference forms them (the backward is VALU /
                // gather bound, not atomic bound: the factorised window of the forward would only save exps here, at the
                // price of a 4e-6 absolute error in per-particle gradients from the table's rounding)
#pragma unroll
                for (int j = -RT; j <= RT; ++j) {
                    const int k = kc + j;
                    const int kk = min(max(k, 0), B - 1);
                    const float r = (u - cl[kk]) * inv_sigma;
                    const float term = img[q * B + kk] * gauss_weight(r) * (-r * inv_sigma);
                    du += (k == kk) ? term : 0.0f;
                }
            } else {
                for (int j = -R; j <= R; ++j) {
                    const int k = kc + j;
                    if (k >= 0 && k < B) {
                        const float r = (u - cl[k]) * inv_sigma;
                        du = fmaf(img[q * B + k] * gauss_weight(r), -r * inv_sigma, du);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < KDE_DMAX; ++j) gv[j] = fmaf(du, vq[j], gv[j]);
        }
    }
    for (int m = 1; m < CH; m <<= 1) {
#pragma unroll
        for (int j = 0; j < KDE_DMAX; ++j) gv[j] += __shfl_xor(gv[j], m);
    }
    if (valid && chunk == 0) {
#pragma unroll
        for (int j = 0; j < KDE_DMAX; ++j)
            if (j < d) gx[p * d + j] = accumulate ? gx[p * d + j] + gv[j] : gv[j];
    }
}

// ------------------------------------------------------------------------------------------------ 2-D forward
// Window cells (i, j) bins away from the centre cell whose weight is below the 2^-50 quantum WHATEVER the position of
// the particle inside its cell: (|i| - 1/2)^2 + (|j| - 1/2)^2 > 2 * 50 ln 2 / s^2 with s >= 2 (radius-4 windows), i.e.
// > 17.33 bins^2 — the four corners (4,4) and the eight (4,3)/(3,4) cells.  They are never visited.
__device__ __forceinline__ constexpr bool kde2d_dead_cell(int i, int j) {
    const int a = i < 0 ? -i : i, b = j < 0 ? -j : j;
    return (a >= 1 && b >= 1) && ((2 * a - 1) * (2 * a - 1) + (2 * b - 1) * (2 * b - 1) > 69);   // 4 * 17.33
}

// grid (G, ngroups); workgroup (bx, by): particles [bx * per_wg, ...), projections of group by.
// LDS: [Pg*Bx*By] u64 image | [Pg*8] V0 | [Pg*8] V1 | [Bx] cx | [By] cy
// waves_per_eu(8, 8): the kernel is LDS-atomic bound and lives on occupancy (two 1024-thread workgroups per CU).  With the
// run-time window guard both window variants sit in one kernel and the allocator took 106 SGPRs for it: 7 waves per SIMD,
// i.e. ONE workgroup per CU, 5.7 -> 6.8 ms at C5.  Capping the wave at the 8-wave budget (30 scalars spill to VGPR lanes,
// none in the specialised window loop) restores two workgroups per CU (5.6 ms; A/B in profiles/README.md, round 3).
template <int RT, int BLOCK>   // RT == 4: both radii are 4 (factorised weights, dead corners skipped);  RT == 0: generic
__global__ __launch_bounds__(BLOCK) MF_WAVES_PER_SIMD(8, 8) void proj_kde2d_fwd_kernel(
    const float* __restrict__ x, int64_t n, int d, const float* __restrict__ V0, const float* __restrict__ V1, int P,
    int Pg, const float* __restrict__ coords_x, int Bx, float inv_sx, int Rx, const float* __restrict__ coords_y,
    int By, float inv_sy, int Ry, u64* __restrict__ Sacc, int per_wg, int gshift) {
    MF_DYN_SMEM(u64, lds);
    const int BB = Bx * By;
    u64* img = lds;
    float* V0l = reinterpret_cast<float*>(img + (size_t)Pg * BB);
    float* V1l = V0l + Pg * KDE_VS;
    float* cxl = V1l + Pg * KDE_VS;
    float* cyl = cxl + Bx;
    const int p_begin = blockIdx.y * Pg;
    const int np = min(Pg, P - p_begin);
    for (int i = threadIdx.x; i < np * BB; i += BLOCK) img[i] = 0;
    stage_vrows<BLOCK>(V0l, V0, p_begin, np, d);
    stage_vrows<BLOCK>(V1l, V1, p_begin, np, d);
    for (int i = threadIdx.x; i < Bx; i += BLOCK) cxl[i] = coords_x[i];
    for (int i = threadIdx.x; i < By; i += BLOCK) cyl[i] = coords_y[i];
    __syncthreads();
    const float cx0 = cxl[0], dx = cxl[1] - cxl[0], inv_dx = 1.0f / dx, ssx = dx * inv_sx;
    const float cy0 = cyl[0], dy = cyl[1] - cyl[0], inv_dy = 1.0f / dy, ssy = dy * inv_sy;
    constexpr int RW = RT > 0 ? RT : 2;
    const bool fact_ok = RT > 0 && ssx >= 2.0f && ssx <= 2.6f && ssy >= 2.0f && ssy <= 2.6f;     // see proj_kde1d_fwd_kernel
    const GaussGamma<RW> gmx = gauss_gamma<RW>(fact_ok ? ssx : 2.0f), gmy = gauss_gamma<RW>(fact_ok ? ssy : 2.0f);
    const int64_t p_lo = (int64_t)blockIdx.x * per_wg;
    const int64_t p_hi = min(n, p_lo + per_wg);
    for (int64_t p = p_lo + threadIdx.x; p < p_hi; p += BLOCK) {
        float xv[KDE_DMAX];
        load_row(x, p, d, xv);
        for (int q = 0; q < np; ++q) {
            float v0[KDE_DMAX], v1[KDE_DMAX];
            load_vrow(V0l, q, v0);
            load_vrow(V1l, q, v1);
            const float u0 = project8(xv, v0);
            const float u1 = project8(xv, v1);
            const int ka = centre_bin(u0, cx0, inv_dx, Bx, Rx);
            const int kb = centre_bin(u1, cy0, inv_dy, By, Ry);
            u64* im = img + (size_t)q * BB;
            if (fact_ok) {
                float wx[2 * RW + 1], wy[2 * RW + 1];
                gauss_window<RW>(u0, cxl, ka, Bx, inv_sx, ssx, gmx, wx);
                gauss_window<RW>(u1, cyl, kb, By, inv_sy, ssy, gmy, wy);
#pragma unroll
                for (int i = -RT; i <= RT; ++i) {
                    const int a = ka + i;
                    if (a < 0 || a >= Bx) continue;
                    u64* row = im + a * By;
#pragma unroll
                    for (int j = -RT; j <= RT; ++j) {
                        if (kde2d_dead_cell(i, j)) continue;
                        const int b = kb + j;
                        if (b >= 0 && b < By) {
                            const u64 f = to_fix(wx[i + RW] * wy[j + RW]);
                            if (f != 0) atomicAdd(&row[b], f);
                        }
                    }
                }
            } else {
                float wy[2 * KDE_RMAX2D + 1];
#pragma unroll
                for (int j = 0; j < 2 * KDE_RMAX2D + 1; ++j) {
                    const int b = kb - Ry + j;
                    float w = 0.0f;
                    if (j <= 2 * Ry && b >= 0 && b < By) w = gauss_weight((u1 - cyl[b]) * inv_sy);
                    wy[j] = w;
                }
                for (int i = 0; i <= 2 * Rx; ++i) {
                    const int a = ka - Rx + i;
                    if (a < 0 || a >= Bx) continue;
                    const float wx = gauss_weight((u0 - cxl[a]) * inv_sx);
                    u64* row = im + a * By;
#pragma unroll
                    for (int j = 0; j < 2 * KDE_RMAX2D + 1; ++j) {
                        const int b = kb - Ry + j;
                        if (j <= 2 * Ry && b >= 0 && b < By) {
                            const u64 f = to_fix(wx * wy[j]);
                            if (f != 0) atomicAdd(&row[b], f);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < np * BB; i += BLOCK) fix_flush(Sacc + (int64_t)p_begin * BB + i, img[i], gshift);
}

// ------------------------------------------------------------------------------------------------ 2-D backward
// A workgroup owns NPT * BLOCK particles and walks the projection groups once.  Every thread keeps the rows and the
// gradient rows of its NPT particles in registers for the whole walk (no read-modify-write of gx per group); the gS
// images of a group are staged into LDS once per workgroup and group.
template <int RT, int BLOCK, int NPT>
__global__ __launch_bounds__(BLOCK) void proj_kde2d_bwd_kernel(
    const float* __restrict__ x, int64_t n, int d, const float* __restrict__ V0, const float* __restrict__ V1, int P,
    int Pg, const float* __restrict__ coords_x, int Bx, float inv_sx, int Rx, const float* __restrict__ coords_y,
    int By, float inv_sy, int Ry, const float* __restrict__ gS, float* __restrict__ gx, int accumulate) {
    MF_DYN_SMEM(float, lds);
    const int BB = Bx * By;
    float* img = lds;
    float* V0l = img + Pg * BB;
    float* V1l = V0l + Pg * KDE_VS;
    float* cxl = V1l + Pg * KDE_VS;
    float* cyl = cxl + Bx;
    for (int i = threadIdx.x; i < Bx; i += BLOCK) cxl[i] = coords_x[i];
    for (int i = threadIdx.x; i < By; i += BLOCK) cyl[i] = coords_y[i];
    __syncthreads();
    const float cx0 = cxl[0], inv_dx = 1.0f / (cxl[1] - cxl[0]);
    const float cy0 = cyl[0], inv_dy = 1.0f / (cyl[1] - cyl[0]);
    constexpr int RW = RT > 0 ? RT : 1;
    // the same cells as the forward: the dead-corner skip of the RT = 4 window is only taken where the forward takes it
    const float ssx = (cxl[1] - cxl[0]) * inv_sx, ssy = (cyl[1] - cyl[0]) * inv_sy;
    const bool fact_ok = RT > 0 && ssx >= 2.0f && ssx <= 2.6f && ssy >= 2.0f && ssy <= 2.6f;
    const int64_t base = (int64_t)blockIdx.x * BLOCK * NPT + threadIdx.x;
    float xv[NPT][KDE_DMAX], gv[NPT][KDE_DMAX];
#pragma unroll
    for (int t = 0; t < NPT; ++t) {
        const int64_t p = base + (int64_t)t * BLOCK;
        load_row(x, p < n ? p : n - 1, d, xv[t]);
#pragma unroll
        for (int j = 0; j < KDE_DMAX; ++j) gv[t][j] = 0.0f;
    }
    for (int p_begin = 0; p_begin < P; p_begin += Pg) {
        const int np = min(Pg, P - p_begin);
        __syncthreads();
        for (int i = threadIdx.x; i < np * BB; i += BLOCK) img[i] = gS[(int64_t)p_begin * BB + i];
        stage_vrows<BLOCK>(V0l, V0, p_begin, np, d);
        stage_vrows<BLOCK>(V1l, V1, p_begin, np, d);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NPT; ++t) {
            for (int q = 0; q < np; ++q) {
                float v0[KDE_DMAX], v1[KDE_DMAX];
                load_vrow(V0l, q, v0);
                load_vrow(V1l, q, v1);
                const float u0 = project8(xv[t], v0);
                const float u1 = project8(xv[t], v1);
                const int ka = centre_bin(u0, cx0, inv_dx, Bx, Rx);
                const int kb = centre_bin(u1, cy0, inv_dy, By, Ry);
                const float* im = img + q * BB;
                float du0 = 0.0f, du1 = 0.0f;
                if (fact_ok) {
                    // same cells as the forward (dead corners skipped); out-of-range bins read a clamped bin, masked
                    // residuals from the tables of bin centres, exactly as the reference forms them
                    float wy[2 * RW + 1], dyw[2 * RW + 1];
                    int bidx[2 * RW + 1];
#pragma unroll
                    for (int j = -RT; j <= RT; ++j) {
                        const int b = kb + j;
                        const int bb = min(max(b, 0), By - 1);
                        const bool ok = b == bb;
                        const float r = (u1 - cyl[bb]) * inv_sy;
                        const float w = ok ? gauss_weight(r) : 0.0f;       // select, not multiply: NaN rows stay out
                        wy[j + RW] = w;
                        dyw[j + RW] = ok ? -r * inv_sy * w : 0.0f;
                        bidx[j + RW] = bb;
                    }
#pragma unroll
                    for (int i = -RT; i <= RT; ++i) {
                        const int a = ka + i;
                        const int aa = min(max(a, 0), Bx - 1);
                        const float* row = im + aa * By;
                        float sa = 0.0f, sb = 0.0f;
#pragma unroll
                        for (int j = -RT; j <= RT; ++j) {
                            if (kde2d_dead_cell(i, j)) continue;
                            const float g = row[bidx[j + RW]];
                            sa = fmaf(g, wy[j + RW], sa);
                            sb = fmaf(g, dyw[j + RW], sb);
                        }
                        const float r = (u0 - cxl[aa]) * inv_sx;
                        const float w = gauss_weight(r);
                        const float tx = -r * inv_sx * w * sa, ty = w * sb;
                        du0 += (a == aa) ? tx : 0.0f;
                        du1 += (a == aa) ? ty : 0.0f;
                    }
                } else {
                    float wy[2 * KDE_RMAX2D + 1], dyv[2 * KDE_RMAX2D + 1];
#pragma unroll
                    for (int j = 0; j < 2 * KDE_RMAX2D + 1; ++j) {
                        const int b = kb - Ry + j;
                        float w = 0.0f, dw = 0.0f;
                        if (j <= 2 * Ry && b >= 0 && b < By) {
                            const float r = (u1 - cyl[b]) * inv_sy;
                            w = gauss_weight(r);
                            dw = -r * inv_sy * w;
                        }
                        wy[j] = w;
                        dyv[j] = dw;
                    }
                    for (int i = 0; i <= 2 * Rx; ++i) {
                        const int a = ka - Rx + i;
                        if (a < 0 || a >= Bx) continue;
                        const float r = (u0 - cxl[a]) * inv_sx;
                        const float wx = gauss_weight(r);
                        const float dwx = -r * inv_sx * wx;
                        const float* row = im + a * By;
                        float sa = 0.0f, sb = 0.0f;
#pragma unroll
                        for (int j = 0; j < 2 * KDE_RMAX2D + 1; ++j) {
                            const int b = kb - Ry + j;
                            if (j <= 2 * Ry && b >= 0 && b < By) {
                                const float g = row[b];
                                sa = fmaf(g, wy[j], sa);
                                sb = fmaf(g, dyv[j], sb);
                            }
                        }
                        du0 = fmaf(dwx, sa, du0);
                        du1 = fmaf(wx, sb, du1);
                    }
                }
#pragma unroll
                for (int j = 0; j < KDE_DMAX; ++j) gv[t][j] = fmaf(du0, v0[j], fmaf(du1, v1[j], gv[t][j]));
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NPT; ++t) {
        const int64_t p = base + (int64_t)t * BLOCK;
        if (p < n) {
#pragma unroll
            for (int j = 0; j < KDE_DMAX; ++j)
                if (j < d) gx[p * d + j] = accumulate ? gx[p * d + j] + gv[t][j] : gv[t][j];
        }
    }
}

// ------------------------------------------------------------------------------------------------ hard-binned counts
// torch.histogram / np.histogramdd bin search: bin k iff edges[k] <= u < edges[k+1], last bin right-inclusive.
__device__ __forceinline__ int edge_bin(float u, const float* e, int B, float inv_delta) {
    if (!(u >= e[0]) || !(u <= e[B])) return -1;
    int k = (int)((u - e[0]) * inv_delta);
    k = max(0, min(k, B - 1));
    while (k > 0 && u < e[k]) --k;
    while (k < B - 1 && u >= e[k + 1]) ++k;
    return k;
}

// grid (G, ngroups); LDS: [Pg*B] int image | [Pg*d] V | [B+1] edges
__global__ __launch_bounds__(KDE_BLOCK) void proj_hist1d_kernel(
    const float* __restrict__ x, int64_t n, int d, const float* __restrict__ V, int P, int Pg,
    const float* __restrict__ edges, int B, int* __restrict__ counts) {
    MF_DYN_SMEM(float, lds);
    int* img = reinterpret_cast<int*>(lds);
    float* Vl = lds + Pg * B;
    float* el = Vl + Pg * d;
    const int p_begin = blockIdx.y * Pg;
    const int np = min(Pg, P - p_begin);
    for (int i = threadIdx.x; i < np * B; i += KDE_BLOCK) img[i] = 0;
    for (int i = threadIdx.x; i < np * d; i += KDE_BLOCK) Vl[i] = V[p_begin * d + i];
    for (int i = threadIdx.x; i <= B; i += KDE_BLOCK) el[i] = edges[i];
    __syncthreads();
    const float inv_delta = 1.0f / (el[1] - el[0]);
    for (int64_t p = (int64_t)blockIdx.x * KDE_BLOCK + threadIdx.x; p < n; p += (int64_t)gridDim.x * KDE_BLOCK) {
        float xv[KDE_DMAX];
        load_row(x, p, d, xv);
        for (int q = 0; q < np; ++q) {
            const int k = edge_bin(project(xv, Vl + q * d, d), el, B, inv_delta);
            if (k >= 0) atomicAdd(&img[q * B + k], 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < np * B; i += KDE_BLOCK)
        if (img[i]) atomicAdd(&counts[(int64_t)p_begin * B + i], img[i]);
}

__global__ __launch_bounds__(KDE_BLOCK) void proj_hist2d_kernel(
    const float* __restrict__ x, int64_t n, int d, const float* __restrict__ V0, const float* __restrict__ V1, int P,
    int Pg, const float* __restrict__ edges_x, int Bx, const float* __restrict__ edges_y, int By,
    int* __restrict__ counts) {
    MF_DYN_SMEM(float, lds);
    const int BB = Bx * By;
    int* img = reinterpret_cast<int*>(lds);
    float* V0l = lds + Pg * BB;
    float* V1l = V0l + Pg * d;
    float* exl = V1l + Pg * d;
    float* eyl = exl + Bx + 1;
    const int p_begin = blockIdx.y * Pg;
    const int np = min(Pg, P - p_begin);
    for (int i = threadIdx.x; i < np * BB; i += KDE_BLOCK) img[i] = 0;
    for (int i = threadIdx.x; i < np * d; i += KDE_BLOCK) {
        V0l[i] = V0[p_begin * d + i];
        V1l[i] = V1[p_begin * d + i];
    }
    for (int i = threadIdx.x; i <= Bx; i += KDE_BLOCK) exl[i] = edges_x[i];
    for (int i = threadIdx.x; i <= By; i += KDE_BLOCK) eyl[i] = edges_y[i];
    __syncthreads();
    const float inv_dx = 1.0f / (exl[1] - exl[0]);
    const float inv_dy = 1.0f / (eyl[1] - eyl[0]);
    for (int64_t p = (int64_t)blockIdx.x * KDE_BLOCK + threadIdx.x; p < n; p += (int64_t)gridDim.x * KDE_BLOCK) {
        float xv[KDE_DMAX];
        load_row(x, p, d, xv);
        for (int q = 0; q < np; ++q) {
            const int a = edge_bin(project(xv, V0l + q * d, d), exl, Bx, inv_dx);
            const int b = edge_bin(project(xv, V1l + q * d, d), eyl, By, inv_dy);
            if (a >= 0 && b >= 0) atomicAdd(&img[q * BB + a * By + b], 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < np * BB; i += KDE_BLOCK)
        if (img[i]) atomicAdd(&counts[(int64_t)p_begin * BB + i], img[i]);
}

// ------------------------------------------------------------------------------------------------ tail
__device__ __forceinline__ double block_sum(double v, double* red) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = KDE_BLOCK / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

// one workgroup per projection.  See include/mentflow_hip.h for the formulas.
__device__ __forceinline__ float disc_term(int kind, float g, float m, float pad) {
    if (kind == 0) {
        const float xl = (m > 0.0f) ? m * logf(m) : 0.0f;     // xlogy(m, m); targets are >= 0
        return xl - m * logf(g + pad);
    }
    if (kind == 1) return fabsf(g - m);
    return (g - m) * (g - m);
}
__device__ __forceinline__ float disc_grad(int kind, float g, float m, float pad) {
    if (kind == 0) return -m / (g + pad);
    if (kind == 1) return (g > m) ? 1.0f : ((g < m) ? -1.0f : 0.0f);
    return 2.0f * (g - m);
}

__global__ __launch_bounds__(KDE_BLOCK) void hist_norm_disc_fwd_kernel(
    const float* __restrict__ S, int bins, int normalize, float pre_scale, float cell, float eps,
    const float* __restrict__ meas, int kind, float pad, float batch_div, float* __restrict__ ghat,
    float* __restrict__ D) {
    __shared__ double red[KDE_BLOCK];
    const int p = blockIdx.x;
    const float* Sp = S + (int64_t)p * bins;
    float norm = 1.0f;
    if (normalize) {
        double acc = 0.0;
        for (int k = threadIdx.x; k < bins; k += KDE_BLOCK) acc += (double)(Sp[k] * pre_scale);
        const double tot = block_sum(acc, red);
        norm = (float)tot * cell + eps;
    }
    double dsum = 0.0;
    for (int k = threadIdx.x; k < bins; k += KDE_BLOCK) {
        const float g = normalize ? (Sp[k] * pre_scale) / norm : Sp[k];
        if (ghat) ghat[(int64_t)p * bins + k] = g;
        if (meas) dsum += (double)disc_term(kind, g, meas[(int64_t)p * bins + k], pad);
    }
    if (meas) {
        const double dt = block_sum(dsum, red);
        if (threadIdx.x == 0) D[p] = (float)(dt / (double)batch_div);
    }
}

// a_k = dL/dghat_k = gD_p * dD_p/dghat_k (+ gghat_k);  normalize: gS_k = (a_k - cell * sum_j a_j ghat_j) / norm * pre_scale
__global__ __launch_bounds__(KDE_BLOCK) void hist_norm_disc_bwd_kernel(
    const float* __restrict__ S, int bins, int normalize, float pre_scale, float cell, float eps,
    const float* __restrict__ meas, int kind, float pad, float batch_div, const float* __restrict__ gD,
    const float* __restrict__ gghat, float* __restrict__ gS) {
    __shared__ double red[KDE_BLOCK];
    const int p = blockIdx.x;
    const float* Sp = S + (int64_t)p * bins;
    float norm = 1.0f;
    if (normalize) {
        double acc = 0.0;
        for (int k = threadIdx.x; k < bins; k += KDE_BLOCK) acc += (double)(Sp[k] * pre_scale);
        const double tot = block_sum(acc, red);
        norm = (float)tot * cell + eps;
    }
    const float gd = (meas && gD) ? gD[p] / batch_div : 0.0f;
    double dot = 0.0;
    for (int k = threadIdx.x; k < bins; k += KDE_BLOCK) {
        const float g = normalize ? (Sp[k] * pre_scale) / norm : Sp[k];
        float a = (meas && gD) ? gd * disc_grad(kind, g, meas[(int64_t)p * bins + k], pad) : 0.0f;
        if (gghat) a += gghat[(int64_t)p * bins + k];
        dot += (double)a * (double)g;
    }
    float adot = 0.0f;
    if (normalize) adot = (float)block_sum(dot, red);
    for (int k = threadIdx.x; k < bins; k += KDE_BLOCK) {
        const float g = normalize ? (Sp[k] * pre_scale) / norm : Sp[k];
        float a = (meas && gD) ? gd * disc_grad(kind, g, meas[(int64_t)p * bins + k], pad) : 0.0f;
        if (gghat) a += gghat[(int64_t)p * bins + k];
        gS[(int64_t)p * bins + k] = normalize ? (a - cell * adot) / norm * pre_scale : a;
    }
}

// ------------------------------------------------------------------------------------------------ entropy sums
__global__ __launch_bounds__(KDE_BLOCK) void mc_entropy_sums_kernel(const float* __restrict__ x,
                                                                     const float* __restrict__ logp, int64_t n, int d,
                                                                     double* __restrict__ acc2) {
    __shared__ double red[KDE_BLOCK];
    double sl = 0.0, sq = 0.0;
    for (int64_t p = (int64_t)blockIdx.x * KDE_BLOCK + threadIdx.x; p < n; p += (int64_t)gridDim.x * KDE_BLOCK) {
        sl += (double)logp[p];
        float q = 0.0f;
        for (int j = 0; j < d; ++j) {
            const float v = x[p * d + j];
            q = fmaf(v, v, q);
        }
        sq += (double)q;
    }
    const double tl = block_sum(sl, red);
    const double tq = block_sum(sq, red);
    if (threadIdx.x == 0) {                       // per-workgroup partials, summed in a fixed order by the finish kernel
        acc2[2 * blockIdx.x] = tl;
        acc2[2 * blockIdx.x + 1] = tq;
    }
}

// One wave: lane l sums the partials l, l + 64, ... of both components (independent loads: the serial two-thread version of
// rounds 1-2 took 100 us for 1 024 partials, a dependent global load each), then the 64 lane sums are added in lane order by
// lane 0.  Fixed order whatever the timing: the result is reproducible.
__global__ __launch_bounds__(64) void entropy_finish_kernel(const double* __restrict__ acc2, int nparts,
                                                            float* __restrict__ out2) {
    __shared__ double part[2][64];
    double t0 = 0.0, t1 = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 64) {
        t0 += acc2[2 * i];
        t1 += acc2[2 * i + 1];
    }
    part[0][threadIdx.x] = t0;
    part[1][threadIdx.x] = t1;
    __syncthreads();
    if (threadIdx.x < 2) {
        double t = 0.0;
        for (int l = 0; l < 64; ++l) t += part[threadIdx.x][l];
        out2[threadIdx.x] = (float)t;
    }
}

__global__ __launch_bounds__(KDE_BLOCK) void scale_rows_kernel(const float* __restrict__ x, int64_t total,
                                                                const float* __restrict__ coef, float cscale,
                                                                float* __restrict__ gx, int accumulate) {
    const float cx = coef[0] * cscale;
    for (int64_t i = (int64_t)blockIdx.x * KDE_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * KDE_BLOCK)
        gx[i] = accumulate ? fmaf(cx, x[i], gx[i]) : cx * x[i];
}

__global__ __launch_bounds__(KDE_BLOCK) void gather_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx,
                                                            float* __restrict__ dst, int64_t n, int accumulate) {
    for (int64_t i = (int64_t)blockIdx.x * KDE_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * KDE_BLOCK) {
        const int32_t j = idx[i];
        const float v = (j >= 0) ? src[j] : 0.0f;
        dst[i] = accumulate ? dst[i] + v : v;
    }
}

// ------------------------------------------------------------------------------------------------ multipole kick
// mentflow/simulate/transform.py:78-146 (MultipoleTransform.forward): z = x + i y (x = X[:,0], y = X[:,2] or 0),
// zn = z^(order-1);  non-skew: U[:,1] = X[:,1] - k Re(zn), U[:,3] = X[:,1] + k Im(zn)  (sic: the reference writes
// X[:,1], not X[:,3]);  skew: U[:,1] = X[:,1] + k Im(zn), U[:,3] = X[:,3] + k Re(zn);  k = strength / (order-1)!.
__device__ __forceinline__ void cpow(float x, float y, int m, float& re, float& im) {
    re = 1.0f;
    im = 0.0f;
    for (int i = 0; i < m; ++i) {
        const float t = re * x - im * y;
        im = re * y + im * x;
        re = t;
    }
}

__global__ __launch_bounds__(KDE_BLOCK) void multipole_fwd_kernel(const float* __restrict__ X, int64_t n, int d, int m,
                                                                   float k, int skew, float* __restrict__ U) {
    for (int64_t p = (int64_t)blockIdx.x * KDE_BLOCK + threadIdx.x; p < n; p += (int64_t)gridDim.x * KDE_BLOCK) {
        const float* xr = X + p * d;
        float* ur = U + p * d;
        const float x = xr[0], y = d > 2 ? xr[2] : 0.0f;
        float re, im;
        cpow(x, y, m, re, im);
        for (int j = 0; j < d; ++j) ur[j] = xr[j];
        if (skew) {
            ur[1] = xr[1] + k * im;
            if (d > 2) ur[3] = xr[3] + k * re;
        } else {
            ur[1] = xr[1] - k * re;
            if (d > 2) ur[3] = xr[1] + k * im;
        }
    }
}

// gX = J^T gU with d Re(zn)/dx = m Re(z^(m-1)), d Re/dy = -m Im(z^(m-1)), d Im/dx = m Im(z^(m-1)), d Im/dy = m Re(z^(m-1))
__global__ __launch_bounds__(KDE_BLOCK) void multipole_bwd_kernel(const float* __restrict__ X, int64_t n, int d, int m,
                                                                   float k, int skew, const float* __restrict__ gU,
                                                                   float* __restrict__ gX) {
    for (int64_t p = (int64_t)blockIdx.x * KDE_BLOCK + threadIdx.x; p < n; p += (int64_t)gridDim.x * KDE_BLOCK) {
        const float* xr = X + p * d;
        const float* gu = gU + p * d;
        float* gx = gX + p * d;
        const float x = xr[0], y = d > 2 ? xr[2] : 0.0f;
        float re1, im1;
        cpow(x, y, m - 1, re1, im1);
        const float dre_dx = m * re1, dre_dy = -(float)m * im1, dim_dx = m * im1, dim_dy = m * re1;
        for (int j = 0; j < d; ++j) gx[j] = gu[j];
        const float g1 = gu[1], g3 = d > 2 ? gu[3] : 0.0f;
        if (skew) {
            gx[0] += k * (g1 * dim_dx + g3 * dre_dx);
            if (d > 2) gx[2] += k * (g1 * dim_dy + g3 * dre_dy);
        } else {
            gx[0] += k * (-g1 * dre_dx + g3 * dim_dx);
            if (d > 2) {
                gx[2] += k * (-g1 * dre_dy + g3 * dim_dy);
                gx[1] += g3;          // U[:,3] reads X[:,1] in the reference
                gx[3] -= g3;          // ... and not X[:,3]
            }
        }
    }
}

static int grid_for(int64_t n, int per_block, int cap) {
    int64_t g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

}  // namespace mf

using namespace mf;

// ================================================================================================= C ABI
extern "C" int mf_gather_f32(const float* src, const int32_t* idx, float* dst, int64_t n, int accumulate, void* stream) {
    if (n <= 0) return 0;
    MF_LAUNCH(gather_kernel, grid_for(n, KDE_BLOCK, 2048), KDE_BLOCK, 0, stream, src, idx, dst, n, accumulate);
    return check_launch("mf_gather_f32");
}

static int kde_check(int64_t n, int d, int P, int B) {
    if (n < 0 || d < 1 || d > KDE_DMAX) return fail("projection kernels support 1 <= d <= %d (got d=%d)", KDE_DMAX, d);
    if (P < 1 || B < 2) return fail("need P >= 1 and at least 2 bins (P=%d, bins=%d)", P, B);
    return 0;
}

// workspace of the forward kernels: one 64-bit integer accumulator per bin
extern "C" int64_t mf_proj_kde_ws_bytes(int P, int bins) { return (int64_t)P * bins * (int64_t)sizeof(u64); }

static int fix_finish(const u64* Sacc, float* S, int64_t total, int shift, void* stream) {
    MF_LAUNCH(acc_to_float_kernel, grid_for(total, KDE_BLOCK, 1024), KDE_BLOCK, 0, stream, Sacc, S, total,
              KDE_FIX_INV * (double)((u64)1 << shift));
    return check_launch("acc_to_float");
}

static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

constexpr size_t KDE_LDS_BYTES = 156 * 1024;          // dynamic LDS a workgroup may ask for (of 160 KiB)

// particles per forward workgroup: enough workgroups to cover the chip `waves` times over, a multiple of the block
// size, at most KDE_MAX_PER_WG (fixed-point headroom)
static int kde_per_wg(int64_t n, int ngroups, int block, int waves) {
    int64_t want = (n * ngroups + (int64_t)waves * NUM_CU - 1) / ((int64_t)waves * NUM_CU);
    want = ((want + block - 1) / block) * block;
    if (want < block) want = block;
    if (want > KDE_MAX_PER_WG) want = (KDE_MAX_PER_WG / block) * block;
    return (int)want;
}

extern "C" int mf_proj_kde1d_fwd(const float* x, int64_t n, int d, const float* V, int P, const float* coords, int B,
                                  float sigma, int radius, float* S, void* ws, void* stream) {
    if (kde_check(n, d, P, B)) return 1;
    u64* Sfix = reinterpret_cast<u64*>(ws);
    if (hipMemsetAsync(Sfix, 0, sizeof(u64) * (size_t)P * B, (hipStream_t)stream) != hipSuccess) return fail("memset ws");
    if (n > 0) {
        const int R = radius < 0 ? 0 : (radius > B ? B : radius);
        const size_t per_proj = sizeof(u64) * (size_t)B + sizeof(float) * KDE_VS;
        if (per_proj + sizeof(float) * B > KDE_LDS_BYTES) return fail("too many bins for the LDS image (%d)", B);
        // Large batches: all projections that fit one image (C4: 100 x 64 bins = 51 KiB), big workgroups.  Small batches
        // (the reference's 25 000 particles): 256 threads and many small projection groups so that ~4 * NUM_CU
        // workgroups exist.
        const bool small = n <= 65536;
        // tuned on MI355X at C4 (tools/kde_sweep.py, profiles/r02_kde_sweep_1d.txt): 1024 threads, ~52 KiB images (two
        // workgroups = 32 waves per CU): 1.11-1.13 ms against 1.19-1.51 ms for 256 threads; 4 workgroup waves over the
        // chip (4096 particles per workgroup at C4) rather than 8: 2 % slower, half the flush atomics
        static const int block_env = env_int("MENTFLOW_KDE1D_BLOCK", 0), waves_env = env_int("MENTFLOW_KDE1D_WAVES", 4);
        static const int lds_env = env_int("MENTFLOW_KDE1D_LDS", 0);
        const int block = block_env ? block_env : (small ? 256 : 1024);
        size_t budget = lds_env ? (size_t)lds_env : (size_t)52 * 1024;
        if (budget > KDE_LDS_BYTES) budget = KDE_LDS_BYTES;
        int Pg = (int)((budget - sizeof(float) * B) / per_proj);
        if (Pg < 1) Pg = 1;
        if (Pg > P) Pg = P;
        if (small) {
            const int64_t wg_particles = (n + block - 1) / block;
            const int64_t want_groups = (4 * NUM_CU + wg_particles - 1) / wg_particles;
            if (want_groups > 1) {
                int pg_small = (int)((P + want_groups - 1) / want_groups);
                if (pg_small < 4) pg_small = P < 4 ? P : 4;
                if (pg_small < Pg) Pg = pg_small;
            }
        } else {
            const int ng = (P + Pg - 1) / Pg;                 // equal-sized groups
            Pg = (P + ng - 1) / ng;
        }
        const int ngroups = (P + Pg - 1) / Pg;
        const size_t smem = per_proj * Pg + sizeof(float) * B;
        const int per_wg = small ? block : kde_per_wg(n, ngroups, block, waves_env);
        const int64_t G = (n + per_wg - 1) / per_wg;
        ProfScope prof(PK_KDE1D_FWD, stream);
        bool launched = false;
#define KDE1D_CASE(RTV, BL)                                                                                           \
    if (!launched && (R == 4) == (RTV == 4) && block == BL) {                                                         \
        MF_ALLOW_DYN_SMEM((proj_kde1d_fwd_kernel<RTV, BL>), smem);                                                    \
        MF_LAUNCH((proj_kde1d_fwd_kernel<RTV, BL>), dim3((unsigned)G, ngroups), BL, smem, stream, x, n, d, V, P, Pg,   \
                  coords, B, 1.0f / sigma, R, Sfix, per_wg, kde_global_shift(n));                                                          \
        launched = true;                                                                                              \
    }
        KDE1D_CASE(4, 256) KDE1D_CASE(0, 256) KDE1D_CASE(4, 512) KDE1D_CASE(0, 512) KDE1D_CASE(4, 1024) KDE1D_CASE(0, 1024)
#undef KDE1D_CASE
        if (!launched) return fail("no 1-D KDE forward instance for block=%d (256, 512, 1024)", block);
        if (check_launch("mf_proj_kde1d_fwd")) return 1;
    }
    return fix_finish(Sfix, S, (int64_t)P * B, kde_global_shift(n), stream);
}

extern "C" int mf_proj_kde1d_bwd(const float* x, int64_t n, int d, const float* V, int P, const float* coords, int B,
                                  float sigma, int radius, const float* gS, float* gx, int accumulate, void* stream) {
    if (kde_check(n, d, P, B)) return 1;
    const size_t per_proj = sizeof(float) * ((size_t)B + KDE_VS);
    if (per_proj + sizeof(float) * B > KDE_LDS_BYTES) return fail("too many bins for the LDS image (%d)", B);
    if (n == 0) return 0;
    const int R = radius < 0 ? 0 : (radius > B ? B : radius);
    static const int block_env = env_int("MENTFLOW_KDE1D_BWD_BLOCK", 0), lds_env = env_int("MENTFLOW_KDE1D_BWD_LDS", 0);
    const int block = block_env ? block_env : (n >= 262144 ? 512 : 256);
    size_t budget = lds_env ? (size_t)lds_env : (size_t)52 * 1024;
    if (budget > KDE_LDS_BYTES) budget = KDE_LDS_BYTES;
    int Pg = (int)((budget - sizeof(float) * B) / per_proj);
    if (Pg < 1) Pg = 1;
    if (Pg > P) Pg = P;
    const size_t smem = per_proj * Pg + sizeof(float) * B;
    // lanes per particle: enough workgroups to give every SIMD a wave (>= 1024 workgroups of 4 waves), at most 8
    int CH = 1;
    while (CH < 8 && CH * 2 <= P && (n * CH + block - 1) / block < 4 * NUM_CU) CH *= 2;
    const int64_t G = (n * CH + block - 1) / block;
    ProfScope prof(PK_KDE1D_BWD, stream);
    bool launched = false;
#define KDE1DB_CASE(RTV, BL)                                                                                          \
    if (!launched && (R == 4) == (RTV == 4) && block == BL) {                                                         \
        MF_ALLOW_DYN_SMEM((proj_kde1d_bwd_kernel<RTV, BL>), smem);                                                    \
        MF_LAUNCH((proj_kde1d_bwd_kernel<RTV, BL>), dim3((unsigned)G), BL, smem, stream, x, n, d, V, P, Pg, coords, B,  \
                  1.0f / sigma, R, gS, gx, accumulate, CH);                                                           \
        launched = true;                                                                                              \
    }
    KDE1DB_CASE(4, 256) KDE1DB_CASE(0, 256) KDE1DB_CASE(4, 512) KDE1DB_CASE(0, 512) KDE1DB_CASE(4, 1024) KDE1DB_CASE(0, 1024)
#undef KDE1DB_CASE
    if (!launched) return fail("no 1-D KDE backward instance for block=%d (256, 512, 1024)", block);
    return check_launch("mf_proj_kde1d_bwd");
}

static int kde2d_check(int d, int P, int Bx, int By, int rx, int ry, size_t entry_bytes) {
    if (rx > KDE_RMAX2D || ry > KDE_RMAX2D)
        return fail("2-D KDE kernel supports a truncation radius <= %d bins (bandwidth <= 0.6 bin widths)", KDE_RMAX2D);
    if (entry_bytes * (size_t)Bx * By + sizeof(float) * (2 * KDE_VS + Bx + By) > KDE_LDS_BYTES)
        return fail("2-D histogram image %dx%d exceeds the LDS budget", Bx, By);
    return 0;
}

extern "C" int mf_proj_kde2d_fwd(const float* x, int64_t n, int d, const float* V0, const float* V1, int P,
                                  const float* coords_x, int Bx, float sigma_x, int radius_x, const float* coords_y,
                                  int By, float sigma_y, int radius_y, float* S, void* ws, void* stream) {
    if (kde_check(n, d, P, Bx) || kde_check(n, d, P, By)) return 1;
    if (kde2d_check(d, P, Bx, By, radius_x, radius_y, sizeof(u64))) return 1;
    const int BB = Bx * By;
    u64* Sfix = reinterpret_cast<u64*>(ws);
    if (hipMemsetAsync(Sfix, 0, sizeof(u64) * (size_t)P * BB, (hipStream_t)stream) != hipSuccess) return fail("memset ws");
    if (n > 0) {
        // tuned on MI355X at C5 (tools/kde_sweep.py): one 85 x 85 image (58 KiB) per 1024-thread workgroup, two
        // workgroups per CU: 6.0 ms against 6.7 (512 threads) / 9-16 ms (256 threads, the r01 shape)
        static const int lds_env = env_int("MENTFLOW_KDE2D_FWD_LDS", 60 * 1024), block_env = env_int("MENTFLOW_KDE2D_BLOCK", 1024);
        static const int waves_env = env_int("MENTFLOW_KDE2D_WAVES", 4);
        const size_t per_proj = sizeof(u64) * (size_t)BB + sizeof(float) * 2 * KDE_VS;
        const size_t fixed = sizeof(float) * ((size_t)Bx + By);
        size_t budget = (size_t)lds_env;
        if (budget > KDE_LDS_BYTES) budget = KDE_LDS_BYTES;
        if (budget < per_proj + fixed) budget = per_proj + fixed;
        int Pg = (int)((budget - fixed) / per_proj);
        if (Pg > P) Pg = P;
        const int block = block_env;
        const size_t smem = per_proj * Pg + fixed;
        const int ngroups = (P + Pg - 1) / Pg;
        const int per_wg = kde_per_wg(n, ngroups, block, waves_env);
        const int64_t G = (n + per_wg - 1) / per_wg;
        const int RT = (radius_x == 4 && radius_y == 4) ? 4 : 0;
        ProfScope prof(PK_KDE2D_FWD, stream);
        bool launched = false;
#define KDE2D_CASE(RTV, BL)                                                                                           \
    if (!launched && RT == RTV && block == BL) {                                                                      \
        MF_ALLOW_DYN_SMEM((proj_kde2d_fwd_kernel<RTV, BL>), smem);                                                    \
        MF_LAUNCH((proj_kde2d_fwd_kernel<RTV, BL>), dim3((unsigned)G, ngroups), BL, smem, stream, x, n, d, V0, V1, P,  \
                  Pg, coords_x, Bx, 1.0f / sigma_x, radius_x, coords_y, By, 1.0f / sigma_y, radius_y, Sfix, per_wg,        \
                  kde_global_shift(n));   \
        launched = true;                                                                                              \
    }
        KDE2D_CASE(4, 256) KDE2D_CASE(0, 256) KDE2D_CASE(4, 512) KDE2D_CASE(0, 512) KDE2D_CASE(4, 1024) KDE2D_CASE(0, 1024)
#undef KDE2D_CASE
        if (!launched) return fail("no 2-D KDE forward instance for block=%d (256, 512, 1024)", block);
        if (check_launch("mf_proj_kde2d_fwd")) return 1;
    }
    return fix_finish(Sfix, S, (int64_t)P * BB, kde_global_shift(n), stream);
}

extern "C" int mf_proj_kde2d_bwd(const float* x, int64_t n, int d, const float* V0, const float* V1, int P,
                                  const float* coords_x, int Bx, float sigma_x, int radius_x, const float* coords_y,
                                  int By, float sigma_y, int radius_y, const float* gS, float* gx, int accumulate,
                                  void* stream) {
    if (kde_check(n, d, P, Bx) || kde_check(n, d, P, By)) return 1;
    if (kde2d_check(d, P, Bx, By, radius_x, radius_y, sizeof(float))) return 1;
    if (n == 0) return 0;
    const int BB = Bx * By;
    // tuned on MI355X at C5 (tools/kde_sweep.py): 1024 threads x 4 particles each with four 85 x 85 images (116 KiB)
    // staged per pass — 3.8 ms against 5.3 ms for the r01 shape (256 threads x 4, one image); staging an image costs
    // L2 -> LDS traffic per workgroup, so fat workgroups win once there are enough particles to fill the chip with them.
    // Smaller batches step down to (1024, 2), (1024, 1), (512, 1), (256, 1) until >= 2 workgroups per CU exist.
    static const int lds_env = env_int("MENTFLOW_KDE2D_BWD_LDS", 0), block_env = env_int("MENTFLOW_KDE2D_BWD_BLOCK", 0);
    static const int npt_env = env_int("MENTFLOW_KDE2D_BWD_NPT", 0);
    int block_h = 256, npt_h = 1;
    {
        static const int shapes[5][2] = {{1024, 4}, {1024, 2}, {1024, 1}, {512, 1}, {256, 1}};
        for (int k = 0; k < 5; ++k) {
            block_h = shapes[k][0];
            npt_h = shapes[k][1];
            if (n / ((int64_t)block_h * npt_h) >= 2 * NUM_CU) break;
        }
    }
    const size_t per_proj = sizeof(float) * ((size_t)BB + 2 * KDE_VS);
    const size_t fixed = sizeof(float) * ((size_t)Bx + By);
    const int block = block_env ? block_env : block_h, npt = npt_env ? npt_env : npt_h;
    size_t budget = lds_env ? (size_t)lds_env : (block >= 1024 ? (size_t)118 * 1024 : (block >= 512 ? (size_t)59 * 1024 : (size_t)30 * 1024));
    if (budget > KDE_LDS_BYTES) budget = KDE_LDS_BYTES;
    if (budget < per_proj + fixed) budget = per_proj + fixed;
    int Pg = (int)((budget - fixed) / per_proj);
    if (Pg > P) Pg = P;
    const size_t smem = per_proj * Pg + fixed;
    const int64_t per_wg = (int64_t)block * npt;
    const int64_t G = (n + per_wg - 1) / per_wg;
    const int RT = (radius_x == 4 && radius_y == 4) ? 4 : 0;
    ProfScope prof(PK_KDE2D_BWD, stream);
    bool launched = false;
#define KDE2DB_CASE(RTV, BL, NP)                                                                                      \
    if (!launched && RT == RTV && block == BL && npt == NP) {                                                         \
        MF_ALLOW_DYN_SMEM((proj_kde2d_bwd_kernel<RTV, BL, NP>), smem);                                                \
        MF_LAUNCH((proj_kde2d_bwd_kernel<RTV, BL, NP>), dim3((unsigned)G), BL, smem, stream, x, n, d, V0, V1, P, Pg,   \
                  coords_x, Bx, 1.0f / sigma_x, radius_x, coords_y, By, 1.0f / sigma_y, radius_y, gS, gx, accumulate); \
        launched = true;                                                                                              \
    }
#define KDE2DB_CASES(BL, NP) KDE2DB_CASE(4, BL, NP) KDE2DB_CASE(0, BL, NP)
    KDE2DB_CASES(256, 1) KDE2DB_CASES(256, 2) KDE2DB_CASES(256, 4) KDE2DB_CASES(512, 1) KDE2DB_CASES(512, 2) KDE2DB_CASES(512, 4)
    KDE2DB_CASES(1024, 1) KDE2DB_CASES(1024, 2) KDE2DB_CASES(1024, 4)
#undef KDE2DB_CASES
#undef KDE2DB_CASE
    if (!launched) return fail("no 2-D KDE backward instance for block=%d npt=%d", block, npt);
    return check_launch("mf_proj_kde2d_bwd");
}

// geometry of the hard-binned 2-D counts kernel: as many int images as fit the budget
static int kde2d_geometry(int d, int P, int Bx, int By, int rx, int ry, int* Pg, size_t* smem, int extra,
                          int budget_floats = KDE_LDS_FLOATS) {
    (void)rx; (void)ry;
    const int BB = Bx * By;
    if (BB > KDE_LDS_FLOATS) return fail("2-D histogram image %dx%d exceeds the LDS budget", Bx, By);
    int g = budget_floats / BB;
    if (g < 1) g = 1;
    if (g > P) g = P;
    *Pg = g;
    *smem = sizeof(float) * ((size_t)g * BB + 2 * (size_t)g * d + Bx + By + extra);
    return 0;
}

extern "C" int mf_proj_hist1d_counts(const float* x, int64_t n, int d, const float* V, int P, const float* edges, int B,
                                      int32_t* counts, void* stream) {
    if (kde_check(n, d, P, B)) return 1;
    if (B + 1 > KDE_LDS_FLOATS) return fail("too many bins");
    if (hipMemsetAsync(counts, 0, sizeof(int32_t) * (size_t)P * B, (hipStream_t)stream) != hipSuccess) return fail("memset");
    if (n == 0) return 0;
    const int Pg = (KDE_LDS_FLOATS / B) < P ? (KDE_LDS_FLOATS / B) : P;
    const int ngroups = (P + Pg - 1) / Pg;
    const size_t smem = sizeof(float) * ((size_t)Pg * B + (size_t)Pg * d + B + 1);
    const int G = grid_for(n, KDE_BLOCK, (NUM_CU * 8 + ngroups - 1) / ngroups);   // >= 256 particles per workgroup
    MF_ALLOW_DYN_SMEM(proj_hist1d_kernel, smem);
    MF_LAUNCH(proj_hist1d_kernel, dim3(G, ngroups), KDE_BLOCK, smem, stream, x, n, d, V, P, Pg, edges, B, counts);
    return check_launch("mf_proj_hist1d_counts");
}

extern "C" int mf_proj_hist2d_counts(const float* x, int64_t n, int d, const float* V0, const float* V1, int P,
                                      const float* edges_x, int Bx, const float* edges_y, int By, int32_t* counts,
                                      void* stream) {
    if (kde_check(n, d, P, Bx) || kde_check(n, d, P, By)) return 1;
    int Pg;
    size_t smem;
    if (kde2d_geometry(d, P, Bx, By, 0, 0, &Pg, &smem, 2)) return 1;
    if (hipMemsetAsync(counts, 0, sizeof(int32_t) * (size_t)P * Bx * By, (hipStream_t)stream) != hipSuccess) return fail("memset");
    if (n == 0) return 0;
    const int ngroups = (P + Pg - 1) / Pg;
    const int G = grid_for(n, KDE_BLOCK, (NUM_CU * 8 + ngroups - 1) / ngroups);   // >= 256 particles per workgroup
    MF_ALLOW_DYN_SMEM(proj_hist2d_kernel, smem);
    MF_LAUNCH(proj_hist2d_kernel, dim3(G, ngroups), KDE_BLOCK, smem, stream, x, n, d, V0, V1, P, Pg, edges_x, Bx, edges_y,
              By, counts);
    return check_launch("mf_proj_hist2d_counts");
}

extern "C" int mf_hist_norm_discrepancy_fwd(const float* S, int P, int bins, int normalize, float pre_scale, float cell,
                                             float eps, const float* meas, int kind, float pad, float batch_div,
                                             float* ghat, float* D, void* stream) {
    if (P < 1 || bins < 1 || kind < 0 || kind > 2) return fail("bad arguments (P=%d bins=%d kind=%d)", P, bins, kind);
    if (meas && !D) return fail("D must be given when meas is");
    MF_LAUNCH(hist_norm_disc_fwd_kernel, P, KDE_BLOCK, 0, stream, S, bins, normalize, pre_scale, cell, eps, meas, kind, pad,
              batch_div, ghat, D);
    return check_launch("mf_hist_norm_discrepancy_fwd");
}

extern "C" int mf_hist_norm_discrepancy_bwd(const float* S, int P, int bins, int normalize, float pre_scale, float cell,
                                             float eps, const float* meas, int kind, float pad, float batch_div,
                                             const float* gD, const float* gghat, float* gS, void* stream) {
    if (P < 1 || bins < 1 || kind < 0 || kind > 2) return fail("bad arguments (P=%d bins=%d kind=%d)", P, bins, kind);
    MF_LAUNCH(hist_norm_disc_bwd_kernel, P, KDE_BLOCK, 0, stream, S, bins, normalize, pre_scale, cell, eps, meas, kind, pad,
              batch_div, gD, gghat, gS);
    return check_launch("mf_hist_norm_discrepancy_bwd");
}

extern "C" int mf_mc_entropy_sums(const float* x, const float* logp, int64_t n, int d, float* out2, double* acc2,
                                   void* stream) {
    if (d < 1) return fail("d must be >= 1");
    static_assert(2 * NUM_CU * 4 <= MF_ENTROPY_SCRATCH_DOUBLES, "scratch of mf_mc_entropy_sums");
    int parts = 0;
    if (n > 0) {
        parts = grid_for(n, KDE_BLOCK * 4, NUM_CU * 4);
        MF_LAUNCH(mc_entropy_sums_kernel, parts, KDE_BLOCK, 0, stream, x, logp, n, d, acc2);
        if (check_launch("mf_mc_entropy_sums")) return 1;
    }
    MF_LAUNCH(entropy_finish_kernel, 1, 64, 0, stream, (const double*)acc2, parts, out2);
    return check_launch("mf_mc_entropy_sums(finish)");
}

extern "C" int mf_scale_rows(const float* x, int64_t n, int d, const float* coef, float cscale, float* gx, int accumulate,
                             void* stream) {
    const int64_t total = n * d;
    if (total <= 0) return 0;
    MF_LAUNCH(scale_rows_kernel, grid_for(total, KDE_BLOCK * 4, 4096), KDE_BLOCK, 0, stream, x, total, coef, cscale, gx,
              accumulate);
    return check_launch("mf_scale_rows");
}

extern "C" int mf_multipole_kick_fwd(const float* x, int64_t n, int d, int order, float k, int skew, float* u, void* stream) {
    if (order < 3 || order > 5) return fail("MultipoleTransform requires 3 <= order <= 5 (reference: transform.py:116-132)");
    if (d < 2 || (d > 2 && d < 4)) return fail("multipole kick needs d = 2 or d >= 4");
    if (n <= 0) return 0;
    MF_LAUNCH(multipole_fwd_kernel, grid_for(n, KDE_BLOCK, 4096), KDE_BLOCK, 0, stream, x, n, d, order - 1, k, skew, u);
    return check_launch("mf_multipole_kick_fwd");
}

extern "C" int mf_multipole_kick_bwd(const float* x, int64_t n, int d, int order, float k, int skew, const float* gu, float* gx,
                                      void* stream) {
    if (order < 3 || order > 5) return fail("MultipoleTransform requires 3 <= order <= 5");
    if (d < 2 || (d > 2 && d < 4)) return fail("multipole kick needs d = 2 or d >= 4");
    if (n <= 0) return 0;
    MF_LAUNCH(multipole_bwd_kernel, grid_for(n, KDE_BLOCK, 4096), KDE_BLOCK, 0, stream, x, n, d, order - 1, k, skew, gu, gx);
    return check_launch("mf_multipole_kick_bwd");
}
