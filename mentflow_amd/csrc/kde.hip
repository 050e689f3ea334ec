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
// Design (HBM-bound byte work; no MFMA here): one lane per particle, particle rows read once (coalesced,
// d floats per lane), projection vectors / bin centres / the histogram image of a group of projections live in
// LDS; every particle touches only the 2R+1 bins whose Gaussian weight is above fp32 resolution (sigma = bw*delta,
// R = ceil(9 bw - 1/2): dropped weights < 3e-18).
// Accumulation inside a workgroup is FIXED POINT: each weight (<= 1) is converted to a 2^-50 integer and added with
// 64-bit integer LDS atomics (order independent, 8.9e-16 quantum: far-tail bins, whose log enters the KL
// discrepancy, keep their relative accuracy); the
// per-workgroup sums are flushed with one fp64 global atomic per bin and rounded to fp32 once at the end.
// Measured on MI355X (tools/ubench_lds_atomics.hip): ds_add_f32 sustains 0.33 lane-ops/clk/CU, ds_add_u64 2.7 — the
// float LDS atomic is 8x slower.
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
constexpr int KDE_MAX_PER_WG = 4096;                     // particles per workgroup: 2^12 * 2^50 < 2^63, no overflow
constexpr float KDE_EXP2_SCALE = 0.7213475204444817f;    // log2(e) / 2:  exp(-r^2/2) = exp2(-KDE_EXP2_SCALE r^2)

__device__ __forceinline__ float gauss_weight(float r) { return __builtin_amdgcn_exp2f(-KDE_EXP2_SCALE * r * r); }

// w in [0, 1] -> round-down 2^-50 fixed point (exact split: high 18 bits and low 32 bits of w * 2^50)
__device__ __forceinline__ u64 to_fix(float w) {
    const float t = w * 262144.0f;
    const unsigned hi = (unsigned)t;
    const unsigned lo = (unsigned)((t - (float)hi) * 4294967296.0f);
    return ((u64)hi << 32) | (u64)lo;
}

// grid (G, ngroups); LDS: [Pg*B] u64 image | [Pg*ds] V (row stride ds = d|1, odd) | [B] coords
template <int RT>   // RT > 0: compile-time window radius (unrolled);  RT == 0: runtime radius
__global__ __launch_bounds__(KDE_BLOCK) void proj_kde1d_fwd_kernel(
    const float* __restrict__ x, int64_t n, int d, const float* __restrict__ V, int P, int Pg,
    const float* __restrict__ coords, int B, float inv_sigma, int Rrt, double* __restrict__ Sacc) {
    MF_DYN_SMEM(u64, lds);
    u64* img = lds;
    const int ds = d | 1;
    float* Vl = reinterpret_cast<float*>(img + Pg * B);
    float* cl = Vl + Pg * ds;
    const int p_begin = blockIdx.y * Pg;
    const int np = min(Pg, P - p_begin);
    const int R = RT > 0 ? RT : Rrt;
    for (int i = threadIdx.x; i < np * B; i += KDE_BLOCK) img[i] = 0;
    for (int i = threadIdx.x; i < np * d; i += KDE_BLOCK) Vl[(i / d) * ds + (i % d)] = V[p_begin * d + i];
    for (int i = threadIdx.x; i < B; i += KDE_BLOCK) cl[i] = coords[i];
    __syncthreads();
    const float c0 = cl[0];
    const float inv_delta = 1.0f / (cl[1] - cl[0]);
    for (int64_t p = (int64_t)blockIdx.x * KDE_BLOCK + threadIdx.x; p < n; p += (int64_t)gridDim.x * KDE_BLOCK) {
        float xv[KDE_DMAX];
        load_row(x, p, d, xv);
        // every lane starts at a different projection: the 64 lanes of a wave then add into different rows of the
        // image instead of piling onto the few populated bins of one projection
        int q = (int)(threadIdx.x % (unsigned)np);
        for (int it = 0; it < np; ++it) {
            const float u = project(xv, Vl + q * ds, d);
            const int kc = centre_bin(u, c0, inv_delta, B, R);
            u64* row = img + q * B;
            if (RT > 0) {
#pragma unroll
                for (int j = -RT; j <= RT; ++j) {
                    const int k = kc + j;
                    if (k >= 0 && k < B) {
                        const float w = gauss_weight((u - cl[k]) * inv_sigma);
                        atomicAdd(&row[k], to_fix(w));
                    }
                }
            } else {
                for (int j = -R; j <= R; ++j) {
                    const int k = kc + j;
                    if (k >= 0 && k < B) {
                        const float w = gauss_weight((u - cl[k]) * inv_sigma);
                        atomicAdd(&row[k], to_fix(w));
                    }
                }
            }
            q = (q + 1 == np) ? 0 : q + 1;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < np * B; i += KDE_BLOCK) {
        const u64 v = img[i];
        if (v != 0) atomicAdd(&Sacc[(int64_t)p_begin * B + i], (double)v * KDE_FIX_INV);
    }
}

__global__ __launch_bounds__(KDE_BLOCK) void acc_to_float_kernel(const double* __restrict__ Sacc, float* __restrict__ S,
                                                                  int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * KDE_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * KDE_BLOCK)
        S[i] = (float)Sacc[i];
}

// ------------------------------------------------------------------------------------------------ 1-D backward
// grid (G); loops over projection groups; LDS: [Pg*B] gS | [Pg*d] V | [B] coords.
// CH (1, 2, 4 or 8) adjacent lanes share a particle and take every CH-th projection; their partial gradient rows are
// summed with a fixed butterfly (deterministic, no atomics).  Small batches use CH > 1: one lane per particle walks
// all P projections serially, which leaves most of the chip idle at the reference's 25 000-particle batch.
template <int RT>   // RT > 0: compile-time window radius (unrolled, branch-free);  RT == 0: runtime radius
__global__ __launch_bounds__(KDE_BLOCK) void proj_kde1d_bwd_kernel(
    const float* __restrict__ x, int64_t n, int d, const float* __restrict__ V, int P, int Pg,
    const float* __restrict__ coords, int B, float inv_sigma, int R, const float* __restrict__ gS,
    float* __restrict__ gx, int accumulate, int CH) {
    MF_DYN_SMEM(float, lds);
    float* img = lds;
    float* Vl = img + Pg * B;
    float* cl = Vl + Pg * d;
    for (int i = threadIdx.x; i < B; i += KDE_BLOCK) cl[i] = coords[i];
    __syncthreads();
    const float c0 = cl[0];
    const float inv_delta = 1.0f / (cl[1] - cl[0]);
    const int per_wg = KDE_BLOCK / CH;
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
        for (int i = threadIdx.x; i < np * B; i += KDE_BLOCK) img[i] = gS[(int64_t)p_begin * B + i];
        for (int i = threadIdx.x; i < np * d; i += KDE_BLOCK) Vl[i] = V[p_begin * d + i];
        __syncthreads();
        for (int q = chunk; q < np; q += CH) {
            const float u = project(xv, Vl + q * d, d);
            const int kc = centre_bin(u, c0, inv_delta, B, R);
            float du = 0.0f;
            if (RT > 0) {
                // window bins outside [0, B) read a clamped bin with weight 0: no divergent branch in the unrolled loop
#pragma unroll
                for (int j = -RT; j <= RT; ++j) {
                    const int k = kc + j;
                    const int kk = min(max(k, 0), B - 1);
                    const float r = (k == kk) ? (u - cl[kk]) * inv_sigma : 0.0f;   // masked too: u may be inf / NaN
                    const float g = (k == kk) ? img[q * B + kk] : 0.0f;
                    du = fmaf(g * gauss_weight(r), -r * inv_sigma, du);
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
            for (int j = 0; j < KDE_DMAX; ++j)
                if (j < d) gv[j] = fmaf(du, Vl[q * d + j], gv[j]);
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
// grid (G, ngroups); LDS: [Pg*Bx*By] u64 image | [Pg*ds] V0 | [Pg*ds] V1 | [Bx] cx | [By] cy
__global__ __launch_bounds__(KDE_BLOCK) void proj_kde2d_fwd_kernel(
    const float* __restrict__ x, int64_t n, int d, const float* __restrict__ V0, const float* __restrict__ V1, int P,
    int Pg, const float* __restrict__ coords_x, int Bx, float inv_sx, int Rx, const float* __restrict__ coords_y,
    int By, float inv_sy, int Ry, double* __restrict__ Sacc) {
    MF_DYN_SMEM(u64, lds);
    const int BB = Bx * By;
    const int ds = d | 1;
    u64* img = lds;
    float* V0l = reinterpret_cast<float*>(img + Pg * BB);
    float* V1l = V0l + Pg * ds;
    float* cxl = V1l + Pg * ds;
    float* cyl = cxl + Bx;
    const int p_begin = blockIdx.y * Pg;
    const int np = min(Pg, P - p_begin);
    for (int i = threadIdx.x; i < np * BB; i += KDE_BLOCK) img[i] = 0;
    for (int i = threadIdx.x; i < np * d; i += KDE_BLOCK) {
        V0l[(i / d) * ds + (i % d)] = V0[p_begin * d + i];
        V1l[(i / d) * ds + (i % d)] = V1[p_begin * d + i];
    }
    for (int i = threadIdx.x; i < Bx; i += KDE_BLOCK) cxl[i] = coords_x[i];
    for (int i = threadIdx.x; i < By; i += KDE_BLOCK) cyl[i] = coords_y[i];
    __syncthreads();
    const float cx0 = cxl[0], inv_dx = 1.0f / (cxl[1] - cxl[0]);
    const float cy0 = cyl[0], inv_dy = 1.0f / (cyl[1] - cyl[0]);
    for (int64_t p = (int64_t)blockIdx.x * KDE_BLOCK + threadIdx.x; p < n; p += (int64_t)gridDim.x * KDE_BLOCK) {
        float xv[KDE_DMAX];
        load_row(x, p, d, xv);
        for (int q = 0; q < np; ++q) {
            const float u0 = project(xv, V0l + q * ds, d);
            const float u1 = project(xv, V1l + q * ds, d);
            const int ka = centre_bin(u0, cx0, inv_dx, Bx, Rx);
            const int kb = centre_bin(u1, cy0, inv_dy, By, Ry);
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
                u64* row = img + q * BB + a * By;
#pragma unroll
                for (int j = 0; j < 2 * KDE_RMAX2D + 1; ++j) {
                    const int b = kb - Ry + j;
                    if (j <= 2 * Ry && b >= 0 && b < By) atomicAdd(&row[b], to_fix(wx * wy[j]));
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < np * BB; i += KDE_BLOCK) {
        const u64 v = img[i];
        if (v != 0) atomicAdd(&Sacc[(int64_t)p_begin * BB + i], (double)v * KDE_FIX_INV);
    }
}

// ------------------------------------------------------------------------------------------------ 2-D backward
// A workgroup owns KDE2D_BWD_NPT * 256 particles and walks the projection groups once.  Every thread keeps the rows and
// the gradient rows of its KDE2D_BWD_NPT particles in registers for the whole walk (no read-modify-write of gx per
// group); the gS images of a group are staged into LDS once per workgroup and group.  The group size is chosen so that
// the image stays under ~40 KB (one 85 x 85 image): several workgroups then share a CU and hide the latency of the
// data-dependent LDS reads (one workgroup per CU with five images was 2.7x slower).
constexpr int KDE2D_BWD_NPT = 4;
__global__ __launch_bounds__(KDE_BLOCK) void proj_kde2d_bwd_kernel(
    const float* __restrict__ x, int64_t n, int d, const float* __restrict__ V0, const float* __restrict__ V1, int P,
    int Pg, const float* __restrict__ coords_x, int Bx, float inv_sx, int Rx, const float* __restrict__ coords_y,
    int By, float inv_sy, int Ry, const float* __restrict__ gS, float* __restrict__ gx, int accumulate) {
    MF_DYN_SMEM(float, lds);
    const int BB = Bx * By;
    float* img = lds;
    float* V0l = img + Pg * BB;
    float* V1l = V0l + Pg * d;
    float* cxl = V1l + Pg * d;
    float* cyl = cxl + Bx;
    for (int i = threadIdx.x; i < Bx; i += KDE_BLOCK) cxl[i] = coords_x[i];
    for (int i = threadIdx.x; i < By; i += KDE_BLOCK) cyl[i] = coords_y[i];
    __syncthreads();
    const float cx0 = cxl[0], inv_dx = 1.0f / (cxl[1] - cxl[0]);
    const float cy0 = cyl[0], inv_dy = 1.0f / (cyl[1] - cyl[0]);
    const int64_t base = (int64_t)blockIdx.x * KDE_BLOCK * KDE2D_BWD_NPT + threadIdx.x;
    float xv[KDE2D_BWD_NPT][KDE_DMAX], gv[KDE2D_BWD_NPT][KDE_DMAX];
#pragma unroll
    for (int t = 0; t < KDE2D_BWD_NPT; ++t) {
        const int64_t p = base + (int64_t)t * KDE_BLOCK;
        load_row(x, p < n ? p : n - 1, d, xv[t]);
#pragma unroll
        for (int j = 0; j < KDE_DMAX; ++j) gv[t][j] = 0.0f;
    }
    for (int p_begin = 0; p_begin < P; p_begin += Pg) {
        const int np = min(Pg, P - p_begin);
        __syncthreads();
        for (int i = threadIdx.x; i < np * BB; i += KDE_BLOCK) img[i] = gS[(int64_t)p_begin * BB + i];
        for (int i = threadIdx.x; i < np * d; i += KDE_BLOCK) {
            V0l[i] = V0[p_begin * d + i];
            V1l[i] = V1[p_begin * d + i];
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < KDE2D_BWD_NPT; ++t) {
            for (int q = 0; q < np; ++q) {
                const float u0 = project(xv[t], V0l + q * d, d);
                const float u1 = project(xv[t], V1l + q * d, d);
                const int ka = centre_bin(u0, cx0, inv_dx, Bx, Rx);
                const int kb = centre_bin(u1, cy0, inv_dy, By, Ry);
                float wy[2 * KDE_RMAX2D + 1], dy[2 * KDE_RMAX2D + 1];
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
                    dy[j] = dw;
                }
                float du0 = 0.0f, du1 = 0.0f;
                for (int i = 0; i <= 2 * Rx; ++i) {
                    const int a = ka - Rx + i;
                    if (a < 0 || a >= Bx) continue;
                    const float r = (u0 - cxl[a]) * inv_sx;
                    const float wx = gauss_weight(r);
                    const float dwx = -r * inv_sx * wx;
                    const float* row = img + q * BB + a * By;
                    float sa = 0.0f, sb = 0.0f;
#pragma unroll
                    for (int j = 0; j < 2 * KDE_RMAX2D + 1; ++j) {
                        const int b = kb - Ry + j;
                        if (j <= 2 * Ry && b >= 0 && b < By) {
                            const float g = row[b];
                            sa = fmaf(g, wy[j], sa);
                            sb = fmaf(g, dy[j], sb);
                        }
                    }
                    du0 = fmaf(dwx, sa, du0);
                    du1 = fmaf(wx, sb, du1);
                }
#pragma unroll
                for (int j = 0; j < KDE_DMAX; ++j)
                    if (j < d) gv[t][j] = fmaf(du0, V0l[q * d + j], fmaf(du1, V1l[q * d + j], gv[t][j]));
            }
        }
    }
#pragma unroll
    for (int t = 0; t < KDE2D_BWD_NPT; ++t) {
        const int64_t p = base + (int64_t)t * KDE_BLOCK;
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
    if (threadIdx.x == 0) {
        atomicAdd(&acc2[0], tl);
        atomicAdd(&acc2[1], tq);
    }
}

__global__ void entropy_finish_kernel(const double* __restrict__ acc2, float* __restrict__ out2) {
    if (threadIdx.x < 2) out2[threadIdx.x] = (float)acc2[threadIdx.x];
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

extern "C" int64_t mf_proj_kde_ws_bytes(int P, int bins) { return (int64_t)P * bins * (int64_t)sizeof(double); }

static int fix_finish(const double* Sacc, float* S, int64_t total, void* stream) {
    MF_LAUNCH(acc_to_float_kernel, grid_for(total, KDE_BLOCK, 1024), KDE_BLOCK, 0, stream, Sacc, S, total);
    return check_launch("acc_to_float");
}

extern "C" int mf_proj_kde1d_fwd(const float* x, int64_t n, int d, const float* V, int P, const float* coords, int B,
                                  float sigma, int radius, float* S, void* ws, void* stream) {
    if (kde_check(n, d, P, B)) return 1;
    const int budget = KDE_LDS_FLOATS / 2;            // 64-bit bins
    if (B > budget) return fail("too many bins for the LDS image (%d)", B);
    double* Sfix = reinterpret_cast<double*>(ws);
    if (hipMemsetAsync(Sfix, 0, sizeof(double) * (size_t)P * B, (hipStream_t)stream) != hipSuccess) return fail("memset ws");
    if (n > 0) {
        const int R = radius < 0 ? 0 : (radius > B ? B : radius);
        int Pg = (budget / B) < P ? (budget / B) : P;
        {   // small batches: more (smaller) projection groups, so that ~4 * NUM_CU workgroups exist
            const int64_t wg_particles = (n + KDE_BLOCK - 1) / KDE_BLOCK;
            const int64_t want_groups = (4 * NUM_CU + wg_particles - 1) / wg_particles;
            if (want_groups > 1) {
                int pg_small = (int)((P + want_groups - 1) / want_groups);
                if (pg_small < 4) pg_small = P < 4 ? P : 4;
                if (pg_small < Pg) Pg = pg_small;
            }
        }
        const int ngroups = (P + Pg - 1) / Pg;
        const int ds = d | 1;
        const size_t smem = sizeof(u64) * (size_t)Pg * B + sizeof(float) * ((size_t)Pg * ds + B);
        int G = grid_for(n, KDE_BLOCK, (NUM_CU * 8 + ngroups - 1) / ngroups);   // >= 256 particles per workgroup
        if ((n + G - 1) / G > KDE_MAX_PER_WG) G = (int)((n + KDE_MAX_PER_WG - 1) / KDE_MAX_PER_WG);
        ProfScope prof(PK_KDE1D_FWD, stream);
        if (R == 4) {
            MF_ALLOW_DYN_SMEM(proj_kde1d_fwd_kernel<4>, smem);
            MF_LAUNCH(proj_kde1d_fwd_kernel<4>, dim3(G, ngroups), KDE_BLOCK, smem, stream, x, n, d, V, P, Pg, coords, B,
                      1.0f / sigma, R, Sfix);
        } else {
            MF_ALLOW_DYN_SMEM(proj_kde1d_fwd_kernel<0>, smem);
            MF_LAUNCH(proj_kde1d_fwd_kernel<0>, dim3(G, ngroups), KDE_BLOCK, smem, stream, x, n, d, V, P, Pg, coords, B,
                      1.0f / sigma, R, Sfix);
        }
        if (check_launch("mf_proj_kde1d_fwd")) return 1;
    }
    return fix_finish(Sfix, S, (int64_t)P * B, stream);
}

extern "C" int mf_proj_kde1d_bwd(const float* x, int64_t n, int d, const float* V, int P, const float* coords, int B,
                                  float sigma, int radius, const float* gS, float* gx, int accumulate, void* stream) {
    if (kde_check(n, d, P, B)) return 1;
    if (B > KDE_LDS_FLOATS) return fail("too many bins for the LDS image (%d)", B);
    if (n == 0) return 0;
    const int R = radius < 0 ? 0 : (radius > B ? B : radius);
    const int Pg = (KDE_LDS_FLOATS / B) < P ? (KDE_LDS_FLOATS / B) : P;
    const size_t smem = sizeof(float) * ((size_t)Pg * B + (size_t)Pg * d + B);
    // lanes per particle: enough workgroups to give every SIMD a wave (>= 1024 workgroups of 4 waves), at most 8
    int CH = 1;
    while (CH < 8 && CH * 2 <= P && (n * CH + KDE_BLOCK - 1) / KDE_BLOCK < 4 * NUM_CU) CH *= 2;
    const int64_t G = (n * CH + KDE_BLOCK - 1) / KDE_BLOCK;
    ProfScope prof(PK_KDE1D_BWD, stream);
    if (R == 4) {
        MF_ALLOW_DYN_SMEM(proj_kde1d_bwd_kernel<4>, smem);
        MF_LAUNCH(proj_kde1d_bwd_kernel<4>, dim3((unsigned)G), KDE_BLOCK, smem, stream, x, n, d, V, P, Pg, coords, B,
                  1.0f / sigma, R, gS, gx, accumulate, CH);
    } else {
        MF_ALLOW_DYN_SMEM(proj_kde1d_bwd_kernel<0>, smem);
        MF_LAUNCH(proj_kde1d_bwd_kernel<0>, dim3((unsigned)G), KDE_BLOCK, smem, stream, x, n, d, V, P, Pg, coords, B,
                  1.0f / sigma, R, gS, gx, accumulate, CH);
    }
    return check_launch("mf_proj_kde1d_bwd");
}

static int kde2d_geometry(int d, int P, int Bx, int By, int rx, int ry, int* Pg, size_t* smem, int extra,
                          int budget_floats = KDE_LDS_FLOATS) {
    if (rx > KDE_RMAX2D || ry > KDE_RMAX2D)
        return fail("2-D KDE kernel supports a truncation radius <= %d bins (bandwidth <= 0.6 bin widths)", KDE_RMAX2D);
    const int BB = Bx * By;
    if (BB > KDE_LDS_FLOATS) return fail("2-D histogram image %dx%d exceeds the LDS budget", Bx, By);
    int g = budget_floats / BB;
    if (g < 1) g = 1;
    if (g > P) g = P;
    *Pg = g;
    *smem = sizeof(float) * ((size_t)g * BB + 2 * (size_t)g * d + Bx + By + extra);
    return 0;
}

extern "C" int mf_proj_kde2d_fwd(const float* x, int64_t n, int d, const float* V0, const float* V1, int P,
                                  const float* coords_x, int Bx, float sigma_x, int radius_x, const float* coords_y,
                                  int By, float sigma_y, int radius_y, float* S, void* ws, void* stream) {
    if (kde_check(n, d, P, Bx) || kde_check(n, d, P, By)) return 1;
    if (radius_x > KDE_RMAX2D || radius_y > KDE_RMAX2D)
        return fail("2-D KDE kernel supports a truncation radius <= %d bins (bandwidth <= 0.6 bin widths)", KDE_RMAX2D);
    const int BB = Bx * By;
    const int budget_max = 18432;                     // 144 KiB of 64-bit bins
    if (BB > budget_max) return fail("2-D histogram image %dx%d exceeds the LDS budget", Bx, By);
    // images per workgroup: ~72 KiB, so that two workgroups share a CU
    static const int fwd_budget = [] { const char* e = getenv("MENTFLOW_KDE2D_FWD_BINS"); return e ? atoi(e) : 9216; }();
    const int budget = fwd_budget > BB ? (fwd_budget > budget_max ? budget_max : fwd_budget) : BB;
    double* Sfix = reinterpret_cast<double*>(ws);
    if (hipMemsetAsync(Sfix, 0, sizeof(double) * (size_t)P * BB, (hipStream_t)stream) != hipSuccess) return fail("memset ws");
    if (n > 0) {
        int Pg = budget / BB;
        if (Pg > P) Pg = P;
        const int ds = d | 1;
        const size_t smem = sizeof(u64) * (size_t)Pg * BB + sizeof(float) * (2 * (size_t)Pg * ds + Bx + By);
        const int ngroups = (P + Pg - 1) / Pg;
        int G = grid_for(n, KDE_BLOCK, (NUM_CU * 8 + ngroups - 1) / ngroups);   // >= 256 particles per workgroup
        if ((n + G - 1) / G > KDE_MAX_PER_WG) G = (int)((n + KDE_MAX_PER_WG - 1) / KDE_MAX_PER_WG);
        ProfScope prof(PK_KDE2D_FWD, stream);
        MF_ALLOW_DYN_SMEM(proj_kde2d_fwd_kernel, smem);
        MF_LAUNCH(proj_kde2d_fwd_kernel, dim3(G, ngroups), KDE_BLOCK, smem, stream, x, n, d, V0, V1, P, Pg, coords_x, Bx,
                  1.0f / sigma_x, radius_x, coords_y, By, 1.0f / sigma_y, radius_y, Sfix);
        if (check_launch("mf_proj_kde2d_fwd")) return 1;
    }
    return fix_finish(Sfix, S, (int64_t)P * BB, stream);
}

extern "C" int mf_proj_kde2d_bwd(const float* x, int64_t n, int d, const float* V0, const float* V1, int P,
                                  const float* coords_x, int Bx, float sigma_x, int radius_x, const float* coords_y,
                                  int By, float sigma_y, int radius_y, const float* gS, float* gx, int accumulate,
                                  void* stream) {
    if (kde_check(n, d, P, Bx) || kde_check(n, d, P, By)) return 1;
    int Pg;
    size_t smem;
    // ~40 KB of images per workgroup: four workgroups per CU (see the kernel's header comment)
    static const int bwd_budget = [] { const char* e = getenv("MENTFLOW_KDE2D_BWD_FLOATS"); return e ? atoi(e) : 10240; }();
    if (kde2d_geometry(d, P, Bx, By, radius_x, radius_y, &Pg, &smem, 0, bwd_budget)) return 1;
    if (n == 0) return 0;
    const int64_t per_wg = (int64_t)KDE_BLOCK * KDE2D_BWD_NPT;
    const int64_t G = (n + per_wg - 1) / per_wg;
    ProfScope prof(PK_KDE2D_BWD, stream);
    MF_ALLOW_DYN_SMEM(proj_kde2d_bwd_kernel, smem);
    MF_LAUNCH(proj_kde2d_bwd_kernel, dim3((unsigned)G), KDE_BLOCK, smem, stream, x, n, d, V0, V1, P, Pg, coords_x, Bx,
              1.0f / sigma_x, radius_x, coords_y, By, 1.0f / sigma_y, radius_y, gS, gx, accumulate);
    return check_launch("mf_proj_kde2d_bwd");
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
    if (hipMemsetAsync(acc2, 0, 2 * sizeof(double), (hipStream_t)stream) != hipSuccess) return fail("memset");
    if (n > 0) {
        MF_LAUNCH(mc_entropy_sums_kernel, grid_for(n, KDE_BLOCK * 4, NUM_CU * 4), KDE_BLOCK, 0, stream, x, logp, n, d, acc2);
        if (check_launch("mf_mc_entropy_sums")) return 1;
    }
    MF_LAUNCH(entropy_finish_kernel, 1, 64, 0, stream, (const double*)acc2, out2);
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
