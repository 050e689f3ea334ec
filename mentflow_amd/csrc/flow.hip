// Autoregressive flow layer kernels for gfx950: masked-MLP conditioner on fp32 MFMA fused with the
// rational-quadratic-spline (NSF) or affine (MAF) univariate transform and its log-det.
//
// What this replaces (reference = austin-hoover/ment-flow, arithmetic in zuko 1.3.1):
//   WrappedZukoFlow.sample_and_log_prob / forward        mentflow/generate/flows/zuko.py:24-29
//   build_flow (NSF / MAF, inverted)                     mentflow/generate/build.py:13-46
//   zuko MaskedMLP (4x F.linear(x, mask*W, b) + ReLU), MonotonicRQSTransform / MonotonicAffineTransform
//   call_and_ladj, DependentTransform ladj sum, DiagNormal.log_prob — and their autograd backward.
//
// Layout of the computation (one launch per flow layer, forward and backward):
//   * the layer's masked weights ("image", <= 154 KiB) are staged once per workgroup into LDS in natural
//     [out][in] order with an odd row stride (65 / d|1): both the forward fragments A[out][k] and the
//     transposed fragments A[in][k=out] needed by the backward are then conflict-free ds_read_b32 streams.
//   * a wave owns a tile of 32 particles: particle = MFMA column (lane & 31), features = MFMA rows
//     (accumulator registers).  v_mfma_f32_32x32x2_f32 with the WEIGHTS as the A operand and the activations as
//     the B operand: the accumulator of one layer (lane half hh holds rows (r&3)+8(r>>2)+4hh) is fed back as the
//     B operand of the next layer with the k-pairing (row of half 0, row of half 1) — activations never leave
//     registers, no LDS round trip, no transposes.
//   * the 3K-1 spline parameters of one feature land in the two lanes (col, col+32) of the particle: widths (+ the
//     first half of the derivatives) in half 0, heights (+ the rest) in half 1 — the output rows of the last
//     linear layer are permuted for that when the image is packed (mentflow_amd/generate/packing.py).  Both
//     softmaxes run in the same instructions; five ds_bpermute exchanges finish a spline evaluation.
//   * backward recomputes the forward from the layer input (nothing but x[N,d] per layer is saved), produces
//     dL/dx in registers the same way (transposed fragments), and writes the per-particle pre-activation gradients
//     and activations as 256-byte rows to an HBM scratch, laid out so that the parameter-gradient contraction over
//     particles (outer_accum kernel, particles = MFMA k) reads both operands as coalesced fragments.
#include "common.h"
#include <stdlib.h>

namespace mf {

constexpr int HID = 64;              // hidden width of the conditioner (reference default, config/gen/flow.yaml:3)
constexpr int WS = HID + 1;          // LDS row stride of the 64-wide weight matrices (odd: conflict-free)
constexpr int FLOW_BLOCK = 512;      // 8 waves, one workgroup per CU (the image fills most of the 160 KiB LDS)
constexpr int FLOW_WAVES = FLOW_BLOCK / WAVE;
constexpr int FLOW_DMAX = 7;         // d*64*65 + trunk must fit LDS
constexpr float RQS_BOUND = 5.0f;
constexpr float LOG_SLOPE_INV = 1.0f / 6.907755278982137f;   // 1/|log(1e-3)|

// row (within a 32-row MFMA tile) held by accumulator register r of lane half hh
__device__ __forceinline__ constexpr int rowmap(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

struct ImageLayout {
    int S0, offW0, offB0, offWh, offW3, offB3, total;
};
__host__ __device__ inline ImageLayout image_layout(int d, int L, int nblk) {
    ImageLayout g;
    g.S0 = d | 1;
    g.offW0 = 0;
    g.offB0 = HID * g.S0;
    g.offWh = g.offB0 + HID;
    g.offW3 = g.offWh + (L - 1) * (HID * WS + HID);
    g.offB3 = g.offW3 + nblk * HID * WS;
    g.total = g.offB3 + nblk * HID;
    return g;
}

typedef float f32x16_t __attribute__((ext_vector_type(16)));

// k-step ranges that skip the all-zero blocks of the autoregressive masks (hidden units are placed sorted by
// dependency class, two per k-step: mentflow_amd/generate/packing.py).  Computed on the host from (d, order).
struct Sparsity {
    int kend_h[2];              // hidden->hidden, output tile rt: k-steps [0, kend_h[rt])
    int kbeg_ht[2];             // transposed hidden->hidden, output (=input-unit) tile rt: k-steps [kbeg_ht[rt], 32)
    int kend3[FLOW_DMAX + 1];   // last layer, output block i: k-steps [0, kend3[i])   (0: the block is pure bias)
    int rt1[FLOW_DMAX + 1];     // transposed last layer, block i: hidden tile 1 receives anything?
};

static Sparsity make_sparsity(int d, const int32_t* order, int nblk) {
    Sparsity sp;
    sp.kend_h[0] = sp.kend_h[1] = 32;
    sp.kbeg_ht[0] = sp.kbeg_ht[1] = 0;
    for (int i = 0; i <= FLOW_DMAX; ++i) { sp.kend3[i] = 32; sp.rt1[i] = 1; }
    if (order == nullptr || d < 2) return sp;                       // dense
    int cum[FLOW_DMAX + 2];
    for (int c = 0; c <= d; ++c) {
        int cnt = 0;
        for (int u = 0; u < HID; ++u) cnt += (1 + u % (d - 1)) <= c;
        cum[c] = cnt;
    }
    auto class_of = [&](int j) { int c = 1; while (cum[c] <= j) ++c; return c; };
    for (int rt = 0; rt < 2; ++rt) {
        sp.kend_h[rt] = (cum[class_of(32 * rt + 31)] + 1) / 2;
        sp.kbeg_ht[rt] = cum[class_of(32 * rt) - 1] / 2;
    }
    if (nblk == d) {                                                  // one output block per feature (RQS)
        for (int i = 0; i < d; ++i) {
            sp.kend3[i] = (cum[order[i]] + 1) / 2;
            sp.rt1[i] = cum[order[i]] > 32;
        }
    } else {                                                          // single block holding every feature (affine)
        sp.kend3[0] = 32;
        sp.rt1[0] = 1;
    }
    return sp;
}

// The weight image in LDS is constant for the whole kernel, so the compiler hoists the (loop-invariant) bias and
// weight-fragment loads of every layer out of the particle-tile loop and then spills them (104 VGPRs spilled in the
// forward kernel).  A compiler-only memory barrier at the top of each tile keeps the loads next to their MFMAs.
#define MF_NO_HOIST() asm volatile("" ::: "memory")

__device__ __forceinline__ f32x16_t mfma(float a, float b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

template <int BLOCK = FLOW_BLOCK>
__device__ __forceinline__ void stage_image(float* lds, const float* __restrict__ image, int total) {
    for (int i = threadIdx.x * 4; i < total; i += BLOCK * 4)
        *reinterpret_cast<float4*>(lds + i) = *reinterpret_cast<const float4*>(image + i);
    __syncthreads();
}

// bias[32*rt + row] broadcast into the accumulator layout
__device__ __forceinline__ f32x16_t bias_tile(const float* b, int rt, int hh) {
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = b[32 * rt + rowmap(r, hh)];
    return acc;
}

// out[2] = W[64 x 64] * in[2]  (+ bias), weights natural [out][in] with stride WS
// k-steps [0, kend0) for output tile 0 and [0, kend1) for tile 1 (wave-uniform bounds: masked-out blocks skipped).
// Steps are issued in groups of four under ONE branch, so that the four weight-fragment ds_reads are in flight together
// instead of one exposed LDS round trip per MFMA (bounds are rounded to the group: the extra steps multiply zeros).
// (Measured alternatives: one branch per step 8 % slower; a two-deep software-pipelined chain with per-step bounds
// 14 % slower — too many tiny basic blocks; requesting group g + 1 before the MFMAs of group g 4 % slower.)
__device__ __forceinline__ constexpr int kcol(int s) { return 32 * (s >> 4) + rowmap(s & 15, 0); }

__device__ __forceinline__ void linear64s(const float* W, int stride, const float* b, const f32x16_t (&in)[2],
                                          f32x16_t (&out)[2], int col, int hh, int kend0, int kend1) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        f32x16_t acc = bias_tile(b, rt, hh);
        const float* wrow = W + (32 * rt + col) * stride + 4 * hh;
        const int kend = rt ? kend1 : kend0;
#pragma unroll
        for (int s4 = 0; s4 < 32; s4 += 4) {
            if (s4 < kend) {
                float a[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) a[j] = wrow[kcol(s4 + j)];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = mfma(a[j], in[(s4 + j) >> 4][(s4 + j) & 15], acc);
            }
        }
        out[rt] = acc;
    }
}
// out[2] = W[64 x 64] * in[2]  (+ bias), weights natural [out][in] with stride WS
__device__ __forceinline__ void linear64(const float* W, const float* b, const f32x16_t (&in)[2], f32x16_t (&out)[2],
                                         int col, int hh, int kend0, int kend1) {
    linear64s(W, WS, b, in, out, col, hh, kend0, kend1);
}

// out[2] += W^T * in[2]   (out rows = input units of W, contraction over W's output units)
// k-steps [kbeg0, 32) for output tile 0 and [kbeg1, 32) for tile 1 (rounded down to a group of four)
__device__ __forceinline__ void linear64_t(const float* W, const f32x16_t (&in)[2], f32x16_t (&out)[2], int col, int hh,
                                           int kbeg0, int kbeg1) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        f32x16_t acc = out[rt];
        const float* wcol = W + 4 * hh * WS + 32 * rt + col;
        const int kbeg = rt ? kbeg1 : kbeg0;
#pragma unroll
        for (int s4 = 0; s4 < 32; s4 += 4) {
            if (s4 + 4 > kbeg) {
                float a[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) a[j] = wcol[kcol(s4 + j) * WS];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = mfma(a[j], in[(s4 + j) >> 4][(s4 + j) & 15], acc);
            }
        }
        out[rt] = acc;
    }
}

// gh[2] += Wblk^T * gv   for one last-layer block stored with row stride `stride` (dense image: WS; compact image: only
// columns < stride - 1 exist, other lanes contribute zero)
__device__ __forceinline__ void linear64s_t(const float* W, int stride, const float (&gv)[32], f32x16_t (&out)[2], int col,
                                            int hh, bool need0, bool need1) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        if (rt ? need1 : need0) {
            f32x16_t acc = out[rt];
            const bool stored = 32 * rt + col < stride - 1;            // this lane's hidden column exists in the block
            const float* wcol = W + 4 * hh * stride + (stored ? 32 * rt + col : 0);
            const float keep = stored ? 1.0f : 0.0f;
#pragma unroll
            for (int s4 = 0; s4 < 32; s4 += 4) {
                float a[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) a[j] = wcol[kcol(s4 + j) * stride] * keep;
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = mfma(a[j], gv[s4 + j], acc);
            }
            out[rt] = acc;
        }
    }
}

// max(x, 0) in ONE instruction.  `fmaxf(x, 0.0f)` costs two under the IEEE mode the kernels run in: the compiler first
// quiets a possible signalling NaN with `v_max_f32 t, x, x` — 32 extra VALU instructions per ReLU of a 64-row tile, 96 per
// tile in the forward kernel (3.4 % of its VALU work), 128 per group in the fused backward.  A signed INTEGER maximum of the
// bit pattern with 0 is the same function (positive floats order like their patterns, every negative float and -0 has the
// sign bit set) and has no NaN rule to honour: v_max_i32.  (+NaN stays NaN, as in torch.relu; fmaxf returned 0.)  It must
// stay a compiler-visible instruction: an inline-asm v_max_f32 would need the MFMA -> VALU wait states placed by hand.
__device__ __forceinline__ float relu1(float x) {
    int b;
    memcpy(&b, &x, 4);
    b = b > 0 ? b : 0;
    float y;
    memcpy(&y, &b, 4);
    return y;
}
__device__ __forceinline__ void relu2(f32x16_t (&h)[2]) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) h[rt][r] = relu1(h[rt][r]);
}

// input layer: h = relu(W0[64 x d] * x + b0);  xb[s] = x[2s + hh] (0 beyond d)
__device__ __forceinline__ void input_layer(const float* W0, const float* b0, int S0, int d, const float (&xb)[4],
                                            f32x16_t (&h)[2], int col, int hh) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        f32x16_t acc = bias_tile(b0, rt, hh);
        const float* wrow = W0 + (32 * rt + col) * S0 + hh;
#pragma unroll
        for (int s = 0; s < 4; ++s)
            if (2 * s < d) acc = mfma(wrow[2 * s], xb[s], acc);
        h[rt] = acc;
    }
    relu2(h);
}

// exp(x) as one v_exp_f32: exp2(x * log2(e)).  Only used on soft-clipped arguments (|x| < 6.91), where the rounding of
// the product costs at most |x| * 1.44 * 2^-24 < 6e-7 relative — the same order as the fp32 rounding of everything
// downstream; a compensated argument (hi + lo split) was measured to cost ~5 % of the forward kernel.
constexpr float LOG2E = 1.4426950408889634f;
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * LOG2E); }

__device__ __forceinline__ float soft_clip(float v, float a) { return v * fast_rcp(fmaf(fabsf(v), a, 1.0f)); }
__device__ __forceinline__ float soft_clip_grad(float v, float a) {
    const float ia = fast_rcp(fmaf(fabsf(v), a, 1.0f));
    return ia * ia;
}

// The two lanes (col, col + 32) of a particle exchange a value: lo = the value held by lane col, hi = the one held by
// lane col + 32, both results in both lanes.  gfx950: ONE v_permlane32_swap_b32 (a VALU instruction; the generic
// __shfl_xor(v, 32) is a ds_bpermute_b32, i.e. an LDS round trip that a lone wave per SIMD sits through, five times per
// spline).  v_permlane32_swap vdst, vsrc swaps lanes 32..63 of vdst with lanes 0..31 of vsrc; with the same value in
// both operands vdst becomes {lo, lo} and vsrc {hi, hi}.
__device__ __forceinline__ void half_pair(float v, int hh, float& lo, float& hi) {
#if defined(MF_EMU) || defined(MF_NO_PERMLANE)
    const float o = __shfl_xor(v, 32);
    lo = hh ? o : v;
    hi = hh ? v : o;
#else
    (void)hh;
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    lo = __builtin_bit_cast(float, (unsigned)r[0]);
    hi = __builtin_bit_cast(float, (unsigned)r[1]);
#endif
}
__device__ __forceinline__ void half_pair(int v, int hh, int& lo, int& hi) {
#if defined(MF_EMU) || defined(MF_NO_PERMLANE)
    const int o = __shfl_xor(v, 32);
    lo = hh ? o : v;
    hi = hh ? v : o;
#else
    (void)hh;
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    lo = (int)r[0];
    hi = (int)r[1];
#endif
}

// float -> unsigned with saturation: v_cvt_u32_f32 (negative / NaN -> 0, >= 2^32 -> 0xffffffff).  Written as the
// instruction itself: a C++ cast leaves out-of-range conversions undefined, and x is unbounded.
__device__ __forceinline__ unsigned cvt_u32_sat(float f) {
#ifdef MF_EMU
    return !(f > 0.0f) ? 0u : (f >= 4294967296.0f ? 0xffffffffu : (unsigned)f);
#else
    unsigned r;
    asm("v_cvt_u32_f32_e32 %0, %1" : "=v"(r) : "v"(f));
    return r;
#endif
}

// (t[idx], t[idx + 1]) of a table of N + 1 registers for a per-lane index idx in [0, N - 1], as a binary tree of selects on
// the BITS of idx: the bit masks are formed once (log2 N compares), every select then reads a mask that was written long
// before.  The linear form "for j: if (idx == j) ..." costs a compare per entry, and on gfx950 a VALU instruction that
// reads a mask needs two wait states after the VALU compare that wrote it — the compiler pads each compare / select pair
// with s_nop (80 issue slots for the 21 spline knots; this tree: 29 selects + 9 for the masks).
// The tree runs over the triples (t[2j], t[2j+1], t[2j+2]); bit 0 of idx picks the pair out of the surviving triple.
template <int N, typename T>
__device__ __forceinline__ void select_pair(const T (&t)[N + 1], int idx, T& lo, T& hi) {
    constexpr int M = (N + 1) / 2;
    T a[M][3];
#pragma unroll
    for (int j = 0; j < M; ++j) {
        a[j][0] = t[2 * j];
        a[j][1] = t[2 * j + 1 <= N ? 2 * j + 1 : N];
        a[j][2] = t[2 * j + 2 <= N ? 2 * j + 2 : N];
    }
    const int j2 = idx >> 1;
#pragma unroll
    for (int b = 0; (1 << b) < M; ++b) {
        const bool bit = ((j2 >> b) & 1) != 0;
#pragma unroll
        for (int j = 0; j + (1 << b) < M; j += 2 << b)
#pragma unroll
            for (int c = 0; c < 3; ++c) a[j][c] = bit ? a[j + (1 << b)][c] : a[j][c];
    }
    const bool odd = (idx & 1) != 0;
    lo = odd ? a[0][1] : a[0][0];
    hi = odd ? a[0][2] : a[0][1];
}

// ------------------------------------------------------------------------------------------------------------
// Rational-quadratic spline of one feature, evaluated cooperatively by the two lanes (col, col+32) of a particle.
// v[32]: this lane's slots of the conditioner output (half 0: K widths, then derivatives 0..KD0-1;
//        half 1: K heights, then derivatives KD0..K-2).  zuko MonotonicRQSTransform.
// MODE 0: forward (y, ladj).  MODE 1: forward + adjoint: also returns g[32] = dL/dv (same slot layout) and
// gx = dL/dx (direct path) for upstream gy = dL/dy, gl = dL/dladj.  MODE 2: inverse — `x` is the transformed value,
// the bin search runs on the heights (half 1) and y_out returns the pre-image (zuko MonotonicRQSTransform._inverse).
//
// Cumulative bin probabilities.  torch.cumsum on the CPU accumulates float32 in double and rounds every prefix to
// float32 (the reference path).  Here the normalised probabilities are quantised to 2^-31 (round to nearest) and summed as
// 32-bit integers: the prefix sums are EXACT sums of the quantised terms (|error| <= 20 * 2^-32 = 5e-9, an order of
// magnitude under the float32 rounding of the prefix itself), converted to float32 once — the same "exact sum, one
// rounding" the double accumulator gives, for 5 full-rate instructions per bin (fma, cvt, add, compare, add-carry) instead
// of the 44 cycles of cvt_f64 / add_f64 / cvt_f32.  The bin search compares the integer prefixes with the (exactly
// converted) query, so it is a total order consistent with the knots that are then used.
// K > 0: bins known at compile time (the built fast instances: 8 and 20).  K == RQS_ANY: any 2 <= bins <= 21 at run time
// (`kbins`), for configurations outside the fast instances (experiments/setup.py:119-121 takes bins from the config): the
// slots are laid out for the maximum — logits in slots 0..20 (slots >= bins unused), derivatives from slot 21 — and the
// loops over 21 bins mask the unused ones.  Slower (all 21 iterations run), same arithmetic.
constexpr int RQS_ANY = -1;
constexpr int RQS_KMAX = 21;
struct NoSink {
    __device__ __forceinline__ void operator()(int, float) const {}
};
// sink(m, g[m]) is called as soon as slot m of the adjoint is final (MODE 1): the fused backward stores it into the LDS
// staging tile right there, so that the 8 KB a wave stages per feature trickle through the LDS write path underneath the
// adjoint's VALU work instead of as one burst in front of the next MFMA chain (whose fragment reads queue behind it)
template <int K, int MODE, class Sink = NoSink>
__device__ __forceinline__ void rqs_apply(const float (&v)[32], float x, int hh, float& y_out, float& ladj_out,
                                          float gy, float gl, float (&g)[32], float& gx_out, int kbins = 0,
                                          const Sink& sink = Sink()) {
    constexpr int KM = K > 0 ? K : RQS_KMAX;                   // loop / table bound; derivative slots start at KM
    constexpr int KD0M = KM / 2;                               // table bound of the derivatives per half
    const int kb = K > 0 ? K : kbins;                          // bins
    const int KD0 = kb / 2;               // derivatives owned by half 0 (interior knots 1..KD0)
    const int KD1 = kb - 1 - KD0;         // derivatives owned by half 1
    constexpr float A2 = 2.0f * LOG_SLOPE_INV;
    constexpr float A1 = LOG_SLOPE_INV;
    constexpr float FIX = 2147483648.0f;  // 2^31: prefix sums <= 1 + 20 * 2^-32 fit 32 bits with headroom
    static_assert(KM + KD0M <= 32, "spline does not fit the 32 slots of a lane half");
#define MF_BIN_ON(m) (K > 0 || (m) < kb)

    // soft clip + softmax over this half's K logits.  The clipped logits lie in (-3.46, 3.46), so exp() cannot
    // overflow and the usual max subtraction (a no-op mathematically) is not needed.  log2(e) is folded into the
    // reciprocal: ia = log2(e) / (1 + |v| A2), p = exp2(v * ia).  The adjoint needs d soft_clip / dv = (ia / log2 e)^2:
    // pq = p * ia^2 is formed here, the constant joins the three per-feature factors below.
    float p[KM], pq[MODE == 1 ? KM : 1];
    float sum = 0.0f;
#pragma unroll
    for (int m = 0; m < KM; ++m) {
        const float ia = fast_rcp(fmaf(fabsf(v[m]), A2 / LOG2E, 1.0f / LOG2E));
        p[m] = MF_BIN_ON(m) ? __builtin_amdgcn_exp2f(v[m] * ia) : 0.0f;
        if (MODE == 1) pq[m] = p[m] * (ia * ia);
        sum += p[m];
    }
    const float inv = fast_rcp(sum);
    // bin search on the cumulative probabilities:  knot_j < x  <=>  c_j < (x / bound + 1) / 2
    unsigned cj[KM + 1];
    cj[0] = 0u;
    const float xc = fmaf(x, 0.5f / RQS_BOUND, 0.5f);
    const unsigned xq = cvt_u32_sat(xc * FIX);          // exact for 0 <= xc < 2 (a power-of-two scaling), saturating outside
    const float fscale = inv * FIX;
    unsigned c = 0u;
    int cnt = (-RQS_BOUND < x) ? 1 : 0;
#pragma unroll
    for (int j = 0; j < KM; ++j) {
        c += cvt_u32_sat(fmaf(p[j], fscale, 0.5f));
        cj[j + 1] = c;
        cnt += (MF_BIN_ON(j) && c < xq) ? 1 : 0;
    }
    {   // both lanes use the count of the half that owns the searched knots: widths (half 0), heights for the inverse
        int lo, hi;
        half_pair(cnt, hh, lo, hi);
        cnt = (MODE == 2) ? hi : lo;
    }
    const int k = cnt - 1;
    const bool inrange = (cnt >= 1) && (cnt <= kb);
    unsigned qk, qk1;                     // knots k and k + 1 (out of range: the first / last bin, never used)
    select_pair<KM>(cj, min(max(k, 0), kb - 1), qk, qk1);
    const float ck = (float)qk * (1.0f / FIX), ck1 = (float)qk1 * (1.0f / FIX);
    const float kn0 = RQS_BOUND * (2.0f * ck - 1.0f);
    const float kn1 = RQS_BOUND * (2.0f * ck1 - 1.0f);
    float x0, x1, y0, y1;                 // widths live in half 0, heights in half 1
    half_pair(kn0, hh, x0, y0);
    half_pair(kn1, hh, x1, y1);
    // raw derivative logits at knots k and k+1 (0 at the boundary knots: exp(0) = 1)
    const int base = hh ? KD0 : 0;
    const int nown = hh ? KD1 : KD0;
    // table of this half's logits by knot: G[e] = logit of interior knot (base + e - 1), zero for knots the half does
    // not own (and for the boundary knots); knots k and k + 1 are the adjacent pair at e = k - base + 1
    float r0, r1;
    {
        float G[KD0M + 4];
        G[0] = G[1] = G[KD0M + 2] = G[KD0M + 3] = 0.0f;
#pragma unroll
        for (int j = 0; j < KD0M; ++j) G[2 + j] = (j < nown) ? v[KM + j] : 0.0f;
        select_pair<KD0M + 3>(G, min(max(k - base + 1, 0), KD0M + 2), r0, r1);
    }
    {
        float a, b;
        half_pair(r0, hh, a, b);
        r0 = a + b;
        half_pair(r1, hh, a, b);
        r1 = a + b;
    }
    const float d0 = fast_exp(soft_clip(r0, A1));
    const float d1 = fast_exp(soft_clip(r1, A1));

    const float w = x1 - x0;
    const float iw = fast_rcp(w);
    const float hgt = y1 - y0;
    const float s = hgt * iw;
    if (MODE == 2) {
        const float yb = x - y0;
        const float bet = d0 + d1 - 2.0f * s;
        const float qa = hgt * (s - d0) + yb * bet;
        const float qb = hgt * d0 - yb * bet;
        const float qc = -s * yb;
        const float zi = 2.0f * qc / (-qb - sqrtf(qb * qb - 4.0f * qa * qc));
        y_out = inrange ? fmaf(zi, w, x0) : x;
        ladj_out = 0.0f;
        return;
    }
    const float z = (x - x0) * iw;
    const float omz = 1.0f - z;
    const float z1 = z * omz;
    const float beta = d0 + d1 - 2.0f * s;
    const float num = s * z * z + d0 * z1;
    const float den = fmaf(beta, z1, s);
    const float iden = fast_rcp(den);
    const float R = num * iden;
    const float Q = 2.0f * s * z1 + d0 * omz * omz + d1 * z * z;
    const float jac = s * s * Q * iden * iden;
    y_out = inrange ? fmaf(hgt, R, y0) : x;
    // jac is a ratio of O(1) positive quantities (slopes in (1e-3, 1e3) squared at most): never denormal, so the bare
    // v_log_f32 (log2, 1 ulp) times ln 2 replaces logf's denormal / infinity handling (12 instructions -> 2)
    ladj_out = inrange ? __builtin_amdgcn_logf(jac) * 0.6931471805599453f : 0.0f;

    if (MODE == 1) {
        const float tz = 1.0f - 2.0f * z;
        const float num_z = 2.0f * s * z + d0 * tz;
        const float den_z = beta * tz;
        const float R_z = (num_z - R * den_z) * iden;
        const float den_s = 1.0f - 2.0f * z1;
        const float R_s = (z * z - R * den_s) * iden;
        const float R_d0 = z1 * (1.0f - R) * iden;
        const float R_d1 = -R * z1 * iden;
        const float iQ = fast_rcp(Q);
        const float Q_z = 2.0f * s * tz - 2.0f * d0 * omz + 2.0f * d1 * z;
        const float l_z = Q_z * iQ - 2.0f * den_z * iden;
        const float l_s = 2.0f * fast_rcp(s) + 2.0f * z1 * iQ - 2.0f * den_s * iden;
        const float l_d0 = omz * omz * iQ - 2.0f * z1 * iden;
        const float l_d1 = z * z * iQ - 2.0f * z1 * iden;
        const float gyh = gy * hgt;
        const float Gz = gyh * R_z + gl * l_z;
        const float Gs = gyh * R_s + gl * l_s;
        const float Gd0 = gyh * R_d0 + gl * l_d0;
        const float Gd1 = gyh * R_d1 + gl * l_d1;
        const float Gh = gy * R;
        gx_out = inrange ? Gz * iw : gy;
        const float gx0 = (Gz * (z - 1.0f) + Gs * s) * iw;
        const float gx1 = -(Gz * z + Gs * s) * iw;
        const float gy0 = gy - Gh - Gs * iw;
        const float gy1 = Gh + Gs * iw;
        // knots -> cumulative probabilities -> softmax -> soft clip (this half's own K logits)
        const float gcA = inrange ? 2.0f * RQS_BOUND * (hh ? gy0 : gx0) : 0.0f;
        const float gcB = inrange ? 2.0f * RQS_BOUND * (hh ? gy1 : gx1) : 0.0f;
        const float dot = gcA * ck + gcB * ck1;
        // d/d(cumulative probability) reaches logit m through every knot >= m + 1: both knots (m < k), the upper one
        // (m == k) or none; three candidates, two selects per logit.  pq holds the UNNORMALISED probability times
        // (log2(e) * d soft_clip/dv): the normalisation and the constant ride on the three candidates.
        // Written without compares: u_m = clamp(k - m, 0, 1) is 1 for m < k and 0 otherwise (one v_sub with the clamp
        // modifier), "m == k" is u_{m-1} - u_m, so the factor is t_gt + u_m (t_lt - t_eq) + u_{m-1} (t_eq - t_gt): a
        // subtract, two FMAs and the product per logit, no mask registers (a VALU compare followed by the select that
        // reads its mask costs two wait states on gfx950, which the compiler fills with s_nop).
        const float nrm = inv * (1.0f / (LOG2E * LOG2E));
        const float t_gt = (0.0f - dot) * nrm, d_eq = gcB * nrm, d_lt = gcA * nrm;   // t_eq - t_gt, t_lt - t_eq
        const float kf = (float)k;
        float u_prev = __builtin_amdgcn_fmed3f(kf + 1.0f, 0.0f, 1.0f);                // u_{-1}: k >= 0
#pragma unroll
        for (int m = 0; m < KM; ++m) {
            const float u = __builtin_amdgcn_fmed3f(kf - (float)m, 0.0f, 1.0f);
            g[m] = pq[m] * fmaf(u, d_lt, fmaf(u_prev, d_eq, t_gt));
            sink(m, g[m]);
            u_prev = u;
        }
        const float gr0 = inrange ? Gd0 * d0 * soft_clip_grad(r0, A1) : 0.0f;
        const float gr1 = inrange ? Gd1 * d1 * soft_clip_grad(r1, A1) : 0.0f;
#pragma unroll
        for (int j = 0; j < 32 - KM; ++j) {
            const bool own = j < nown;
            float t = 0.0f;
            t = (own && (k - 1 == base + j)) ? gr0 : t;
            t = (own && (k == base + j)) ? t + gr1 : t;
            g[KM + j] = t;
            sink(KM + j, t);
        }
    }
}
#undef MF_BIN_ON

// A[in rows of tile] fragment of one output block (64 padded rows) of the last linear layer
__device__ __forceinline__ void block_linear(const float* W, const float* b, const f32x16_t (&in)[2], float (&v)[32],
                                             int col, int hh, int kend) {
    f32x16_t phi[2];
    linear64(W, b, in, phi, col, hh, kend, kend);
#pragma unroll
    for (int m = 0; m < 32; ++m) v[m] = phi[m >> 4][m & 15];
}

__device__ __forceinline__ float base_log_prob(const float* xp, int d) {
    float q = 0.0f;
    for (int j = 0; j < d; ++j) q = fmaf(xp[j], xp[j], q);
    return -0.5f * q - 0.9189385332046727f * (float)d;
}

// =========================================================================================== forward, RQS
template <int K, int L, int BLOCK>
__global__ __launch_bounds__(BLOCK) void rqs_layer_fwd_kernel(const float* __restrict__ image, int d,
                                                                   const float* __restrict__ x, int64_t n,
                                                                   float* __restrict__ y,
                                                                   const float* __restrict__ logp_in,
                                                                   float* __restrict__ logp_out, int init_logp,
                                                                   Sparsity sp, int bins_rt) {
    MF_DYN_SMEM(float, lds);
    const ImageLayout g = image_layout(d, L, d);
    stage_image<BLOCK>(lds, image, g.total);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, col = lane & 31, hh = lane >> 5;
    const int64_t ntiles = (n + 31) / 32;
    for (int64_t tile = (int64_t)blockIdx.x * (BLOCK / 64) + wid; tile < ntiles; tile += (int64_t)gridDim.x * (BLOCK / 64)) {
        MF_NO_HOIST();
        const int64_t p = tile * 32 + col;
        const bool valid = p < n;
        const float* xp = x + (valid ? p : n - 1) * d;
        float xb[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) xb[s] = (2 * s + hh < d) ? xp[2 * s + hh] : 0.0f;
        // The mask bounds are loop invariants: left alone, the compiler evaluates every "group < bound" test once, ahead of
        // the tile loop, keeps the 40 results as lane masks in SGPR pairs and re-derives each branch condition from them
        // with a v_cndmask / v_cmp_ne pair per group (54 VALU instructions per tile).  Opaque copies of the three scalars
        // keep the tests where they are used: one s_cmp each.
        int d_s = d, kh0 = sp.kend_h[0], kh1 = sp.kend_h[1];
#ifndef MF_EMU
        asm volatile("" : "+s"(d_s), "+s"(kh0), "+s"(kh1));
#endif
        f32x16_t h[2];
        input_layer(lds + g.offW0, lds + g.offB0, g.S0, d_s, xb, h, col, hh);
#pragma unroll
        for (int l = 1; l < L; ++l) {
            f32x16_t t[2];
            const float* W = lds + g.offWh + (l - 1) * (HID * WS + HID);
            linear64(W, W + HID * WS, h, t, col, hh, kh0, kh1);
            relu2(t);
            h[0] = t[0];
            h[1] = t[1];
        }
        float ladj = 0.0f;
#pragma unroll 1
        for (int i = 0; i < d; ++i) {
            float v[32], gdummy[32];
            block_linear(lds + g.offW3 + i * HID * WS, lds + g.offB3 + i * HID, h, v, col, hh, sp.kend3[i]);
            float yi, li, gxd;
            rqs_apply<K, 0>(v, xp[i], hh, yi, li, 0.0f, 0.0f, gdummy, gxd, bins_rt);
            ladj += li;
            if (valid && hh == 0) y[p * d + i] = yi;
        }
        if (valid && hh == 0) {
            const float lp0 = init_logp ? base_log_prob(xp, d) : logp_in[p];
            logp_out[p] = lp0 - ladj;
        }
    }
}

// scratch buffers are stored as transposed 32-particle tiles:  X[tile][c][particle]  (64 x 32 floats = 8 KiB per
// tile), column c = 32*rt + 16*hh + r <-> accumulator register r of row tile rt of lane half hh = MFMA row
// 32*rt + rowmap(r, hh).  The parameter-gradient contraction (particles = MFMA k) then reads, per lane, 16 consecutive
// particles of one column as four 16-byte loads, and memory tile (c >> 5) == MFMA row tile so masked-out tiles can be
// skipped.  A store instruction writes two 128-byte segments (the two lane halves).
__device__ __forceinline__ void store_tile(float* __restrict__ dst, int64_t tile, int col, int hh, const float (&v)[32]) {
    float* base = dst + tile * 2048 + (16 * hh) * 32 + col;
#pragma unroll
    for (int m = 0; m < 32; ++m) base[(32 * (m >> 4) + (m & 15)) * 32] = v[m];
}
__device__ __forceinline__ void store_tile(float* __restrict__ dst, int64_t tile, int col, int hh, const f32x16_t (&a)[2]) {
    float* base = dst + tile * 2048 + (16 * hh) * 32 + col;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) base[(32 * rt + r) * 32] = a[rt][r];
}

// =========================================================================================== backward, RQS
// scratch: ACT[L][npad][64] | GPRE[L][npad][64] | GPHI[d][npad][64],  npad = ntiles*32
template <int K, int L, int BLOCK>
__global__ __launch_bounds__(BLOCK) void rqs_layer_bwd_kernel(const float* __restrict__ image, int d,
                                                                   const float* __restrict__ x, int64_t n,
                                                                   const float* __restrict__ gy,
                                                                   const float* __restrict__ glogp,
                                                                   float* __restrict__ gx, float* __restrict__ scratch,
                                                                   Sparsity sp, int bins_rt) {
    MF_DYN_SMEM(float, lds);
    const ImageLayout g = image_layout(d, L, d);
    stage_image<BLOCK>(lds, image, g.total);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, col = lane & 31, hh = lane >> 5;
    const int64_t ntiles = (n + 31) / 32;
    const int64_t npad = ntiles * 32;
    float* ACT = scratch;
    float* GPRE = ACT + (int64_t)L * npad * 64;
    float* GPHI = GPRE + (int64_t)L * npad * 64;
    for (int64_t tile = (int64_t)blockIdx.x * (BLOCK / 64) + wid; tile < ntiles; tile += (int64_t)gridDim.x * (BLOCK / 64)) {
        MF_NO_HOIST();
        const int64_t p = tile * 32 + col;
        const bool valid = p < n;
        const int64_t pc = valid ? p : n - 1;
        const float* xp = x + pc * d;
        float xb[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) xb[s] = (2 * s + hh < d) ? xp[2 * s + hh] : 0.0f;
        // ---- recompute the trunk, keep every activation
        f32x16_t h[L][2];
        input_layer(lds + g.offW0, lds + g.offB0, g.S0, d, xb, h[0], col, hh);
        store_tile(ACT, tile, col, hh, h[0]);
#pragma unroll
        for (int l = 1; l < L; ++l) {
            const float* W = lds + g.offWh + (l - 1) * (HID * WS + HID);
            linear64(W, W + HID * WS, h[l - 1], h[l], col, hh, sp.kend_h[0], sp.kend_h[1]);
            relu2(h[l]);
            store_tile(ACT + (int64_t)l * npad * 64, tile, col, hh, h[l]);
        }
        // ---- output blocks: spline forward + adjoint, accumulate dL/dh_last
        f32x16_t gh[2];
        f32x16_t gacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            gh[0][r] = 0.0f;
            gh[1][r] = 0.0f;
            gacc[r] = 0.0f;
        }
        const float gl = valid ? -glogp[pc] : 0.0f;
#pragma unroll 1
        for (int i = 0; i < d; ++i) {
            float v[32], gv[32];
            const float* W3 = lds + g.offW3 + i * HID * WS;
            block_linear(W3, lds + g.offB3 + i * HID, h[L - 1], v, col, hh, sp.kend3[i]);
            const float gyi = valid ? gy[pc * d + i] : 0.0f;
            float yi, li, gxd;
            rqs_apply<K, 1>(v, xp[i], hh, yi, li, gyi, gl, gv, gxd, bins_rt);
            // direct path dL/dx_i goes into row i of the dL/dx accumulator tile (row = 4*hh + reg for rows < 8)
#pragma unroll
            for (int j = 0; j < 4; ++j) gacc[j] += ((hh == ((i >> 2) & 1)) && ((i & 3) == j)) ? gxd : 0.0f;
            store_tile(GPHI + (int64_t)i * npad * 64, tile, col, hh, gv);
            // gh += W3_i^T gphi   (contraction over the 64 padded output rows = slots of both halves); hidden tile 1
            // only receives something if block i sees more than 32 hidden units, nothing at all for a pure-bias block
            linear64s_t(W3, WS, gv, gh, col, hh, sp.kend3[i] > 0, sp.rt1[i] != 0);
        }
        // ---- trunk backward
#pragma unroll
        for (int l = L - 1; l >= 1; --l) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) gh[rt][r] = (h[l][rt][r] > 0.0f) ? gh[rt][r] : 0.0f;
            store_tile(GPRE + (int64_t)l * npad * 64, tile, col, hh, gh);
            f32x16_t t[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                t[0][r] = 0.0f;
                t[1][r] = 0.0f;
            }
            linear64_t(lds + g.offWh + (l - 1) * (HID * WS + HID), gh, t, col, hh, sp.kbeg_ht[0], sp.kbeg_ht[1]);
            gh[0] = t[0];
            gh[1] = t[1];
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) gh[rt][r] = (h[0][rt][r] > 0.0f) ? gh[rt][r] : 0.0f;
        store_tile(GPRE, tile, col, hh, gh);
        if (gx != nullptr) {
            // gacc += W0^T gpre0 : rows = input features (lanes col < d carry weights, others 0)
            const float* wcol = lds + g.offW0 + 4 * hh * g.S0 + col;
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                const int kk = 32 * (s >> 4) + rowmap(s & 15, 0);
                const float a = (col < d) ? wcol[kk * g.S0] : 0.0f;
                gacc = mfma(a, gh[s >> 4][s & 15], gacc);
            }
            if (valid) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (4 * hh + j < d) gx[p * d + 4 * hh + j] = gacc[j];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Hand-scheduled MFMA groups for the one-wave-per-SIMD fused kernel.  The compiler's code for `linear64s` is
// ds_read2 -> s_waitcnt lgkmcnt(0) -> 2 MFMAs -> ds_read2 -> ... with one reused register pair: harmless when a second
// wave fills the LDS round trips, ~45 % MFMA efficiency when the wave is alone on its SIMD.  These blocks request the
// weight fragments of group g + 1, issue the four (dependent) MFMAs of group g, and only then wait for LDS — by which
// time the data has long arrived.  `KS` = distance in floats between consecutive k of one row/column in LDS
// (1: row-major [out][in];  WS: column access).  All LDS offsets are immediates.
#if !defined(MF_EMU)
#define MF_ASM_CHAIN 1
__device__ __forceinline__ unsigned lds_addr(const float* p) { return (unsigned)(size_t)p; }

// group g: request the fragments of the group starting at k-step S4N into n[] (unconditionally: a skipped group's
// fragments are never used), multiply-accumulate the four k-steps held in a[].
// BA: the B operands (activations) are taken from AGPRs.  An MFMA reads A / B / C from either register file; an operand
// constraint "v" on values that live across the VALU-heavy spline makes the compiler shuttle them between the files
// (v_accvgpr_write to park, v_accvgpr_read to bring back: 64 VALU per 32-row tile and use), "a" lets them stay parked.
#define MF_DEF_MFMA4(SUF, BC)                                                                                         \
    template <int KS, int S4N>                                                                                        \
    __device__ __forceinline__ void mfma4_pf_##SUF(f32x16_t& acc, const float (&a)[4], float (&n)[4], unsigned addr,  \
                                                   float b0, float b1, float b2, float b3) {                          \
        asm volatile(                                                                                                 \
            "ds_read_b32 %1, %9 offset:%14\n\t"                                                                       \
            "ds_read_b32 %2, %9 offset:%15\n\t"                                                                       \
            "ds_read_b32 %3, %9 offset:%16\n\t"                                                                       \
            "ds_read_b32 %4, %9 offset:%17\n\t"                                                                       \
            "v_mfma_f32_32x32x2_f32 %0, %5, %10, %0\n\t"                                                              \
            "v_mfma_f32_32x32x2_f32 %0, %6, %11, %0\n\t"                                                              \
            "v_mfma_f32_32x32x2_f32 %0, %7, %12, %0\n\t"                                                              \
            "v_mfma_f32_32x32x2_f32 %0, %8, %13, %0\n\t"                                                              \
            "s_waitcnt lgkmcnt(0)"                                                                                    \
            : "+v"(acc), "=&v"(n[0]), "=&v"(n[1]), "=&v"(n[2]), "=&v"(n[3])                                           \
            : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(addr), BC(b0), BC(b1), BC(b2), BC(b3),                  \
              "n"(kcol(S4N) * KS * 4), "n"(kcol(S4N + 1) * KS * 4), "n"(kcol(S4N + 2) * KS * 4),                      \
              "n"(kcol(S4N + 3) * KS * 4));                                                                           \
    }                                                                                                                 \
    /* last group of a chain: no prefetch */                                                                          \
    __device__ __forceinline__ void mfma4_last_##SUF(f32x16_t& acc, const float (&a)[4], float b0, float b1, float b2, \
                                                     float b3) {                                                      \
        asm volatile(                                                                                                 \
            "s_nop 1\n\t"                                                                                             \
            "v_mfma_f32_32x32x2_f32 %0, %1, %5, %0\n\t"                                                               \
            "v_mfma_f32_32x32x2_f32 %0, %2, %6, %0\n\t"                                                               \
            "v_mfma_f32_32x32x2_f32 %0, %3, %7, %0\n\t"                                                               \
            "v_mfma_f32_32x32x2_f32 %0, %4, %8, %0"                                                                   \
            : "+v"(acc)                                                                                               \
            : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), BC(b0), BC(b1), BC(b2), BC(b3));                            \
    }
#define MF_BC_V(x) "v"(x)
#define MF_BC_A(x) "a"(x)
MF_DEF_MFMA4(v, MF_BC_V)
MF_DEF_MFMA4(a, MF_BC_A)
#undef MF_DEF_MFMA4
template <int KS, int S4N, bool BA = false>
__device__ __forceinline__ void mfma4_pf(f32x16_t& acc, const float (&a)[4], float (&n)[4], unsigned addr, float b0, float b1,
                                         float b2, float b3) {
    if constexpr (BA) mfma4_pf_a<KS, S4N>(acc, a, n, addr, b0, b1, b2, b3);
    else mfma4_pf_v<KS, S4N>(acc, a, n, addr, b0, b1, b2, b3);
}
template <bool BA = false>
__device__ __forceinline__ void mfma4_last(f32x16_t& acc, const float (&a)[4], float b0, float b1, float b2, float b3) {
    if constexpr (BA) mfma4_last_a(acc, a, b0, b1, b2, b3);
    else mfma4_last_v(acc, a, b0, b1, b2, b3);
}
// the compiler cannot see the MFMAs inside the blocks: pad the MFMA -> VALU read distance (18 wait states) by hand
__device__ __forceinline__ void mfma_drain(f32x16_t& acc) { asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc)); }

#endif

// B operand accessors of a chain: k-step S of an accumulator pair, or of a 32-slot vector
struct BTile {
    static constexpr bool agpr = false;
    const f32x16_t (&t)[2];
    template <int S>
    __device__ __forceinline__ float get() const { return t[S >> 4][S & 15]; }
};
struct BTileA {                          // the same, operands constrained to AGPRs (long-lived activations)
    static constexpr bool agpr = true;
    const f32x16_t (&t)[2];
    template <int S>
    __device__ __forceinline__ float get() const { return t[S >> 4][S & 15]; }
};
struct BVec {
    static constexpr bool agpr = false;
    const float (&v)[32];
    template <int S>
    __device__ __forceinline__ float get() const { return v[S]; }
};

// acc += sum over the k-step groups [g0, g1) (four k-steps each) of A * B;  A fragment of k-step s = wl[kcol(s) * KS]
// (wl: this lane's LDS row/column, KS = 1 for contiguous k, WS for a column walk), B(s) = b.get<s>().
// gfx950: hand-scheduled blocks, fragment registers ping-pong between a0[] and a1[];  emulator: the plain loop.
template <int KS, class BOp>
__device__ __forceinline__ void chain64(f32x16_t& acc, const float* wl, int g0, int g1, const BOp& b) {
#ifdef MF_ASM_CHAIN
    const unsigned addr = lds_addr(wl);
    float a0[4], a1[4];
#define MF_GRP(G, CUR, NXT)                                                                                           \
    if (G == g0 && G < g1) {                                                                                          \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) CUR[j] = wl[kcol(4 * G + j) * KS];                              \
    }                                                                                                                 \
    if (G >= g0 && G < g1)                                                                                            \
        mfma4_pf<KS, (4 * G + 4) & 31, BOp::agpr>(acc, CUR, NXT, addr, b.template get<4 * G>(), b.template get<4 * G + 1>(), \
                                       b.template get<4 * G + 2>(), b.template get<4 * G + 3>());
    MF_GRP(0, a0, a1) MF_GRP(1, a1, a0) MF_GRP(2, a0, a1) MF_GRP(3, a1, a0) MF_GRP(4, a0, a1) MF_GRP(5, a1, a0) MF_GRP(6, a0, a1)
#undef MF_GRP
    if (7 == g0 && 7 < g1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) a1[j] = wl[kcol(28 + j) * KS];
    }
    if (7 >= g0 && 7 < g1)
        mfma4_last<BOp::agpr>(acc, a1, b.template get<28>(), b.template get<29>(), b.template get<30>(), b.template get<31>());
    mfma_drain(acc);
#else
#define MF_STEP(S) if ((S) >= 4 * g0 && (S) < 4 * g1) acc = mfma(wl[kcol(S) * KS], b.template get<S>(), acc);
#define MF_STEP4(S) MF_STEP(S) MF_STEP(S + 1) MF_STEP(S + 2) MF_STEP(S + 3)
    MF_STEP4(0) MF_STEP4(4) MF_STEP4(8) MF_STEP4(12) MF_STEP4(16) MF_STEP4(20) MF_STEP4(24) MF_STEP4(28)
#undef MF_STEP4
#undef MF_STEP
#endif
}

#ifdef MF_ASM_CHAIN
__device__ __forceinline__ void mfma4_half(f32x16_t& acc, const float (&a)[4], float b0, float b1) {
    asm volatile(
        "s_nop 1\n\t"
        "v_mfma_f32_32x32x2_f32 %0, %1, %3, %0\n\t"
        "v_mfma_f32_32x32x2_f32 %0, %2, %4, %0"
        : "+v"(acc)
        : "v"(a[0]), "v"(a[1]), "v"(b0), "v"(b1));
}
#endif
// acc += A * B over 7 full groups and the first two k-steps of the eighth (30 of 32 k-steps): the last-layer transposed
// product of a spline with 20 bins, whose slots 30 and 31 are padding in both lane halves
template <int KS, class BOp>
__device__ __forceinline__ void chain64_30(f32x16_t& acc, const float* wl, const BOp& b) {
#ifdef MF_ASM_CHAIN
    const unsigned addr = lds_addr(wl);
    float a0[4], a1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) a0[j] = wl[kcol(j) * KS];
#define MF_G(G, CUR, NXT) mfma4_pf<KS, 4 * G + 4, BOp::agpr>(acc, CUR, NXT, addr, b.template get<4 * G>(), b.template get<4 * G + 1>(), \
                                                             b.template get<4 * G + 2>(), b.template get<4 * G + 3>());
    MF_G(0, a0, a1) MF_G(1, a1, a0) MF_G(2, a0, a1) MF_G(3, a1, a0) MF_G(4, a0, a1) MF_G(5, a1, a0) MF_G(6, a0, a1)
#undef MF_G
    mfma4_half(acc, a1, b.template get<28>(), b.template get<29>());
    mfma_drain(acc);
#else
    chain64<KS>(acc, wl, 0, 8, b);
#endif
}

// Two chains over the SAME B operand back to back (the two 32-row tiles of one product): acc0 += A0 * B over the k-step
// groups [S0, E0), acc1 += A1 * B over [S1, E1) (mask-bounded ranges: the skipped groups only multiply zeros).  The last
// group of the first chain requests the first fragments of the second (no exposed LDS round trip between them) and only
// one MFMA -> VALU drain is paid.  The ranges are template parameters: the fragment buffers ping-pong by position in the
// sequence, which must stay a compile-time register choice.
#ifdef MF_ASM_CHAIN
template <int KS, int G, int NEXT_S, class BOp>     // NEXT_S: first k-step of the group to prefetch, -1: none
__device__ __forceinline__ void chain_grp(f32x16_t& acc, const float (&cur)[4], float (&nxt)[4], unsigned addr_next, const BOp& b) {
    if constexpr (NEXT_S >= 0)
        mfma4_pf<KS, NEXT_S, BOp::agpr>(acc, cur, nxt, addr_next, b.template get<4 * G>(), b.template get<4 * G + 1>(),
                             b.template get<4 * G + 2>(), b.template get<4 * G + 3>());
    else
        mfma4_last<BOp::agpr>(acc, cur, b.template get<4 * G>(), b.template get<4 * G + 1>(), b.template get<4 * G + 2>(),
                   b.template get<4 * G + 3>());
}
#endif
template <int KS, int S0, int E0, int S1, int E1, class BOp>
__device__ __forceinline__ void chain64x2r(f32x16_t& acc0, f32x16_t& acc1, const float* wl0, const float* wl1, const BOp& b) {
    static_assert(0 <= S0 && S0 < E0 && E0 <= 8 && 0 <= S1 && S1 < E1 && E1 <= 8, "non-empty group ranges within 0..8");
#ifdef MF_ASM_CHAIN
    const unsigned addr0 = lds_addr(wl0), addr1 = lds_addr(wl1);
    float a0[4], a1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) a0[j] = wl0[kcol(4 * S0 + j) * KS];
#define MF_A(G)                                                                                                       \
    if constexpr (G >= S0 && G < E0) {                                                                               \
        constexpr bool last_ = (G + 1 == E0);                                                                        \
        constexpr int nxt_ = last_ ? 4 * S1 : 4 * G + 4;                                                             \
        if constexpr (((G - S0) & 1) == 0) chain_grp<KS, G, nxt_>(acc0, a0, a1, last_ ? addr1 : addr0, b);           \
        else chain_grp<KS, G, nxt_>(acc0, a1, a0, last_ ? addr1 : addr0, b);                                         \
    }
    MF_A(0) MF_A(1) MF_A(2) MF_A(3) MF_A(4) MF_A(5) MF_A(6) MF_A(7)
#undef MF_A
#define MF_B(G)                                                                                                       \
    if constexpr (G >= S1 && G < E1) {                                                                               \
        constexpr int nxt_ = (G + 1 == E1) ? -1 : 4 * G + 4;                                                         \
        if constexpr ((((E0 - S0) + (G - S1)) & 1) == 0) chain_grp<KS, G, nxt_>(acc1, a0, a1, addr1, b);             \
        else chain_grp<KS, G, nxt_>(acc1, a1, a0, addr1, b);                                                         \
    }
    MF_B(0) MF_B(1) MF_B(2) MF_B(3) MF_B(4) MF_B(5) MF_B(6) MF_B(7)
#undef MF_B
    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc0), "+v"(acc1));      // MFMA -> VALU read distance, once for both
#else
    chain64<KS>(acc0, wl0, S0, E0, b);
    chain64<KS>(acc1, wl1, S1, E1, b);
#endif
}
template <int KS, int NG, class BOp>
__device__ __forceinline__ void chain64x2(f32x16_t& acc0, f32x16_t& acc1, const float* wl0, const float* wl1, const BOp& b) {
    chain64x2r<KS, 0, NG, 0, NG>(acc0, acc1, wl0, wl1, b);
}
// Same pair, equal ranges [0, NG), with the MFMAs of the two chains INTERLEAVED (acc0, acc1, acc0, ...): consecutive
// MFMAs are independent, which removes the ~3.5 cycles of issue stall a dependent fp32 MFMA pays
// (tools/ubench_chain.hip: 64 MFMAs in 4445 instead of 4671 cycles).  Costs eight more fragment registers.
#ifdef MF_ASM_CHAIN
#define MF_DEF_MFMA8(SUF, BC)                                                                                         \
    template <int KS, int S4N>                                                                                        \
    __device__ __forceinline__ void mfma8_pf_##SUF(f32x16_t& acc0, f32x16_t& acc1, const float (&a)[8], float (&n)[8], \
                                                   unsigned addr0, unsigned addr1, float b0, float b1, float b2,      \
                                                   float b3) {                                                        \
        asm volatile(                                                                                                 \
            "ds_read_b32 %2, %18 offset:%24\n\t"                                                                      \
            "ds_read_b32 %3, %18 offset:%25\n\t"                                                                      \
            "ds_read_b32 %4, %18 offset:%26\n\t"                                                                      \
            "ds_read_b32 %5, %18 offset:%27\n\t"                                                                      \
            "ds_read_b32 %6, %19 offset:%24\n\t"                                                                      \
            "ds_read_b32 %7, %19 offset:%25\n\t"                                                                      \
            "ds_read_b32 %8, %19 offset:%26\n\t"                                                                      \
            "ds_read_b32 %9, %19 offset:%27\n\t"                                                                      \
            "v_mfma_f32_32x32x2_f32 %0, %10, %20, %0\n\t"                                                             \
            "v_mfma_f32_32x32x2_f32 %1, %14, %20, %1\n\t"                                                             \
            "v_mfma_f32_32x32x2_f32 %0, %11, %21, %0\n\t"                                                             \
            "v_mfma_f32_32x32x2_f32 %1, %15, %21, %1\n\t"                                                             \
            "v_mfma_f32_32x32x2_f32 %0, %12, %22, %0\n\t"                                                             \
            "v_mfma_f32_32x32x2_f32 %1, %16, %22, %1\n\t"                                                             \
            "v_mfma_f32_32x32x2_f32 %0, %13, %23, %0\n\t"                                                             \
            "v_mfma_f32_32x32x2_f32 %1, %17, %23, %1\n\t"                                                             \
            "s_waitcnt lgkmcnt(0)"                                                                                    \
            : "+v"(acc0), "+v"(acc1), "=&v"(n[0]), "=&v"(n[1]), "=&v"(n[2]), "=&v"(n[3]), "=&v"(n[4]), "=&v"(n[5]),   \
              "=&v"(n[6]), "=&v"(n[7])                                                                                \
            : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(addr0),     \
              "v"(addr1), BC(b0), BC(b1), BC(b2), BC(b3), "n"(kcol(S4N) * KS * 4), "n"(kcol(S4N + 1) * KS * 4),       \
              "n"(kcol(S4N + 2) * KS * 4), "n"(kcol(S4N + 3) * KS * 4));                                              \
    }                                                                                                                 \
    __device__ __forceinline__ void mfma8_last_##SUF(f32x16_t& acc0, f32x16_t& acc1, const float (&a)[8], float b0,   \
                                                     float b1, float b2, float b3) {                                  \
        asm volatile(                                                                                                 \
            "s_nop 1\n\t"                                                                                             \
            "v_mfma_f32_32x32x2_f32 %0, %2, %10, %0\n\t"                                                              \
            "v_mfma_f32_32x32x2_f32 %1, %6, %10, %1\n\t"                                                              \
            "v_mfma_f32_32x32x2_f32 %0, %3, %11, %0\n\t"                                                              \
            "v_mfma_f32_32x32x2_f32 %1, %7, %11, %1\n\t"                                                              \
            "v_mfma_f32_32x32x2_f32 %0, %4, %12, %0\n\t"                                                              \
            "v_mfma_f32_32x32x2_f32 %1, %8, %12, %1\n\t"                                                              \
            "v_mfma_f32_32x32x2_f32 %0, %5, %13, %0\n\t"                                                              \
            "v_mfma_f32_32x32x2_f32 %1, %9, %13, %1"                                                                  \
            : "+v"(acc0), "+v"(acc1)                                                                                  \
            : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), BC(b0), BC(b1), \
              BC(b2), BC(b3));                                                                                        \
    }
MF_DEF_MFMA8(v, MF_BC_V)
MF_DEF_MFMA8(a, MF_BC_A)
#undef MF_DEF_MFMA8
// last group of a chain with only TWO k-steps left (the spline uses 30 of a lane half's 32 slots: the k-steps of the two
// padding slots would multiply zeros)
__device__ __forceinline__ void mfma8_half(f32x16_t& acc0, f32x16_t& acc1, const float (&a)[8], float b0, float b1) {
    asm volatile(
        "s_nop 1\n\t"
        "v_mfma_f32_32x32x2_f32 %0, %2, %6, %0\n\t"
        "v_mfma_f32_32x32x2_f32 %1, %4, %6, %1\n\t"
        "v_mfma_f32_32x32x2_f32 %0, %3, %7, %0\n\t"
        "v_mfma_f32_32x32x2_f32 %1, %5, %7, %1"
        : "+v"(acc0), "+v"(acc1)
        : "v"(a[0]), "v"(a[1]), "v"(a[4]), "v"(a[5]), "v"(b0), "v"(b1));
}
template <int KS, int S4N, bool BA = false>
__device__ __forceinline__ void mfma8_pf(f32x16_t& acc0, f32x16_t& acc1, const float (&a)[8], float (&n)[8], unsigned addr0,
                                         unsigned addr1, float b0, float b1, float b2, float b3) {
    if constexpr (BA) mfma8_pf_a<KS, S4N>(acc0, acc1, a, n, addr0, addr1, b0, b1, b2, b3);
    else mfma8_pf_v<KS, S4N>(acc0, acc1, a, n, addr0, addr1, b0, b1, b2, b3);
}
template <bool BA = false>
__device__ __forceinline__ void mfma8_last(f32x16_t& acc0, f32x16_t& acc1, const float (&a)[8], float b0, float b1, float b2,
                                           float b3) {
    if constexpr (BA) mfma8_last_a(acc0, acc1, a, b0, b1, b2, b3);
    else mfma8_last_v(acc0, acc1, a, b0, b1, b2, b3);
}
#endif
// TAIL2: the last of the NG groups only has its first two k-steps (k-steps 4 NG - 2, 4 NG - 1 multiply padding)
template <int KS, int NG, bool TAIL2 = false, class BOp>
__device__ __forceinline__ void chain64x2i(f32x16_t& acc0, f32x16_t& acc1, const float* wl0, const float* wl1, const BOp& b) {
    static_assert(NG >= 1 && NG <= 8, "1..8 groups");
#ifdef MF_ASM_CHAIN
    const unsigned addr0 = lds_addr(wl0), addr1 = lds_addr(wl1);
    float a0[8], a1[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        a0[j] = wl0[kcol(j) * KS];
        a0[4 + j] = wl1[kcol(j) * KS];
    }
#define MF_I(G, CUR, NXT)                                                                                             \
    if constexpr (G < NG) {                                                                                          \
        if constexpr (G + 1 < NG)                                                                                    \
            mfma8_pf<KS, (4 * G + 4) & 31, BOp::agpr>(acc0, acc1, CUR, NXT, addr0, addr1, b.template get<4 * G>(),   \
                                           b.template get<4 * G + 1>(), b.template get<4 * G + 2>(),                 \
                                           b.template get<4 * G + 3>());                                             \
        else if constexpr (TAIL2)                                                                                    \
            mfma8_half(acc0, acc1, CUR, b.template get<4 * G>(), b.template get<4 * G + 1>());                       \
        else                                                                                                         \
            mfma8_last<BOp::agpr>(acc0, acc1, CUR, b.template get<4 * G>(), b.template get<4 * G + 1>(),               \
                                  b.template get<4 * G + 2>(),                                                       \
                       b.template get<4 * G + 3>());                                                                 \
    }
    MF_I(0, a0, a1) MF_I(1, a1, a0) MF_I(2, a0, a1) MF_I(3, a1, a0) MF_I(4, a0, a1) MF_I(5, a1, a0) MF_I(6, a0, a1) MF_I(7, a1, a0)
#undef MF_I
    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc0), "+v"(acc1));
#else
    chain64<KS>(acc0, wl0, 0, NG, b);
    chain64<KS>(acc1, wl1, 0, NG, b);
#endif
}

// run-time number of groups (wave-uniform): ONE branch into straight-line instances
template <int KS, class BOp>
__device__ __forceinline__ void chain64x2_upto(f32x16_t& acc0, f32x16_t& acc1, const float* wl0, const float* wl1, int ng,
                                               const BOp& b) {
    switch (ng) {
        case 0: break;
        case 1: chain64x2i<KS, 1>(acc0, acc1, wl0, wl1, b); break;
        case 2: chain64x2i<KS, 2>(acc0, acc1, wl0, wl1, b); break;
        case 3: chain64x2i<KS, 3>(acc0, acc1, wl0, wl1, b); break;
        case 4: chain64x2i<KS, 4>(acc0, acc1, wl0, wl1, b); break;
        case 5: chain64x2i<KS, 5>(acc0, acc1, wl0, wl1, b); break;
        case 6: chain64x2i<KS, 6>(acc0, acc1, wl0, wl1, b); break;
        case 7: chain64x2i<KS, 7>(acc0, acc1, wl0, wl1, b); break;
        default: chain64x2i<KS, 8>(acc0, acc1, wl0, wl1, b); break;
    }
}

// input layer as ONE group of four k-steps (features 2s + hh): columns >= d multiply xb = 0 (the image words read there
// are other, finite, weights), so no bounds are needed
__device__ __forceinline__ void input_layer4(const float* W0, const float* b0, int S0, const float (&xb)[4], f32x16_t (&h)[2],
                                             int col, int hh) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        f32x16_t acc = bias_tile(b0, rt, hh);
        const float* wrow = W0 + (32 * rt + col) * S0 + hh;
#ifdef MF_ASM_CHAIN
        float a[4];
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) a[s_] = wrow[2 * s_];
        mfma4_last(acc, a, xb[0], xb[1], xb[2], xb[3]);
        mfma_drain(acc);
#else
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) acc = mfma(wrow[2 * s_], xb[s_], acc);
#endif
        h[rt] = acc;
    }
    relu2(h);
}

// acc += sum over the k-step groups [0, g1) of A * B, g1 wave-uniform at run time, with ONE branch: a switch over g1
// whose cases are separate straight-line chains with compile-time bounds.  (The eight per-group branches of
// chain64(…, 0, g1, …) cost a lone wave ~40 % here; a single fall-through switch entered at group g1 - 1 made the
// compiler copy the accumulator and the fragments at every label.)
template <int KS, class BOp>
__device__ __forceinline__ void chain64_upto(f32x16_t& acc, const float* wl, int g1, const BOp& b) {
#ifdef MF_ASM_CHAIN
    switch (g1) {
        case 0: break;
        case 1: chain64<KS>(acc, wl, 0, 1, b); break;
        case 2: chain64<KS>(acc, wl, 0, 2, b); break;
        case 3: chain64<KS>(acc, wl, 0, 3, b); break;
        case 4: chain64<KS>(acc, wl, 0, 4, b); break;
        case 5: chain64<KS>(acc, wl, 0, 5, b); break;
        case 6: chain64<KS>(acc, wl, 0, 6, b); break;
        case 7: chain64<KS>(acc, wl, 0, 7, b); break;
        default: chain64<KS>(acc, wl, 0, 8, b); break;
    }
#else
    chain64<KS>(acc, wl, 0, g1, b);
#endif
}

// Diagnostic build only (-DMF_WS_DIAG): cycle stamps of pair 0 of every workgroup, read back with mf_debug_ws_read.
#if defined(MF_WS_DIAG) && !defined(MF_EMU)
__device__ unsigned long long g_ws_diag[NUM_CU * 4 * 16];      // [workgroup][wave][slot]
#define WS_T() __builtin_amdgcn_s_memtime()
#define WS_ACC(var, t0) var += WS_T() - (t0)
#else
#define WS_T() 0ull
#define WS_ACC(var, t0) (void)(t0)
#endif

// =========================================================================================== backward, RQS, fused
// Backward of one layer INCLUDING the parameter gradients: no activation / gradient tiles go through HBM.
//
// The contraction dW[a][b] = sum_p G[a][p] H[b][p] needs both operands with the feature on the lane (MFMA rows/columns)
// and the particles along k, while the chain produces them with the particle on the lane.  Here the transposition
// goes through LDS: a workgroup is 4 waves (one per SIMD, up to 512 registers each), wave w walks tile 4*group + w
// through the same chain as rqs_layer_bwd_kernel, and after every stage the four waves write their 64 x 32 operand
// tiles into two staging areas (S_A: the gradient, S_B: the activation it multiplies; 2 x 4 x 8 KiB), meet at a
// barrier, and each wave multiplies ITS share of the 32 x 32 output blocks over all four tiles:
//     stage with a full 64 x 64 product : wave w owns block (w >> 1, w & 1), 4 tiles x 16 k-steps
//     stage with one column tile        : wave w owns block (w & 1, 0) for tiles 2 (w >> 1) .. + 1  (k split)
// so every wave keeps ONE accumulator block per stage (d last-layer blocks + L trunk levels: 9 x 16 registers for
// d = 6, L = 3) for the whole kernel and stores it into the workgroup's slab row at the end (deterministic flush,
// see dw_store), exactly like outer_accum_kernel.  Bias gradients are the row sums of S_A; the waves that share a row tile split the tiles.
// Staging layout: see FB_PS below (row pairs side by side, pair stride 66: one address register per tile and operand,
// bank-conflict free for the producers' 64-float stores and the consumers' ds_read_b64).  LDS: trunk + COMPACT (and
// transposed) last-layer blocks (91 KB for d = 6) + 66 KB staging; d = 7 does not fit and uses the two-kernel path.  With
// one wave per SIMD nothing hides an LDS or HBM round trip, so the MFMA chains are hand-scheduled (chain64) and the
// particle rows of the next group are prefetched.
constexpr int FB_BLOCK = 256;
constexpr int FB_DMAX = 6;
// hidden columns block i keeps (its k-steps rounded up to groups of four, two columns per k-step)
__device__ __forceinline__ int fb_blk_cols(const Sparsity& sp, int i) { return 2 * ((sp.kend3[i] + 3) & ~3); }

// Staged 64 x 32 tile: accumulator register r of row tile rt holds MFMA rows (r & 3) + 8 (r >> 2) + 4 hh for the two
// lane halves hh; the two rows of such a PAIR sit side by side,
//     element (pair 16 rt + r, half hh, particle col)  at  pair * FB_PS + 32 hh + col  =  pair * FB_PS + lane,
// so a producer's store of one register is 64 consecutive floats: ONE address register (4 * lane) for the whole tile, every
// other term an immediate, no bank conflict.  A consumer lane (row i of a 32-row tile, k-half kk) reads the 16 particles
// 16 kk .. 16 kk + 15 of its row as eight ds_read_b64 at immediate offsets of ONE address; with the pair stride 66 the 32
// rows of a tile start on 32 different even banks (2 r + 32 hh mod 64): conflict-free.  (r02 kept rows of 32 floats with
// XOR-swizzled 16-byte chunks: also conflict-free, but eight swizzle registers for the producers and eight chunk
// addresses per product for the consumers — loop invariants that the allocator spilled once the kernel ran at 512.)
constexpr int FB_PS = 66;
constexpr int FB_TILE = 32 * FB_PS;

__device__ __forceinline__ void stage_tile(float* S, int lane, const f32x16_t (&a)[2]) {
    float* q = S + lane;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) q[(16 * rt + r) * FB_PS] = a[rt][r];
}
__device__ __forceinline__ void stage_tile(float* S, int lane, const float (&v)[32]) {
    float* q = S + lane;
#pragma unroll
    for (int m = 0; m < 32; ++m) q[m * FB_PS] = v[m];
}
// rows 0 .. d-1 of a staged tile <- the particle rows x (lanes of half 0 write; row j = pair (j & 3), half (j >> 2))
template <int DMAXR>
__device__ __forceinline__ void stage_x_rows(float* S, int col, int hh, int d, const float (&xr)[DMAXR]) {
    if (hh == 0) {
#pragma unroll
        for (int j = 0; j < DMAXR; ++j)
            if (j < d) S[(j & 3) * FB_PS + 32 * (j >> 2) + col] = xr[j];
    }
}
// ReLU / ReLU-mask fused with the staging store of the same register: the stores are issued between the VALU instructions
// instead of as a burst of 8 KB per wave (x 4 waves through a 64-85 B/clk write path) in front of the next LDS reader
__device__ __forceinline__ void relu2_stage(f32x16_t (&h)[2], float* S, int lane) {
    float* q = S + lane;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            h[rt][r] = relu1(h[rt][r]);
            q[(16 * rt + r) * FB_PS] = h[rt][r];
        }
}
__device__ __forceinline__ void relu_mask_stage(f32x16_t (&gh)[2], const f32x16_t (&h)[2], float* S, int lane) {
    float* q = S + lane;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            gh[rt][r] = (h[rt][r] > 0.0f) ? gh[rt][r] : 0.0f;
            q[(16 * rt + r) * FB_PS] = gh[rt][r];
        }
}
// float offset of (row i of row tile ra, particle 16 kk) within a staged tile
__device__ __forceinline__ int stage_row_offset(int ra, int i, int kk) {
    return (16 * ra + (i & 3) + 4 * (i >> 3)) * FB_PS + 32 * ((i >> 2) & 1) + 16 * kk;
}

struct DwFrag {
    float2 a[8], b[8];
};
__device__ __forceinline__ void dw_load(DwFrag& f, const float* pa, const float* pb) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        f.a[q] = *reinterpret_cast<const float2*>(pa + 2 * q);
        f.b[q] = *reinterpret_cast<const float2*>(pb + 2 * q);
    }
}
__device__ __forceinline__ void dw_mac(const DwFrag& f, bool mm, bool bias, f32x16_t& acc, float& bsum) {
    if (bias) {
#pragma unroll
        for (int q = 0; q < 8; ++q) bsum += f.a[q].x + f.a[q].y;
    }
    if (mm) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            acc = mfma(f.a[q].x, f.b[q].x, acc);
            acc = mfma(f.a[q].y, f.b[q].y, acc);
        }
    }
}
// acc += A[rows 32 ra ..][particles] * B[rows 32 rb ..][particles]^T over the staged tiles [t0, t0 + 2 npair);  bsum += row
// sums of A over the tile pairs selected by bias_pair (-1: every pair; p: pair p only — the two waves that share a row
// tile of a full product split its bias sums between them).  Two fragment sets ping-pong so that the reads of the next
// tile are in flight during the MFMAs of this one.  (Compiler-scheduled form: the emulator build, and the gfx950 build
// without MF_DW_ASM.)
__device__ __forceinline__ void dw_accum(const float* SA, const float* SB, int ra, int rb, int t0, int npair, bool mm,
                                         int bias_pair, int lane, f32x16_t& acc, float& bsum) {
    const int i = lane & 31, kk = lane >> 5;
    const float* pa = SA + t0 * FB_TILE + stage_row_offset(ra, i, kk);
    const float* pb = SB + t0 * FB_TILE + stage_row_offset(rb, i, kk);
    DwFrag f0, f1;
    dw_load(f0, pa, pb);
#pragma unroll 1
    for (int u = 0; u < npair; ++u) {
        const bool bias = bias_pair < 0 || bias_pair == u;
        dw_load(f1, pa + FB_TILE, pb + FB_TILE);
        dw_mac(f0, mm, bias, acc, bsum);
        if (u + 1 < npair) dw_load(f0, pa + 2 * FB_TILE, pb + 2 * FB_TILE);
        dw_mac(f1, mm, bias, acc, bsum);
        pa += 2 * FB_TILE;
        pb += 2 * FB_TILE;
    }
}

// ---- the same product, hand-scheduled (gfx950 build; -DMF_DW_COMPILER keeps the compiler-scheduled form for A/B) -------
// The compiler's code for dw_accum costs a lone wave ~1.3 k cycles per product on top of its MFMAs (r02 ablation:
// 40.8 k cycles for 26.6 k of matrix-pipe time): SLP-packed bias sums (v_pk_add_f32 fed by v_mov / v_accvgpr_read
// shuffles) under exec-mask branches, address arithmetic per call, four waits per tile.  dw_product_asm.inc (generated by
// tools/gen_dw_asm.py) holds ONE asm block per product shape: a ring of 8-byte fragment loads runs a few k-steps ahead
// of the MFMAs, every pair of MFMAs waits for exactly its two loads (counted lgkmcnt; LDS returns in order), the bias row
// sums are plain v_add_f32, nothing branches, and the only operands are the accumulator and ONE address per matrix.
// Fragment registers are fixed physical registers (clobbers): ds_read_b64 fills 2-register tuples whose single registers
// the MFMAs name, which operand constraints cannot express.
#if defined(MF_ASM_CHAIN) && !defined(MF_DW_COMPILER)
#define MF_DW_ASM 1
#include "dw_product_asm.inc"
// This wave's share of one stage's product.  full: block (fra, frb) of a 64 x 64 product over all four tiles, bias row sums
// from tile pair frb (the two waves of a row tile split them); otherwise block (hra, 0) of a one-column-tile product over
// the wave's tile pair ht0, ht0 + 1 (mm = false: the block is all zero, only the bias sums are needed).
struct DwRole {
    int fra, frb, hra, ht0;
};
__device__ __forceinline__ void dw_product_stage(const float* SA, const float* SB, const DwRole& ro, bool full, bool mm, int lane,
                                                 f32x16_t& acc, float& bsum) {
    const int i = lane & 31, kk = lane >> 5;
    if (full) {
        const int t1 = 2 * ro.frb, t2 = 2 - t1;
        const unsigned oa = lds_addr(SA + stage_row_offset(ro.fra, i, kk)), ob = lds_addr(SB + stage_row_offset(ro.frb, i, kk));
        if (mm) dw_product_full(acc, bsum, oa + t1 * FB_TILE * 4, ob + t1 * FB_TILE * 4, oa + t2 * FB_TILE * 4, ob + t2 * FB_TILE * 4);
        else dw_product_bias(acc, bsum, oa + t1 * FB_TILE * 4, ob);
    } else {
        const unsigned oa = lds_addr(SA + ro.ht0 * FB_TILE + stage_row_offset(ro.hra, i, kk));
        const unsigned ob = lds_addr(SB + ro.ht0 * FB_TILE + stage_row_offset(0, i, kk));
        if (mm) dw_product_half(acc, bsum, oa, ob);
        else dw_product_bias(acc, bsum, oa, ob);
    }
}
#endif

// ---- deterministic flush of the per-workgroup parameter-gradient accumulators -------------------------------------
// Every workgroup owns one ROW of a slab buffer gslab[rows][image floats] and writes its accumulator blocks there with
// plain coalesced stores (accumulate != 0: read-modify-write of its own row — later chunks of one backward pass);
// mf_flow_grad_reduce then sums the rows in a fixed order (fp64 accumulation).  No float atomics: the parameter
// gradients are bitwise reproducible run to run, and nothing contends at the end of the kernel.
// gW[(32 ra + row) * stride + 32 rb + col] (+)= acc   (columns < ncols only)
__device__ __forceinline__ void dw_store(float* gW, int stride, int ncols, int ra, int rb, int lane, const f32x16_t& acc,
                                         int accumulate) {
    const int j = lane & 31, hh = lane >> 5;
    if (32 * rb + j < ncols) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float* q = &gW[(32 * ra + rowmap(r, hh)) * stride + 32 * rb + j];
            *q = accumulate ? *q + acc[r] : acc[r];
        }
    }
}
// bsum: per-lane partial row sums (row = lane & 31, the two lane halves hold the two k-halves)
__device__ __forceinline__ void bias_store(float* gB, int ra, int lane, float bsum, int accumulate) {
    bsum += __shfl_xor(bsum, 32);
    if (lane < 32) gB[32 * ra + lane] = accumulate ? gB[32 * ra + lane] + bsum : bsum;
}
// Two waves that split the k range (tile pairs) of ONE output block — waves w and w ^ 2 — meet in LDS: on return the
// accumulators of waves 0 / 1 hold the sum (fixed order: low wave + high wave).  X: 2 * FB_XCH floats, free at kernel end.
constexpr int FB_XCH = 16 * 64 + 64;
__device__ __forceinline__ void pair_reduce_k(float* X, int wid, int lane, f32x16_t& acc, float& bsum) {
    __syncthreads();
    if (wid >= 2) {
        float* q = X + (wid - 2) * FB_XCH;
#pragma unroll
        for (int r = 0; r < 16; ++r) q[r * 64 + lane] = acc[r];
        q[1024 + lane] = bsum;
    }
    __syncthreads();
    if (wid < 2) {
        const float* q = X + wid * FB_XCH;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += q[r * 64 + lane];
        bsum += q[1024 + lane];
    }
}
// The two waves that share a ROW tile of a full 64 x 64 product (waves 2 ra and 2 ra + 1) split its bias row sums:
// the even wave ends up with the total.
__device__ __forceinline__ void pair_reduce_bias(float* X, int wid, int lane, float& bsum) {
    __syncthreads();
    if (wid & 1) X[(wid >> 1) * 64 + lane] = bsum;
    __syncthreads();
    if (!(wid & 1)) bsum += X[(wid >> 1) * 64 + lane];
}

// timing-ablation switches of the diagnostic build (tools/fb_diag.py, WS_DIAG_FLAGS): results are WRONG with any of them
#ifdef MF_FB_NO_BARRIER
#define FB_SYNC() ((void)0)
#else
#define FB_SYNC() __syncthreads()
#endif
#ifdef MF_FB_NO_DW
#define FB_DW(...) ((void)0)
#else
#define FB_DW(...) dw_accum(__VA_ARGS__)
#endif
#ifdef MF_FB_NO_SPLINE
#define FB_SPLINE(...)                                                                                               \
    do {                                                                                                             \
        _Pragma("unroll") for (int m_ = 0; m_ < 32; ++m_) gv[m_] = v[m_] * gyi;                                       \
        gxd = gl; yi = xi; li = 0.0f;                                                                                \
    } while (0)
#else
#define FB_SPLINE(...) __VA_ARGS__
#endif

template <int K, int L>
__global__ __launch_bounds__(FB_BLOCK) void rqs_layer_bwd_fused_kernel(const float* __restrict__ image, int d,
                                                                        const float* __restrict__ x, int64_t n,
                                                                        const float* __restrict__ gy,
                                                                        const float* __restrict__ glogp,
                                                                        float* __restrict__ gx, float* __restrict__ gslab,
                                                                        int accumulate, Sparsity sp, int bins_rt) {
    MF_DYN_SMEM(float, lds);
    const ImageLayout g = image_layout(d, L, d);
    float* gimage = gslab + (int64_t)blockIdx.x * g.total;     // this workgroup's slab row
    // ---- stage the image: trunk as is; last-layer block i TRANSPOSED and compacted to the hidden columns its mask
    // leaves non-zero: T_i[c][m] = W3_i[m][c], c < ncols_i, row stride WS.  Both products then walk LDS with immediate
    // offsets: phi = W3 h reads column m of T (stride WS), gh += W3^T gphi reads row c of T (contiguous).
    for (int i = threadIdx.x * 4; i < g.offW3; i += FB_BLOCK * 4)
        *reinterpret_cast<float4*>(lds + i) = *reinterpret_cast<const float4*>(image + i);
    int off = g.offW3;
    for (int i = 0; i < d; ++i) {
        const int nc = fb_blk_cols(sp, i);
        for (int e = threadIdx.x; e < HID * nc; e += FB_BLOCK) {        // coalesced reads along c, LDS writes stride WS
            const int r = e / nc, c = e - r * nc;
            lds[off + c * WS + r] = image[g.offW3 + (i * HID + r) * WS + c];
        }
        off += nc * WS;
    }
    const int offB3c = off;
    for (int e = threadIdx.x; e < d * HID; e += FB_BLOCK) lds[offB3c + e] = image[g.offB3 + e];
    float* zrow = lds + offB3c + d * HID;                  // 64 zeros: the "row" of a hidden column a block does not store
    if (threadIdx.x < HID) zrow[threadIdx.x] = 0.0f;
    float* SA = lds + ((offB3c + d * HID + HID + 3) & ~3);
    float* SB = SA + 4 * FB_TILE;
    __syncthreads();

    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, col = lane & 31, hh = lane >> 5;
    float* myA = SA + wid * FB_TILE;
    float* myB = SB + wid * FB_TILE;
    // product roles of this wave
    const int fra = wid >> 1, frb = wid & 1;              // full 64 x 64 product
    const int hra = wid & 1, ht0 = 2 * (wid >> 1);        // single column tile, k split over tile pairs
#ifdef MF_DW_ASM
    const DwRole role{fra, frb, hra, ht0};
#endif

    f32x16_t accO[FB_DMAX], accT[L];
    float bsO[FB_DMAX], bsT[L];
#pragma unroll
    for (int i = 0; i < FB_DMAX; ++i) {
        bsO[i] = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) accO[i][r] = 0.0f;
    }
#pragma unroll
    for (int l = 0; l < L; ++l) {
        bsT[l] = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) accT[l][r] = 0.0f;
    }

    unsigned long long c_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t0_, tl_, t1_;
    const int64_t ntiles = (n + 31) / 32;
    const int64_t ngroups = (ntiles + 3) / 4;
    // the particle rows (x, dL/dy, dL/dlog_prob) of the NEXT group are requested at the top of the current one: a lone
    // wave per SIMD would otherwise sit through every HBM round trip
    float xn[FB_DMAX], gyn[FB_DMAX], gln;
    auto load_rows = [&](int64_t grp2, float (&xo)[FB_DMAX], float (&go)[FB_DMAX], float& glo) {
        const int64_t p2 = (grp2 * 4 + wid) * 32 + col;
        const bool v2 = p2 < n;
        const int64_t pc2 = v2 ? p2 : n - 1;
#pragma unroll
        for (int j = 0; j < FB_DMAX; ++j) {
            xo[j] = (j < d) ? x[pc2 * d + j] : 0.0f;
            go[j] = (v2 && j < d) ? gy[pc2 * d + j] : 0.0f;
        }
        glo = v2 ? -glogp[pc2] : 0.0f;
    };
    load_rows(blockIdx.x, xn, gyn, gln);
    for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        MF_NO_HOIST();
        tl_ = t0_ = WS_T();
        const int64_t tile = grp * 4 + wid;
        const int64_t p = tile * 32 + col;
        const bool valid = p < n;
        float xr[FB_DMAX], gyr[FB_DMAX];
#pragma unroll
        for (int j = 0; j < FB_DMAX; ++j) {
            xr[j] = xn[j];
            gyr[j] = gyn[j];
        }
        const float gl = gln;
        {
            const int64_t gnext = grp + gridDim.x;
            load_rows(gnext < ngroups ? gnext : grp, xn, gyn, gln);
        }
        float xb[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) xb[s] = (2 * s < FB_DMAX) ? (hh ? (2 * s + 1 < FB_DMAX ? xr[2 * s + 1] : 0.0f) : xr[2 * s]) : 0.0f;
        // ---- recompute the trunk; h[0] is not kept (it is 8 MFMAs to recompute, and 32 registers to keep)
        f32x16_t h[L][2];
        {
            f32x16_t h0[2];
            input_layer4(lds + g.offW0, lds + g.offB0, g.S0, xb, h0, col, hh);
#pragma unroll
            for (int l = 1; l < L; ++l) {
                const float* W = lds + g.offWh + (l - 1) * (HID * WS + HID);
                // (bias by ONE MFMA per tile — A = the bias in k = 0, B = 1, C = 0 — instead of bias_tile's 16 ds_read_b32 and
                // their exposed round trip was measured in round 3: 16.82 against 16.77 ms per step, not kept)
                h[l][0] = bias_tile(W + HID * WS, 0, hh);
                h[l][1] = bias_tile(W + HID * WS, 1, hh);
                // trunk chains run DENSE here: 8 straight-line groups (64 MFMAs) beat the 5 + 8 mask-bounded groups
                // (52 MFMAs) whose wave-uniform branches break the ds_read / MFMA pipelining of a lone wave
                {
                    // mask-bounded: output tile 0 only sees the k-steps [0, kend_h[0]) (4, 5, 6 or 8 groups of four for
                    // d = 3/5, 6, 4, 2); tile 1 sees all of them.  One switch into straight-line pairs.
                    const float* w0 = W + col * WS + 4 * hh;
                    const float* w1 = W + (32 + col) * WS + 4 * hh;
                    const BTile bt{l == 1 ? h0 : h[l - 1]};
                    switch ((sp.kend_h[0] + 3) >> 2) {
                        case 4: chain64x2r<1, 0, 4, 0, 8>(h[l][0], h[l][1], w0, w1, bt); break;
                        case 5: chain64x2r<1, 0, 5, 0, 8>(h[l][0], h[l][1], w0, w1, bt); break;
                        case 6: chain64x2r<1, 0, 6, 0, 8>(h[l][0], h[l][1], w0, w1, bt); break;
                        default: chain64x2i<1, 8>(h[l][0], h[l][1], w0, w1, bt); break;       // equal ranges: interleaved
                    }
                }
                if (l == L - 1) {
                    FB_SYNC();                           // the previous group's last product has read S_A / S_B
                    relu2_stage(h[l], myB, lane);        // the stores ride between the ReLUs: no 8 KB burst per wave
                } else {
                    relu2(h[l]);
                }
            }
            if (L == 1) { h[0][0] = h0[0]; h[0][1] = h0[1]; }
        }
        WS_ACC(c_[0], t0_);
        if (L == 1) {
            FB_SYNC();                                   // the previous group's last product has read S_A / S_B
            stage_tile(myB, lane, h[L - 1]);
        }
        // ---- output blocks: spline forward + adjoint, dL/dh_last, last-layer weight gradients
        f32x16_t gh[2];
        f32x16_t gacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            gh[0][r] = 0.0f;
            gh[1][r] = 0.0f;
            gacc[r] = 0.0f;
        }
        int w3off = g.offW3;
        // one feature: phi_i, spline forward + adjoint, stage gphi_i, this wave's share of dW3_i, gh += W3_i^T gphi_i
        auto feature = [&](int i, f32x16_t& accF, float& bsF) {
            float v[32], gv[32];
            // per-feature mask bounds by scalar selects over values held in SGPRs: indexing the kernel-argument struct
            // with the run-time feature index costs an exposed scalar-memory round trip per use (three per feature)
            int kend3_i = sp.kend3[0], rt1_i = sp.rt1[0];
#pragma unroll
            for (int j = 1; j < FB_DMAX; ++j) {
                kend3_i = (i == j) ? sp.kend3[j] : kend3_i;
                rt1_i = (i == j) ? sp.rt1[j] : rt1_i;
            }
            const int nc = 2 * ((kend3_i + 3) & ~3);
            const float* W3 = lds + w3off;                 // T_i[c][m]
            w3off += nc * WS;
            t0_ = WS_T();
            {
                f32x16_t phi[2];
                phi[0] = bias_tile(lds + offB3c + i * HID, 0, hh);
                phi[1] = bias_tile(lds + offB3c + i * HID, 1, hh);
                chain64x2_upto<WS>(phi[0], phi[1], W3 + 4 * hh * WS + col, W3 + 4 * hh * WS + 32 + col, (kend3_i + 3) >> 2,
                                   BTile{h[L - 1]});
#pragma unroll
                for (int m = 0; m < 32; ++m) v[m] = phi[m >> 4][m & 15];
            }
            WS_ACC(c_[1], t0_);
            t0_ = WS_T();
            float xi = xr[0], gyi = gyr[0];
#pragma unroll
            for (int j = 1; j < FB_DMAX; ++j) {
                xi = (i == j) ? xr[j] : xi;
                gyi = (i == j) ? gyr[j] : gyi;
            }
            float yi, li, gxd;
            // Barrier A ("product i-1 has read S_A") sits AHEAD of the spline, and the staging stores are issued from inside
            // the adjoint as each slot becomes final: the 32 KB of a staging event drain through the 64-85 B/clk LDS write
            // path underneath the adjoint's VALU work instead of as one burst in front of the next chain, whose fragment
            // reads queue behind it (17.12 -> 16.78 ms per step at C4).
            if (i > 0) FB_SYNC();
#if defined(MF_FB_NO_SPLINE)
            FB_SPLINE(rqs_apply<K, 1>(v, xi, hh, yi, li, gyi, gl, gv, gxd, bins_rt));
            stage_tile(myA, lane, gv);
#else
            {
                float* const qA = myA + lane;
                auto to_stage = [qA](int m, float val) { qA[m * FB_PS] = val; };
                rqs_apply<K, 1>(v, xi, hh, yi, li, gyi, gl, gv, gxd, bins_rt, to_stage);     // its two loops cover all 32 slots
            }
#endif
            WS_ACC(c_[2], t0_);
#pragma unroll
            for (int j = 0; j < 4; ++j) gacc[j] += ((hh == ((i >> 2) & 1)) && ((i & 3) == j)) ? gxd : 0.0f;
            t0_ = WS_T();
            WS_ACC(c_[4], t0_);
            // gh += W3_i^T gphi_i BEFORE the meeting point of the product: the chain gives the four waves ~3 k cycles of
            // slack at barrier B, and gphi (32 registers) is dead by the time the product's fragments are live
            t0_ = WS_T();
            if (kend3_i > 0) {                                     // hidden tile 0 receives something from block i
                const float* r0 = (col < nc ? W3 + col * WS : zrow) + 4 * hh;
                if (rt1_i != 0) {                                  // ... and so does hidden tile 1
                    const float* r1 = (32 + col < nc ? W3 + (32 + col) * WS : zrow) + 4 * hh;
                    if constexpr (K == 20) chain64x2i<1, 8, true>(gh[0], gh[1], r0, r1, BVec{gv});     // 30 slots used
                    else chain64x2i<1, 8>(gh[0], gh[1], r0, r1, BVec{gv});
                } else {
                    if constexpr (K == 20) chain64_30<1>(gh[0], r0, BVec{gv});
                    else chain64<1>(gh[0], r0, 0, 8, BVec{gv});
                }
            }
            WS_ACC(c_[8], t0_);
            t0_ = WS_T();
            FB_SYNC();
            WS_ACC(c_[5], t0_);
            t0_ = WS_T();
            {
                const bool full = rt1_i != 0;
                const bool mm = kend3_i > 0;
#if defined(MF_DW_ASM) && !defined(MF_FB_NO_DW)
                dw_product_stage(SA, SB, role, full, mm, lane, accF, bsF);
#else
                FB_DW(SA, SB, full ? fra : hra, full ? frb : 0, full ? 0 : ht0, full ? 2 : 1, mm, full ? frb : -1, lane, accF, bsF);
#endif
                WS_ACC(c_[6], t0_);
            }
        };
        // The feature loop stays rolled (the spline is ~8 KB of code), so the accumulator of "the current feature"
        // cannot be indexed by i: two features per iteration use accO[0] and accO[1], then the array is rotated by two
        // (register moves; FB_DMAX / 2 iterations bring every block back to its place).  A switch on i whose cases name
        // accO[0..5] removes the 96 moves per pair but costs more than it saves (r03: +290 scalar instructions per group
        // for the dispatch, 17.46 ms against 17.27 per step); rotating after every feature cost twice the moves; unrolling
        // the three iterations (no rotation at all, 1.7 MB of code object instead of 0.96) thrashes the instruction cache:
        // 18.06 ms.
        static_assert(FB_DMAX % 2 == 0, "two features per iteration");
#pragma unroll 1
        for (int i = 0; i < FB_DMAX; i += 2) {
            if (i < d) feature(i, accO[0], bsO[0]);
            if (i + 1 < d) feature(i + 1, accO[1], bsO[1]);
            const f32x16_t ta0 = accO[0], ta1 = accO[1];
            const float tb0 = bsO[0], tb1 = bsO[1];
#pragma unroll
            for (int k = 0; k + 2 < FB_DMAX; ++k) {
                accO[k] = accO[k + 2];
                bsO[k] = bsO[k + 2];
            }
            accO[FB_DMAX - 2] = ta0;
            accO[FB_DMAX - 1] = ta1;
            bsO[FB_DMAX - 2] = tb0;
            bsO[FB_DMAX - 1] = tb1;
        }
        t0_ = WS_T();
        // ---- trunk backward
#pragma unroll
        for (int l = L - 1; l >= 1; --l) {
            t1_ = WS_T();
            if (l == 1) input_layer4(lds + g.offW0, lds + g.offB0, g.S0, xb, h[0], col, hh);
            FB_SYNC();                               // the previous product has read S_A / S_B
            stage_tile(myB, lane, h[l - 1]);
            relu_mask_stage(gh, h[l], myA, lane);    // ReLU mask fused with the staging stores of the masked gradient
            FB_SYNC();
            WS_ACC(c_[12], t1_);
            t1_ = WS_T();
#if defined(MF_DW_ASM) && !defined(MF_FB_NO_DW)
            dw_product_stage(SA, SB, role, true, !(fra == 0 && frb == 1 && sp.kend_h[0] <= 16), lane, accT[l], bsT[l]);
#else
            FB_DW(SA, SB, fra, frb, 0, 2, !(fra == 0 && frb == 1 && sp.kend_h[0] <= 16), frb, lane, accT[l], bsT[l]);
#endif
            WS_ACC(c_[13], t1_);
            t1_ = WS_T();
            f32x16_t t[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                t[0][r] = 0.0f;
                t[1][r] = 0.0f;
            }
            {
                const float* W = lds + g.offWh + (l - 1) * (HID * WS + HID);
                // transposed: input-unit tile 1 only receives from the k-steps [kbeg_ht[1], 32)
                const float* w0 = W + 4 * hh * WS + col;
                const float* w1 = w0 + 32;
                switch (sp.kbeg_ht[1] >> 2) {
                    case 2: chain64x2r<WS, 0, 8, 2, 8>(t[0], t[1], w0, w1, BTile{gh}); break;
                    case 3: chain64x2r<WS, 0, 8, 3, 8>(t[0], t[1], w0, w1, BTile{gh}); break;
                    case 4: chain64x2r<WS, 0, 8, 4, 8>(t[0], t[1], w0, w1, BTile{gh}); break;
                    default: chain64x2i<WS, 8>(t[0], t[1], w0, w1, BTile{gh}); break;      // equal ranges: interleaved
                }
            }
            gh[0] = t[0];
            gh[1] = t[1];
            WS_ACC(c_[14], t1_);
        }
        t1_ = WS_T();
        FB_SYNC();
        relu_mask_stage(gh, h[0], myA, lane);
        stage_x_rows(myB, col, hh, d, xr);                  // S_B rows 0..d-1 <- x (rows >= d: stale finite values, never flushed)
        FB_SYNC();
#if defined(MF_DW_ASM) && !defined(MF_FB_NO_DW)
        dw_product_stage(SA, SB, role, false, true, lane, accT[0], bsT[0]);
#else
        FB_DW(SA, SB, hra, 0, ht0, 1, true, -1, lane, accT[0], bsT[0]);
#endif
        WS_ACC(c_[15], t1_);
        WS_ACC(c_[9], t0_);
        t0_ = WS_T();
        if (gx != nullptr) {
            // rows >= d of the result are never stored, so the lanes col >= d may multiply whatever W0 words they read
            const float* wcol = lds + g.offW0 + 4 * hh * g.S0 + col;
            if (g.S0 == 7) chain64<7>(gacc, wcol, 0, 8, BTile{gh});
            else if (g.S0 == 5) chain64<5>(gacc, wcol, 0, 8, BTile{gh});
            else if (g.S0 == 3) chain64<3>(gacc, wcol, 0, 8, BTile{gh});
            else chain64<1>(gacc, wcol, 0, 8, BTile{gh});
            if (valid) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (4 * hh + j < d) gx[p * d + 4 * hh + j] = gacc[j];
            }
        }
        WS_ACC(c_[10], t0_);
        WS_ACC(c_[11], tl_);
    }
    t0_ = WS_T();
    // ---- flush the accumulators (image coordinates) into this workgroup's slab row: plain stores, fixed order
#pragma unroll
    for (int i = 0; i < FB_DMAX; ++i) {
        if (i < d) {
            const bool full = sp.rt1[i] != 0;
            if (full) {
                pair_reduce_bias(SA, wid, lane, bsO[i]);
                if (sp.kend3[i] > 0) dw_store(gimage + g.offW3 + i * HID * WS, WS, HID, fra, frb, lane, accO[i], accumulate);
                if (frb == 0) bias_store(gimage + g.offB3 + i * HID, fra, lane, bsO[i], accumulate);
            } else {
                pair_reduce_k(SA, wid, lane, accO[i], bsO[i]);
                if (wid < 2) {
                    if (sp.kend3[i] > 0) dw_store(gimage + g.offW3 + i * HID * WS, WS, HID, hra, 0, lane, accO[i], accumulate);
                    bias_store(gimage + g.offB3 + i * HID, hra, lane, bsO[i], accumulate);
                }
            }
        }
    }
#pragma unroll
    for (int l = 1; l < L; ++l) {
        float* gW = gimage + g.offWh + (l - 1) * (HID * WS + HID);
        pair_reduce_bias(SA, wid, lane, bsT[l]);
        if (!(fra == 0 && frb == 1 && sp.kend_h[0] <= 16)) dw_store(gW, WS, HID, fra, frb, lane, accT[l], accumulate);
        if (frb == 0) bias_store(gW + HID * WS, fra, lane, bsT[l], accumulate);
    }
    pair_reduce_k(SA, wid, lane, accT[0], bsT[0]);
    if (wid < 2) {
        dw_store(gimage + g.offW0, g.S0, d, hra, 0, lane, accT[0], accumulate);
        bias_store(gimage + g.offB0, hra, lane, bsT[0], accumulate);
    }
#if defined(MF_WS_DIAG) && !defined(MF_EMU)
    __builtin_amdgcn_s_waitcnt(0);
    c_[7] = WS_T() - t0_;                                  // slot 7: the final flush, once per workgroup
    if (lane == 0)
        for (int q = 0; q < 16; ++q) g_ws_diag[(blockIdx.x * 4 + wid) * 16 + q] = c_[q];
#endif
}

// =========================================================================================== inverse (density of a point)
// x = T^-1(y) for one autoregressive layer (zuko AutoregressiveTransform._inverse: "x = 0; repeat d times
// x = meta(x).inv(y)").  Feature of order t only depends on features of order < t, so the d passes are done in order:
// pass t recomputes the conditioner on the current x^ and inverts the single feature of order t (identical values to
// zuko's d full passes).  x^ lives in a per-wave LDS strip so that it can be re-read as MFMA B operands.
struct InvOrder {
    int feat[FLOW_DMAX + 1];    // feat[t] = feature whose order is t
};

#ifdef MF_EMU
#define MF_WAVE_SYNC() emu::wave_sync()
#else
#define MF_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#endif

constexpr int INV_BLOCK = 512;
template <int K, int L>     // K > 0 or RQS_ANY: rational-quadratic spline;  K == 0: affine
__global__ __launch_bounds__(INV_BLOCK) void layer_inv_kernel(const float* __restrict__ image, int d,
                                                              const float* __restrict__ y, int64_t n,
                                                              float* __restrict__ x, Sparsity sp, InvOrder io, int bins_rt) {
    MF_DYN_SMEM(float, lds);
    const int nblk = (K != 0) ? d : 1;
    const ImageLayout g = image_layout(d, L, nblk);
    stage_image<INV_BLOCK>(lds, image, g.total);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, col = lane & 31, hh = lane >> 5;
    float* xs = lds + g.total + wid * (32 * 8) + col * 8;        // this particle's x^[0..7]
    const int64_t ntiles = (n + 31) / 32;
    for (int64_t tile = (int64_t)blockIdx.x * (INV_BLOCK / 64) + wid; tile < ntiles;
         tile += (int64_t)gridDim.x * (INV_BLOCK / 64)) {
        MF_NO_HOIST();
        const int64_t p = tile * 32 + col;
        const bool valid = p < n;
        const float* yp = y + (valid ? p : n - 1) * d;
        if (hh == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) xs[j] = 0.0f;
        }
        MF_WAVE_SYNC();
        for (int t = 0; t < d; ++t) {
            MF_NO_HOIST();
            const int i = io.feat[t];
            const int blk = (K != 0) ? i : 0;
            float v[32];
            const bool pure_bias = (K != 0) && (sp.kend3[i] == 0);
            if (pure_bias) {
#pragma unroll
                for (int m = 0; m < 32; ++m) v[m] = lds[g.offB3 + blk * HID + 32 * (m >> 4) + rowmap(m & 15, hh)];
            } else {
                float xb[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) xb[s] = (2 * s + hh < d) ? xs[2 * s + hh] : 0.0f;
                f32x16_t h[2];
                input_layer(lds + g.offW0, lds + g.offB0, g.S0, d, xb, h, col, hh);
#pragma unroll
                for (int l = 1; l < L; ++l) {
                    f32x16_t tt[2];
                    const float* W = lds + g.offWh + (l - 1) * (HID * WS + HID);
                    linear64(W, W + HID * WS, h, tt, col, hh, sp.kend_h[0], sp.kend_h[1]);
                    relu2(tt);
                    h[0] = tt[0];
                    h[1] = tt[1];
                }
                f32x16_t phi[2];
                linear64(lds + g.offW3 + blk * HID * WS, lds + g.offB3 + blk * HID, h, phi, col, hh, sp.kend3[blk],
                         (K != 0) ? sp.kend3[blk] : 0);
#pragma unroll
                for (int m = 0; m < 32; ++m) v[m] = phi[m >> 4][m & 15];
            }
            float xi;
            if (K != 0) {
                float li, gxd, gdummy[32];
                rqs_apply<(K != 0 ? K : 8), 2>(v, yp[i], hh, xi, li, 0.0f, 0.0f, gdummy, gxd, bins_rt);
            } else {
                // slot i of half 0 = shift_i, of half 1 = scale_i (runtime i: select among the 8 candidate slots)
                float mine = 0.0f;
#pragma unroll
                for (int j = 0; j < FLOW_DMAX + 1; ++j) mine = (j == i) ? v[j] : mine;
                const float other = __shfl_xor(mine, 32);
                const float shift = hh ? other : mine, scale = hh ? mine : other;
                xi = (yp[i] - shift) * fast_exp(-soft_clip(scale, LOG_SLOPE_INV));
            }
            MF_WAVE_SYNC();
            if (hh == 0) xs[i] = xi;
            MF_WAVE_SYNC();
        }
        if (valid && hh == 0) {
            for (int j = 0; j < d; ++j) x[p * d + j] = xs[j];
        }
        MF_WAVE_SYNC();
    }
}

// =========================================================================================== affine (MAF) layers
// zuko MonotonicAffineTransform: y = x * exp(s~) + t, s~ = s / (1 + |s / log(1e-3)|), ladj = s~  (build.py:28 "maf").
// One output block: slot i of lane half 0 = shift_i, of half 1 = scale_i (row tile 0 only; tile 1 is padding).
template <int L, int BLOCK>
__global__ __launch_bounds__(BLOCK) void affine_layer_fwd_kernel(const float* __restrict__ image, int d,
                                                                 const float* __restrict__ x, int64_t n,
                                                                 float* __restrict__ y, const float* __restrict__ logp_in,
                                                                 float* __restrict__ logp_out, int init_logp, Sparsity sp) {
    MF_DYN_SMEM(float, lds);
    const ImageLayout g = image_layout(d, L, 1);
    stage_image<BLOCK>(lds, image, g.total);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, col = lane & 31, hh = lane >> 5;
    const int64_t ntiles = (n + 31) / 32;
    for (int64_t tile = (int64_t)blockIdx.x * (BLOCK / 64) + wid; tile < ntiles; tile += (int64_t)gridDim.x * (BLOCK / 64)) {
        MF_NO_HOIST();
        const int64_t p = tile * 32 + col;
        const bool valid = p < n;
        const float* xp = x + (valid ? p : n - 1) * d;
        float xb[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) xb[s] = (2 * s + hh < d) ? xp[2 * s + hh] : 0.0f;
        f32x16_t h[2];
        input_layer(lds + g.offW0, lds + g.offB0, g.S0, d, xb, h, col, hh);
#pragma unroll
        for (int l = 1; l < L; ++l) {
            f32x16_t t[2];
            const float* W = lds + g.offWh + (l - 1) * (HID * WS + HID);
            linear64(W, W + HID * WS, h, t, col, hh, sp.kend_h[0], sp.kend_h[1]);
            relu2(t);
            h[0] = t[0];
            h[1] = t[1];
        }
        f32x16_t phi[2];
        linear64(lds + g.offW3, lds + g.offB3, h, phi, col, hh, 32, 0);
        float ladj = 0.0f;
#pragma unroll
        for (int i = 0; i < FLOW_DMAX + 1; ++i) {
            if (i < d) {
                const float mine = phi[0][i];
                const float other = __shfl_xor(mine, 32);
                const float shift = hh ? other : mine, scale = hh ? mine : other;
                const float ls = soft_clip(scale, LOG_SLOPE_INV);
                ladj += ls;
                if (valid && hh == 0) y[p * d + i] = fmaf(xp[i], fast_exp(ls), shift);
            }
        }
        if (valid && hh == 0) {
            const float lp0 = init_logp ? base_log_prob(xp, d) : logp_in[p];
            logp_out[p] = lp0 - ladj;
        }
    }
}

// scratch: ACT[L][npad][64] | GPRE[L][npad][64] | GPHI[1][npad][64]
template <int L>
__global__ __launch_bounds__(FLOW_BLOCK) void affine_layer_bwd_kernel(const float* __restrict__ image, int d,
                                                                      const float* __restrict__ x, int64_t n,
                                                                      const float* __restrict__ gy,
                                                                      const float* __restrict__ glogp, float* __restrict__ gx,
                                                                      float* __restrict__ scratch, Sparsity sp) {
    MF_DYN_SMEM(float, lds);
    const ImageLayout g = image_layout(d, L, 1);
    stage_image(lds, image, g.total);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, col = lane & 31, hh = lane >> 5;
    const int64_t ntiles = (n + 31) / 32;
    const int64_t npad = ntiles * 32;
    float* ACT = scratch;
    float* GPRE = ACT + (int64_t)L * npad * 64;
    float* GPHI = GPRE + (int64_t)L * npad * 64;
    for (int64_t tile = (int64_t)blockIdx.x * FLOW_WAVES + wid; tile < ntiles; tile += (int64_t)gridDim.x * FLOW_WAVES) {
        MF_NO_HOIST();
        const int64_t p = tile * 32 + col;
        const bool valid = p < n;
        const int64_t pc = valid ? p : n - 1;
        const float* xp = x + pc * d;
        float xb[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) xb[s] = (2 * s + hh < d) ? xp[2 * s + hh] : 0.0f;
        f32x16_t h[L][2];
        input_layer(lds + g.offW0, lds + g.offB0, g.S0, d, xb, h[0], col, hh);
        store_tile(ACT, tile, col, hh, h[0]);
#pragma unroll
        for (int l = 1; l < L; ++l) {
            const float* W = lds + g.offWh + (l - 1) * (HID * WS + HID);
            linear64(W, W + HID * WS, h[l - 1], h[l], col, hh, sp.kend_h[0], sp.kend_h[1]);
            relu2(h[l]);
            store_tile(ACT + (int64_t)l * npad * 64, tile, col, hh, h[l]);
        }
        f32x16_t phi[2];
        linear64(lds + g.offW3, lds + g.offB3, h[L - 1], phi, col, hh, 32, 0);
        const float gl = valid ? -glogp[pc] : 0.0f;
        f32x16_t gacc;
        float gv[32];
#pragma unroll
        for (int r = 0; r < 16; ++r) gacc[r] = 0.0f;
#pragma unroll
        for (int m = 0; m < 32; ++m) gv[m] = 0.0f;
#pragma unroll
        for (int i = 0; i < FLOW_DMAX + 1; ++i) {
            if (i < d) {
                const float mine = phi[0][i];
                const float other = __shfl_xor(mine, 32);
                const float scale = hh ? mine : other;
                const float ls = soft_clip(scale, LOG_SLOPE_INV);
                const float e = fast_exp(ls);
                const float gyi = valid ? gy[pc * d + i] : 0.0f;
                const float gls = fmaf(gyi * xp[i], e, gl);               // dL/ds~ : through y and through ladj
                gv[i] = hh ? gls * soft_clip_grad(scale, LOG_SLOPE_INV) : gyi;
                const float gxd = gyi * e;
#pragma unroll
                for (int j = 0; j < 4; ++j) gacc[j] += ((hh == ((i >> 2) & 1)) && ((i & 3) == j)) ? gxd : 0.0f;
            }
        }
        store_tile(GPHI, tile, col, hh, gv);
        // gh = W3^T gphi: only the 16 slots of row tile 0 are populated
        f32x16_t gh[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            f32x16_t acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            const float* wcol = lds + g.offW3 + 4 * hh * WS + 32 * rt + col;
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = mfma(wcol[rowmap(s, 0) * WS], gv[s], acc);
            gh[rt] = acc;
        }
#pragma unroll
        for (int l = L - 1; l >= 1; --l) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) gh[rt][r] = (h[l][rt][r] > 0.0f) ? gh[rt][r] : 0.0f;
            store_tile(GPRE + (int64_t)l * npad * 64, tile, col, hh, gh);
            f32x16_t t[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                t[0][r] = 0.0f;
                t[1][r] = 0.0f;
            }
            linear64_t(lds + g.offWh + (l - 1) * (HID * WS + HID), gh, t, col, hh, sp.kbeg_ht[0], sp.kbeg_ht[1]);
            gh[0] = t[0];
            gh[1] = t[1];
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) gh[rt][r] = (h[0][rt][r] > 0.0f) ? gh[rt][r] : 0.0f;
        store_tile(GPRE, tile, col, hh, gh);
        if (gx != nullptr) {
            const float* wcol = lds + g.offW0 + 4 * hh * g.S0 + col;
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                const int kk = 32 * (s >> 4) + rowmap(s & 15, 0);
                const float a = (col < d) ? wcol[kk * g.S0] : 0.0f;
                gacc = mfma(a, gh[s >> 4][s & 15], gacc);
            }
            if (valid) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (4 * hh + j < d) gx[p * d + 4 * hh + j] = gacc[j];
            }
        }
    }
}

// =========================================================================================== parameter gradients
// C[a][b] = sum_p A[p][a] * B[p][b] over particles (MFMA k = particle), bias[a] = sum_p A[p][a], for every linear
// layer of the conditioner.  grid (G, 2): blockIdx.y = 0 -> the `nblk` output blocks of the last layer (wave w owns
// block w: A = GPHI[w], B = ACT[L-1] shared by all waves of the workgroup through L1/L2);  blockIdx.y = 1 -> the trunk
// (wave 0: A = GPRE[0], B = x;  wave l: A = GPRE[l], B = ACT[l-1]).  Every wave keeps its 64x64 result in 64
// accumulator registers over all the tiles it visits and stores it into its slab row (image coordinates, see dw_store).
constexpr int OA_MAX_WAVES = 8;
__global__ __launch_bounds__(64 * OA_MAX_WAVES) void outer_accum_kernel(const float* __restrict__ scratch,
                                                                        const float* __restrict__ x, int64_t n, int d,
                                                                        int L, int nblk, float* __restrict__ gslab,
                                                                        int accumulate, Sparsity sp) {
    const ImageLayout g = image_layout(d, L, nblk);
    float* gimage = gslab + (int64_t)blockIdx.x * g.total;     // this workgroup column's slab row (blockIdx.y: disjoint parts)
    const int64_t ntiles = (n + 31) / 32;
    const int64_t npad = ntiles * 32;
    const float* ACT = scratch;
    const float* GPRE = ACT + (int64_t)L * npad * 64;
    const float* GPHI = GPRE + (int64_t)L * npad * 64;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, col = lane & 31, hh = lane >> 5;
    const float* A;
    const float* B = nullptr;
    int offW, offB, strideW;
    bool need_b0 = true, need_b1 = true;       // which 32-column tiles of B can be non-zero
    bool from_x = false;
    if (blockIdx.y == 0) {
        if (wid >= nblk) return;
        A = GPHI + (int64_t)wid * npad * 64;
        B = ACT + (int64_t)(L - 1) * npad * 64;
        offW = g.offW3 + wid * HID * WS; offB = g.offB3 + wid * HID; strideW = WS;
        need_b0 = sp.kend3[wid] > 0;
        need_b1 = sp.rt1[wid] != 0;
    } else {
        if (wid >= L) return;
        A = GPRE + (int64_t)wid * npad * 64;
        if (wid == 0) {
            from_x = true;
            need_b1 = false;
            offW = g.offW0; offB = g.offB0; strideW = g.S0;
        } else {
            B = ACT + (int64_t)(wid - 1) * npad * 64;
            offW = g.offWh + (wid - 1) * (HID * WS + HID); offB = offW + HID * WS; strideW = WS;
        }
    }
    f32x16_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    float bsum0 = 0.0f, bsum1 = 0.0f;
    // lane (col, hh): column 32*t + col, particles 16*hh + s (s = 0..15): k-step s pairs particles (s, 16 + s).
    // Work is issued in half tiles (8 particles per lane half: eight 16-byte loads, 32 MFMAs) so that the kernel stays
    // under 128 VGPRs: four waves per SIMD hide the HBM latency better than a deeper per-wave prefetch did.
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            const float4* pa0 = reinterpret_cast<const float4*>(A + tile * 2048 + col * 32 + 16 * hh) + 2 * half;
            const float4* pa1 = reinterpret_cast<const float4*>(A + tile * 2048 + (32 + col) * 32 + 16 * hh) + 2 * half;
            float4 a0[2], a1[2], b0[2], b1[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                a0[q] = pa0[q];
                a1[q] = pa1[q];
            }
            if (from_x) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    float t[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int64_t p = tile * 32 + 16 * hh + 4 * (2 * half + q) + e;
                        t[e] = (col < d && p < n) ? x[p * d + col] : 0.0f;
                    }
                    b0[q] = make_float4(t[0], t[1], t[2], t[3]);
                    b1[q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                }
            } else {
                const float4* pb0 = reinterpret_cast<const float4*>(B + tile * 2048 + col * 32 + 16 * hh) + 2 * half;
                const float4* pb1 = reinterpret_cast<const float4*>(B + tile * 2048 + (32 + col) * 32 + 16 * hh) + 2 * half;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    b0[q] = need_b0 ? pb0[q] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    b1[q] = need_b1 ? pb1[q] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float av0[4] = {a0[q].x, a0[q].y, a0[q].z, a0[q].w};
                const float av1[4] = {a1[q].x, a1[q].y, a1[q].z, a1[q].w};
                const float bv0[4] = {b0[q].x, b0[q].y, b0[q].z, b0[q].w};
                const float bv1[4] = {b1[q].x, b1[q].y, b1[q].z, b1[q].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bsum0 += av0[e];
                    bsum1 += av1[e];
                    if (need_b0) {
                        acc[0][0] = mfma(av0[e], bv0[e], acc[0][0]);
                        acc[1][0] = mfma(av1[e], bv0[e], acc[1][0]);
                    }
                    if (need_b1) {
                        acc[0][1] = mfma(av0[e], bv1[e], acc[0][1]);
                        acc[1][1] = mfma(av1[e], bv1[e], acc[1][1]);
                    }
                }
            }
        }
    }
    // memory column c = 32*rt + 16*hc + r  <->  image row rho = 32*rt + rowmap(r, hc)
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            if (!(tb ? need_b1 : need_b0)) continue;
            const int rhoB = from_x ? col : (32 * tb + rowmap(col & 15, col >> 4));
            if (from_x && col >= d) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int wA = rowmap(r, hh);                       // C row within the tile = memory column of A
                const int rhoA = 32 * ta + rowmap(wA & 15, wA >> 4);
                float* q = &gimage[offW + rhoA * strideW + rhoB];
                *q = accumulate ? *q + acc[ta][tb][r] : acc[ta][tb][r];
            }
        }
    bsum0 += __shfl_xor(bsum0, 32);
    bsum1 += __shfl_xor(bsum1, 32);
    if (hh == 0) {
        const int rho = rowmap(col & 15, col >> 4);
        gimage[offB + rho] = accumulate ? gimage[offB + rho] + bsum0 : bsum0;
        gimage[offB + 32 + rho] = accumulate ? gimage[offB + 32 + rho] + bsum1 : bsum1;
    }
}

// =========================================================================================== backward, affine, fused
// The MAF counterpart of rqs_layer_bwd_fused_kernel: same 4-wave workgroups, LDS staging of the parameter-gradient
// operands and hand-scheduled chains; one output block (slot i of lane half 0 = shift_i, of half 1 = scale_i, all in
// row tile 0), so the last-layer product is two 32 x 32 blocks (0, w & 1) with k split over the tile pairs (w >> 1).
// The dense image fits LDS next to the staging areas for every d <= 7 (53 KB + 64 KB).
template <int L>
__global__ __launch_bounds__(FB_BLOCK) void affine_layer_bwd_fused_kernel(const float* __restrict__ image, int d,
                                                                           const float* __restrict__ x, int64_t n,
                                                                           const float* __restrict__ gy,
                                                                           const float* __restrict__ glogp,
                                                                           float* __restrict__ gx, float* __restrict__ gslab,
                                                                           int accumulate) {
    MF_DYN_SMEM(float, lds);
    const ImageLayout g = image_layout(d, L, 1);
    float* gimage = gslab + (int64_t)blockIdx.x * g.total;     // this workgroup's slab row
    stage_image<FB_BLOCK>(lds, image, g.total);
    float* SA = lds + ((g.total + 3) & ~3);
    float* SB = SA + 4 * FB_TILE;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, col = lane & 31, hh = lane >> 5;
    float* myA = SA + wid * FB_TILE;
    float* myB = SB + wid * FB_TILE;
    const int fra = wid >> 1, frb = wid & 1;              // full 64 x 64 product
    const int hra = wid & 1, ht0 = 2 * (wid >> 1);        // single column / row tile, k split over tile pairs
    constexpr int DM = FLOW_DMAX + 1;

    f32x16_t accO, accT[L];
    float bsO = 0.0f, bsT[L];
#pragma unroll
    for (int r = 0; r < 16; ++r) accO[r] = 0.0f;
#pragma unroll
    for (int l = 0; l < L; ++l) {
        bsT[l] = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) accT[l][r] = 0.0f;
    }
    const int64_t ntiles = (n + 31) / 32;
    const int64_t ngroups = (ntiles + 3) / 4;
    float xn[DM], gyn[DM], gln;
    auto load_rows = [&](int64_t grp2, float (&xo)[DM], float (&go)[DM], float& glo) {
        const int64_t p2 = (grp2 * 4 + wid) * 32 + col;
        const bool v2 = p2 < n;
        const int64_t pc2 = v2 ? p2 : n - 1;
#pragma unroll
        for (int j = 0; j < DM; ++j) {
            xo[j] = (j < d) ? x[pc2 * d + j] : 0.0f;
            go[j] = (v2 && j < d) ? gy[pc2 * d + j] : 0.0f;
        }
        glo = v2 ? -glogp[pc2] : 0.0f;
    };
    load_rows(blockIdx.x, xn, gyn, gln);
    for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        MF_NO_HOIST();
        const int64_t p = (grp * 4 + wid) * 32 + col;
        const bool valid = p < n;
        float xr[DM], gyr[DM];
#pragma unroll
        for (int j = 0; j < DM; ++j) {
            xr[j] = xn[j];
            gyr[j] = gyn[j];
        }
        const float gl = gln;
        {
            const int64_t gnext = grp + gridDim.x;
            load_rows(gnext < ngroups ? gnext : grp, xn, gyn, gln);
        }
        float xb[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) xb[s] = hh ? xr[2 * s + 1] : xr[2 * s];
        // ---- trunk (h[0] is recomputed later instead of kept)
        f32x16_t h[L][2];
        {
            f32x16_t h0[2];
            input_layer4(lds + g.offW0, lds + g.offB0, g.S0, xb, h0, col, hh);
#pragma unroll
            for (int l = 1; l < L; ++l) {
                const float* W = lds + g.offWh + (l - 1) * (HID * WS + HID);
                h[l][0] = bias_tile(W + HID * WS, 0, hh);
                h[l][1] = bias_tile(W + HID * WS, 1, hh);
                // dense image (the affine kernel takes no mask structure): both row tiles over all 8 groups, interleaved
                chain64x2i<1, 8>(h[l][0], h[l][1], W + col * WS + 4 * hh, W + (32 + col) * WS + 4 * hh, BTile{l == 1 ? h0 : h[l - 1]});
                relu2(h[l]);
            }
            if (L == 1) { h[0][0] = h0[0]; h[0][1] = h0[1]; }
        }
        __syncthreads();                                   // the previous group's last product has read S_A / S_B
        stage_tile(myB, lane, h[L - 1]);
        // ---- output block (row tile 0 only) and the affine adjoint
        const float* W3 = lds + g.offW3;
        f32x16_t phi = bias_tile(lds + g.offB3, 0, hh);
        chain64<1>(phi, W3 + col * WS + 4 * hh, 0, 8, BTile{h[L - 1]});
        f32x16_t gacc;
        float gv[32];
#pragma unroll
        for (int r = 0; r < 16; ++r) gacc[r] = 0.0f;
#pragma unroll
        for (int m = 0; m < 32; ++m) gv[m] = 0.0f;
#pragma unroll
        for (int i = 0; i < DM; ++i) {
            if (i < d) {
                const float mine = phi[i];
                const float other = __shfl_xor(mine, 32);
                const float scale = hh ? mine : other;
                const float ls = soft_clip(scale, LOG_SLOPE_INV);
                const float e = fast_exp(ls);
                const float gyi = gyr[i];
                const float gls = fmaf(gyi * xr[i], e, gl);               // dL/ds~ : through y and through ladj
                gv[i] = hh ? gls * soft_clip_grad(scale, LOG_SLOPE_INV) : gyi;
                const float gxd = gyi * e;
#pragma unroll
                for (int j = 0; j < 4; ++j) gacc[j] += ((hh == ((i >> 2) & 1)) && ((i & 3) == j)) ? gxd : 0.0f;
            }
        }
        stage_tile(myA, lane, gv);
        __syncthreads();
        dw_accum(SA, SB, 0, frb, ht0, 1, true, frb == 0 ? -1 : 99, lane, accO, bsO);
        // gh = W3^T gphi: only the 16 slots of row tile 0 are populated (k-step groups 0..3)
        f32x16_t gh[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) gh[rt][r] = 0.0f;
        chain64x2i<WS, 4>(gh[0], gh[1], W3 + 4 * hh * WS + col, W3 + 4 * hh * WS + 32 + col, BVec{gv});
        // ---- trunk backward
#pragma unroll
        for (int l = L - 1; l >= 1; --l) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) gh[rt][r] = (h[l][rt][r] > 0.0f) ? gh[rt][r] : 0.0f;
            if (l == 1) input_layer4(lds + g.offW0, lds + g.offB0, g.S0, xb, h[0], col, hh);
            __syncthreads();
            stage_tile(myA, lane, gh);
            stage_tile(myB, lane, h[l - 1]);
            __syncthreads();
            dw_accum(SA, SB, fra, frb, 0, 2, true, frb, lane, accT[l], bsT[l]);
            f32x16_t t[2];
            const float* W = lds + g.offWh + (l - 1) * (HID * WS + HID);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) t[rt][r] = 0.0f;
            chain64x2i<WS, 8>(t[0], t[1], W + 4 * hh * WS + col, W + 4 * hh * WS + 32 + col, BTile{gh});
            gh[0] = t[0];
            gh[1] = t[1];
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) gh[rt][r] = (h[0][rt][r] > 0.0f) ? gh[rt][r] : 0.0f;
        __syncthreads();
        stage_tile(myA, lane, gh);
        stage_x_rows(myB, col, hh, d, xr);
        __syncthreads();
        dw_accum(SA, SB, hra, 0, ht0, 1, true, -1, lane, accT[0], bsT[0]);
        if (gx != nullptr) {
            const float* wcol = lds + g.offW0 + 4 * hh * g.S0 + col;
            if (g.S0 == 7) chain64<7>(gacc, wcol, 0, 8, BTile{gh});
            else if (g.S0 == 5) chain64<5>(gacc, wcol, 0, 8, BTile{gh});
            else if (g.S0 == 3) chain64<3>(gacc, wcol, 0, 8, BTile{gh});
            else chain64<1>(gacc, wcol, 0, 8, BTile{gh});
            if (valid) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (4 * hh + j < d) gx[p * d + 4 * hh + j] = gacc[j];
            }
        }
    }
    // ---- deterministic flush into this workgroup's slab row (see dw_store)
    pair_reduce_k(SA, wid, lane, accO, bsO);               // block (0, frb): k split over waves w, w ^ 2; bias in wave 0
    if (wid < 2) {
        dw_store(gimage + g.offW3, WS, HID, 0, frb, lane, accO, accumulate);
        if (frb == 0) bias_store(gimage + g.offB3, 0, lane, bsO, accumulate);
    }
#pragma unroll
    for (int l = 1; l < L; ++l) {
        float* gW = gimage + g.offWh + (l - 1) * (HID * WS + HID);
        pair_reduce_bias(SA, wid, lane, bsT[l]);
        dw_store(gW, WS, HID, fra, frb, lane, accT[l], accumulate);
        if (frb == 0) bias_store(gW + HID * WS, fra, lane, bsT[l], accumulate);
    }
    pair_reduce_k(SA, wid, lane, accT[0], bsT[0]);
    if (wid < 2) {
        dw_store(gimage + g.offW0, g.S0, d, hra, 0, lane, accT[0], accumulate);
        bias_store(gimage + g.offB0, hra, lane, bsT[0], accumulate);
    }
}

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

// Built spline instances: bins 20 (the reference's value, experiments/setup.py:119-121) and 8 (zuko's default) at compile
// time; every other 2 <= bins <= 21 through the run-time instance (RQS_ANY: slots laid out for 21 bins, slower).
#define MF_RQS_CASES(X) X(20, 3) X(20, 2) X(8, 3) X(8, 2) X(RQS_ANY, 3) X(RQS_ANY, 2)
static bool rqs_case_matches(int KK, int bins) {
    return KK == RQS_ANY ? (bins != 20 && bins != 8 && bins >= 2 && bins <= RQS_KMAX) : bins == KK;
}
// slot of this lane half's first derivative logit in the packed last-layer block (mentflow_amd/generate/packing.py): the
// number of bins for the compile-time instances, 21 for the run-time one; -1: no kernel for this number of bins
extern "C" int mf_flow_rqs_deriv_slot(int bins) {
    if (bins == 20 || bins == 8) return bins;
    return (bins >= 2 && bins <= RQS_KMAX) ? RQS_KMAX : -1;
}

extern "C" int mf_flow_rqs_layer_fwd(const float* image, int d, int hidden_layers, int bins, const int32_t* order,
                                      const float* x, int64_t n, float* y, const float* logp_in, float* logp_out,
                                      int init_logp, void* stream) {
    if (flow_check(d, hidden_layers, n)) return 1;
    if (n == 0) return 0;
    const Sparsity sp = make_sparsity(d, order, d);
    const size_t smem = sizeof(float) * (size_t)image_layout(d, hidden_layers, d).total;
    // Workgroup size: 1024 threads (4 waves per SIMD at <= 128 VGPRs) for big batches; small batches (the reference's
    // 25 000 particles = 782 tiles) use fewer waves per workgroup so that the tiles spread over all 256 CUs.
    const int64_t nt = (n + 31) / 32;
    static const int fwd_block_env = [] { const char* e = getenv("MENTFLOW_FWD_BLOCK"); return e ? atoi(e) : 0; }();
    const int fwd_block = fwd_block_env ? fwd_block_env : (nt <= 4 * NUM_CU ? 256 : (nt <= 8 * NUM_CU ? 512 : 1024));
#define XB(KK, LL, BB)                                                                                                \
    if (fwd_block == BB) {                                                                                            \
        MF_ALLOW_DYN_SMEM((rqs_layer_fwd_kernel<KK, LL, BB>), smem);                                                  \
        MF_LAUNCH((rqs_layer_fwd_kernel<KK, LL, BB>), flow_grid(n, BB / 64), BB, smem, stream, image, d, x, n, y,      \
                  logp_in, logp_out, init_logp, sp, bins);                                            \
        return check_launch("mf_flow_rqs_layer_fwd");                                                                 \
    }
#define X(KK, LL)                                                                                                     \
    if (rqs_case_matches(KK, bins) && hidden_layers == LL) {                                                          \
        ProfScope prof(PK_FLOW_FWD, stream);                                                                          \
        XB(KK, LL, 256) XB(KK, LL, 512) XB(KK, LL, 1024)                                                              \
        return fail("MENTFLOW_FWD_BLOCK must be 256, 512 or 1024");                                                   \
    }
    MF_RQS_CASES(X)
#undef X
    return fail("no RQS kernel instance for bins=%d hidden_layers=%d (built: 2 <= bins <= 21, hidden_layers in {2,3})", bins,
                hidden_layers);
}

// grid sizes of the backward kernels: the number of slab rows a call writes (one per workgroup column)
static int fused_grid(int64_t n) {
    const int64_t ngroups = ((n + 31) / 32 + 3) / 4;
    return (int)(ngroups > NUM_CU ? NUM_CU : (ngroups < 1 ? 1 : ngroups));
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
    const size_t smem = sizeof(float) * (size_t)image_layout(d, hidden_layers, d).total;
    bool launched = false;
    // fused backward + parameter gradients (no scratch traffic): 19.2 ms against 11.1 + 9.3 ms at 2 M particles (C4)
    {
        size_t smem_f = 0;
        if (!launched && rqs_bwd_fused(n, d, hidden_layers, order, sp, &smem_f)) {
            {
                const int gf = fused_grid(n);
#define XF(KK, LL)                                                                                                    \
    if (!launched && rqs_case_matches(KK, bins) && hidden_layers == LL) {                                             \
        ProfScope prof(PK_FLOW_BWD, stream);                                                                          \
        MF_ALLOW_DYN_SMEM((rqs_layer_bwd_fused_kernel<KK, LL>), smem_f);                                              \
        MF_LAUNCH((rqs_layer_bwd_fused_kernel<KK, LL>), gf, FB_BLOCK, smem_f, stream, image, d, x, n, gy, glogp, gx,   \
                  gslab, accumulate, sp, bins);                                                                          \
        launched = true;                                                                                              \
    }
                MF_RQS_CASES(XF)
#undef XF
                if (launched) return check_launch("mf_flow_rqs_layer_bwd(fused)");
                return fail("no RQS kernel instance for bins=%d hidden_layers=%d (built: 2 <= bins <= 21, hidden_layers in {2,3})",
                            bins, hidden_layers);
            }
        }
    }
    const int64_t ntb = (n + 31) / 32;
    static const int bwd_block_env = [] { const char* e = getenv("MENTFLOW_BWD_BLOCK"); return e ? atoi(e) : 0; }();
    // small batches: one tile per SIMD on as many CUs as possible
    const int bwd_block = bwd_block_env == 256 || bwd_block_env == 512 ? bwd_block_env : (ntb <= 4 * NUM_CU ? 256 : 512);
#define X(KK, LL)                                                                                                     \
    if (!launched && rqs_case_matches(KK, bins) && hidden_layers == LL) {                                             \
        ProfScope prof(PK_FLOW_BWD, stream);                                                                          \
        if (bwd_block == 256) {                                                                                       \
            MF_ALLOW_DYN_SMEM((rqs_layer_bwd_kernel<KK, LL, 256>), smem);                                             \
            MF_LAUNCH((rqs_layer_bwd_kernel<KK, LL, 256>), flow_grid(n, 4), 256, smem, stream, image, d, x, n, gy,     \
                      glogp, gx, scratch, sp, bins);                                                  \
        } else {                                                                                                      \
            MF_ALLOW_DYN_SMEM((rqs_layer_bwd_kernel<KK, LL, 512>), smem);                                             \
            MF_LAUNCH((rqs_layer_bwd_kernel<KK, LL, 512>), flow_grid(n, 8), 512, smem, stream, image, d, x, n, gy,     \
                      glogp, gx, scratch, sp, bins);                                                  \
        }                                                                                                             \
        launched = true;                                                                                              \
    }
    MF_RQS_CASES(X)
#undef X
    if (!launched)
        return fail("no RQS kernel instance for bins=%d hidden_layers=%d (built: 2 <= bins <= 21, hidden_layers in {2,3})",
                    bins, hidden_layers);
    if (check_launch("mf_flow_rqs_layer_bwd")) return 1;
    const int nwaves = d > hidden_layers ? d : hidden_layers;
    if (nwaves > OA_MAX_WAVES) return fail("too many linear blocks for the gradient kernel");
    ProfScope prof(PK_OUTER_ACCUM, stream);
    MF_LAUNCH(outer_accum_kernel, dim3((unsigned)outer_accum_grid(n), 2), 64 * nwaves, 0, stream, (const float*)scratch, x, n,
              d, hidden_layers, d, gslab, accumulate, sp);
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

#define MF_AFFINE_CASES(X) X(3) X(2)

extern "C" int mf_flow_affine_layer_fwd(const float* image, int d, int hidden_layers, const int32_t* order, const float* x,
                                         int64_t n, float* y, const float* logp_in, float* logp_out, int init_logp,
                                         void* stream) {
    if (flow_check(d, hidden_layers, n)) return 1;
    if (n == 0) return 0;
    const Sparsity sp = make_sparsity(d, order, 1);
    const size_t smem = sizeof(float) * (size_t)image_layout(d, hidden_layers, 1).total;
#define X(LL)                                                                                                         \
    if (hidden_layers == LL) {                                                                                        \
        ProfScope prof(PK_FLOW_FWD, stream);                                                                          \
        MF_ALLOW_DYN_SMEM((affine_layer_fwd_kernel<LL, 1024>), smem);                                                 \
        MF_LAUNCH((affine_layer_fwd_kernel<LL, 1024>), flow_grid(n, 16), 1024, smem, stream, image, d, x, n, y, logp_in, \
                  logp_out, init_logp, sp);                                                                           \
        return check_launch("mf_flow_affine_layer_fwd");                                                              \
    }
    MF_AFFINE_CASES(X)
#undef X
    return fail("no affine kernel instance for hidden_layers=%d (built: 2, 3)", hidden_layers);
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
    const size_t smem = sizeof(float) * (size_t)image_layout(d, hidden_layers, 1).total;
    bool launched = false;
    if (affine_bwd_fused(n)) {
        const size_t smem_f = sizeof(float) * ((((size_t)image_layout(d, hidden_layers, 1).total + 3) & ~(size_t)3) + 8 * (size_t)FB_TILE);
        const int gf = fused_grid(n);
#define XF(LL)                                                                                                        \
    if (!launched && hidden_layers == LL) {                                                                           \
        ProfScope prof(PK_FLOW_BWD, stream);                                                                          \
        MF_ALLOW_DYN_SMEM((affine_layer_bwd_fused_kernel<LL>), smem_f);                                               \
        MF_LAUNCH((affine_layer_bwd_fused_kernel<LL>), gf, FB_BLOCK, smem_f, stream, image, d, x, n, gy, glogp, gx,    \
                  gslab, accumulate);                                                                                 \
        launched = true;                                                                                              \
    }
        MF_AFFINE_CASES(XF)
#undef XF
        if (launched) return check_launch("mf_flow_affine_layer_bwd(fused)");
        return fail("no affine kernel instance for hidden_layers=%d (built: 2, 3)", hidden_layers);
    }
#define X(LL)                                                                                                         \
    if (!launched && hidden_layers == LL) {                                                                           \
        ProfScope prof(PK_FLOW_BWD, stream);                                                                          \
        MF_ALLOW_DYN_SMEM((affine_layer_bwd_kernel<LL>), smem);                                                       \
        MF_LAUNCH((affine_layer_bwd_kernel<LL>), flow_grid(n), FLOW_BLOCK, smem, stream, image, d, x, n, gy, glogp, gx, \
                  scratch, sp);                                                                                       \
        launched = true;                                                                                              \
    }
    MF_AFFINE_CASES(X)
#undef X
    if (!launched) return fail("no affine kernel instance for hidden_layers=%d (built: 2, 3)", hidden_layers);
    if (check_launch("mf_flow_affine_layer_bwd")) return 1;
    ProfScope prof(PK_OUTER_ACCUM, stream);
    MF_LAUNCH(outer_accum_kernel, dim3((unsigned)outer_accum_grid(n), 2), 64 * hidden_layers, 0, stream, (const float*)scratch,
              x, n, d, hidden_layers, 1, gslab, accumulate, sp);
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
#define X(KK, LL)                                                                                                     \
    if (rqs_case_matches(KK, bins) && hidden_layers == LL) {                                                          \
        MF_ALLOW_DYN_SMEM((layer_inv_kernel<KK, LL>), smem);                                                          \
        MF_LAUNCH((layer_inv_kernel<KK, LL>), flow_grid(n, INV_BLOCK / 64), INV_BLOCK, smem, stream, image, d, y, n, x,  \
                  sp, io, bins);                                                                                            \
        return check_launch("mf_flow_rqs_layer_inv");                                                                 \
    }
    MF_RQS_CASES(X)
#undef X
    return fail("no RQS kernel instance for bins=%d hidden_layers=%d (built: 2 <= bins <= 21, hidden_layers in {2,3})", bins,
                hidden_layers);
}

extern "C" int mf_flow_affine_layer_inv(const float* image, int d, int hidden_layers, const int32_t* order, const float* y,
                                         int64_t n, float* x, void* stream) {
    if (flow_check(d, hidden_layers, n)) return 1;
    InvOrder io;
    if (inv_order(d, order, &io)) return 1;
    if (n == 0) return 0;
    const Sparsity sp = make_sparsity(d, order, 1);
    const size_t smem = sizeof(float) * ((size_t)image_layout(d, hidden_layers, 1).total + (INV_BLOCK / 64) * 32 * 8);
#define X(LL)                                                                                                         \
    if (hidden_layers == LL) {                                                                                        \
        MF_ALLOW_DYN_SMEM((layer_inv_kernel<0, LL>), smem);                                                           \
        MF_LAUNCH((layer_inv_kernel<0, LL>), flow_grid(n, INV_BLOCK / 64), INV_BLOCK, smem, stream, image, d, y, n, x, sp, \
                  io, 0);                                                                                                \
        return check_launch("mf_flow_affine_layer_inv");                                                              \
    }
    MF_AFFINE_CASES(X)
#undef X
    return fail("no affine kernel instance for hidden_layers=%d (built: 2, 3)", hidden_layers);
}

#if defined(MF_WS_DIAG) && !defined(MF_EMU)
extern "C" int mf_debug_ws_read(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(mf::g_ws_diag), sizeof(unsigned long long) * NUM_CU * 4 * 16) == hipSuccess ? 0 : 1;
}
#endif
