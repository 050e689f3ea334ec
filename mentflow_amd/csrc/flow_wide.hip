// Wide conditioners: the flow layers for hidden widths 65 .. 128 and / or 8 .. 16 features — configurations whose weights do not
// fit one CU's LDS (160 KiB; the last layer of a 128-wide conditioner at d = 6 alone is 198 KB), which is what the tuned 64-wide
// kernels of flow_kernels.inc are built around.  mentflow/generate/build.py:36-38 takes hidden_units from the config
// (config/gen/flow.yaml:3 sets 64) and zuko accepts any width.
//
// Same computation, same wave <-> particle-tile map, same spline code (rqs_apply) as the 64-wide kernels; what differs:
//   * the weights stay in global memory, packed on the host in FRAGMENT ORDER: one 32 x 32 block of a weight matrix
//     (16 k-steps of v_mfma_f32_32x32x2_f32) is 4 KiB laid out so that lane l reads the four A operands of k-steps 4g .. 4g+3 as
//     ONE 16-byte load at block + 16 (64 g + l): a wave instruction moves 1 KiB of consecutive bytes and every byte fetched is
//     used.  The whole image of a layer (683 KB at 128 units, d = 6, three hidden layers) is L2-resident (4 MiB per XCD) and is
//     re-read by every wave: 16 B / clk / CU at the rate the matrix pipe consumes it, a quarter of the L1's 64 B / clk.  The
//     transposed products of the backward read a second, transposed-fragment copy of every matrix from the same image.
//   * mask sparsity at TILE granularity: hidden units are placed sorted by dependency class (packing.py), so the
//     autoregressive masks are block-triangular over 32-unit tiles; wave-uniform tile counts skip the all-zero blocks
//     (13 of 16 hidden blocks and 14 of 24 last-layer blocks remain at 128 units, d = 6).
//   * the backward is the two-kernel form: a per-tile kernel runs the adjoint, produces dL/dx and writes the pre-activation
//     gradients to an HBM scratch; wide_outer_accum_kernel contracts them with the activations over particles into slab rows of a
//     NATURAL-layout gradient image (mf_flow_wide_grad_floats), summed by mf_flow_grad_reduce.  The activations come from the
//     training forward (hand-off through HBM: hidden tiles in the scratch layout, conditioner outputs, ReLU sign bits —
//     mf_flow_wide_layer_fwd_save / _bwd_saved) or, without it, are recomputed and written by the per-tile kernel.
//   * hidden_layers is a run-time argument (1 .. 4).
#include "flow_launch.h"
#include <stdlib.h>

namespace mf {

constexpr int WIDE_HT = 4;                   // hidden tiles of 32 units
constexpr int WIDE_HP = 32 * WIDE_HT;        // padded hidden width
constexpr int WIDE_DMAX = 16;                // features: 8 k-steps of the input layer, 16 rows of the dL/dx tile
constexpr int WIDE_LMAX = 4;
constexpr int WIDE_FB = 1024;                // floats of one 32 x 32 fragment block
constexpr int WIDE_TS = WIDE_HT * 1024;      // scratch floats of one hidden tile row (32 particles x 128 columns)
constexpr int WIDE_BLOCK = 256;              // 4 waves per workgroup; two workgroups per CU (<= 256 registers per lane)
constexpr int WIDE_MAXJOBS = 48;

typedef float wf4 __attribute__((ext_vector_type(4)));

// The weight image is read through pointers that are (a) typed as GLOBAL memory — global_load with a scalar base; a generic
// pointer would make every fragment load a flat_load on a per-lane 64-bit address — and (b) re-derived per particle tile from a
// wave-uniform value the optimiser cannot see through (MF_OPAQUE_BASE), so that the block addresses are formed by scalar adds
// next to their loads instead of being hoisted out of the tile loop as ~200 loop-invariant per-lane pointer pairs, and spilled.
#ifdef MF_EMU
#define MF_GLOBAL
#define MF_OPAQUE_BASE(u) asm volatile("" : "+r"(u))
#else
#define MF_GLOBAL __attribute__((address_space(1)))
#define MF_OPAQUE_BASE(u) asm volatile("" : "+s"(u))
#endif
typedef const MF_GLOBAL float* gfp;

// ---- weight image (fragment order) -------------------------------------------------------------------------------------
//   W0F [HT][2][256]      input layer, forward:  lane l, group gx, j -> W0[32 rt + col][2 (4 gx + j) + hh]
//   B0  [HT][2][16]       bias fragments:        (rt, hh, r) -> b0[32 rt + rowmap(r, hh)]
//   W0T [HT][1024]        input layer, transposed (k = hidden tile kt): (g, l, j) -> W0[32 kt + rowmap(4 g + j, hh)][col]
//   per hidden layer l = 1 .. L-1:   F [HT out][HT in][1024] | T [HT in][HT out][1024] | B [HT][2][16]
//       F block (rt, it): (g, l, j) -> W[32 rt + col][32 it + rowmap(4 g + j, hh)]
//       T block (it, kt): (g, l, j) -> W[32 kt + rowmap(4 g + j, hh)][32 it + col]
//   per output block (RQS: one per feature, 64 permuted rows; affine: one):   F [2][HT][1024] | T [HT][2][1024] | B [2][2][16]
struct WideLayout {
    int offW0F, offB0, offW0T, offH, strideH, off3, stride3, total;
};
__host__ __device__ inline WideLayout wide_layout(int L, int nblk) {
    WideLayout g;
    g.offW0F = 0;
    g.offB0 = g.offW0F + WIDE_HT * 512;
    g.offW0T = g.offB0 + WIDE_HT * 32;
    g.offH = g.offW0T + WIDE_HT * WIDE_FB;
    g.strideH = 2 * WIDE_HT * WIDE_HT * WIDE_FB + WIDE_HT * 32;
    g.off3 = g.offH + (L - 1) * g.strideH;
    g.stride3 = 2 * WIDE_HT * WIDE_FB + WIDE_HT * 2 * WIDE_FB + 64;
    g.total = g.off3 + nblk * g.stride3;
    return g;
}
// ---- gradient image (natural order, physical rows / columns) ----------------------------------------------------------
//   gW0 [HP][16] | gb0 [HP] | per hidden layer: gW [HP][HP] | gb [HP] | per output block: gW [64][HP] | gb [64]
struct WideGradLayout {
    int offW0, offB0, offH, strideH, off3, stride3, total;
};
__host__ __device__ inline WideGradLayout wide_grad_layout(int L, int nblk) {
    WideGradLayout g;
    g.offW0 = 0;
    g.offB0 = WIDE_HP * WIDE_DMAX;
    g.offH = g.offB0 + WIDE_HP;
    g.strideH = WIDE_HP * WIDE_HP + WIDE_HP;
    g.off3 = g.offH + (L - 1) * g.strideH;
    g.stride3 = 64 * WIDE_HP + 64;
    g.total = g.off3 + nblk * g.stride3;
    return g;
}

// tile-granular mask structure (wave-uniform loop bounds)
struct WideSp {
    int ht;                     // hidden tiles in use: ceil(width / 32)
    int nin_h[WIDE_HT];         // hidden -> hidden, output tile rt: input tiles [0, nin_h[rt])
    int kbeg_t[WIDE_HT];        // transposed hidden -> hidden, output (= input-unit) tile it: k tiles [kbeg_t[it], ht)
    int nin3[WIDE_DMAX];        // last layer, block i: input tiles [0, nin3[i])   (0: the block is pure bias)
};

static WideSp make_wide_sp(int d, const int32_t* order, int width, int nblk) {
    WideSp sp;
    sp.ht = (width + 31) / 32;
    for (int t = 0; t < WIDE_HT; ++t) {
        sp.nin_h[t] = t < sp.ht ? sp.ht : 0;
        sp.kbeg_t[t] = 0;
    }
    for (int i = 0; i < WIDE_DMAX; ++i) sp.nin3[i] = sp.ht;
    if (order == nullptr || d < 2) return sp;                         // dense
    int cum[WIDE_DMAX + 2];                                           // cum[c] = hidden units of dependency class <= c
    for (int c = 0; c <= d; ++c) {
        int cnt = 0;
        for (int u = 0; u < width; ++u) cnt += (1 + u % (d - 1)) <= c;
        cum[c] = cnt;
    }
    auto class_of = [&](int j) { int c = 1; while (cum[c] <= j) ++c; return c; };
    for (int rt = 0; rt < sp.ht; ++rt) {
        const int last = 32 * rt + 31 < width ? 32 * rt + 31 : width - 1;
        sp.nin_h[rt] = (cum[class_of(last)] + 31) / 32;
        sp.kbeg_t[rt] = cum[class_of(32 * rt) - 1] / 32;
    }
    if (nblk == d)
        for (int i = 0; i < d; ++i) sp.nin3[i] = (cum[order[i]] + 31) / 32;
    return sp;
}

// ---- fragment helpers ---------------------------------------------------------------------------------------------------
// acc += (one 32 x 32 fragment block) x (16 k-steps of B): four 16-byte loads, sixteen MFMAs
struct TileB {
    const f32x16_t& t;
    __device__ __forceinline__ float operator[](int i) const { return t[i]; }
};
template <int BASE>
struct SlotB {
    const float (&v)[32];
    __device__ __forceinline__ float operator[](int i) const { return v[BASE + i]; }
};
// Addressing: the block base is wave-uniform (scalar registers), the lane's part is ONE 32-bit byte offset shared by every
// block (global_load_dwordx4 v, v_off, s[base:base+1]).  Written with per-lane 64-bit pointers, the compiler hoists one
// loop-invariant pointer pair per 4 KiB block out of the particle-tile loop — some 200 of them in the backward — and spills.
struct Frag {                 // the A operands of one fragment block: 16 k-steps
    wf4 a[4];
};
__device__ __forceinline__ void frag_load(Frag& f, gfp blk, int lane) {
    const MF_GLOBAL char* q = (const MF_GLOBAL char*)blk;
    const unsigned off = (unsigned)lane * 16u;
#pragma unroll
    for (int g = 0; g < 4; ++g) f.a[g] = *(const MF_GLOBAL wf4*)(q + (off + 1024u * g));
}
template <class B>
__device__ __forceinline__ void frag_mfma(f32x16_t& acc, const Frag& f, const B& b) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = mfma(f.a[g][j], b[4 * g + j], acc);
}
// two blocks that are neighbours in the image (consecutive input tiles of one output tile / consecutive k tiles): 8 KiB
struct Frag2 {
    wf4 a[8];
};
__device__ __forceinline__ void frag2_load(Frag2& f, gfp blk, int lane) {
    const MF_GLOBAL char* q = (const MF_GLOBAL char*)blk;
    const unsigned off = (unsigned)lane * 16u;
#pragma unroll
    for (int g = 0; g < 8; ++g) f.a[g] = *(const MF_GLOBAL wf4*)(q + (off + 1024u * g));
}
template <int HALF, class B>
__device__ __forceinline__ void frag2_mfma(f32x16_t& acc, const Frag2& f, const B& b) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = mfma(f.a[4 * HALF + g][j], b[4 * g + j], acc);
}
// One block at a time (load, wait, 16 MFMAs): only where a single block is needed.  Chains of blocks go through the rolling
// prefetch below, in PAIRS of neighbouring blocks: the fragments of pair k + 1 (8 KiB) are requested before the MFMAs of pair k
// (up to 2 048 cycles of matrix pipe against an L2 round trip that exceeds the 1 024 cycles of a single block when 256 CUs read
// the same image), across the output tiles of a layer — the mask bounds are wave-uniform, so "the next active pair" is known when
// the current one starts; the second block of a pair is fetched even when the masks switch its MFMAs off (zeros, never used).
// -DMF_WIDE_NO_PREFETCH builds the plain form (A/B runs).
template <class B>
__device__ __forceinline__ void wide_block(f32x16_t& acc, gfp blk, int lane, const B& b) {
    Frag f;
    frag_load(f, blk, lane);
    frag_mfma(acc, f, b);
}

__device__ __forceinline__ f32x16_t wide_bias(gfp B, int rt, int hh) {
    const MF_GLOBAL wf4* q = (const MF_GLOBAL wf4*)(B + (rt * 2 + hh) * 16);
    f32x16_t acc;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const wf4 t = q[g];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[4 * g + j] = t[j];
    }
    return acc;
}

__device__ __forceinline__ f32x16_t wide_zero() {
    f32x16_t z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.0f;
    return z;
}

__device__ __forceinline__ void wide_relu(f32x16_t& h) {
#pragma unroll
    for (int r = 0; r < 16; ++r) h[r] = relu1(h[r]);
}

// h = relu(W0 x + b0);  xb[s] = x[2 s + hh] (0 beyond d)
__device__ __forceinline__ void wide_input(gfp img, const WideLayout& g, int d, const float (&xb)[8],
                                           f32x16_t (&h)[WIDE_HT], int lane, int hh, int ht) {
#pragma unroll
    for (int rt = 0; rt < WIDE_HT; ++rt) {
        if (rt < ht) {
            f32x16_t acc = wide_bias(img + g.offB0, rt, hh);
            const MF_GLOBAL char* q = (const MF_GLOBAL char*)(img + g.offW0F + rt * 512);
            const unsigned off = (unsigned)lane * 16u;
            const wf4 a0 = *(const MF_GLOBAL wf4*)(q + off);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = mfma(a0[j], xb[j], acc);
            if (d > 8) {
                const wf4 a1 = *(const MF_GLOBAL wf4*)(q + (off + 1024u));
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = mfma(a1[j], xb[4 + j], acc);
            }
            wide_relu(acc);
            h[rt] = acc;
        } else {
            h[rt] = wide_zero();
        }
    }
}

// out = relu(W in + b) of one hidden layer (H = that layer's F | T | B)
template <bool PAIR>
__device__ __forceinline__ void wide_hidden(gfp H, const f32x16_t (&in)[WIDE_HT], f32x16_t (&out)[WIDE_HT],
                                            int lane, int hh, const WideSp& sp) {
    const gfp B = H + 2 * WIDE_HT * WIDE_HT * WIDE_FB;
#ifdef MF_WIDE_NO_PREFETCH
#pragma unroll
    for (int rt = 0; rt < WIDE_HT; ++rt) {
        if (rt < sp.ht) {
            f32x16_t acc = wide_bias(B, rt, hh);
#pragma unroll
            for (int it = 0; it < WIDE_HT; ++it)
                if (it < sp.nin_h[rt]) wide_block(acc, H + (rt * WIDE_HT + it) * WIDE_FB, lane, TileB{in[it]});
            wide_relu(acc);
            out[rt] = acc;
        } else {
            out[rt] = wide_zero();
        }
    }
#else
    if constexpr (PAIR) {
        Frag2 cur, nxt;
        frag2_load(cur, H, lane);                                 // pair (0, 0): every tile in use has at least one input tile
#pragma unroll
        for (int rt = 0; rt < WIDE_HT; ++rt) {
            if (rt < sp.ht) {
                f32x16_t acc = wide_bias(B, rt, hh);
#pragma unroll
                for (int pr = 0; pr < WIDE_HT / 2; ++pr)
                    if (2 * pr < sp.nin_h[rt]) {
                        if (pr + 1 < WIDE_HT / 2 && 2 * (pr + 1) < sp.nin_h[rt]) frag2_load(nxt, H + (rt * WIDE_HT + 2 * (pr + 1)) * WIDE_FB, lane);
                        else if (rt + 1 < WIDE_HT && rt + 1 < sp.ht) frag2_load(nxt, H + (rt + 1) * WIDE_HT * WIDE_FB, lane);
                        frag2_mfma<0>(acc, cur, TileB{in[2 * pr]});
                        if (2 * pr + 1 < sp.nin_h[rt]) frag2_mfma<1>(acc, cur, TileB{in[2 * pr + 1]});
                        cur = nxt;
                    }
                wide_relu(acc);
                out[rt] = acc;
            } else {
                out[rt] = wide_zero();
            }
    }
    } else {
    Frag cur, nxt;
    frag_load(cur, H, lane);                                  // block (0, 0): every tile in use has at least one input tile
#pragma unroll
    for (int rt = 0; rt < WIDE_HT; ++rt) {
        if (rt < sp.ht) {
            f32x16_t acc = wide_bias(B, rt, hh);
#pragma unroll
            for (int it = 0; it < WIDE_HT; ++it)
                if (it < sp.nin_h[rt]) {
                    if (it + 1 < WIDE_HT && it + 1 < sp.nin_h[rt]) frag_load(nxt, H + (rt * WIDE_HT + it + 1) * WIDE_FB, lane);
                    else if (rt + 1 < WIDE_HT && rt + 1 < sp.ht) frag_load(nxt, H + (rt + 1) * WIDE_HT * WIDE_FB, lane);
                    frag_mfma(acc, cur, TileB{in[it]});
                    cur = nxt;
                }
            wide_relu(acc);
            out[rt] = acc;
        } else {
            out[rt] = wide_zero();
        }
    }
    }
#endif
}

// v = (output block) h + b : NRT row tiles of the block (2: the 64 spline slots of both halves; 1: the affine block)
template <int NRT, bool PAIR>
__device__ __forceinline__ void wide_out_block(gfp blk, const f32x16_t (&h)[WIDE_HT], float (&v)[32], int lane,
                                               int hh, int nin) {
    const gfp B = blk + 4 * WIDE_HT * WIDE_FB;
#ifdef MF_WIDE_NO_PREFETCH
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        if (rt < NRT) {
            f32x16_t acc = wide_bias(B, rt, hh);
#pragma unroll
            for (int it = 0; it < WIDE_HT; ++it)
                if (it < nin) wide_block(acc, blk + (rt * WIDE_HT + it) * WIDE_FB, lane, TileB{h[it]});
#pragma unroll
            for (int r = 0; r < 16; ++r) v[16 * rt + r] = acc[r];
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[16 * rt + r] = 0.0f;
        }
    }
#else
    if constexpr (PAIR) {
        Frag2 cur, nxt;
        if (nin > 0) frag2_load(cur, blk, lane);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            if (rt < NRT) {
                f32x16_t acc = wide_bias(B, rt, hh);
#pragma unroll
                for (int pr = 0; pr < WIDE_HT / 2; ++pr)
                    if (2 * pr < nin) {
                        if (pr + 1 < WIDE_HT / 2 && 2 * (pr + 1) < nin) frag2_load(nxt, blk + (rt * WIDE_HT + 2 * (pr + 1)) * WIDE_FB, lane);
                        else if (rt + 1 < NRT) frag2_load(nxt, blk + (rt + 1) * WIDE_HT * WIDE_FB, lane);
                        frag2_mfma<0>(acc, cur, TileB{h[2 * pr]});
                        if (2 * pr + 1 < nin) frag2_mfma<1>(acc, cur, TileB{h[2 * pr + 1]});
                        cur = nxt;
                    }
#pragma unroll
                for (int r = 0; r < 16; ++r) v[16 * rt + r] = acc[r];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[16 * rt + r] = 0.0f;
            }
    }
    } else {
    Frag cur, nxt;
    if (nin > 0) frag_load(cur, blk, lane);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        if (rt < NRT) {
            f32x16_t acc = wide_bias(B, rt, hh);
#pragma unroll
            for (int it = 0; it < WIDE_HT; ++it)
                if (it < nin) {
                    if (it + 1 < WIDE_HT && it + 1 < nin) frag_load(nxt, blk + (rt * WIDE_HT + it + 1) * WIDE_FB, lane);
                    else if (rt + 1 < NRT) frag_load(nxt, blk + (rt + 1) * WIDE_HT * WIDE_FB, lane);
                    frag_mfma(acc, cur, TileB{h[it]});
                    cur = nxt;
                }
#pragma unroll
            for (int r = 0; r < 16; ++r) v[16 * rt + r] = acc[r];
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[16 * rt + r] = 0.0f;
        }
    }
    }
#endif
}

// the affine transform's parameters of feature i out of the block's slots: slot i of half 0 = shift_i, of half 1 = scale_i
__device__ __forceinline__ void wide_affine_params(const float (&v)[32], int i, int hh, float& shift, float& scale) {
    float mine = 0.0f;
#pragma unroll
    for (int j = 0; j < WIDE_DMAX; ++j) mine = (j == i) ? v[j] : mine;
    half_pair(mine, hh, shift, scale);
}

// ------------------------------------------------------------------------------------------------------------
// scratch tiles:  X[tile][c][particle], column c = 32 rt + 16 hh + r  <->  accumulator register r of row tile rt of lane half hh
// (the layout of the 64-wide two-kernel path with more columns):  ACT[L][ntiles][HT*1024] | GPRE[L][ntiles][HT*1024] |
// GPHI[nblk][ntiles][2048]
__device__ __forceinline__ void wide_store(float* __restrict__ dst, int rt, int col, int hh, const f32x16_t& a) {
    float* base = dst + (32 * rt + 16 * hh) * 32 + col;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float t = a[r];
#ifdef MF_WIDE_PLAIN_STORES
        base[r * 32] = t;
#else
        MF_ACT_ST(base + r * 32, t);           // written once, read once by another kernel: non-temporal (26.6 vs 26.9 ms per step)
#endif
    }
}

// sign bits of a post-ReLU tile (bit r = register r is positive) and their use as the ReLU mask of a gradient tile
__device__ __forceinline__ unsigned wide_bits(const f32x16_t& h) {
    unsigned b = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float t = h[r];
        b |= (t > 0.0f ? 1u : 0u) << r;
    }
    return b;
}
__device__ __forceinline__ void wide_mask(f32x16_t& gh, unsigned bits) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float t = gh[r];
        gh[r] = ((bits >> r) & 1u) ? t : 0.0f;
    }
}

// Activation hand-off of the wide family (forward -> backward through HBM, as §4.2 of DESIGN.md does for the 64-wide kernels): a
// training forward stores, per layer,  ACTH[L][ntiles][HT*1024]  — every hidden level in the SCRATCH-TILE layout, so that the
// contraction kernel reads its B operands straight from it and the backward neither recomputes nor re-writes them — and
// PHI[nblk][ntiles][2048]  — the conditioner outputs of every output block in the register layout of the wave (float4[8][lane]).
// 3 KB per particle and layer at 128 units, d = 6, three hidden layers.
// ... and  BITS[L][ntiles][HT][64]  — the sign bits of every hidden tile (bit r of lane l's word = register r is positive): the
// backward needs the hidden values themselves only as ReLU masks, 12 words per lane instead of 192 loads.
__host__ __device__ inline int64_t wide_act_floats(int64_t ntiles, int L, int nblk) {
    return ntiles * ((int64_t)L * WIDE_TS + (int64_t)nblk * 2048 + (int64_t)L * WIDE_HT * 64);
}
__device__ __forceinline__ void wide_phi_store(float* __restrict__ blk, int lane, const float (&v)[32]) {
    act_f4* q = reinterpret_cast<act_f4*>(blk) + lane;
#pragma unroll
    for (int g4 = 0; g4 < 8; ++g4) {
        const act_f4 t = {v[4 * g4], v[4 * g4 + 1], v[4 * g4 + 2], v[4 * g4 + 3]};
        MF_ACT_ST(q + g4 * 64, t);
    }
}
__device__ __forceinline__ void wide_phi_load(const float* __restrict__ blk, int lane, float (&v)[32]) {
    const act_f4* q = reinterpret_cast<const act_f4*>(blk) + lane;
#pragma unroll
    for (int g4 = 0; g4 < 8; ++g4) {
        const act_f4 t = MF_ACT_LD(q + g4 * 64);
        v[4 * g4] = t.x;
        v[4 * g4 + 1] = t.y;
        v[4 * g4 + 2] = t.z;
        v[4 * g4 + 3] = t.w;
    }
}

// =========================================================================================== forward
template <int K>      // K > 0 or RQS_ANY: rational-quadratic spline;  K == 0: affine
__global__ __launch_bounds__(WIDE_BLOCK) MF_WAVES_PER_SIMD(2, 2) void wide_fwd_kernel(
    const float* __restrict__ image_arg, int d, int L, const float* __restrict__ x, int64_t n, float* __restrict__ y,
    const float* __restrict__ logp_in, float* __restrict__ logp_out, int init_logp, WideSp sp, int bins_rt,
    float* __restrict__ act) {          // act != nullptr: training forward, hands the activations over (wide_act_floats)
    const int nblk = (K != 0) ? d : 1;
    const WideLayout g = wide_layout(L, nblk);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, col = lane & 31, hh = lane >> 5;
    const int64_t ntiles = (n + 31) / 32;
    float* const PHI = act + (int64_t)L * ntiles * WIDE_TS;
    unsigned* const BITS = reinterpret_cast<unsigned*>(PHI + (int64_t)nblk * ntiles * 2048);
    for (int64_t tile = (int64_t)blockIdx.x * (WIDE_BLOCK / 64) + wid; tile < ntiles; tile += (int64_t)gridDim.x * (WIDE_BLOCK / 64)) {
        MF_NO_HOIST();                 // the image is loop invariant: keep its loads next to their MFMAs (flow_kernels.inc)
        uint64_t image_u = (uint64_t)image_arg;
        MF_OPAQUE_BASE(image_u);       // ... and its block addresses next to the loads (scalar adds), see MF_GLOBAL
        const gfp image = (gfp)image_u;
        const int64_t p = tile * 32 + col;
        const bool valid = p < n;
        const float* xp = x + (valid ? p : n - 1) * d;
        float xb[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) xb[s] = (2 * s + hh < d) ? xp[2 * s + hh] : 0.0f;
        f32x16_t h[WIDE_HT];
        wide_input(image, g, d, xb, h, lane, hh, sp.ht);
        if (act != nullptr) {
#pragma unroll
            for (int rt = 0; rt < WIDE_HT; ++rt) {
                wide_store(act + tile * WIDE_TS, rt, col, hh, h[rt]);
                BITS[(tile * WIDE_HT + rt) * 64 + lane] = wide_bits(h[rt]);
            }
        }
#pragma unroll 1
        for (int l = 1; l < L; ++l) {
            f32x16_t t[WIDE_HT];
            wide_hidden<false>(image + g.offH + (l - 1) * g.strideH, h, t, lane, hh, sp);
#pragma unroll
            for (int rt = 0; rt < WIDE_HT; ++rt) h[rt] = t[rt];
            if (act != nullptr) {
#pragma unroll
                for (int rt = 0; rt < WIDE_HT; ++rt) {
                    wide_store(act + ((int64_t)l * ntiles + tile) * WIDE_TS, rt, col, hh, h[rt]);
                    BITS[(((int64_t)l * ntiles + tile) * WIDE_HT + rt) * 64 + lane] = wide_bits(h[rt]);
                }
            }
        }
        float ladj = 0.0f;
        if constexpr (K != 0) {
#pragma unroll 1
            for (int i = 0; i < d; ++i) {
                float v[32], gdummy[32];
                wide_out_block<2, false>(image + g.off3 + i * g.stride3, h, v, lane, hh, sp.nin3[i]);
                if (act != nullptr) wide_phi_store(PHI + ((int64_t)i * ntiles + tile) * 2048, lane, v);
                float yi, li, gxd;
                rqs_apply<K, 0>(v, xp[i], hh, yi, li, 0.0f, 0.0f, gdummy, gxd, bins_rt);
                ladj += li;
                if (valid && hh == 0) y[p * d + i] = yi;
            }
        } else {
            float v[32];
            wide_out_block<1, false>(image + g.off3, h, v, lane, hh, sp.ht);
            if (act != nullptr) wide_phi_store(PHI + tile * 2048, lane, v);
#pragma unroll 1
            for (int i = 0; i < d; ++i) {
                float shift, scale;
                wide_affine_params(v, i, hh, shift, scale);
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

// =========================================================================================== backward
template <int K, bool SAVED>      // SAVED: the forward handed its activations over (act): nothing is recomputed
#ifndef MF_WIDE_BWD_WAVES
#define MF_WIDE_BWD_WAVES 1            // one wave per SIMD: ~400 registers (the level-L-1 tile, dL/dh, the spline's adjoint state)
#endif
__global__ __launch_bounds__(WIDE_BLOCK) MF_WAVES_PER_SIMD(MF_WIDE_BWD_WAVES, MF_WIDE_BWD_WAVES) void wide_bwd_kernel(
    const float* __restrict__ image_arg, int d, int L, const float* __restrict__ x, int64_t n, const float* __restrict__ gy,
    const float* __restrict__ glogp, float* __restrict__ gx, float* __restrict__ scratch, WideSp sp, int bins_rt,
    const float* __restrict__ act) {
    const int nblk = (K != 0) ? d : 1;
    const WideLayout g = wide_layout(L, nblk);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, col = lane & 31, hh = lane >> 5;
    const int64_t ntiles = (n + 31) / 32;
    float* ACT = scratch;
    float* GPRE = ACT + (int64_t)L * ntiles * WIDE_TS;
    float* GPHI = GPRE + (int64_t)L * ntiles * WIDE_TS;
    const float* const PHI = act + (int64_t)L * ntiles * WIDE_TS;      // (SAVED only)
    const unsigned* const BITS = reinterpret_cast<const unsigned*>(PHI + (int64_t)nblk * ntiles * 2048);
    for (int64_t tile = (int64_t)blockIdx.x * (WIDE_BLOCK / 64) + wid; tile < ntiles; tile += (int64_t)gridDim.x * (WIDE_BLOCK / 64)) {
        MF_NO_HOIST();                 // the image is loop invariant: keep its loads next to their MFMAs (flow_kernels.inc)
        uint64_t image_u = (uint64_t)image_arg;
        MF_OPAQUE_BASE(image_u);       // ... and its block addresses next to the loads (scalar adds), see MF_GLOBAL
        const gfp image = (gfp)image_u;
        const int64_t p = tile * 32 + col;
        const bool valid = p < n;
        const int64_t pc = valid ? p : n - 1;
        const float* xp = x + pc * d;
        float xb[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) xb[s] = (2 * s + hh < d) ? xp[2 * s + hh] : 0.0f;
        unsigned bits[WIDE_LMAX][WIDE_HT];
        f32x16_t h[WIDE_HT];
        if constexpr (SAVED) {
            // ---- of the hidden tiles the backward itself only needs the sign bits (the ReLU masks): the contraction reads the
            // values from the hand-off buffer on its own, and the conditioner outputs are handed over as well
#pragma unroll
            for (int l = 0; l < WIDE_LMAX; ++l)
#pragma unroll
                for (int rt = 0; rt < WIDE_HT; ++rt)
                    bits[l][rt] = (l < L) ? BITS[(((int64_t)l * ntiles + tile) * WIDE_HT + rt) * 64 + lane] : 0u;
        } else {
            // ---- recompute the trunk: every activation goes to the scratch, its sign bits stay in registers
            wide_input(image, g, d, xb, h, lane, hh, sp.ht);
#pragma unroll
            for (int rt = 0; rt < WIDE_HT; ++rt) {
                wide_store(ACT + tile * WIDE_TS, rt, col, hh, h[rt]);
                bits[0][rt] = wide_bits(h[rt]);
            }
#pragma unroll
            for (int l = 1; l < WIDE_LMAX; ++l) {
                if (l < L) {
                    f32x16_t t[WIDE_HT];
                    wide_hidden<true>(image + g.offH + (l - 1) * g.strideH, h, t, lane, hh, sp);
#pragma unroll
                    for (int rt = 0; rt < WIDE_HT; ++rt) {
                        h[rt] = t[rt];
                        wide_store(ACT + ((int64_t)l * ntiles + tile) * WIDE_TS, rt, col, hh, h[rt]);
                        bits[l][rt] = wide_bits(h[rt]);
                    }
                } else {
#pragma unroll
                    for (int rt = 0; rt < WIDE_HT; ++rt) bits[l][rt] = 0;
                }
            }
        }
        // ---- output blocks: transform forward + adjoint, accumulate dL/dh of the last hidden level
        f32x16_t gh[WIDE_HT];
#pragma unroll
        for (int rt = 0; rt < WIDE_HT; ++rt) gh[rt] = wide_zero();
        f32x16_t gacc = wide_zero();                     // dL/dx tile: row = feature
        const float gl = valid ? -glogp[pc] : 0.0f;
        if constexpr (K != 0) {
#pragma unroll 1
            for (int i = 0; i < d; ++i) {
                float v[32], gv[32];
                const gfp blk = image + g.off3 + i * g.stride3;
                const int nin = sp.nin3[i];
                if constexpr (SAVED) wide_phi_load(PHI + ((int64_t)i * ntiles + tile) * 2048, lane, v);
                else wide_out_block<2, true>(blk, h, v, lane, hh, nin);
                const float gyi = valid ? gy[pc * d + i] : 0.0f;
                float yi, li, gxd;
                rqs_apply<K, 1>(v, xp[i], hh, yi, li, gyi, gl, gv, gxd, bins_rt);
#pragma unroll
                for (int j = 0; j < 8; ++j) gacc[j] += (rowmap(j, hh) == i) ? gxd : 0.0f;
                float* gp = GPHI + ((int64_t)i * ntiles + tile) * 2048 + (16 * hh) * 32 + col;
#pragma unroll
                for (int m = 0; m < 32; ++m) gp[(32 * (m >> 4) + (m & 15)) * 32] = gv[m];
                // gh += W3_i^T gphi  (contraction over the 64 padded rows of the block = the 32 slots of both halves)
                const gfp T = blk + 2 * WIDE_HT * WIDE_FB;
#ifdef MF_WIDE_NO_PREFETCH
#pragma unroll
                for (int it = 0; it < WIDE_HT; ++it)
                    if (it < nin) {
                        wide_block(gh[it], T + (it * 2 + 0) * WIDE_FB, lane, SlotB<0>{gv});
                        wide_block(gh[it], T + (it * 2 + 1) * WIDE_FB, lane, SlotB<16>{gv});
                    }
#else
                // (single blocks here: a pair in flight on top of the spline's adjoint state spills 50 registers — 15.6 vs 16.0 ms per
                // step; requesting the NEXT output block's first fragments ahead of this chain measured nothing: 15.7 vs 15.6)
                Frag cur, nxt;
                if (nin > 0) frag_load(cur, T, lane);
#pragma unroll
                for (int it = 0; it < WIDE_HT; ++it)
                    if (it < nin) {
                        frag_load(nxt, T + (it * 2 + 1) * WIDE_FB, lane);
                        frag_mfma(gh[it], cur, SlotB<0>{gv});
                        cur = nxt;
                        if (it + 1 < WIDE_HT && it + 1 < nin) frag_load(nxt, T + (it + 1) * 2 * WIDE_FB, lane);
                        frag_mfma(gh[it], cur, SlotB<16>{gv});
                        cur = nxt;
                    }
#endif
            }
        } else {
            float v[32], gv[32];
            const gfp blk = image + g.off3;
            if constexpr (SAVED) wide_phi_load(PHI + tile * 2048, lane, v);
            else wide_out_block<1, true>(blk, h, v, lane, hh, sp.ht);
#pragma unroll
            for (int m = 0; m < 32; ++m) gv[m] = 0.0f;
#pragma unroll
            for (int i = 0; i < WIDE_DMAX; ++i) {
                if (i < d) {
                    float shift, scale;
                    half_pair(v[i], hh, shift, scale);
                    const float ls = soft_clip(scale, LOG_SLOPE_INV);
                    const float e = fast_exp(ls);
                    const float gyi = valid ? gy[pc * d + i] : 0.0f;
                    const float gls = fmaf(gyi * xp[i], e, gl);               // dL/ds~ : through y and through ladj
                    gv[i] = hh ? gls * soft_clip_grad(scale, LOG_SLOPE_INV) : gyi;
                    const float gxd = gyi * e;
#pragma unroll
                    for (int j = 0; j < 8; ++j) gacc[j] += (rowmap(j, hh) == i) ? gxd : 0.0f;
                }
            }
            float* gp = GPHI + tile * 2048 + (16 * hh) * 32 + col;
#pragma unroll
            for (int m = 0; m < 32; ++m) gp[(32 * (m >> 4) + (m & 15)) * 32] = gv[m];
            const gfp T = blk + 2 * WIDE_HT * WIDE_FB;
#pragma unroll
            for (int it = 0; it < WIDE_HT; ++it)
                if (it < sp.ht) wide_block(gh[it], T + (it * 2 + 0) * WIDE_FB, lane, SlotB<0>{gv});
        }
        // ---- trunk backward
#pragma unroll
        for (int l = WIDE_LMAX - 1; l >= 1; --l) {
            if (l < L) {
#pragma unroll
                for (int rt = 0; rt < WIDE_HT; ++rt) {
                    wide_mask(gh[rt], bits[l][rt]);
                    wide_store(GPRE + ((int64_t)l * ntiles + tile) * WIDE_TS, rt, col, hh, gh[rt]);
                }
                const gfp T = image + g.offH + (l - 1) * g.strideH + WIDE_HT * WIDE_HT * WIDE_FB;
                f32x16_t t[WIDE_HT];
#ifdef MF_WIDE_NO_PREFETCH
#pragma unroll
                for (int it = 0; it < WIDE_HT; ++it) {
                    t[it] = wide_zero();
                    if (it < sp.ht) {
#pragma unroll
                        for (int kt = 0; kt < WIDE_HT; ++kt)
                            if (kt >= sp.kbeg_t[it] && kt < sp.ht) wide_block(t[it], T + (it * WIDE_HT + kt) * WIDE_FB, lane, TileB{gh[kt]});
                    }
                }
#else
                Frag2 cur, nxt;
                frag2_load(cur, T, lane);                         // pair (0, 0): kbeg_t[0] = 0
#pragma unroll
                for (int it = 0; it < WIDE_HT; ++it) {
                    t[it] = wide_zero();
                    if (it < sp.ht) {
#pragma unroll
                        for (int pr = 0; pr < WIDE_HT / 2; ++pr)
                            if (2 * pr + 1 >= sp.kbeg_t[it] && 2 * pr < sp.ht) {
                                if (pr + 1 < WIDE_HT / 2 && 2 * (pr + 1) < sp.ht) frag2_load(nxt, T + (it * WIDE_HT + 2 * (pr + 1)) * WIDE_FB, lane);
                                else if (it + 1 < WIDE_HT && it + 1 < sp.ht)
                                    frag2_load(nxt, T + ((it + 1) * WIDE_HT + (sp.kbeg_t[it + 1 < WIDE_HT ? it + 1 : it] & ~1)) * WIDE_FB, lane);
                                if (2 * pr >= sp.kbeg_t[it]) frag2_mfma<0>(t[it], cur, TileB{gh[2 * pr]});
                                if (2 * pr + 1 < sp.ht) frag2_mfma<1>(t[it], cur, TileB{gh[2 * pr + 1]});
                                cur = nxt;
                            }
                    }
                }
#endif
#pragma unroll
                for (int it = 0; it < WIDE_HT; ++it) gh[it] = t[it];
            }
        }
#pragma unroll
        for (int rt = 0; rt < WIDE_HT; ++rt) {
            wide_mask(gh[rt], bits[0][rt]);
            wide_store(GPRE + tile * WIDE_TS, rt, col, hh, gh[rt]);
        }
        if (gx != nullptr) {
            // gacc += W0^T gpre0 : rows = input features (lanes col < d carry weights, the image holds zeros elsewhere)
            Frag2 cur, nxt;
            frag2_load(cur, image + g.offW0T, lane);
#pragma unroll
            for (int pr = 0; pr < WIDE_HT / 2; ++pr)
                if (2 * pr < sp.ht) {
                    if (pr + 1 < WIDE_HT / 2 && 2 * (pr + 1) < sp.ht) frag2_load(nxt, image + g.offW0T + 2 * (pr + 1) * WIDE_FB, lane);
                    frag2_mfma<0>(gacc, cur, TileB{gh[2 * pr]});
                    if (2 * pr + 1 < sp.ht) frag2_mfma<1>(gacc, cur, TileB{gh[2 * pr + 1]});
                    cur = nxt;
                }
            if (valid) {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (rowmap(j, hh) < d) gx[p * d + rowmap(j, hh)] = gacc[j];
            }
        }
    }
}

// =========================================================================================== parameter gradients
// C[a][b] = sum_p A[p][a] B[p][b] over particles (MFMA k = particle) and bias[a] = sum_p A[p][a] for every linear layer, cut
// into JOBS of 64 x 64 (two column tiles of A times one or two column tiles of B); a wave owns one job, keeps the block in 64
// accumulator registers over all the particle tiles it visits (grid-stride over blockIdx.x) and stores it into its slab row.
struct WideJob {
    int64_t a_base, b_base;     // float offsets of column tile 0 of the job: A into the scratch, B into the buffer that holds the
                                // activations (the scratch, or the forward's hand-off buffer); b_base < 0: B = the layer input x
    int a_ts, b_ts;             // floats per particle tile
    int offW, strideW, offB;    // gradient-image position of C[0][0], its row stride, of the bias sums (-1: another job's)
    int nb;                     // column tiles of B (1 or 2)
};
struct WideJobs {
    int count;
    WideJob job[WIDE_MAXJOBS];
};
constexpr int WIDE_OA_WAVES = 4;
// B column tiles per job.  2: 64 x 64 blocks, 136 registers, three waves per SIMD.  -DMF_WIDE_OA_FAT: 64 x 128 blocks (each A pair read
// once per layer: 0.76 of the bytes) at 216 registers, two waves per SIMD — measured slower (11.7 vs 10.4 ms per step at 128 units).
#ifdef MF_WIDE_OA_FAT
constexpr int WIDE_OA_NB = WIDE_HT;
#define MF_WIDE_OA_OCC 2
#else
constexpr int WIDE_OA_NB = 2;
#define MF_WIDE_OA_OCC 3
#endif

__global__ __launch_bounds__(64 * WIDE_OA_WAVES) MF_WAVES_PER_SIMD(MF_WIDE_OA_OCC, MF_WIDE_OA_OCC) void wide_outer_accum_kernel(const float* __restrict__ scratch,
                                                                              const float* __restrict__ actbuf,
                                                                              const float* __restrict__ x, int64_t n, int d,
                                                                              float* __restrict__ gslab, int64_t gtotal,
                                                                              int accumulate, WideJobs jobs) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, col = lane & 31, hh = lane >> 5;
    const int jid = blockIdx.y * WIDE_OA_WAVES + wid;
    if (jid >= jobs.count) return;
    const WideJob jb = jobs.job[jid];
    float* gimage = gslab + (int64_t)blockIdx.x * gtotal;
    const int64_t ntiles = (n + 31) / 32;
    const bool from_x = jb.b_base < 0;
    const int nb = jb.nb;
    f32x16_t acc[2][WIDE_OA_NB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < WIDE_OA_NB; ++b) acc[a][b] = wide_zero();
    float bsum0 = 0.0f, bsum1 = 0.0f;
    const float* A = scratch + jb.a_base;
    const float* B = actbuf + (from_x ? 0 : jb.b_base);
    // lane (col, hh): column 32 t + col, particles 16 hh + s (s = 0..15): k-step s pairs particles (s, 16 + s)
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            const float4* pa0 = reinterpret_cast<const float4*>(A + tile * jb.a_ts + col * 32 + 16 * hh) + 2 * half;
            const float4* pa1 = reinterpret_cast<const float4*>(A + tile * jb.a_ts + (32 + col) * 32 + 16 * hh) + 2 * half;
            float4 a0[2], a1[2], bq[WIDE_OA_NB][2];
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
                    bq[0][q] = make_float4(t[0], t[1], t[2], t[3]);
                }
            } else {
#pragma unroll
                for (int tb = 0; tb < WIDE_OA_NB; ++tb)
                    if (tb < nb) {
                        const float4* pb = reinterpret_cast<const float4*>(B + tile * jb.b_ts + (32 * tb + col) * 32 + 16 * hh) + 2 * half;
                        bq[tb][0] = pb[0];
                        bq[tb][1] = pb[1];
                    }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float av0[4] = {a0[q].x, a0[q].y, a0[q].z, a0[q].w};
                const float av1[4] = {a1[q].x, a1[q].y, a1[q].z, a1[q].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bsum0 += av0[e];
                    bsum1 += av1[e];
                }
#pragma unroll
                for (int tb = 0; tb < WIDE_OA_NB; ++tb)
                    if (tb < nb) {
                        const float bv[4] = {bq[tb][q].x, bq[tb][q].y, bq[tb][q].z, bq[tb][q].w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            acc[0][tb] = mfma(av0[e], bv[e], acc[0][tb]);
                            acc[1][tb] = mfma(av1[e], bv[e], acc[1][tb]);
                        }
                    }
            }
        }
    }
    // memory column c = 32 t + 16 hc + r  <->  physical row / column 32 t + rowmap(r, hc)
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < WIDE_OA_NB; ++tb) {
            if (tb >= nb) continue;
            if (from_x && col >= d) continue;
            const int rhoB = from_x ? col : (32 * tb + rowmap(col & 15, col >> 4));
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int wA = rowmap(r, hh);                       // C row within the tile = memory column of A
                const int rhoA = 32 * ta + rowmap(wA & 15, wA >> 4);
                float* q = &gimage[jb.offW + rhoA * jb.strideW + rhoB];
                *q = accumulate ? *q + acc[ta][tb][r] : acc[ta][tb][r];
            }
        }
    bsum0 += __shfl_xor(bsum0, 32);
    bsum1 += __shfl_xor(bsum1, 32);
    if (hh == 0 && jb.offB >= 0) {
        const int rho = rowmap(col & 15, col >> 4);
        gimage[jb.offB + rho] = accumulate ? gimage[jb.offB + rho] + bsum0 : bsum0;
        gimage[jb.offB + 32 + rho] = accumulate ? gimage[jb.offB + 32 + rho] + bsum1 : bsum1;
    }
}

// every position of the gradient image that a parameter maps to is written by exactly one job: the jobs cover all hidden tile
// pairs in use, whatever the masks say (masked-out blocks receive the products of real activations and gradients; their
// entries carry no parameter and are never read back)
static WideJobs make_wide_jobs(int64_t n, int L, int nblk, const WideSp& sp) {
    WideJobs J;
    J.count = 0;
    const int64_t ntiles = (n + 31) / 32;
    const WideGradLayout gg = wide_grad_layout(L, nblk);
    const int64_t ACT = 0, GPRE = (int64_t)L * ntiles * WIDE_TS, GPHI = 2 * GPRE;
    const int npair = (sp.ht + 1) / 2;
    auto add = [&](int64_t a, int a_ts, int64_t b, int b_ts, int offW, int strideW, int offB, int nb) {
        WideJob& j = J.job[J.count++];
        j.a_base = a; j.b_base = b; j.a_ts = a_ts; j.b_ts = b_ts;
        j.offW = offW; j.strideW = strideW; j.offB = offB; j.nb = nb;
    };
    for (int ap = 0; ap < npair; ++ap)                                                       // input layer: A = GPRE[0], B = x
        add(GPRE + ap * 2048, WIDE_TS, -1, 0, gg.offW0 + 64 * ap * WIDE_DMAX, WIDE_DMAX, gg.offB0 + 64 * ap, 1);
    const int nbp = (sp.ht + WIDE_OA_NB - 1) / WIDE_OA_NB;                                   // B groups of WIDE_OA_NB column tiles
    for (int l = 1; l < L; ++l)                                                              // hidden: A = GPRE[l], B = ACT[l-1]
        for (int ap = 0; ap < npair; ++ap)
            for (int bp = 0; bp < nbp; ++bp) {
                const int nb = sp.ht - WIDE_OA_NB * bp < WIDE_OA_NB ? sp.ht - WIDE_OA_NB * bp : WIDE_OA_NB;
                add(GPRE + (int64_t)l * ntiles * WIDE_TS + ap * 2048, WIDE_TS,
                    ACT + (int64_t)(l - 1) * ntiles * WIDE_TS + bp * WIDE_OA_NB * 1024, WIDE_TS,
                    gg.offH + (l - 1) * gg.strideH + 64 * ap * WIDE_HP + 32 * WIDE_OA_NB * bp, WIDE_HP,
                    bp == 0 ? gg.offH + (l - 1) * gg.strideH + WIDE_HP * WIDE_HP + 64 * ap : -1, nb);
            }
    for (int blk = 0; blk < nblk; ++blk)                                                     // last layer: A = GPHI[blk], B = ACT[L-1]
        for (int bp = 0; bp < nbp; ++bp) {
            // only the input tiles the block's mask lets through (bp 0 always runs: it carries the bias sums)
            int nb = sp.nin3[blk] - WIDE_OA_NB * bp;
            if (nb > WIDE_OA_NB) nb = WIDE_OA_NB;
            if (nb <= 0) {
                if (bp > 0) continue;
                nb = 1;
            }
            add(GPHI + (int64_t)blk * ntiles * 2048, 2048, ACT + (int64_t)(L - 1) * ntiles * WIDE_TS + bp * WIDE_OA_NB * 1024, WIDE_TS,
                gg.off3 + blk * gg.stride3 + 32 * WIDE_OA_NB * bp, WIDE_HP, bp == 0 ? gg.off3 + blk * gg.stride3 + 64 * WIDE_HP : -1, nb);
        }
    return J;
}

// =========================================================================================== inverse
struct WideInvOrder {
    int feat[WIDE_DMAX];        // feat[t] = feature whose order is t
};

template <int K>
__global__ __launch_bounds__(WIDE_BLOCK) MF_WAVES_PER_SIMD(2, 2) void wide_inv_kernel(
    const float* __restrict__ image_arg, int d, int L, const float* __restrict__ y, int64_t n, float* __restrict__ x, WideSp sp,
    WideInvOrder io, int bins_rt) {
    MF_DYN_SMEM(float, lds);
    const int nblk = (K != 0) ? d : 1;
    const WideLayout g = wide_layout(L, nblk);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, col = lane & 31, hh = lane >> 5;
    float* xs = lds + wid * (32 * WIDE_DMAX) + col * WIDE_DMAX;        // this particle's x^[0..15]
    const int64_t ntiles = (n + 31) / 32;
    for (int64_t tile = (int64_t)blockIdx.x * (WIDE_BLOCK / 64) + wid; tile < ntiles; tile += (int64_t)gridDim.x * (WIDE_BLOCK / 64)) {
        MF_NO_HOIST();                 // the image is loop invariant: keep its loads next to their MFMAs (flow_kernels.inc)
        uint64_t image_u = (uint64_t)image_arg;
        MF_OPAQUE_BASE(image_u);       // ... and its block addresses next to the loads (scalar adds), see MF_GLOBAL
        const gfp image = (gfp)image_u;
        const int64_t p = tile * 32 + col;
        const bool valid = p < n;
        const float* yp = y + (valid ? p : n - 1) * d;
        if (hh == 0) {
#pragma unroll
            for (int j = 0; j < WIDE_DMAX; ++j) xs[j] = 0.0f;
        }
        MF_WAVE_SYNC();
#pragma unroll 1
        for (int t = 0; t < d; ++t) {
            const int i = io.feat[t];
            const int blk = (K != 0) ? i : 0;
            const int nin = (K != 0) ? sp.nin3[i] : sp.ht;
            f32x16_t h[WIDE_HT];
            if (nin > 0) {
                float xb[8];
#pragma unroll
                for (int s = 0; s < 8; ++s) xb[s] = (2 * s + hh < d) ? xs[2 * s + hh] : 0.0f;
                wide_input(image, g, d, xb, h, lane, hh, sp.ht);
#pragma unroll 1
                for (int l = 1; l < L; ++l) {
                    f32x16_t tt[WIDE_HT];
                    wide_hidden<false>(image + g.offH + (l - 1) * g.strideH, h, tt, lane, hh, sp);
#pragma unroll
                    for (int rt = 0; rt < WIDE_HT; ++rt) h[rt] = tt[rt];
                }
            } else {
#pragma unroll
                for (int rt = 0; rt < WIDE_HT; ++rt) h[rt] = wide_zero();
            }
            float v[32];
            float xi;
            if constexpr (K != 0) {
                wide_out_block<2, false>(image + g.off3 + blk * g.stride3, h, v, lane, hh, nin);
                float li, gxd, gdummy[32];
                rqs_apply<K, 2>(v, yp[i], hh, xi, li, 0.0f, 0.0f, gdummy, gxd, bins_rt);
            } else {
                wide_out_block<1, false>(image + g.off3, h, v, lane, hh, nin);
                float shift, scale;
                wide_affine_params(v, i, hh, shift, scale);
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

// =========================================================================================== launch + C ABI
static int wide_check(int d, int hidden, int L, int bins, int64_t n) {
    if (d < 1 || d > WIDE_DMAX) return fail("wide flow kernels support 1 <= d <= %d (got %d)", WIDE_DMAX, d);
    if (hidden < 1 || hidden > WIDE_HP) return fail("wide flow kernels support hidden_units <= %d (got %d)", WIDE_HP, hidden);
    if (L < 1 || L > WIDE_LMAX) return fail("wide flow kernels support 1 <= hidden_layers <= %d (got %d)", WIDE_LMAX, L);
    if (bins != 0 && (bins < 2 || bins > RQS_KMAX)) return fail("wide flow kernels: bins = 0 (affine) or 2 .. %d (got %d)", RQS_KMAX, bins);
    if (n < 0) return fail("negative particle count");
    return 0;
}

static int wide_grid(int64_t n) {
    const int64_t ntiles = (n + 31) / 32;
    int64_t g = (ntiles + WIDE_BLOCK / 64 - 1) / (WIDE_BLOCK / 64);
    int64_t cap = 2 * NUM_CU;
#ifdef MF_EMU
    // emulator build only (tests): a small cap makes a wave walk several tiles at test sizes — the grid-stride path that a
    // 512-workgroup grid only reaches past 65 536 particles
    if (const char* e = getenv("MENTFLOW_EMU_WIDE_GRID")) cap = atoi(e) > 0 ? atoi(e) : cap;
#endif
    if (g > cap) g = cap;
    return (int)(g < 1 ? 1 : g);
}

// K instance of a number of bins: 20 and 8 at compile time, any other through the run-time instance, 0 = affine
#define MF_WIDE_DISPATCH(bins, CALL)                   \
    do {                                               \
        if ((bins) == 0) { CALL(0); }                  \
        else if ((bins) == 20) { CALL(20); }           \
        else if ((bins) == 8) { CALL(8); }             \
        else { CALL(RQS_ANY); }                        \
    } while (0)

}  // namespace mf

using namespace mf;

extern "C" int mf_flow_wide_limits(int* max_features, int* max_hidden, int* max_hidden_layers) {
    *max_features = WIDE_DMAX;
    *max_hidden = WIDE_HP;
    *max_hidden_layers = WIDE_LMAX;
    return 0;
}
extern "C" int64_t mf_flow_wide_image_floats(int hidden_layers, int nblk) { return wide_layout(hidden_layers, nblk).total; }
extern "C" int64_t mf_flow_wide_grad_floats(int hidden_layers, int nblk) { return wide_grad_layout(hidden_layers, nblk).total; }

static int wide_layer_fwd_impl(const float* image, int d, int hidden, int hidden_layers, int bins, const int32_t* order,
                               const float* x, int64_t n, float* y, const float* logp_in, float* logp_out, int init_logp, float* act,
                               void* stream) {
    if (wide_check(d, hidden, hidden_layers, bins, n)) return 1;
    if (n == 0) return 0;
    const WideSp sp = make_wide_sp(d, order, hidden, bins ? d : 1);
    ProfScope prof(PK_FLOW_FWD, stream);
#define CALL(KK) MF_LAUNCH((wide_fwd_kernel<KK>), wide_grid(n), WIDE_BLOCK, 0, stream, image, d, hidden_layers, x, n, y, logp_in, \
                           logp_out, init_logp, sp, bins, act)
    MF_WIDE_DISPATCH(bins, CALL);
#undef CALL
    return check_launch("mf_flow_wide_layer_fwd");
}

extern "C" int mf_flow_wide_layer_fwd(const float* image, int d, int hidden, int hidden_layers, int bins, const int32_t* order,
                                       const float* x, int64_t n, float* y, const float* logp_in, float* logp_out, int init_logp,
                                       void* stream) {
    return wide_layer_fwd_impl(image, d, hidden, hidden_layers, bins, order, x, n, y, logp_in, logp_out, init_logp, nullptr, stream);
}

// ---- activation hand-off (training forward -> backward through HBM; wide_act_floats) -------------------------------------------
extern "C" int64_t mf_flow_wide_act_floats(int64_t n, int d, int hidden_layers, int bins) {
    if (n <= 0 || d < 1 || hidden_layers < 1) return 0;
    return wide_act_floats((n + 31) / 32, hidden_layers, bins ? d : 1);
}

extern "C" int mf_flow_wide_layer_fwd_save(const float* image, int d, int hidden, int hidden_layers, int bins, const int32_t* order,
                                            const float* x, int64_t n, float* y, const float* logp_in, float* logp_out,
                                            int init_logp, float* act, int64_t act_floats, void* stream) {
    if (n > 0 && (act == nullptr || act_floats < mf_flow_wide_act_floats(n, d, hidden_layers, bins)))
        return fail("act buffer too small: %lld floats, need %lld (mf_flow_wide_act_floats)", (long long)act_floats,
                    (long long)mf_flow_wide_act_floats(n, d, hidden_layers, bins));
    return wide_layer_fwd_impl(image, d, hidden, hidden_layers, bins, order, x, n, y, logp_in, logp_out, init_logp, act, stream);
}

extern "C" int64_t mf_flow_wide_bwd_scratch_floats(int64_t n, int d, int hidden_layers, int bins) {
    const int64_t ntiles = (n + 31) / 32;
    return (2 * (int64_t)hidden_layers * WIDE_TS + (int64_t)(bins ? d : 1) * 2048) * ntiles;
}

extern "C" int mf_flow_wide_bwd_slab_rows(int64_t n) {
    if (n <= 0) return 0;
    const int64_t ntiles = (n + 31) / 32;
    int64_t G = (ntiles + 7) / 8;                    // at least 8 tiles of work per workgroup column
    if (G > NUM_CU) G = NUM_CU;
    return (int)(G < 1 ? 1 : G);
}

static int wide_layer_bwd_impl(const float* image, int d, int hidden, int hidden_layers, int bins, const int32_t* order,
                               const float* x, int64_t n, const float* gy, const float* glogp, float* gx, float* gslab,
                               int slab_rows, int accumulate, float* scratch, int64_t scratch_floats, const float* act, void* stream) {
    if (wide_check(d, hidden, hidden_layers, bins, n)) return 1;
    if (n == 0) return 0;
    if (scratch_floats < mf_flow_wide_bwd_scratch_floats(n, d, hidden_layers, bins)) return fail("scratch too small");
    if (slab_rows != mf_flow_wide_bwd_slab_rows(n))
        return fail("gslab has %d rows, this call writes %d (mf_flow_wide_bwd_slab_rows)", slab_rows, mf_flow_wide_bwd_slab_rows(n));
    const int nblk = bins ? d : 1;
    const WideSp sp = make_wide_sp(d, order, hidden, nblk);
    {
        ProfScope prof(PK_FLOW_BWD, stream);
#define CALL(KK)                                                                                                                  \
    if (act != nullptr)                                                                                                           \
        MF_LAUNCH((wide_bwd_kernel<KK, true>), wide_grid(n), WIDE_BLOCK, 0, stream, image, d, hidden_layers, x, n, gy, glogp, gx,  \
                  scratch, sp, bins, act);                                                                                        \
    else                                                                                                                          \
        MF_LAUNCH((wide_bwd_kernel<KK, false>), wide_grid(n), WIDE_BLOCK, 0, stream, image, d, hidden_layers, x, n, gy, glogp, gx, \
                  scratch, sp, bins, act)
        MF_WIDE_DISPATCH(bins, CALL);
#undef CALL
    }
    if (check_launch("mf_flow_wide_layer_bwd")) return 1;
    const WideJobs jobs = make_wide_jobs(n, hidden_layers, nblk, sp);
    ProfScope prof(PK_OUTER_ACCUM, stream);
    // the B operands (hidden activations): from the forward's hand-off buffer when there is one — same [L][ntiles][HT*1024] layout as
    // the ACT part of the scratch, which the backward then leaves unwritten
    MF_LAUNCH(wide_outer_accum_kernel, dim3((unsigned)slab_rows, (unsigned)((jobs.count + WIDE_OA_WAVES - 1) / WIDE_OA_WAVES)),
              64 * WIDE_OA_WAVES, 0, stream, scratch, act != nullptr ? act : scratch, x, n, d, gslab,
              (int64_t)wide_grad_layout(hidden_layers, nblk).total, accumulate, jobs);
    return check_launch("mf_flow_wide_layer_bwd(outer_accum)");
}

extern "C" int mf_flow_wide_layer_bwd(const float* image, int d, int hidden, int hidden_layers, int bins, const int32_t* order,
                                       const float* x, int64_t n, const float* gy, const float* glogp, float* gx, float* gslab,
                                       int slab_rows, int accumulate, float* scratch, int64_t scratch_floats, void* stream) {
    return wide_layer_bwd_impl(image, d, hidden, hidden_layers, bins, order, x, n, gy, glogp, gx, gslab, slab_rows, accumulate, scratch,
                               scratch_floats, nullptr, stream);
}

extern "C" int mf_flow_wide_layer_bwd_saved(const float* image, int d, int hidden, int hidden_layers, int bins, const int32_t* order,
                                             const float* x, int64_t n, const float* gy, const float* glogp, float* gx, float* gslab,
                                             int slab_rows, int accumulate, float* scratch, int64_t scratch_floats, const float* act,
                                             int64_t act_floats, void* stream) {
    if (n > 0 && (act == nullptr || act_floats < mf_flow_wide_act_floats(n, d, hidden_layers, bins))) return fail("act buffer too small");
    return wide_layer_bwd_impl(image, d, hidden, hidden_layers, bins, order, x, n, gy, glogp, gx, gslab, slab_rows, accumulate, scratch,
                               scratch_floats, act, stream);
}

extern "C" int mf_flow_wide_layer_inv(const float* image, int d, int hidden, int hidden_layers, int bins, const int32_t* order,
                                       const float* y, int64_t n, float* x, void* stream) {
    if (wide_check(d, hidden, hidden_layers, bins, n)) return 1;
    if (order == nullptr) return fail("the inverse needs the autoregressive order of the layer");
    WideInvOrder io;
    for (int t = 0; t < WIDE_DMAX; ++t) io.feat[t] = 0;
    for (int i = 0; i < d; ++i) {
        if (order[i] < 0 || order[i] >= d) return fail("order[%d] = %d out of range", i, order[i]);
        io.feat[order[i]] = i;
    }
    if (n == 0) return 0;
    const WideSp sp = make_wide_sp(d, order, hidden, bins ? d : 1);
    const size_t smem = sizeof(float) * (WIDE_BLOCK / 64) * 32 * WIDE_DMAX;
#define CALL(KK) MF_LAUNCH((wide_inv_kernel<KK>), wide_grid(n), WIDE_BLOCK, smem, stream, image, d, hidden_layers, y, n, x, sp, io, bins)
    MF_WIDE_DISPATCH(bins, CALL);
#undef CALL
    return check_launch("mf_flow_wide_layer_inv");
}
