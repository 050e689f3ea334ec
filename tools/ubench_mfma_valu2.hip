// Micro-benchmark (development tool), second take: which instruction classes co-execute on ONE gfx950 SIMD?
// Workgroup = 8 waves on one CU; wave w lands on SIMD w % 4, so waves w and w + 4 share a SIMD.  Role of a wave:
//   M = chain of v_mfma_f32_32x32x2_f32,  B = chain of v_mfma_f32_32x32x16_bf16,  V = independent v_fma_f32,
//   F = scalar v_fma_f32 (asm), C = v_cmp + v_cndmask (asm),
//   I = integer/select VALU (v_add_u32 / v_cndmask),  T = transcendental (v_exp_f32),  - = idle (exits at once)
// A "unit" is 4 fp32 MFMAs (256 matrix-pipe cycles), 8 bf16 MFMAs (8 x 32 cycles), 64 FMAs / int ops (256 issue cycles),
// or 16 v_exp (quarter rate: 256 cycles).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <char R>
__device__ __forceinline__ float run(int iters, float seed) {
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = seed;
    float a = 1.0f + seed * 1e-6f, b = 0.5f;
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = seed + j;
    int iv[8];
    for (int j = 0; j < 8; ++j) iv[j] = (int)seed + j;
    bf16x8 ha, hb;
    for (int j = 0; j < 8; ++j) { ha[j] = (__bf16)(seed + j); hb[j] = (__bf16)(j * 0.25f); }
    for (int it = 0; it < iters; ++it) {
        if (R == 'M') {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        } else if (R == 'B') {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, acc, 0, 0, 0);
        } else if (R == 'V') {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], a, b);
        } else if (R == 'I') {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) iv[j] = (iv[j] + it) ^ iv[(j + 1) & 7];
        } else if (R == 'T') {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = __builtin_amdgcn_exp2f(v[j]);
        } else if (R == 'F') {            // 64 scalar (non-packed) v_fma_f32
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(a), "v"(b));
        } else if (R == 'C') {            // 32 x (v_cmp_lt_f32 + v_cndmask_b32): the select chains of the spline
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(v[j]) : "v"(a), "v"(b) : "vcc");
        } else if (R == 'X') {            // one wave interleaving 1 fp32 MFMA : 16 FMA
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], a, b);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (R == 'Y') {            // one wave interleaving 2 bf16 MFMA : 16 FMA
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, acc, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], a, b);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0.0f;
    for (int j = 0; j < 8; ++j) s += v[j] + (float)iv[j];
    for (int r = 0; r < 16; ++r) s += acc[r];
    return s;
}

__device__ float role(char r, int iters, float seed) {
    switch (r) {
        case 'M': return run<'M'>(iters, seed);
        case 'B': return run<'B'>(iters, seed);
        case 'V': return run<'V'>(iters, seed);
        case 'I': return run<'I'>(iters, seed);
        case 'T': return run<'T'>(iters, seed);
        case 'F': return run<'F'>(iters, seed);
        case 'C': return run<'C'>(iters, seed);
        case 'X': return run<'X'>(iters, seed);
        case 'Y': return run<'Y'>(iters, seed);
        default: return 0.0f;
    }
}

// lo = role of waves 0..3, hi = role of waves 4..7 (wave w + 4 shares the SIMD of wave w)
__global__ __launch_bounds__(512) void k(float* out, int iters, char lo, char hi) {
    const int wid = threadIdx.x >> 6;
    const char r = wid < 4 ? lo : hi;          // wave-uniform
    out[blockIdx.x * 512 + threadIdx.x] = role(r, iters, (float)(threadIdx.x & 63));
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 512 * 4);
    const int iters = 20000;
    const char* cfgs[] = {"M-", "V-", "B-", "I-", "T-", "X-", "Y-", "MM", "VV", "MV", "MI", "MT", "BV", "BI", "BT", "XX", "YY",
                          "BB", "BM", "F-", "C-", "FF", "CC", "MF", "MC", "BF", "BC"};
    for (const char* c : cfgs) {
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        k<<<256, 512>>>(out, 100, c[0], c[1]);
        hipDeviceSynchronize();
        hipEventRecord(a);
        k<<<256, 512>>>(out, iters, c[0], c[1]);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%s : %8.3f ms  = %6.0f ns/iter\n", c, ms, ms * 1e6 / iters);
    }
    return 0;
}
