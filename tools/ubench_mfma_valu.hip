// Micro-benchmark (development tool): do v_mfma_f32_32x32x2_f32 and fp32 VALU work co-execute on a gfx950 SIMD?
// SUPERSEDED by ubench_mfma_valu2.hip: mode 2 below (even waves MFMA / odd waves VALU) puts the two kinds on DIFFERENT
// SIMDs (wave w runs on SIMD w % 4), so its "full rate" result says nothing about co-execution on one SIMD.
// Each wave runs a chain of fp32 MFMAs, or a chain of v_fma_f32, or both interleaved; 4 or 8 waves per CU-SIMD mixes.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// mode: 0 = all waves MFMA f32, 1 = all waves VALU, 2 = even waves MFMA / odd waves VALU, 3 = every wave interleaves,
//       4 = all waves bf16 MFMA 32x32x16, 5 = even bf16 MFMA / odd VALU
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
    const int wid = threadIdx.x >> 6;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = (float)threadIdx.x;
    float a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f;
    float v0 = a, v1 = b, v2 = a + b, v3 = a - b, v4 = 1.f, v5 = 2.f, v6 = 3.f, v7 = 4.f;
    s16x8 ha, hb;
    for (int j = 0; j < 8; ++j) { ha[j] = (short)(threadIdx.x + j); hb[j] = (short)(j * 3); }
    const bool do_mfma = (mode == 0) || (mode == 2 && !(wid & 1)) || mode == 3;
    const bool do_valu = (mode == 1) || (mode == 2 && (wid & 1)) || mode == 3 || (mode == 5 && (wid & 1));
    const bool do_bf16 = (mode == 4) || (mode == 5 && !(wid & 1));
    for (int it = 0; it < iters; ++it) {
        if (mode == 6) {      // fine-grained interleave inside every wave: 1 MFMA, 16 FMAs, repeated 4 times
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                v0 = fmaf(v0, a, b); v1 = fmaf(v1, a, b); v2 = fmaf(v2, a, b); v3 = fmaf(v3, a, b);
                v4 = fmaf(v4, a, b); v5 = fmaf(v5, a, b); v6 = fmaf(v6, a, b); v7 = fmaf(v7, a, b);
                v0 = fmaf(v0, a, b); v1 = fmaf(v1, a, b); v2 = fmaf(v2, a, b); v3 = fmaf(v3, a, b);
                v4 = fmaf(v4, a, b); v5 = fmaf(v5, a, b); v6 = fmaf(v6, a, b); v7 = fmaf(v7, a, b);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (do_mfma) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        if (do_bf16) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, acc, 0, 0, 0);
        }
        if (do_valu) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {     // 64 independent-ish FMAs = 256 issue cycles at 4 cyc each
                v0 = fmaf(v0, a, b); v1 = fmaf(v1, a, b); v2 = fmaf(v2, a, b); v3 = fmaf(v3, a, b);
                v4 = fmaf(v4, a, b); v5 = fmaf(v5, a, b); v6 = fmaf(v6, a, b); v7 = fmaf(v7, a, b);
            }
        }
    }
    float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 512 * 4);
    const int iters = 20000;
    const char* names[] = {"all waves f32 MFMA (4/iter)", "all waves VALU (64 fma/iter)", "even MFMA f32 / odd VALU",
                           "every wave both", "all waves bf16 MFMA (8/iter)", "even bf16 MFMA / odd VALU",
                           "every wave interleaved 1:16"};
    for (int mode = 0; mode < 7; ++mode) {
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        k<<<256, 512>>>(out, 100, mode);
        hipDeviceSynchronize();
        hipEventRecord(a);
        k<<<256, 512>>>(out, iters, mode);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-32s : %8.3f ms  (%.0f cycles/iter/SIMD-pair @2.4GHz)\n", names[mode], ms, ms * 1e-3 * 2.4e9 / iters);
    }
    return 0;
}
