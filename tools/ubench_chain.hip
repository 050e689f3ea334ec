// Micro-benchmark (development tool): cycles per hand-scheduled fp32 MFMA chain (the chain64 of flow.hip) with ONE wave
// per SIMD, in isolation: (a) bare dependent chain, (b) two interleaved independent chains, (c) bias init + chain + relu.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16_t __attribute__((ext_vector_type(16)));
constexpr int WS = 65;
__device__ __forceinline__ constexpr int rowmap(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }
__device__ __forceinline__ constexpr int kcol(int s) { return 32 * (s >> 4) + rowmap(s & 15, 0); }

template <int KS, int S4N>
__device__ __forceinline__ void mfma4_pf(f32x16_t& acc, const float (&a)[4], float (&n)[4], unsigned addr, float b0, float b1, float b2, float b3) {
    asm volatile(
        "ds_read_b32 %1, %9 offset:%14\n\t"
        "ds_read_b32 %2, %9 offset:%15\n\t"
        "ds_read_b32 %3, %9 offset:%16\n\t"
        "ds_read_b32 %4, %9 offset:%17\n\t"
        "v_mfma_f32_32x32x2_f32 %0, %5, %10, %0\n\t"
        "v_mfma_f32_32x32x2_f32 %0, %6, %11, %0\n\t"
        "v_mfma_f32_32x32x2_f32 %0, %7, %12, %0\n\t"
        "v_mfma_f32_32x32x2_f32 %0, %8, %13, %0\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "+v"(acc), "=&v"(n[0]), "=&v"(n[1]), "=&v"(n[2]), "=&v"(n[3])
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(addr), "v"(b0), "v"(b1), "v"(b2), "v"(b3),
          "n"(kcol(S4N) * KS * 4), "n"(kcol(S4N + 1) * KS * 4), "n"(kcol(S4N + 2) * KS * 4), "n"(kcol(S4N + 3) * KS * 4));
}
// two independent accumulators per block: 8 fragment reads, 8 MFMAs (alternating accumulators)
template <int KS, int S4N>
__device__ __forceinline__ void mfma8_pf(f32x16_t& acc0, f32x16_t& acc1, const float (&a)[8], float (&n)[8], unsigned addr0, unsigned addr1,
                                         float b0, float b1, float b2, float b3) {
    asm volatile(
        "ds_read_b32 %2, %18 offset:%24\n\t"
        "ds_read_b32 %3, %18 offset:%25\n\t"
        "ds_read_b32 %4, %18 offset:%26\n\t"
        "ds_read_b32 %5, %18 offset:%27\n\t"
        "ds_read_b32 %6, %19 offset:%24\n\t"
        "ds_read_b32 %7, %19 offset:%25\n\t"
        "ds_read_b32 %8, %19 offset:%26\n\t"
        "ds_read_b32 %9, %19 offset:%27\n\t"
        "v_mfma_f32_32x32x2_f32 %0, %10, %20, %0\n\t"
        "v_mfma_f32_32x32x2_f32 %1, %14, %20, %1\n\t"
        "v_mfma_f32_32x32x2_f32 %0, %11, %21, %0\n\t"
        "v_mfma_f32_32x32x2_f32 %1, %15, %21, %1\n\t"
        "v_mfma_f32_32x32x2_f32 %0, %12, %22, %0\n\t"
        "v_mfma_f32_32x32x2_f32 %1, %16, %22, %1\n\t"
        "v_mfma_f32_32x32x2_f32 %0, %13, %23, %0\n\t"
        "v_mfma_f32_32x32x2_f32 %1, %17, %23, %1\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "+v"(acc0), "+v"(acc1), "=&v"(n[0]), "=&v"(n[1]), "=&v"(n[2]), "=&v"(n[3]), "=&v"(n[4]), "=&v"(n[5]), "=&v"(n[6]), "=&v"(n[7])
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(addr0), "v"(addr1),
          "v"(b0), "v"(b1), "v"(b2), "v"(b3),
          "n"(kcol(S4N) * KS * 4), "n"(kcol(S4N + 1) * KS * 4), "n"(kcol(S4N + 2) * KS * 4), "n"(kcol(S4N + 3) * KS * 4));
}
__device__ __forceinline__ void drain(f32x16_t& acc) { asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc)); }

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
    __shared__ float W[64 * WS + 64];
    for (int i = threadIdx.x; i < 64 * WS + 64; i += 256) W[i] = 0.001f * (float)(i % 97);
    __syncthreads();
    const int lane = threadIdx.x & 63, col = lane & 31, hh = lane >> 5;
    f32x16_t h[2], t[2];
    for (int r = 0; r < 16; ++r) { h[0][r] = 0.01f * r + lane; h[1][r] = 0.02f * r - lane; t[0][r] = 0; t[1][r] = 0; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        asm volatile("" ::: "memory");
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                if (MODE == 2) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) t[rt][r] = W[64 * WS + 32 * rt + rowmap(r, hh)];
                }
                const float* wl = W + (32 * rt + col) * WS + 4 * hh;
                const unsigned addr = (unsigned)(size_t)wl;
                float a0[4], a1[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) a0[j] = wl[kcol(j)];
#define G(S, C, N) mfma4_pf<1, (S + 4) & 31>(t[rt], C, N, addr, h[(S) >> 4][(S) & 15], h[(S + 1) >> 4][(S + 1) & 15], h[(S + 2) >> 4][(S + 2) & 15], h[(S + 3) >> 4][(S + 3) & 15]);
                G(0, a0, a1) G(4, a1, a0) G(8, a0, a1) G(12, a1, a0) G(16, a0, a1) G(20, a1, a0) G(24, a0, a1) G(28, a1, a0)
#undef G
                drain(t[rt]);
            }
            if (MODE == 2) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) t[rt][r] = fmaxf(t[rt][r], 0.0f);
            }
        } else {
            const float* wl0 = W + col * WS + 4 * hh;
            const float* wl1 = W + (32 + col) * WS + 4 * hh;
            float a0[8], a1[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) { a0[j] = wl0[kcol(j)]; a0[4 + j] = wl1[kcol(j)]; }
#define G(S, C, N) mfma8_pf<1, (S + 4) & 31>(t[0], t[1], C, N, (unsigned)(size_t)wl0, (unsigned)(size_t)wl1, h[(S) >> 4][(S) & 15], h[(S + 1) >> 4][(S + 1) & 15], h[(S + 2) >> 4][(S + 2) & 15], h[(S + 3) >> 4][(S + 3) & 15]);
            G(0, a0, a1) G(4, a1, a0) G(8, a0, a1) G(12, a1, a0) G(16, a0, a1) G(20, a1, a0) G(24, a0, a1) G(28, a1, a0)
#undef G
            drain(t[0]);
            drain(t[1]);
        }
        // feed back so that iterations depend on each other like the real layers do
#pragma unroll
        for (int r = 0; r < 16; ++r) { h[0][r] = t[0][r] * 1e-3f; h[1][r] = t[1][r] * 1e-3f; }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += h[0][r] + h[1][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 2000;
    k<MODE><<<256, 256>>>(out, cyc, 10);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    k<MODE><<<256, 256>>>(out, cyc, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 256; ++i) avg += (double)h[i]; avg /= 256;
    printf("%-44s %8.3f ms  %9.1f memtime ticks per iteration (64 MFMAs: 4096 cycles ideal)  -> %.1f ns/iter\n", name, ms, avg / iters, ms * 1e6 / iters);
}
int main() {
    run<0>("two dependent chains (rt 0, then rt 1)");
    run<1>("two chains interleaved in one block");
    run<2>("bias init + two chains + relu");
    return 0;
}
