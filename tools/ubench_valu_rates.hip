// Micro-benchmark (development tool): issue cost in cycles of the VALU instructions the spline code leans on, one wave per
// SIMD, 8 independent chains per lane so that latency does not matter: v_fma_f32 (reference, 4 cycles), v_exp_f32,
// v_rcp_f32, v_cvt_f64_f32, v_cvt_f32_f64, v_add_f64, v_cndmask, and the whole fp64 cumulative-sum step against a
// compensated fp32 one.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
    float a[8];
    double d[8];
    for (int j = 0; j < 8; ++j) { a[j] = 0.5f + 0.01f * (threadIdx.x + j); d[j] = a[j]; }
    float c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MODE == 0) a[j] = fmaf(a[j], 1.0001f, 0.0001f);
            if (MODE == 1) a[j] = __builtin_amdgcn_exp2f(a[j] * 0.001f) ;
            if (MODE == 2) a[j] = __builtin_amdgcn_rcpf(a[j]) + 1.0f;
            if (MODE == 3) { d[j] = (double)a[j]; asm volatile("" : "+v"(d[j])); a[j] = a[j] + 1e-7f; }
            if (MODE == 4) { a[j] = (float)d[j]; asm volatile("" : "+v"(a[j])); }
            if (MODE == 5) d[j] = d[j] + 1.0000001;
            if (MODE == 6) a[j] = (a[j] > 0.7f) ? a[j] * 0.999f : c[j];
            if (MODE == 7) {                      // fp64 cumulative-sum step: cvt, add, cvt, compare
                d[j] += (double)a[j];
                c[j] = (float)d[j];
                a[j] = (c[j] < 1e30f) ? a[j] : 0.0f;
            }
            if (MODE == 8) {                      // compensated fp32 step (ordered two-sum), same outputs
                const float s = c[j], p = a[j];
                const float hi = fmaxf(s, p), lo = fminf(s, p);
                const float t = hi + lo;
                const float e = lo - (t - hi);
                float comp = (float)d[j];         // (stand-in for the running compensation register)
                comp += e;
                c[j] = t;
                a[j] = ((t + comp) < 1e30f) ? a[j] : 0.0f;
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int j = 0; j < 8; ++j) s += a[j] + (float)d[j] + c[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 4096;
    k<MODE><<<256, 256>>>(out, cyc, 16);
    hipDeviceSynchronize();
    k<MODE><<<256, 256>>>(out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 256; ++i) avg += (double)h[i]; avg /= 256;
    printf("%-44s %7.2f cycles per wave-step (8 steps per iteration)\n", name, avg / iters / 8);
}
int main() {
    run<0>("v_fma_f32");
    run<1>("v_mul + v_exp_f32");
    run<2>("v_rcp_f32 + v_add");
    run<3>("v_cvt_f64_f32 + v_add_f32");
    run<4>("v_cvt_f32_f64");
    run<5>("v_add_f64");
    run<6>("v_cmp + v_cndmask + v_mul");
    run<7>("fp64 cumsum step (cvt, add_f64, cvt, cmp, sel)");
    run<8>("compensated fp32 step (max, min, 3 add, cmp..)");
    return 0;
}
