// Micro-benchmark (development tool, not part of the library): throughput of LDS atomics on gfx950 for the access
// patterns of the KDE scatter: float add vs u32 add vs plain read-modify-write, random vs same-address lanes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, int spread) {
    __shared__ float img[8192];
    unsigned* imgu = reinterpret_cast<unsigned*>(img);
    for (int i = threadIdx.x; i < 8192; i += 256) img[i] = 0.f;
    __syncthreads();
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        int a = (spread == 0) ? ((it * 7) & 8191) : (int)((s >> 8) % (unsigned)spread);
        float w = (float)(s & 255) * (1.0f / 256.0f);
        if (MODE == 0) atomicAdd(&img[a], w);
        if (MODE == 1) atomicAdd(&imgu[a], (unsigned)(w * 65536.0f));
        if (MODE == 2) { img[a] += w; }                      // racy plain RMW (upper bound of ds traffic)
        if (MODE == 3) acc += img[a] * w;                    // gather only
        if (MODE == 4) atomicAdd(reinterpret_cast<unsigned long long*>(img) + (a >> 1), (unsigned long long)(w * 4294967296.0f));
    }
    __syncthreads();
    float t = acc;
    for (int i = threadIdx.x; i < 8192; i += 256) t += img[i];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}

template <int MODE>
void run(const char* name, int spread) {
    float* out;
    hipMalloc(&out, 1024 * 256 * 4);
    const int iters = 4096;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<1024, 256>>>(out, 16, spread);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE><<<1024, 256>>>(out, iters, spread);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double ops = 1024.0 * 256 * iters;
    printf("%-28s spread %5d : %8.3f ms  %7.2f Gops/s  (%.2f lane-ops/clk/CU @2.4GHz)\n", name, spread, ms, ops / ms * 1e-6,
           ops / (ms * 1e-3) / 256 / 2.4e9);
    hipFree(out);
}

int main() {
    for (int spread : {0, 64, 640, 6400}) {
        run<0>("ds_add_f32", spread);
        run<1>("ds_add_u32", spread);
        run<4>("ds_add_u64", spread);
        run<2>("plain rmw (racy)", spread);
        run<3>("gather read", spread);
    }
    return 0;
}
