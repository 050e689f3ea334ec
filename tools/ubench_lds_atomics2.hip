// Micro-benchmark (development tool, not part of the library): LDS atomic-add throughput on gfx950 as a function of the
// ADDRESS PATTERN of a wave.  Question behind it: is the 2.7 lane-ops/clk/CU that ds_add_u64 sustains with random
// addresses (tools/ubench_lds_atomics.hip) the instruction's ceiling, or the cost of random bank conflicts?
// entry(lane, it) = (rnd % spread) * mult + (lane & lmask) * lmul   (64-bit entries for the u64 modes)
//   spread=1, lmask=63, lmul=1            : 64 consecutive entries per wave-instruction (conflict-free by construction)
//   mult=16, lmask=15, lmul=1             : "one private copy per lane of a 16-lane group" layout, random bin
//   mult=1, lmask=0                       : plain random entries
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned long long u64;

template <int MODE>   // 0: ds_add_u64   1: ds_add_u32   2: ds_add_u64 x1 + nothing else, address precomputed (no VALU in loop)
__global__ __launch_bounds__(256) void k(float* out, int iters, int spread, int mult, int lmask, int lmul, int entries) {
    extern __shared__ u64 img[];
    unsigned* img32 = reinterpret_cast<unsigned*>(img);
    for (int i = threadIdx.x; i < entries; i += 256) img[i] = 0;
    __syncthreads();
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    const int lane_off = (threadIdx.x & lmask) * lmul;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        const int a = (int)((s >> 8) % (unsigned)spread) * mult + lane_off;
        const unsigned w = s & 0xffffu;
        if (MODE == 0) atomicAdd(&img[a], (u64)w << 20);
        if (MODE == 1) atomicAdd(&img32[a], w);
    }
    __syncthreads();
    u64 t = 0;
    for (int i = threadIdx.x; i < entries; i += 256) t += img[i];
    out[blockIdx.x * 256 + threadIdx.x] = (float)t;
}

template <int MODE>
void run(const char* name, int spread, int mult, int lmask, int lmul, int grid) {
    float* out;
    hipMalloc(&out, (size_t)grid * 256 * 4);
    const int entries = spread * mult + 64 * (lmul > 0 ? lmul : 1) + 64;
    const size_t smem = (size_t)entries * 8;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    const int iters = 4096;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<grid, 256, smem>>>(out, 16, spread, mult, lmask, lmul, entries);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE><<<grid, 256, smem>>>(out, iters, spread, mult, lmask, lmul, entries);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double ops = (double)grid * 256 * iters;
    printf("%-10s spread %5d mult %3d lmask %2d lmul %2d  LDS %6zu B grid %4d : %8.3f ms  %8.2f Gops/s  %5.2f lane-ops/clk/CU\n", name,
           spread, mult, lmask, lmul, smem, grid, ms, ops / ms * 1e-6, ops / (ms * 1e-3) / 256 / 2.4e9);
    hipFree(out);
}

int main() {
    for (int grid : {1024, 2048}) {
        run<0>("u64", 1, 1, 63, 1, grid);          // 64 consecutive entries: conflict-free
        run<0>("u64", 100, 64, 63, 1, grid);       // random row, lane = column: conflict-free rows
        run<0>("u64", 6400, 1, 0, 0, grid);        // random entries (the r01 number)
        run<0>("u64", 1280, 1, 0, 0, grid);        // random, smaller image
        run<0>("u64", 1280, 16, 15, 1, grid);      // 16 private copies (one per lane of a 16-lane group)
        run<0>("u64", 1280, 8, 7, 1, grid);        // 8 copies
        run<0>("u64", 1280, 4, 3, 1, grid);        // 4 copies
        run<0>("u64", 1280, 2, 1, 1, grid);        // 2 copies
        run<0>("u64", 400, 17, 15, 1, grid);       // 16 copies, odd stride
        run<0>("u64", 64, 1, 0, 0, grid);          // one hot row
        run<0>("u64", 16, 1, 0, 0, grid);          // very hot
        run<1>("u32", 1, 1, 63, 1, grid);
        run<1>("u32", 6400, 1, 0, 0, grid);
        run<1>("u32", 1280, 32, 31, 1, grid);      // 32 private copies
        run<1>("u32", 1280, 16, 15, 1, grid);
        run<1>("u32", 64, 1, 0, 0, grid);
    }
    return 0;
}
