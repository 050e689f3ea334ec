// Micro-benchmark (development tool): cost of one staging event of the fused flow backward — 4 waves (one per SIMD) each
// write a 64 x 32 fp32 tile held particle-on-the-lane (32 registers per lane) into LDS — for
//   (a) ds_write_b32 with per-lane XOR-swizzled addresses (the r02 layout: row*32 + swizzle(particle)),
//   (b) ds_write_addtid_b32 (address = M0 + imm + 4*lane, no address register): row pairs (r, r+4) adjacent, pair stride 66,
// followed by the s_waitcnt + barrier the kernel needs before anybody reads.  Also checks what (b) wrote and times the
// two matching fragment-read patterns of the consumer (4 x ds_read_b128 swizzled vs 8 x ds_read_b64 plain) with 16 MFMAs.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_stage.hip -o tools/bin/ubench_stage && tools/bin/ubench_stage
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16_t __attribute__((ext_vector_type(16)));
constexpr int PS = 66;                  // floats per row pair
constexpr int TILE_B = 32 * PS;         // floats per staged 64-row tile (layout b)

__device__ __forceinline__ void stage_b32(float* S, const int (&off)[2][4], const float (&v)[32]) {
#pragma unroll
    for (int m = 0; m < 32; ++m) S[(32 * (m >> 4) + 8 * ((m & 15) >> 2)) * 32 + off[((m & 15) >> 2) & 1][m & 3]] = v[m];
}

// 32 stores, pair pr = 16 (m >> 4) + (m & 15) at byte offset pr * PS * 4 from the tile base in M0
__device__ __forceinline__ void stage_addtid(unsigned base_bytes, const float (&v)[32]) {
    asm volatile(
        "s_mov_b32 m0, %32\n\t"
        "ds_write_addtid_b32 %0 offset:0\n\t" "ds_write_addtid_b32 %1 offset:264\n\t" "ds_write_addtid_b32 %2 offset:528\n\t"
        "ds_write_addtid_b32 %3 offset:792\n\t" "ds_write_addtid_b32 %4 offset:1056\n\t" "ds_write_addtid_b32 %5 offset:1320\n\t"
        "ds_write_addtid_b32 %6 offset:1584\n\t" "ds_write_addtid_b32 %7 offset:1848\n\t" "ds_write_addtid_b32 %8 offset:2112\n\t"
        "ds_write_addtid_b32 %9 offset:2376\n\t" "ds_write_addtid_b32 %10 offset:2640\n\t" "ds_write_addtid_b32 %11 offset:2904\n\t"
        "ds_write_addtid_b32 %12 offset:3168\n\t" "ds_write_addtid_b32 %13 offset:3432\n\t" "ds_write_addtid_b32 %14 offset:3696\n\t"
        "ds_write_addtid_b32 %15 offset:3960\n\t" "ds_write_addtid_b32 %16 offset:4224\n\t" "ds_write_addtid_b32 %17 offset:4488\n\t"
        "ds_write_addtid_b32 %18 offset:4752\n\t" "ds_write_addtid_b32 %19 offset:5016\n\t" "ds_write_addtid_b32 %20 offset:5280\n\t"
        "ds_write_addtid_b32 %21 offset:5544\n\t" "ds_write_addtid_b32 %22 offset:5808\n\t" "ds_write_addtid_b32 %23 offset:6072\n\t"
        "ds_write_addtid_b32 %24 offset:6336\n\t" "ds_write_addtid_b32 %25 offset:6600\n\t" "ds_write_addtid_b32 %26 offset:6864\n\t"
        "ds_write_addtid_b32 %27 offset:7128\n\t" "ds_write_addtid_b32 %28 offset:7392\n\t" "ds_write_addtid_b32 %29 offset:7656\n\t"
        "ds_write_addtid_b32 %30 offset:7920\n\t" "ds_write_addtid_b32 %31 offset:8184\n\t"
        :
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]), "v"(v[10]),
          "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]), "v"(v[16]), "v"(v[17]), "v"(v[18]), "v"(v[19]), "v"(v[20]),
          "v"(v[21]), "v"(v[22]), "v"(v[23]), "v"(v[24]), "v"(v[25]), "v"(v[26]), "v"(v[27]), "v"(v[28]), "v"(v[29]), "v"(v[30]),
          "v"(v[31]), "s"(base_bytes)
        : "memory", "m0");
}

// MODE 0: swizzled ds_write_b32;  1: ds_write_addtid_b32;  2: nothing (loop overhead: the VALU that makes new values)
template <int MODE>
__global__ __launch_bounds__(256) void k_stage(float* out, unsigned long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, col = lane & 31, hh = lane >> 5;
    int off[2][4];
    for (int e = 0; e < 2; ++e)
        for (int j = 0; j < 4; ++j) {
            const int sw = (4 * e + ((j + 4 * hh) >> 1)) & 7;
            off[e][j] = (((col >> 2) ^ sw) & 7) * 4 + (col & 3) + 32 * (j + 4 * hh);
        }
    float v[32];
    for (int m = 0; m < 32; ++m) v[m] = (float)(lane + 64 * m + 1);
    float* mine = lds + wid * (MODE == 1 ? TILE_B : 2048);
    const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)mine);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) stage_b32(mine, off, v);
        if (MODE == 1) stage_addtid(base, v);
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 32; ++m) v[m] += 1.0f;       // new values every time (32 VALU: same in every mode)
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 4 + wid] = t1 - t0;
    float s = 0.f;
    for (int m = 0; m < 32; ++m) s += v[m];
    out[blockIdx.x * 256 + threadIdx.x] = s + lds[threadIdx.x];
}

// layout check of (b): every wave stages v[m] = code(tile, m, lane); the block dumps its LDS
__global__ __launch_bounds__(256) void k_check(float* dump) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float v[32];
    for (int m = 0; m < 32; ++m) v[m] = (float)(wid * 100000 + m * 100 + lane);
    stage_addtid((unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lds + wid * TILE_B)), v);
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * TILE_B; i += 256) dump[i] = lds[i];
}

// consumer: one 32-row A fragment stream + B stream over NT tiles, 16 MFMAs per tile.  MODE 0: 4 + 4 ds_read_b128 at
// swizzled chunk addresses; MODE 1: 8 + 8 ds_read_b64 at immediate offsets of ONE address register per operand.
template <int MODE>
__global__ __launch_bounds__(256) void k_read(float* out, unsigned long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 8 * TILE_B; i += 256) lds[i] = 0.001f * (float)(i % 113);
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, i = lane & 31, kk = lane >> 5;
    f32x16_t acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 0) {
        const int sw = (i >> 1) & 7;
        unsigned aa[4], ab[4];
        for (int q = 0; q < 4; ++q) {
            aa[q] = (unsigned)(size_t)(lds + ((wid >> 1) * 32 + i) * 32 + ((4 * kk + q) ^ sw) * 4);
            ab[q] = aa[q] + 32768 + ((wid & 1) - (wid >> 1)) * 4096;
        }
        for (int it = 0; it < iters; ++it) {
            asm volatile(
                "ds_read_b128 v[32:35], %1 offset:0\n\t" "ds_read_b128 a[240:243], %5 offset:0\n\t"
                "ds_read_b128 v[36:39], %2 offset:0\n\t" "ds_read_b128 a[244:247], %6 offset:0\n\t"
                "ds_read_b128 v[40:43], %3 offset:0\n\t" "ds_read_b128 a[248:251], %7 offset:0\n\t"
                "ds_read_b128 v[44:47], %4 offset:0\n\t" "ds_read_b128 a[252:255], %8 offset:0\n\t"
                "s_waitcnt lgkmcnt(6)\n\t"
                "v_mfma_f32_32x32x2_f32 %0, v32, a240, %0\n\t" "v_mfma_f32_32x32x2_f32 %0, v33, a241, %0\n\t"
                "v_mfma_f32_32x32x2_f32 %0, v34, a242, %0\n\t" "v_mfma_f32_32x32x2_f32 %0, v35, a243, %0\n\t"
                "ds_read_b128 v[32:35], %1 offset:8192\n\t" "ds_read_b128 a[240:243], %5 offset:8192\n\t"
                "s_waitcnt lgkmcnt(6)\n\t"
                "v_mfma_f32_32x32x2_f32 %0, v36, a244, %0\n\t" "v_mfma_f32_32x32x2_f32 %0, v37, a245, %0\n\t"
                "v_mfma_f32_32x32x2_f32 %0, v38, a246, %0\n\t" "v_mfma_f32_32x32x2_f32 %0, v39, a247, %0\n\t"
                "ds_read_b128 v[36:39], %2 offset:8192\n\t" "ds_read_b128 a[244:247], %6 offset:8192\n\t"
                "s_waitcnt lgkmcnt(6)\n\t"
                "v_mfma_f32_32x32x2_f32 %0, v40, a248, %0\n\t" "v_mfma_f32_32x32x2_f32 %0, v41, a249, %0\n\t"
                "v_mfma_f32_32x32x2_f32 %0, v42, a250, %0\n\t" "v_mfma_f32_32x32x2_f32 %0, v43, a251, %0\n\t"
                "ds_read_b128 v[40:43], %3 offset:8192\n\t" "ds_read_b128 a[248:251], %7 offset:8192\n\t"
                "s_waitcnt lgkmcnt(6)\n\t"
                "v_mfma_f32_32x32x2_f32 %0, v44, a252, %0\n\t" "v_mfma_f32_32x32x2_f32 %0, v45, a253, %0\n\t"
                "v_mfma_f32_32x32x2_f32 %0, v46, a254, %0\n\t" "v_mfma_f32_32x32x2_f32 %0, v47, a255, %0\n\t"
                "ds_read_b128 v[44:47], %4 offset:8192\n\t" "ds_read_b128 a[252:255], %8 offset:8192\n\t"
                "s_waitcnt lgkmcnt(6)\n\t"
                "v_mfma_f32_32x32x2_f32 %0, v32, a240, %0\n\t" "v_mfma_f32_32x32x2_f32 %0, v33, a241, %0\n\t"
                "v_mfma_f32_32x32x2_f32 %0, v34, a242, %0\n\t" "v_mfma_f32_32x32x2_f32 %0, v35, a243, %0\n\t"
                "s_waitcnt lgkmcnt(4)\n\t"
                "v_mfma_f32_32x32x2_f32 %0, v36, a244, %0\n\t" "v_mfma_f32_32x32x2_f32 %0, v37, a245, %0\n\t"
                "v_mfma_f32_32x32x2_f32 %0, v38, a246, %0\n\t" "v_mfma_f32_32x32x2_f32 %0, v39, a247, %0\n\t"
                "s_waitcnt lgkmcnt(2)\n\t"
                "v_mfma_f32_32x32x2_f32 %0, v40, a248, %0\n\t" "v_mfma_f32_32x32x2_f32 %0, v41, a249, %0\n\t"
                "v_mfma_f32_32x32x2_f32 %0, v42, a250, %0\n\t" "v_mfma_f32_32x32x2_f32 %0, v43, a251, %0\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "v_mfma_f32_32x32x2_f32 %0, v44, a252, %0\n\t" "v_mfma_f32_32x32x2_f32 %0, v45, a253, %0\n\t"
                "v_mfma_f32_32x32x2_f32 %0, v46, a254, %0\n\t" "v_mfma_f32_32x32x2_f32 %0, v47, a255, %0\n\t"
                "s_nop 15\n\ts_nop 3"
                : "+a"(acc)
                : "v"(aa[0]), "v"(aa[1]), "v"(aa[2]), "v"(aa[3]), "v"(ab[0]), "v"(ab[1]), "v"(ab[2]), "v"(ab[3])
                : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47",
                  "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253",
                  "a254", "a255");
        }
    } else {
        // row i of row tile ra: pair r = (i & 3) + 4 (i >> 3), half (i >> 2) & 1; this lane's 16 particles start at 16 kk
        const int r = (i & 3) + 4 * (i >> 3), h2 = (i >> 2) & 1;
        const unsigned a0 = (unsigned)(size_t)(lds + (16 * (wid >> 1) + r) * PS + 32 * h2 + 16 * kk);
        const unsigned b0 = a0 + 4 * TILE_B * 4 + ((wid & 1) - (wid >> 1)) * 16 * PS * 4;
        for (int it = 0; it < iters; ++it) {
            // 2 tiles: chunk c (2 particles) of tile 1 refills the registers of chunk c of tile 0
            asm volatile(
                "ds_read_b64 v[32:33], %1 offset:0\n\t"  "ds_read_b64 a[240:241], %2 offset:0\n\t"
                "ds_read_b64 v[34:35], %1 offset:8\n\t"  "ds_read_b64 a[242:243], %2 offset:8\n\t"
                "ds_read_b64 v[36:37], %1 offset:16\n\t" "ds_read_b64 a[244:245], %2 offset:16\n\t"
                "ds_read_b64 v[38:39], %1 offset:24\n\t" "ds_read_b64 a[246:247], %2 offset:24\n\t"
                "ds_read_b64 v[40:41], %1 offset:32\n\t" "ds_read_b64 a[248:249], %2 offset:32\n\t"
                "ds_read_b64 v[42:43], %1 offset:40\n\t" "ds_read_b64 a[250:251], %2 offset:40\n\t"
                "ds_read_b64 v[44:45], %1 offset:48\n\t" "ds_read_b64 a[252:253], %2 offset:48\n\t"
                "ds_read_b64 v[46:47], %1 offset:56\n\t" "ds_read_b64 a[254:255], %2 offset:56\n\t"
                "s_waitcnt lgkmcnt(14)\n\t"
                "v_mfma_f32_32x32x2_f32 %0, v32, a240, %0\n\t" "v_mfma_f32_32x32x2_f32 %0, v33, a241, %0\n\t"
                "ds_read_b64 v[32:33], %1 offset:8448\n\t"  "ds_read_b64 a[240:241], %2 offset:8448\n\t"
                "s_waitcnt lgkmcnt(14)\n\t"
                "v_mfma_f32_32x32x2_f32 %0, v34, a242, %0\n\t" "v_mfma_f32_32x32x2_f32 %0, v35, a243, %0\n\t"
                "ds_read_b64 v[34:35], %1 offset:8456\n\t"  "ds_read_b64 a[242:243], %2 offset:8456\n\t"
                "s_waitcnt lgkmcnt(14)\n\t"
                "v_mfma_f32_32x32x2_f32 %0, v36, a244, %0\n\t" "v_mfma_f32_32x32x2_f32 %0, v37, a245, %0\n\t"
                "ds_read_b64 v[36:37], %1 offset:8464\n\t"  "ds_read_b64 a[244:245], %2 offset:8464\n\t"
                "s_waitcnt lgkmcnt(14)\n\t"
                "v_mfma_f32_32x32x2_f32 %0, v38, a246, %0\n\t" "v_mfma_f32_32x32x2_f32 %0, v39, a247, %0\n\t"
                "ds_read_b64 v[38:39], %1 offset:8472\n\t"  "ds_read_b64 a[246:247], %2 offset:8472\n\t"
                "s_waitcnt lgkmcnt(14)\n\t"
                "v_mfma_f32_32x32x2_f32 %0, v40, a248, %0\n\t" "v_mfma_f32_32x32x2_f32 %0, v41, a249, %0\n\t"
                "ds_read_b64 v[40:41], %1 offset:8480\n\t"  "ds_read_b64 a[248:249], %2 offset:8480\n\t"
                "s_waitcnt lgkmcnt(14)\n\t"
                "v_mfma_f32_32x32x2_f32 %0, v42, a250, %0\n\t" "v_mfma_f32_32x32x2_f32 %0, v43, a251, %0\n\t"
                "ds_read_b64 v[42:43], %1 offset:8488\n\t"  "ds_read_b64 a[250:251], %2 offset:8488\n\t"
                "s_waitcnt lgkmcnt(14)\n\t"
                "v_mfma_f32_32x32x2_f32 %0, v44, a252, %0\n\t" "v_mfma_f32_32x32x2_f32 %0, v45, a253, %0\n\t"
                "ds_read_b64 v[44:45], %1 offset:8496\n\t"  "ds_read_b64 a[252:253], %2 offset:8496\n\t"
                "s_waitcnt lgkmcnt(14)\n\t"
                "v_mfma_f32_32x32x2_f32 %0, v46, a254, %0\n\t" "v_mfma_f32_32x32x2_f32 %0, v47, a255, %0\n\t"
                "ds_read_b64 v[46:47], %1 offset:8504\n\t"  "ds_read_b64 a[254:255], %2 offset:8504\n\t"
                "s_waitcnt lgkmcnt(14)\n\t"
                "v_mfma_f32_32x32x2_f32 %0, v32, a240, %0\n\t" "v_mfma_f32_32x32x2_f32 %0, v33, a241, %0\n\t"
                "s_waitcnt lgkmcnt(12)\n\t"
                "v_mfma_f32_32x32x2_f32 %0, v34, a242, %0\n\t" "v_mfma_f32_32x32x2_f32 %0, v35, a243, %0\n\t"
                "s_waitcnt lgkmcnt(10)\n\t"
                "v_mfma_f32_32x32x2_f32 %0, v36, a244, %0\n\t" "v_mfma_f32_32x32x2_f32 %0, v37, a245, %0\n\t"
                "s_waitcnt lgkmcnt(8)\n\t"
                "v_mfma_f32_32x32x2_f32 %0, v38, a246, %0\n\t" "v_mfma_f32_32x32x2_f32 %0, v39, a247, %0\n\t"
                "s_waitcnt lgkmcnt(6)\n\t"
                "v_mfma_f32_32x32x2_f32 %0, v40, a248, %0\n\t" "v_mfma_f32_32x32x2_f32 %0, v41, a249, %0\n\t"
                "s_waitcnt lgkmcnt(4)\n\t"
                "v_mfma_f32_32x32x2_f32 %0, v42, a250, %0\n\t" "v_mfma_f32_32x32x2_f32 %0, v43, a251, %0\n\t"
                "s_waitcnt lgkmcnt(2)\n\t"
                "v_mfma_f32_32x32x2_f32 %0, v44, a252, %0\n\t" "v_mfma_f32_32x32x2_f32 %0, v45, a253, %0\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "v_mfma_f32_32x32x2_f32 %0, v46, a254, %0\n\t" "v_mfma_f32_32x32x2_f32 %0, v47, a255, %0\n\t"
                "s_nop 15\n\ts_nop 3"
                : "+a"(acc)
                : "v"(a0), "v"(b0)
                : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47",
                  "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253",
                  "a254", "a255");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 4 + wid] = t1 - t0;
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <class K>
static double run(K kern, const char* name, int iters, double ideal, const char* unit) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 4 * 8);
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * TILE_B * 4);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kern, dim3(256), dim3(256), 8 * TILE_B * 4, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto c : h) m += (double)c; m /= h.size();
    printf("%-64s %9.1f cycles per %s (ideal %.0f)  [%s]\n", name, m / iters, unit, ideal, hipGetErrorString(hipGetLastError()));
    hipFree(out); hipFree(cyc);
    return m / iters;
}

int main() {
    // layout check
    float* dump; hipMalloc(&dump, 4 * TILE_B * 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_check), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * TILE_B * 4);
    hipLaunchKernelGGL(k_check, dim3(1), dim3(256), 8 * TILE_B * 4, 0, dump);
    std::vector<float> h(4 * TILE_B);
    hipMemcpy(h.data(), dump, h.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < 4; ++w)
        for (int m = 0; m < 32; ++m)
            for (int lane = 0; lane < 64; ++lane) {
                const float want = (float)(w * 100000 + m * 100 + lane);
                const float got = h[w * TILE_B + m * PS + lane];            // pair m, half lane >> 5, particle lane & 31
                if (want != got && bad++ < 5) printf("  addtid layout mismatch: tile %d pair %d lane %d: got %.0f want %.0f\n", w, m, lane, got, want);
            }
    printf("ds_write_addtid_b32 layout check: %s\n", bad ? "MISMATCH" : "ok (pair*66 + lane)");
    const int iters = 2000;
    const double o = run(k_stage<2>, "staging loop overhead (32 VALU + 2 barriers)", iters, 128, "event");
    const double a = run(k_stage<0>, "stage 64x32 tile: 32 x ds_write_b32 (swizzled addresses)", iters, 128, "event");
    const double b = run(k_stage<1>, "stage 64x32 tile: 32 x ds_write_addtid_b32 (M0 + imm)", iters, 128, "event");
    printf("  -> staging event net of loop overhead: ds_write_b32 %.1f, ds_write_addtid_b32 %.1f cycles\n", a - o, b - o);
    run(k_read<0>, "2-tile product: ds_read_b128 swizzled, 32 MFMAs", iters, 2048, "product");
    run(k_read<1>, "2-tile product: ds_read_b64 plain (pair stride 66), 32 MFMAs", iters, 2048, "product");
    return 0;
}
