// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
//
// Minimal single-OS-thread emulation of the HIP / gfx950 constructs the kernels under
// mentflow_amd/csrc use, so that kernel *logic* (MFMA fragment maps, LDS images, cross-lane exchanges,
// closed-form adjoints) can be debugged and unit-tested in the GPU-less build container.
// Every GPU thread of a workgroup is a ucontext fiber; workgroups run one after another; wave-level
// collectives (MFMA, shuffles) and __syncthreads() are rendezvous points between fibers.
//
// The product library (libmentflow_hip.so) never includes this header; it is force-included only when
// tests/emu/build_emu.sh compiles the same sources with -DMF_EMU into tests/emu/libmentflow_emu.so.
// MFMA semantics follow /opt/skills/guides/cdna_hip_programming.md §3 (lane maps, k-ordered fmaf chain).
#pragma once
#ifndef MF_EMU
#error "hip_emu.h is test infrastructure; compile with -DMF_EMU"
#endif

#include <ucontext.h>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>
#include <algorithm>
using std::min;
using std::max;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
#define hipMemcpyDeviceToDevice 3
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return 0; }
#define hipFuncAttributeMaxDynamicSharedMemorySize 8

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }

namespace emu {

constexpr int WAVE = 64;
constexpr size_t STACK = 512 * 1024;

// AddressSanitizer has to be told about every stack switch of the fibers (tests/emu/build_sanitize.sh)
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define MF_EMU_ASAN 1
#endif
#endif
#ifdef MF_EMU_ASAN
extern "C" void __sanitizer_start_switch_fiber(void** fake_stack_save, const void* bottom, size_t size);
extern "C" void __sanitizer_finish_switch_fiber(void* fake_stack_save, const void** bottom_old, size_t* size_old);
extern "C" void __asan_poison_memory_region(void const volatile* addr, size_t size);
extern "C" void __asan_unpoison_memory_region(void const volatile* addr, size_t size);
#define MF_FIBER_START(save, bottom, size) __sanitizer_start_switch_fiber(save, bottom, size)
#define MF_FIBER_FINISH(save, bottom_old, size_old) __sanitizer_finish_switch_fiber(save, bottom_old, size_old)
#else
#define MF_FIBER_START(save, bottom, size) ((void)0)
#define MF_FIBER_FINISH(save, bottom_old, size_old) ((void)0)
#endif

// Context switch between the scheduler and the fibers.  glibc's swapcontext saves and restores the signal mask with a
// system call on EVERY switch (tens of millions per test run: most of the emulator's wall time was kernel time), so
// on x86-64 a six-register hand-written switch is used instead; other hosts keep ucontext.
#if defined(__x86_64__)
#define MF_EMU_FAST_SWITCH 1
struct Ctx {
    void* sp = nullptr;
};
extern "C" void mf_emu_ctx_switch(Ctx* from, Ctx* to);
#else
typedef ucontext_t Ctx;
#endif

struct Fiber {
    Ctx ctx;
    char* stack = nullptr;
    bool done = false;
    dim3 tid;
    void* fake_stack = nullptr;      // ASan fake-stack handle while this fiber is switched out
};

struct WaveState {
    float a[WAVE], b[WAVE];
    uint64_t u[WAVE];
    int arrived = 0;
    unsigned gen = 0;
};

struct BlockState {
    std::vector<Fiber> fibers;
    std::vector<WaveState> waves;
    int cur = 0;
    int barrier_arrived = 0;
    unsigned barrier_gen = 0;
    Ctx sched;
    std::function<void()> body;
    // dynamic LDS of the workgroup: EXACTLY the requested size, its end on a PROT_NONE guard page (and, under ASan,
    // the alignment slack in front of it poisoned): an out-of-range LDS index in a kernel faults on the CPU
    char* dyn_smem = nullptr;
    void* sched_fake_stack = nullptr;
    const void* sched_bottom = nullptr;
    size_t sched_size = 0;
};

extern BlockState* g_block;
extern dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;

inline void yield() {
    BlockState* b = g_block;
    Fiber& f = b->fibers[b->cur];
    MF_FIBER_START(&f.fake_stack, b->sched_bottom, b->sched_size);
#ifdef MF_EMU_FAST_SWITCH
    mf_emu_ctx_switch(&f.ctx, &b->sched);
#else
    swapcontext(&f.ctx, &b->sched);
#endif
    MF_FIBER_FINISH(f.fake_stack, nullptr, nullptr);
}

inline int linear_tid() { return g_threadIdx.x + g_blockDim.x * (g_threadIdx.y + g_blockDim.y * g_threadIdx.z); }
inline int lane_id() { return linear_tid() % WAVE; }
inline WaveState& wave() { return g_block->waves[linear_tid() / WAVE]; }

inline void wave_sync() {
    WaveState& w = wave();
    unsigned gen = w.gen;
    int nlanes = WAVE;
    int nthreads = (int)g_block->fibers.size();
    int wid = linear_tid() / WAVE;
    if ((wid + 1) * WAVE > nthreads) nlanes = nthreads - wid * WAVE;
    if (++w.arrived == nlanes) {
        w.arrived = 0;
        w.gen++;
    } else {
        while (w.gen == gen) yield();
    }
}

inline void block_sync() {
    BlockState* b = g_block;
    unsigned gen = b->barrier_gen;
    if (++b->barrier_arrived == (int)b->fibers.size()) {
        b->barrier_arrived = 0;
        b->barrier_gen++;
    } else {
        while (b->barrier_gen == gen) yield();
    }
}

void launch(dim3 grid, dim3 block, size_t smem, std::function<void()> body);

}  // namespace emu

#define threadIdx (emu::g_threadIdx)
#define blockIdx (emu::g_blockIdx)
#define blockDim (emu::g_blockDim)
#define gridDim (emu::g_gridDim)

static inline void __syncthreads() { emu::block_sync(); }

// ---- wave collectives -------------------------------------------------------------------------
static inline float __shfl(float v, int src, int width = 64) {
    (void)width;
    emu::WaveState& w = emu::wave();
    int l = emu::lane_id();
    w.a[l] = v;
    emu::wave_sync();
    float r = w.a[src & 63];
    emu::wave_sync();
    return r;
}
static inline int __shfl(int v, int src, int width = 64) {
    float f;
    memcpy(&f, &v, 4);
    f = __shfl(f, src, width);
    memcpy(&v, &f, 4);
    return v;
}
static inline float __shfl_xor(float v, int mask, int width = 64) { return __shfl(v, emu::lane_id() ^ mask, width); }
static inline int __shfl_xor(int v, int mask, int width = 64) { return __shfl(v, emu::lane_id() ^ mask, width); }
static inline float __shfl_down(float v, unsigned delta, int width = 64) {
    int l = emu::lane_id();
    int src = l + (int)delta;
    if (src >= 64) src = l;
    return __shfl(v, src, width);
}

// v_mfma_f32_32x32x2_f32:  A[i=l&31][k=l>>5], B[k=l>>5][j=l&31],
// C/D: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5);  D = fma(a_k1,b_k1, fma(a_k0,b_k0, C)).
static inline f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, f32x16 c, int, int, int) {
    emu::WaveState& w = emu::wave();
    int l = emu::lane_id();
    w.a[l] = a;
    w.b[l] = b;
    emu::wave_sync();
    int col = l & 31, hh = l >> 5;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
        float acc = c[r];
        acc = fmaf(w.a[row], w.b[col], acc);
        acc = fmaf(w.a[row + 32], w.b[col + 32], acc);
        c[r] = acc;
    }
    emu::wave_sync();
    return c;
}

// v_mfma_f32_16x16x4_f32: A[l&15][k=l>>4], B[k=l>>4][l&15]; C/D: col = lane&15, row = 4*(lane>>4)+reg.
static inline f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, f32x4 c, int, int, int) {
    emu::WaveState& w = emu::wave();
    int l = emu::lane_id();
    w.a[l] = a;
    w.b[l] = b;
    emu::wave_sync();
    int col = l & 15, q = l >> 4;
    for (int r = 0; r < 4; ++r) {
        int row = 4 * q + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(w.a[row + 16 * k], w.b[col + 16 * k], acc);
        c[r] = acc;
    }
    emu::wave_sync();
    return c;
}

// ---- atomics (single OS thread: plain read-modify-write) ---------------------------------------
static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline double atomicAdd(double* p, double v) { double o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { auto o = *p; *p = o + v; return o; }

// ---- math intrinsics ---------------------------------------------------------------------------
static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline float __fdividef(float a, float b) { return a / b; }
static inline void __builtin_amdgcn_sched_group_barrier(int, int, int) {}
static inline void __builtin_amdgcn_s_sleep(int) {}
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }
static inline float __builtin_amdgcn_logf(float x) { return log2f(x); }
static inline float __builtin_amdgcn_fmed3f(float a, float b, float c) { return fmaxf(fminf(fmaxf(a, b), c), fminf(a, b)); }
static inline int __builtin_amdgcn_readfirstlane(int v) { return __shfl(v, 0); }
static inline float fminf_(float a, float b) { return a < b ? a : b; }
