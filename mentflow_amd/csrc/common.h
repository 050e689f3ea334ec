// Shared helpers for the gfx950 kernels of libmentflow_hip.so.
#pragma once
#ifndef MF_EMU
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/mentflow_hip.h"

namespace mf {

// error text of the last failing entry point on this host thread
extern thread_local char g_err[512];
int fail(const char* fmt, ...);
int check_launch(const char* what);

constexpr int WAVE = 64;
constexpr int NUM_CU = 256;   // MI355X: 8 XCDs x 32 CUs

#ifdef MF_EMU
#define MF_LAUNCH(kernel, grid, block, smem, stream, ...) \
    emu::launch(dim3(grid), dim3(block), (smem), [&]() { kernel(__VA_ARGS__); })
#define MF_DYN_SMEM(type, name) type* name = reinterpret_cast<type*>(((uintptr_t)emu::g_block->dyn_smem + 63) & ~(uintptr_t)63)
#define MF_ALLOW_DYN_SMEM(kernel, bytes) ((void)0)
#define MF_WAVES_PER_SIMD(lo, hi)
#else
// register budget of a kernel stated as the occupancy it has to keep (waves per SIMD)
#define MF_WAVES_PER_SIMD(lo, hi) __attribute__((amdgpu_waves_per_eu(lo, hi)))
#define MF_LAUNCH(kernel, grid, block, smem, stream, ...) \
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), (smem), (hipStream_t)(stream), __VA_ARGS__)
#define MF_DYN_SMEM(type, name) extern __shared__ __attribute__((aligned(16))) unsigned char name##_raw_[]; \
    type* name = reinterpret_cast<type*>(name##_raw_)
// LDS beyond 64 KiB per workgroup has to be requested explicitly (gfx950 allows 160 KiB)
#define MF_ALLOW_DYN_SMEM(kernel, bytes)                                                                              \
    do {                                                                                                              \
        static size_t allowed_ = 0;        /* raised once per kernel (and again only if a larger size is needed) */    \
        if ((size_t)(bytes) > allowed_) {                                                                             \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)(bytes));                                                                  \
            allowed_ = (size_t)(bytes);                                                                               \
        }                                                                                                             \
    } while (0)
#endif

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// Optional per-kernel timing with HIP events recorded on the launch stream (mf_prof_enable / mf_prof_report).
enum ProfKernel { PK_FLOW_FWD = 0, PK_FLOW_BWD, PK_OUTER_ACCUM, PK_KDE1D_FWD, PK_KDE1D_BWD, PK_KDE2D_FWD, PK_KDE2D_BWD,
                  PK_COUNT };
struct ProfScope {
    int id;
    void* stream;
    void* ev0 = nullptr;
    ProfScope(int id_, void* stream_);
    ~ProfScope();
};

}  // namespace mf
