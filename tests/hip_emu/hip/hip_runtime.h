// TEST INFRASTRUCTURE ONLY — a host stand-in for <hip/hip_runtime.h>, just large enough to compile the
// *element-wise* kernels of tacotron2_amd/csrc (audio.hip, optim.hip) for the CPU and execute them thread by thread
// (tests/hip_emu/emu_runtime.cpp).  It lets the CPU test-suite run the very kernel source the GPU runs — index
// arithmetic, bounds, reductions, rounding — when no GPU is at hand.  It is never built into, linked with or imported
// by the product; MFMA / LDS-DMA / DPP kernels cannot be emulated here (their builtins are stubs that abort).
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static          /* one workgroup runs at a time: block-shared = one static instance */

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct emu_idx { unsigned x, y, z; };
extern thread_local emu_idx threadIdx, blockIdx;
extern dim3 blockDim, gridDim;

typedef void* hipStream_t;
typedef void* hipEvent_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "emulated"; }

// just enough of the host API for tools/probe/gpu_selftest.cpp to be dry-run against the emulated kernels
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost };
struct hipDeviceProp_t { char gcnArchName[64]; int multiProcessorCount; };
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) { *p = (T*)malloc(n); return *p ? hipSuccess : 2; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { strcpy(p->gcnArchName, "host-emulation"); p->multiProcessorCount = 0; return hipSuccess; }

// the vector types the element-wise kernels use
struct float4 { float x, y, z, w; };
struct uint2 { unsigned x, y; };
static inline float4 make_float4(float x, float y, float z, float w) { float4 v = {x, y, z, w}; return v; }
static inline uint2 make_uint2(unsigned x, unsigned y) { uint2 v = {x, y}; return v; }

void emu_launch(dim3 grid, dim3 block, const std::function<void()>& body);
void emu_syncthreads();
void emu_exchange(const void* mine, void* partner, size_t bytes, int mask);   // wave-level xor exchange

#define hipLaunchKernelGGL(kern, grid, block, lds, stream, ...) emu_launch((grid), (block), [=]() { kern(__VA_ARGS__); })
#define __syncthreads() emu_syncthreads()

template <class T>
static inline T __shfl_xor(T v, int mask, int width = 64) {
    (void)width;
    T out;
    emu_exchange(&v, &out, sizeof(T), mask);
    return out;
}

// correctly rounded single operations (compile this target with -ffp-contract=off)
static inline float __fmul_rn(float a, float b) { return a * b; }
static inline float __fadd_rn(float a, float b) { return a + b; }
static inline unsigned __float_as_uint(float x) { unsigned u; memcpy(&u, &x, 4); return u; }
static inline float __uint_as_float(unsigned u) { float x; memcpy(&x, &u, 4); return x; }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __sync_fetch_and_add(p, v); }
static inline unsigned long long wall_clock64() { return 0; }
// scoped atomics (the SC1 forms of csrc/cell_bwd.h, never instantiated by the emulated kernels): plain accesses on the host
#define __HIP_MEMORY_SCOPE_AGENT 4
template <class T> static inline T __hip_atomic_load(const T* p, int, int) { return *p; }
template <class T, class V> static inline void __hip_atomic_store(T* p, V v, int, int) { *p = (T)v; }

// device-only builtins referenced by inline helpers of common.h that the emulated kernels never call
static inline float __builtin_amdgcn_rcpf(float) { abort(); }
static inline float __builtin_amdgcn_exp2f(float) { abort(); }
static inline int __builtin_amdgcn_update_dpp(int, int, int, int, int, bool) { abort(); }
