// Platform layer of the renderer core.
//
// The product build is HIP for gfx950 (hipcc, `make -C raytracer_amd/csrc`).  The same sources also
// compile with g++ under -DRT_EMU into tests/_emu/librtcore_emu.so: a single-threaded HOST EMULATION
// of the kernels used only by the CPU-side unit tests (`pytest -m "not gpu"`) to exercise queue logic
// and host orchestration where no GPU exists.  The emulation is never built into raytracer_amd/lib,
// never loaded by the raytracer_amd package and is not a fallback: the package raises if
// librtcore.so is missing.
#pragma once

#include <stdint.h>

#ifndef RT_EMU
#include <hip/hip_runtime.h>
#define RT_DEV __device__ __forceinline__
#define RT_HD __host__ __device__ __forceinline__
#define RT_MEM __device__ __forceinline__
#define RT_KERNEL(name, ...) __global__ void __launch_bounds__(256) name(__VA_ARGS__)
// a kernel held to `w` waves per SIMD (register budget 512 / w)
#define RT_KERNEL_W(w, name, ...) __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(w, w))) name(__VA_ARGS__)
#define RT_TID ((uint32_t)(blockIdx.x * blockDim.x + threadIdx.x))
#define RT_NTHREADS ((uint32_t)(gridDim.x * blockDim.x))
#define RT_LANE_IN_BLOCK ((uint32_t)threadIdx.x)
#define RT_BLOCK 256
#else
// ---- host emulation -------------------------------------------------------------------------
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#define RT_DEV static inline
#define RT_HD static inline
#define RT_MEM inline
#define RT_KERNEL(name, ...) static void name(__VA_ARGS__)
#define RT_KERNEL_W(w, name, ...) static void name(__VA_ARGS__)
#define RT_TID 0u
#define RT_NTHREADS 1u
#define RT_LANE_IN_BLOCK 0u
#define RT_BLOCK 1
struct float4 { float x, y, z, w; };
struct uint4 { uint32_t x, y, z, w; };
struct uint2 { uint32_t x, y; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float __uint_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline uint32_t __float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
typedef int hipError_t;
typedef void* hipStream_t;
struct EmuEvent { std::chrono::steady_clock::time_point t; };
typedef EmuEvent* hipEvent_t;
#define hipSuccess 0
#define hipMemcpyHostToDevice 1
#define hipMemcpyDeviceToHost 2
#define hipMemcpyDeviceToDevice 3
static inline hipError_t hipSetDevice(int) { return 0; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return 0; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? 0 : 2; }
static inline hipError_t hipFree(void* p) { std::free(p); return 0; }
static inline hipError_t hipHostMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? 0 : 2; }
static inline hipError_t hipHostFree(void* p) { std::free(p); return 0; }
static inline hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return 0; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { std::memcpy(d, s, n); return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { std::memcpy(d, s, n); return 0; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return 0; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return 0; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return 0; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
static inline hipError_t hipStreamQuery(hipStream_t) { return 0; }
static inline hipError_t hipDeviceSynchronize() { return 0; }
static inline hipError_t hipGetLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new EmuEvent; return 0; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return 0; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return 0; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return 0;
}
#endif

// ---- wave-uniform read-only data ------------------------------------------------------------------
// Scene tables that every lane of a wave reads at the same address (instance headers, transform chains, lights) are
// read through the constant address space, so that hipcc emits scalar loads (s_load_*, one per wave, results in
// SGPRs) instead of 64-lane vector loads whose latency sits in every ray's dependency chain.  The tables are
// written only by the host before any kernel runs.
#if !defined(RT_EMU) && !defined(RT_NO_UNIFORM)
#define RT_UTYPE(T) const __attribute__((address_space(4))) T
#define RT_UNIFORM_PTR(T, p) ((const __attribute__((address_space(4))) T*)(p))
#else
#define RT_UTYPE(T) const T
#define RT_UNIFORM_PTR(T, p) ((const T*)(p))
#endif

// ---- atomics used by the kernels (plain adds in the single-threaded emulation) ------------------
#ifndef RT_EMU
RT_DEV uint32_t rt_atomic_add_u32(uint32_t* p, uint32_t v) { return atomicAdd(p, v); }
RT_DEV void rt_atomic_add_u64(unsigned long long* p, unsigned long long v) { atomicAdd(p, v); }
RT_DEV void rt_atomic_add_f32(float* p, float v) { unsafeAtomicAdd(p, v); }
RT_DEV void rt_atomic_min_u32(uint32_t* p, uint32_t v) { atomicMin(p, v); }
RT_DEV void rt_atomic_max_u32(uint32_t* p, uint32_t v) { atomicMax(p, v); }
RT_DEV uint32_t rt_clz(uint32_t v) { return (uint32_t)__clz((int)v); }
#else
RT_DEV uint32_t rt_atomic_add_u32(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o + v; return o; }
RT_DEV void rt_atomic_add_u64(unsigned long long* p, unsigned long long v) { *p += v; }
RT_DEV void rt_atomic_add_f32(float* p, float v) { *p += v; }
RT_DEV void rt_atomic_min_u32(uint32_t* p, uint32_t v) { if (v < *p) *p = v; }
RT_DEV void rt_atomic_max_u32(uint32_t* p, uint32_t v) { if (v > *p) *p = v; }
RT_DEV uint32_t rt_clz(uint32_t v) { return v ? (uint32_t)__builtin_clz(v) : 32u; }
#endif
