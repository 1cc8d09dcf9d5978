// hip_emu.h -- TEST INFRASTRUCTURE ONLY.
//
// A tiny single-threaded emulator of the HIP execution model (grid of blocks, threads as
// ucontext fibers, __syncthreads, static/dynamic __shared__, a malloc-backed "device" heap)
// so that the *same kernel sources* under plonky2_amd/csrc can be compiled with g++ into
// tests/emu/libp2hot_emu.so and their index arithmetic checked against the oracle in the
// GPU-less build container (`pytest -m "not gpu"`).  It is never built into, linked with or
// loaded by the product library (libp2hot.so is hipcc/gfx950 only and fails loudly without a GPU).
// No wave intrinsics are emulated beyond 64-lane __shfl/__shfl_xor/__shfl_down within a block.
#pragma once
#include <ucontext.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct emu_uint3 {
    unsigned x, y, z;
};

namespace emu {
extern emu_uint3 threadIdx_, blockIdx_;
extern dim3 blockDim_, gridDim_;
extern unsigned char *dyn_shared;
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &body);
void barrier();
uint64_t shfl_exchange(uint64_t v, int src_lane);
}  // namespace emu

#define threadIdx emu::threadIdx_
#define blockIdx emu::blockIdx_
#define blockDim emu::blockDim_
#define gridDim emu::gridDim_

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __constant__ static const
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__

static inline void __syncthreads() { emu::barrier(); }
static inline uint64_t __umul64hi(uint64_t a, uint64_t b) {
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
}
static inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
static inline unsigned __brev(unsigned x) {
    unsigned r = 0;
    for (int i = 0; i < 32; ++i) r |= ((x >> i) & 1u) << (31 - i);
    return r;
}
static inline unsigned long long __brevll(unsigned long long x) {
    unsigned long long r = 0;
    for (int i = 0; i < 64; ++i) r |= ((x >> i) & 1ull) << (63 - i);
    return r;
}
static inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }

static inline unsigned long long atomicMin(unsigned long long *p, unsigned long long v) {
    unsigned long long o = *p;
    if (v < o) *p = v;
    return o;
}

static inline unsigned atomicOr(unsigned *p, unsigned v) {
    unsigned o = *p;
    *p = o | v;
    return o;
}

// ---- host API subset (malloc-backed) ----
typedef int hipError_t;
typedef void *hipStream_t;
typedef void *hipEvent_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) {
    *d = 0;
    return hipSuccess;
}
static inline hipError_t hipMalloc(void **p, size_t n) {
    *p = malloc(n ? n : 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
static inline hipError_t hipFree(void *p) {
    free(p);
    return hipSuccess;
}
enum { hipHostMallocDefault = 0 };
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { return hipMalloc(p, n); }
static inline hipError_t hipHostFree(void *p) { return hipFree(p); }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) {
    memmove(d, s, n);
    return hipSuccess;
}
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) {
    memmove(d, s, n);
    return hipSuccess;
}
static inline hipError_t hipMemcpy2DAsync(void *d, size_t dpitch, const void *s, size_t spitch, size_t width, size_t height,
                                          hipMemcpyKind, hipStream_t) {
    for (size_t r = 0; r < height; ++r) memmove((char *)d + r * dpitch, (const char *)s + r * spitch, width);
    return hipSuccess;
}
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) {
    memset(d, v, n);
    return hipSuccess;
}
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char *hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipMemGetInfo(size_t *f, size_t *t) {
    *f = *t = (size_t)1 << 34;
    return hipSuccess;
}
