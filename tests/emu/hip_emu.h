// hip_emu.h -- TEST INFRASTRUCTURE ONLY.
//
// A tiny single-threaded emulator of the HIP execution model (grid of blocks, threads as
// ucontext fibers, __syncthreads, static/dynamic __shared__, a malloc-backed "device" heap)
// so that the *same kernel sources* under plonky2_amd/csrc can be compiled with g++ into
// tests/emu/libp2hot_emu.so and their index arithmetic checked against the oracle in the
// GPU-less build container (`pytest -m "not gpu"`).  It is never built into, linked with or
// loaded by the product library (libp2hot.so is hipcc/gfx950 only and fails loudly without a GPU).
// No wave intrinsics are emulated beyond 64-lane __shfl/__shfl_xor/__shfl_down within a block.  Between two barriers a block's
// threads run one after the other, ascending; P2HOT_EMU_THREADS=reverse visits threads and blocks in descending order, so that a
// dependency on another thread's write inside a barrier interval (a missing __syncthreads) fails in at least one of the two orders.
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

// ---- host API subset: a small model of the HIP RUNTIME semantics the multi-GPU layer depends on (hip_emu.cpp) ----
// * P2HOT_EMU_DEVICES "GPUs" (default 8).  Each host thread has a current device (hipSetDevice / hipGetDevice).
// * hipMalloc returns page-aligned memory tagged with the device that was current; while another device is current the
//   pages are PROT_NONE, so a kernel, a host loop or a memcpy that touches another GPU's memory faults, and the fault handler
//   names the allocation, its device and the current device before aborting (an access from a thread whose OWN current device
//   owns the pages is let through: protection is process-wide, the current device is per thread).  Every allocation ENDS (to 16
//   bytes) at a guard page that is never accessible: an index that runs off the end of a device buffer is a named fault
//   ("DEVICE MEMORY OVERRUN: ... N bytes past the end of a B-byte allocation") at the first word, in any kernel, copy or host loop.
// * Streams and events belong to the device that was current when they were created.  A launch, copy, memset, event record
//   or stream wait issued on a stream of another device than the current one fails with hipErrorInvalidResourceHandle and a
//   message (hipGetErrorString); so does recording an event on a stream of another device, or using a destroyed handle.
// * Copies BETWEEN devices go through hipMemcpyAsync / hipMemcpy2DAsync only (explicit peer copies: counted, see emu::stats).
// * A fake RCCL (ncclCommInitAll / ncclCommInitRank / grouped ncclBroadcast / ncclAllGather, rccl_emu.h) enforces per-rank
//   device, stream and buffer identity and refuses collectives that would hang on the real library.  ncclCommInitRank with
//   nranks > 1 joins PROCESSES (one per rank, device ids node-global) through a shared-memory segment named by the unique id;
//   every collective is announced to all ranks before data moves, and a rank that never arrives is a timeout
//   (P2HOT_EMU_RCCL_TIMEOUT_MS, default 60 s), not a hang.
// * Streams are QUEUES.  Launches, asynchronous copies, memsets, event records, stream waits and collectives are appended to
//   their stream and run only when somebody needs the result (a synchronisation, a synchronous copy, hipFree, another stream's
//   wait on an event recorded behind them): the as-late-as-legal schedule.  A consumer that forgot its hipStreamWaitEvent runs
//   before its producer; a host read before the synchronisation sees old bytes; kernel arguments are captured by value at
//   launch.  P2HOT_EMU_ASYNC=random:<seed> adds random legal steps (other schedules), =0 restores immediate execution.
//   Copies follow the runtime's documented rules (pageable source captured at the call, pageable destination synchronous,
//   pinned / device memory in stream order).  A collective of an in-process communicator executes when EVERY rank's stream has
//   reached it.  One adversarial schedule plus random ones, not an exhaustive exploration.
typedef int hipError_t;
typedef void *hipStream_t;
typedef void *hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorInvalidDevice = 101, hipErrorInvalidResourceHandle = 400 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipHostMallocDefault = 0, hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipEventBlockingSync = 1 };

namespace emu {
struct Stats {
    unsigned long long peer_copies = 0, peer_bytes = 0;        // hipMemcpy* between two devices' allocations
    unsigned long long nccl_broadcasts = 0, nccl_allgathers = 0, nccl_bytes = 0;
    unsigned long long device_switches = 0, violations = 0;    // violations: API calls refused for a device / handle mismatch
};
extern Stats stats;
extern bool fault_drop_waits;        // test hook (p2hot_emu_fault "drop_stream_waits"): hipStreamWaitEvent does nothing
extern bool fault_no_device_guard;  // test hook (p2hot_emu_fault): DeviceGuard becomes a no-op, so a mis-guarded entry point is visible
int current_device();
int device_of(const void *p);       // device owning the allocation that contains p, -1: host / unknown memory
// kernel launch on `stream`: refused (sticky error for hipGetLastError) unless the stream belongs to the current device
void launch_on(hipStream_t stream, dim3 grid, dim3 block, size_t shmem, std::function<void()> body);
// the kernel with its arguments captured BY VALUE at launch time (as a real launch marshals them): the call itself is queued
template <class K, class... A>
std::function<void()> bind_kernel(K k, A... a) {
    return [=]() { k(a...); };
}
}  // namespace emu

hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int *d);
hipError_t hipGetDeviceCount(int *n);
hipError_t hipDeviceEnablePeerAccess(int peer, unsigned flags);
hipError_t hipMalloc(void **p, size_t n);
hipError_t hipFree(void *p);
hipError_t hipHostMalloc(void **p, size_t n, unsigned flags);
hipError_t hipHostFree(void *p);
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind kind, hipStream_t stream);
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind kind);
hipError_t hipMemcpy2DAsync(void *d, size_t dpitch, const void *s, size_t spitch, size_t width, size_t height, hipMemcpyKind kind,
                            hipStream_t stream);
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t stream);
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags);
hipError_t hipEventCreate(hipEvent_t *e);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);
hipError_t hipDeviceSynchronize();
hipError_t hipGetLastError();
const char *hipGetErrorString(hipError_t e);
hipError_t hipMemGetInfo(size_t *f, size_t *t);
