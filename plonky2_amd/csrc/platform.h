// platform.h -- the one switch between the real target (hipcc, gfx950) and the test-only
// kernel-source emulator (tests/emu, g++ -DP2HOT_EMU; never part of the product library).
#pragma once
#include <stddef.h>
#include <stdint.h>

#ifdef P2HOT_EMU
#include "hip_emu.h"
#define P2HOT_LAUNCH(kernel, grid, block, shmem, stream, ...) \
    emu::launch_on((stream), (grid), (block), (shmem), emu::bind_kernel(kernel, __VA_ARGS__))
#define P2HOT_DYN_SHARED(type, name) type *name = reinterpret_cast<type *>(emu::dyn_shared)
#else
#include <hip/hip_runtime.h>
#define P2HOT_LAUNCH(kernel, grid, block, shmem, stream, ...) \
    hipLaunchKernelGGL(kernel, (grid), (block), (shmem), (stream), __VA_ARGS__)
#define P2HOT_DYN_SHARED(type, name) extern __shared__ __attribute__((aligned(16))) unsigned char name##_raw[]; \
    type *name = reinterpret_cast<type *>(name##_raw)
#endif
