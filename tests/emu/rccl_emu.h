// rccl_emu.h -- TEST INFRASTRUCTURE ONLY: the fake RCCL of the kernel emulator (hip_emu_rt.cpp).  The emulator build of
// plonky2_amd/csrc/host_multi.hpp binds these where the product dlopen()s librccl: ncclCommInitAll / ncclCommInitRank /
// grouped ncclBroadcast / ncclAllGather with per-rank device, stream and buffer identity enforced, and the call patterns
// that hang the real library (a multi-rank collective outside a group from one thread, ranks posting different sequences)
// reported as errors.  Multi-rank ncclCommInitRank = one PROCESS per rank over POSIX shared memory (tests/test_emu_rccl_ranks.py).
#pragma once
#include <stddef.h>
extern "C" {
int emu_ncclGetUniqueId(char *id128);
int emu_ncclCommInitRank(void **comm, int nranks, const char *id128, int rank);
int emu_ncclCommInitAll(void **comms, int ndev, const int *devlist);
int emu_ncclCommDestroy(void *comm);
int emu_ncclBroadcast(const void *send, void *recv, size_t count, int dtype, int root, void *comm, void *stream);
int emu_ncclAllGather(const void *send, void *recv, size_t sendcount, int dtype, void *comm, void *stream);
int emu_ncclGroupStart();
int emu_ncclGroupEnd();
const char *emu_ncclGetErrorString(int rc);
void p2hot_emu_stats(unsigned long long out[8]);
int p2hot_emu_fault(const char *what, int on);
int p2hot_emu_set_device(int d);
int p2hot_emu_device_of(const void *p);
}
