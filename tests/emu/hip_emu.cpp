// hip_emu.cpp -- TEST INFRASTRUCTURE ONLY (see hip_emu.h).
#include "hip_emu.h"

namespace emu {
emu_uint3 threadIdx_, blockIdx_;
dim3 blockDim_, gridDim_;
unsigned char *dyn_shared = nullptr;

namespace {
struct Fiber {
    ucontext_t ctx;
    std::vector<unsigned char> stack;
    bool done = false;
    bool started = false;
    emu_uint3 tid;
};
ucontext_t sched_ctx;
Fiber *current = nullptr;
const std::function<void()> *current_body = nullptr;
std::vector<uint64_t> shfl_slots;

void trampoline() {
    (*current_body)();
    current->done = true;
    swapcontext(&current->ctx, &sched_ctx);
}
}  // namespace

void barrier() {
    // yield to the scheduler; it resumes us once every live fiber of the block has yielded
    swapcontext(&current->ctx, &sched_ctx);
}

uint64_t shfl_exchange(uint64_t v, int src_lane) {
    // all threads of the block must call this (wave-uniform control flow)
    unsigned lin = threadIdx_.x;
    shfl_slots[lin] = v;
    barrier();
    unsigned wave_base = lin & ~63u;
    uint64_t r = shfl_slots[wave_base + ((unsigned)src_lane & 63u)];
    barrier();
    return r;
}

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &body) {
    const size_t nthreads = (size_t)block.x * block.y * block.z;
    const size_t stack_bytes = 96 * 1024;
    std::vector<Fiber> fibers(nthreads);
    for (auto &f : fibers) f.stack.resize(stack_bytes);
    std::vector<unsigned char> dyn(shmem + 16);
    shfl_slots.assign(nthreads, 0);
    dyn_shared = dyn.data();
    blockDim_ = block;
    gridDim_ = grid;
    current_body = &body;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                blockIdx_ = {bx, by, bz};
                size_t t = 0;
                for (unsigned tz = 0; tz < block.z; ++tz)
                    for (unsigned ty = 0; ty < block.y; ++ty)
                        for (unsigned tx = 0; tx < block.x; ++tx, ++t) {
                            Fiber &f = fibers[t];
                            f.done = false;
                            f.started = false;
                            f.tid = {tx, ty, tz};
                        }
                size_t live = nthreads;
                while (live) {
                    for (auto &f : fibers) {
                        if (f.done) continue;
                        current = &f;
                        threadIdx_ = f.tid;
                        if (!f.started) {
                            f.started = true;
                            getcontext(&f.ctx);
                            f.ctx.uc_stack.ss_sp = f.stack.data();
                            f.ctx.uc_stack.ss_size = f.stack.size();
                            f.ctx.uc_link = &sched_ctx;
                            makecontext(&f.ctx, (void (*)())trampoline, 0);
                        }
                        swapcontext(&sched_ctx, &f.ctx);
                        if (f.done) --live;
                    }
                }
            }
    dyn_shared = nullptr;
    current_body = nullptr;
}
}  // namespace emu
