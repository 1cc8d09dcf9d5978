// hip_emu.cpp -- TEST INFRASTRUCTURE ONLY (see hip_emu.h).
#include "hip_emu.h"

#include <cstdio>
#include <cstdlib>

namespace emu {
emu_uint3 threadIdx_, blockIdx_;
dim3 blockDim_, gridDim_;
unsigned char *dyn_shared = nullptr;

namespace {
// Context switches: on x86-64 a hand-written callee-saved-register swap (glibc's swapcontext makes two sigprocmask system
// calls per switch, which dominated the CPU test tier); elsewhere, or with -DEMU_UCONTEXT (the ASAN build), ucontext.
#if defined(__x86_64__) && !defined(EMU_UCONTEXT)
#define EMU_ASM_SWITCH 1
extern "C" void emu_switch(void **save_sp, void *load_sp);
asm(".text\n"
    ".globl emu_switch\n"
    ".type emu_switch,@function\n"
    "emu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n"
    "  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n"
    ".size emu_switch,.-emu_switch\n");
#endif

struct Fiber {
#ifdef EMU_ASM_SWITCH
    void *sp = nullptr;
#else
    ucontext_t ctx;
#endif
    std::vector<unsigned char> stack;
    bool done = false;
    bool started = false;
    emu_uint3 tid;
};
#ifdef EMU_ASM_SWITCH
void *sched_sp = nullptr;
#else
ucontext_t sched_ctx;
#endif
Fiber *current = nullptr;
const std::function<void()> *current_body = nullptr;
std::vector<uint64_t> shfl_slots;

void to_scheduler() {
#ifdef EMU_ASM_SWITCH
    emu_switch(&current->sp, sched_sp);
#else
    swapcontext(&current->ctx, &sched_ctx);
#endif
}

void trampoline() {
    (*current_body)();
    current->done = true;
    to_scheduler();
    abort();  // a finished fiber is never resumed
}

void start_or_resume(Fiber &f) {
#ifdef EMU_ASM_SWITCH
    if (!f.started) {
        f.started = true;
        // initial frame: six callee-saved registers, then the return address emu_switch's `ret` jumps to; after the pops and
        // the ret the stack pointer is 8 mod 16, as at any function entry
        uintptr_t top = ((uintptr_t)f.stack.data() + f.stack.size()) & ~(uintptr_t)15;
        void **sp = (void **)(top - 8 * 8);
        for (int i = 0; i < 6; ++i) sp[i] = nullptr;
        sp[6] = (void *)trampoline;
        sp[7] = nullptr;
        f.sp = sp;
    }
    emu_switch(&sched_sp, f.sp);
#else
    if (!f.started) {
        f.started = true;
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack.data();
        f.ctx.uc_stack.ss_size = f.stack.size();
        f.ctx.uc_link = &sched_ctx;
        makecontext(&f.ctx, (void (*)())trampoline, 0);
    }
    swapcontext(&sched_ctx, &f.ctx);
#endif
}
}  // namespace

bool direct_mode = false;     // the running thread is a plain call on the scheduler's stack (no fiber to yield from)
bool fiber_yielded = false;   // the fiber that is running has reached a barrier at least once

void barrier() {
    if (direct_mode) {  // thread 0 of this block finished without a barrier, another thread reached one: not expressible here
        fprintf(stderr, "hip_emu: __syncthreads / shuffle reached by thread (%u,%u,%u) of block (%u,%u,%u) but not by thread 0\n",
                threadIdx_.x, threadIdx_.y, threadIdx_.z, blockIdx_.x, blockIdx_.y, blockIdx_.z);
        abort();
    }
    fiber_yielded = true;
    // yield to the scheduler; it resumes us once every live fiber of the block has yielded
    to_scheduler();
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
    static std::vector<Fiber> fibers;  // stacks are kept between launches (one host thread drives the emulator)
    if (fibers.size() < nthreads) {
        const size_t have = fibers.size();
        fibers.resize(nthreads);
        for (size_t i = have; i < nthreads; ++i) fibers[i].stack.resize(stack_bytes);
    }
    std::vector<unsigned char> dyn(shmem + 16);
    shfl_slots.assign(nthreads, 0);
    dyn_shared = dyn.data();
    blockDim_ = block;
    gridDim_ = grid;
    current_body = &body;
    // P2HOT_EMU_THREADS=reverse: blocks and the threads of a block are visited in DESCENDING order.  Between two barriers the
    // fibers run one after the other, so a read of shared or global memory that another thread writes in the same interval sees
    // "written" in one visiting order and "not yet written" in the other: a missing __syncthreads shows in at least one of them
    static const bool reverse = getenv("P2HOT_EMU_THREADS") && !strcmp(getenv("P2HOT_EMU_THREADS"), "reverse");
    std::vector<size_t> order(nthreads);
    for (size_t i = 0; i < nthreads; ++i) order[i] = reverse ? nthreads - 1 - i : i;
    for (unsigned bz_ = 0; bz_ < grid.z; ++bz_)
        for (unsigned by_ = 0; by_ < grid.y; ++by_)
            for (unsigned bx_ = 0; bx_ < grid.x; ++bx_) {
                const unsigned bx = reverse ? grid.x - 1 - bx_ : bx_, by = reverse ? grid.y - 1 - by_ : by_, bz = reverse ? grid.z - 1 - bz_ : bz_;
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
                bool first_pass = true;
                while (live) {
                    for (size_t k = 0; k < nthreads; ++k) {
                        const size_t i = order[k];
                        Fiber &f = fibers[i];
                        if (f.done) continue;
                        current = &f;
                        threadIdx_ = f.tid;
                        if (first_pass && k == 0) fiber_yielded = false;
                        start_or_resume(f);
                        if (f.done) --live;
                        if (first_pass && k == 0 && f.done && !fiber_yielded) {
                            // the first thread ran to completion without a barrier: the block is barrier-free (a barrier that only
                            // other threads reach aborts loudly), so its other threads run as plain calls -- no context switches
                            direct_mode = true;
                            for (size_t kk = 1; kk < nthreads; ++kk) {
                                const size_t j = order[kk];
                                threadIdx_ = fibers[j].tid;
                                body();
                                fibers[j].done = true;
                            }
                            direct_mode = false;
                            live = 0;
                            break;
                        }
                    }
                    first_pass = false;
                }
            }
    dyn_shared = nullptr;
    current_body = nullptr;
}
}  // namespace emu
