// gcn_asm.h -- TEST INFRASTRUCTURE ONLY (part of the kernel-source emulator, tests/emu).
//
// An interpreter for the handful of gfx950 VALU instructions the product's inline-asm blocks use, so that the CPU test tier
// executes THE INSTRUCTION STREAMS THAT SHIP (the very template strings hipcc assembles: gl_mul3.hpp mul3 / mul1 / fold3 /
// fold1, poseidon.hpp mds_term / mds_first, ntt.hpp mul_pow2_asm) instead of only their C fallbacks, and checks statically
// what the hand scheduling has to guarantee on this target:
//   * the gfx940+ hazard "VALU writes an SGPR -> a VALU reads it: 2 wait states" (LLVM GCNHazardRecognizer,
//     VALUWriteSGPRVALUReadWaitstates = 2, hasVDecCoExecHazard) -- the assembler inserts nothing inside inline asm;
//   * every physical register a block writes is in its clobber list, and none is read before the block wrote it;
//   * no operand is read before it is defined (early-clobber outputs start undefined).
// One lane is interpreted at a time (the emulator's lanes are fibers), so an SGPR pair that holds a wave's carry mask on
// the GPU holds this lane's carry bit here.  The product sources reach this through csrc/asm_block.h (P2_ASM): the real
// build expands to a GNU asm statement, the emulator build to gcn::run(...) with the same template, operands and clobbers.
// The interpreter is OFF by default (the C fallbacks are 30x faster); tests switch it on with p2hot_emu_asm(1).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <map>
#include <string>
#include <vector>

namespace gcn {

struct Arg {
    const char *name;        // "[ra0]" or "" (positional)
    const char *constraint;  // "=&v", "+v", "=&s", "+s", "v", "s", "n"
    void *ptr;               // outputs / in-outs: where the value lives
    uint64_t value;          // inputs: the value
    int bits;                // 32 or 64
};
template <class T>
inline Arg out(const char *name, const char *constraint, T &var) {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "asm operand size");
    return Arg{name, constraint, (void *)&var, 0, (int)sizeof(T) * 8};
}
template <class T>
inline Arg in(const char *name, const char *constraint, T value) {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "asm operand size");
    return Arg{name, constraint, nullptr, (uint64_t)value, (int)sizeof(T) * 8};
}

struct State {
    bool on = false;
    bool quiet = false;                // do not print reports (the checker's own negative tests)
    unsigned long long executed = 0;   // instructions interpreted
    unsigned long long blocks = 0;     // asm blocks interpreted
    unsigned long long errors = 0;
    std::string first_error;
};
State &state();
inline bool interpret() { return state().on; }

void run(const char *tmpl, std::initializer_list<Arg> outs, std::initializer_list<Arg> ins,
         std::initializer_list<const char *> clobbers);

}  // namespace gcn
