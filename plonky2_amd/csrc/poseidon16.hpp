// poseidon16.hpp -- the same Poseidon permutation with ONE STATE WORD PER LANE: the 12 words of a state sit in lanes
// 0..11 of a 16-lane DPP row (lanes 12..15 idle), four states per wave.  For the LATENCY-critical chains: the
// Fiat-Shamir sponge (about 30 dependent permutations per recursion-size proof) and tree levels with a few hundred nodes.
//
// One permutation per lane (poseidon.hpp) is throughput-optimal at ~22 us per permutation; the quad mapping
// (poseidon4.hpp, 3 words per lane) brings a lone permutation to ~15 us.  Here every lane raises ONE word to the 7th power
// (a 4-multiply dependent chain) and computes ONE MDS row: row r = sum_j x_j * C[(j - r) mod 12], where x_j reaches every
// lane of the row through a DPP row broadcast (`row_newbcast:j`, a VALU operand modifier: no LDS, no barrier) and the
// circulant constants are twelve per-lane registers.  A round is ~110 dependent-ish instructions instead of ~270 (quad)
// or ~1010 (lane), at 16 lanes per state: used below a few thousand permutations per launch only.
// Same function as poseidon::permute: the parity tests run all three mappings against the oracle and the reference KATs.
#pragma once
#include "poseidon.hpp"

namespace poseidon16 {
using gl::u32;
using gl::u64;

// the value lane J of this 16-lane row holds, seen by every lane of the row
template <int J>
__device__ __forceinline__ u32 row_bcast(u32 v) {
#ifdef P2HOT_EMU
    const int lane = (int)(threadIdx.x & 63);
    return (u32)emu::shfl_exchange(v, (lane & ~15) | J);
#else
    return (u32)__builtin_amdgcn_mov_dpp((int)v, 0x150 + J, 0xF, 0xF, true);  // row_newbcast:J (gfx90a+)
#endif
}

// gfx950 needs two wait states between a VALU write of a VGPR and a DPP read of it; hipcc does not see the last
// instructions of an asm block (mul1 / fold1), so the word passes through a data-dependent two-state nop first
__device__ __forceinline__ void dpp_guard(u32 &lo, u32 &hi) {
#ifndef P2HOT_EMU
    asm volatile("s_nop 1" : "+v"(lo), "+v"(hi));
#else
    (void)lo;
    (void)hi;
#endif
}

struct RowConsts {
    u32 c[12];  // c[j] = C[(j - r) mod 12] (+ 8 for r = j = 0: the diagonal, poseidon_goldilocks.rs:24-25); zeros for r >= 12
};
__device__ __forceinline__ RowConsts row_consts(unsigned r) {
    constexpr u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    RowConsts k;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        u32 v = 0;
#pragma unroll
        for (int rr = 0; rr < 12; ++rr)  // select by lane: a 12-way v_cndmask chain, once per kernel
            if (r == (unsigned)rr) v = C[(j - rr + 12) % 12] + ((rr == 0 && j == 0) ? 8u : 0u);
        k.c[j] = v;
    }
    return k;
}

// y_r = rc + sum_j x_j * c_r[j]  (MDS row r fused with the next round's constant), w = this lane's word
__device__ __forceinline__ u64 mds_row(u64 w, const RowConsts &k, u64 rc) {
    u32 lo = (u32)w, hi = (u32)(w >> 32);
    dpp_guard(lo, hi);
    u64 al = (u32)rc, ah = (u32)(rc >> 32);
#define P2_T(J)                                   \
    al += (u64)row_bcast<J>(lo) * k.c[J];         \
    ah += (u64)row_bcast<J>(hi) * k.c[J];
    P2_T(0) P2_T(1) P2_T(2) P2_T(3) P2_T(4) P2_T(5) P2_T(6) P2_T(7) P2_T(8) P2_T(9) P2_T(10) P2_T(11)
#undef P2_T
    return gl::fold1(al, ah);  // al, ah < 2^42
}

// the permutation on a row-distributed state: lane r = threadIdx.x & 15 holds word r (r < 12; lanes 12..15 pass 0 and
// ignore the result).  All 16 lanes of the row must call it.  Output not canonicalised.
__device__ inline void permute_row(u64 &w, unsigned r, const RowConsts &k) {
    const unsigned ri = r < 12 ? r : 0;                      // lanes 12..15 read word 0's constants (results unused)
    const u64 *rc = P2_POSEIDON_PUSHED_ROUND_CONSTANTS + ri;  // pushed form: partial rounds add to word 0 only
    w = gl::add_canon(w, rc[0]);
    u64 next = rc[12];  // the constant of round `round + 1`, fetched one round ahead of its use
    int round = 0;
#pragma unroll 1
    for (int kk = 0; kk < 4; ++kk, ++round) {  // rounds 0..3
        const u64 cur = next;
        next = rc[12 * (round + 2)];
        w = mds_row(poseidon::sbox7_asm(w), k, cur);
    }
#pragma unroll 1
    for (int kk = 0; kk < 22; ++kk, ++round) {  // rounds 4..25: S-box on word 0 only
        const u64 cur = next;
        next = rc[12 * (round + 2)];
        const u64 sb = poseidon::sbox7_asm(w);
        w = mds_row(r == 0 ? sb : w, k, cur);
    }
#pragma unroll 1
    for (int kk = 0; kk < 3; ++kk, ++round) {  // rounds 26..28
        const u64 cur = next;
        next = round + 2 < 30 ? rc[12 * (round + 2)] : 0;
        w = mds_row(poseidon::sbox7_asm(w), k, cur);
    }
    w = mds_row(poseidon::sbox7_asm(w), k, 0);  // round 29: nothing follows
}

}  // namespace poseidon16
