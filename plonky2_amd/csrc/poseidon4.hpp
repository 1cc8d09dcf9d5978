// poseidon4.hpp -- the same Poseidon permutation with the 12-word state spread over the 4 lanes of a quad
// (3 words per lane), for LATENCY-bound work: small trees, the top Merkle levels, FRI round trees.
//
// One permutation per lane (poseidon.hpp) is throughput-optimal but a lone wave needs ~45 us per permutation
// (19.7 k mostly dependent instructions); a tree level with a few thousand nodes therefore costs 45 us however few
// nodes it has.  Spreading the state over a quad cuts the dependent chain ~3x:
//   * S-box of a full round: every lane raises its own 3 words with one mul3 stream (poseidon.rs:712-718);
//   * MDS: lane q needs rows 3q..3q+2 = sum_i C[i] * x_{(i + 3q + t) % 12}.  Rotating the quad by s lanes with a DPP
//     quad_perm (a VALU operand modifier, no LDS) hands every lane the words x_{(i + 3q) % 12} in the SAME register
//     order, so the circulant constants stay compile-time immediates; the diagonal term 8*x_0 is a per-lane factor;
//   * partial rounds: the scalar constant and the S-box touch word 0 only (lane 0, slot 0); all lanes execute the
//     stream, the other lanes' results are discarded.
// Costs 4 lanes per permutation (~1.3x the lane-instructions), so it is used only below ~2^15 permutations per launch.
#pragma once
#include "poseidon.hpp"

namespace poseidon4 {
using gl::u32;
using gl::u64;

// value of lane ((q + S) & 3) of this quad, as seen from lane q
template <int S>
__device__ __forceinline__ u32 quad_rot32(u32 v) {
#ifdef P2HOT_EMU
    const int lane = (int)(threadIdx.x & 63);
    return (u32)emu::shfl_exchange(v, (lane & ~3) | ((lane + S) & 3));
#else
    constexpr int ctrl = ((0 + S) & 3) | (((1 + S) & 3) << 2) | (((2 + S) & 3) << 4) | (((3 + S) & 3) << 6);
    return (u32)__builtin_amdgcn_mov_dpp((int)v, ctrl, 0xF, 0xF, true);
#endif
}
template <int S>
__device__ __forceinline__ u64 quad_rot(u64 v) {
    return ((u64)quad_rot32<S>((u32)(v >> 32)) << 32) | quad_rot32<S>((u32)v);
}

// gfx950 needs two wait states between a VALU write of a VGPR and a DPP read of it.  hipcc pads that hazard for its
// own instructions but knows nothing about the last instructions of an asm block (mul3 / mul1 / fold3), so the words
// are passed through an explicit two-state nop that is data-dependent on them before the DPP rotations read them.
__device__ __forceinline__ void dpp_guard(u64 w[3]) {
#ifndef P2HOT_EMU
    asm volatile("s_nop 1" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]));
#else
    (void)w;
#endif
}

// MDS (+ optional constants rc3[t] for this lane's rows) on the quad-distributed state w[3]
__device__ __forceinline__ void mds_quad(u64 w[3], unsigned q, const u64 rc3[3], bool has_rc) {
    constexpr u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    u64 z[12];  // z[i] = x_{(i + 3q) % 12}
    dpp_guard(w);
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        z[u] = w[u];
        z[3 + u] = quad_rot<1>(w[u]);
        z[6 + u] = quad_rot<2>(w[u]);
        z[9 + u] = quad_rot<3>(w[u]);
    }
    u32 zl[12], zh[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        zl[i] = (u32)z[i];
        zh[i] = (u32)(z[i] >> 32);
    }
    const u32 d8 = q == 0 ? 8u : 0u;  // diag [8,0,...]: only row 0 = (lane 0, slot 0); z[0] is x_0 there
    u64 al[3], ah[3], y[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        u32 rl = has_rc ? (u32)rc3[t] : 0u, rh = has_rc ? (u32)(rc3[t] >> 32) : 0u;
        al[t] = rl;
        ah[t] = rh;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int j = (i + t) % 12;
            al[t] += (u64)zl[j] * C[i];
            ah[t] += (u64)zh[j] * C[i];
        }
        if (t == 0) {
            al[t] += (u64)zl[0] * d8;
            ah[t] += (u64)zh[0] * d8;
        }
    }
    gl::fold3(al, ah, y);
    w[0] = y[0];
    w[1] = y[1];
    w[2] = y[2];
}

// the permutation on a quad-distributed state; q = lane & 3 holds words 3q..3q+2.  All four lanes must call it.
__device__ inline void permute_quad(u64 w[3], unsigned q) {
    const u64 *rc = P2_POSEIDON_PUSHED_ROUND_CONSTANTS + 3 * q;  // pushed form: partial rounds add to word 0 only
#pragma unroll
    for (int t = 0; t < 3; ++t) w[t] = gl::add_canon(w[t], rc[t]);
    int round = 0;
#pragma unroll 1
    for (int k = 0; k < 4; ++k, ++round) {  // rounds 0..3 (the constants after round 3 are scalar: zeros elsewhere)
        poseidon::sbox7_x3(w[0], w[1], w[2]);
        mds_quad(w, q, rc + 12 * (round + 1), true);
    }
#pragma unroll 1
    for (int k = 0; k < 22; ++k, ++round) {  // rounds 4..25
        w[0] = q == 0 ? poseidon::sbox7_asm(w[0]) : w[0];
        mds_quad(w, q, rc + 12 * (round + 1), true);
    }
#pragma unroll 1
    for (int k = 0; k < 3; ++k, ++round) {  // rounds 26..28
        poseidon::sbox7_x3(w[0], w[1], w[2]);
        mds_quad(w, q, rc + 12 * (round + 1), true);
    }
    poseidon::sbox7_x3(w[0], w[1], w[2]);
    mds_quad(w, q, rc, false);
}

}  // namespace poseidon4
