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

// value of lane 0 of this quad
__device__ __forceinline__ u64 quad_bcast0(u64 v) {
#ifdef P2HOT_EMU
    const int lane = (int)(threadIdx.x & 63);
    return emu::shfl_exchange(v, lane & ~3);
#else
    const u32 lo = (u32)__builtin_amdgcn_mov_dpp((int)(u32)v, 0, 0xF, 0xF, true);
    const u32 hi = (u32)__builtin_amdgcn_mov_dpp((int)(u32)(v >> 32), 0, 0xF, 0xF, true);
    return ((u64)hi << 32) | lo;
#endif
}

// The constants of the batched partial rounds as THIS lane needs them: its three rows of M^3 in the rotated word order its
// quad rotations deliver (z'[i] = x_{(i + 3q) % 12}; M^3 is not circulant, so unlike the MDS immediates they differ per lane)
// and its rows' entries of the columns M^2 e0 and M e0.  Selected from literals once per permutation.
struct QuadM3 {
    u32 m[36], c2[3], c1[3];
};
__device__ __forceinline__ u32 sel4(unsigned q, u32 a0, u32 a1, u32 a2, u32 a3) { return q == 0 ? a0 : q == 1 ? a1 : q == 2 ? a2 : a3; }
__device__ __forceinline__ QuadM3 quad_m3(unsigned q) {
    QuadM3 k;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
#pragma unroll
        for (int i = 0; i < 12; ++i)
            k.m[12 * t + i] = sel4(q, P2_POSEIDON_M3[12 * (0 + t) + (i + 0) % 12], P2_POSEIDON_M3[12 * (3 + t) + (i + 3) % 12],
                                   P2_POSEIDON_M3[12 * (6 + t) + (i + 6) % 12], P2_POSEIDON_M3[12 * (9 + t) + (i + 9) % 12]);
        k.c2[t] = sel4(q, P2_POSEIDON_MCOL0[12 + t], P2_POSEIDON_MCOL0[15 + t], P2_POSEIDON_MCOL0[18 + t], P2_POSEIDON_MCOL0[21 + t]);
        k.c1[t] = sel4(q, P2_POSEIDON_MCOL0[t], P2_POSEIDON_MCOL0[3 + t], P2_POSEIDON_MCOL0[6 + t], P2_POSEIDON_MCOL0[9 + t]);
    }
    return k;
}

// Three partial rounds in one dense pass on the quad-distributed state: poseidon::partial_rounds3 (see there for the algebra)
// with the word-0 chain -- S-box, the row-0 products (M z)[0] and (M^2 z)[0], the two deltas -- run by every lane on its own
// rotated view (lane 0's view is the true order, the others' results are discarded), the deltas broadcast from lane 0, and
// each lane finishing its own three rows of M^3 z + d1 M^2 e0 + d2 M e0.  ~130 instructions per round instead of the ~190 of
// a round-by-round MDS (the quad kernels spend two thirds of a permutation in the partial rounds).
__device__ __forceinline__ void partial_rounds3_quad(u64 w[3], unsigned q, u64 c0, u64 c1, u64 c2, const QuadM3 &k) {
    const u64 s0 = poseidon::sbox7_asm(gl::add_canon(w[0], c0));
    w[0] = q == 0 ? s0 : w[0];
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
    u64 al = 0, ah = 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        al += (u64)zl[i] * P2_POSEIDON_M1_ROW0[i];
        ah += (u64)zh[i] * P2_POSEIDON_M1_ROW0[i];
    }
    const u64 y1 = gl::fold1(al, ah);
    u64 d1 = gl::sub(poseidon::sbox7_asm(gl::add_canon(y1, c1)), y1);
    al = (u64)(u32)d1 * P2_POSEIDON_MCOL0[0];
    ah = (u64)(u32)(d1 >> 32) * P2_POSEIDON_MCOL0[0];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        al += (u64)zl[i] * P2_POSEIDON_M2_ROW0[i];
        ah += (u64)zh[i] * P2_POSEIDON_M2_ROW0[i];
    }
    const u64 y2 = gl::fold1(al, ah);
    u64 d2 = gl::sub(poseidon::sbox7_asm(gl::add_canon(y2, c2)), y2);
#ifndef P2HOT_EMU
    asm volatile("s_nop 1" : "+v"(d1), "+v"(d2));  // VALU write -> DPP read (see dpp_guard)
#endif
    d1 = quad_bcast0(d1);
    d2 = quad_bcast0(d2);
    const u32 d1l = (u32)d1, d1h = (u32)(d1 >> 32), d2l = (u32)d2, d2h = (u32)(d2 >> 32);
    u64 bl[3], bh[3], y[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        bl[t] = (u64)d1l * k.c2[t] + (u64)d2l * k.c1[t];
        bh[t] = (u64)d1h * k.c2[t] + (u64)d2h * k.c1[t];
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            bl[t] += (u64)zl[i] * k.m[12 * t + i];
            bh[t] += (u64)zh[i] * k.m[12 * t + i];
        }
    }
    gl::fold3(bl, bh, y);
    w[0] = y[0];
    w[1] = y[1];
    w[2] = y[2];
}

// the permutation on a quad-distributed state; q = lane & 3 holds words 3q..3q+2.  All four lanes must call it.
__device__ inline void permute_quad(u64 w[3], unsigned q) {
    const u64 *rc = P2_POSEIDON_PUSHED_ROUND_CONSTANTS + 3 * q;  // pushed form: partial rounds add to word 0 only
    const QuadM3 k = quad_m3(q);
#pragma unroll
    for (int t = 0; t < 3; ++t) w[t] = gl::add_canon(w[t], rc[t]);
    int round = 0;
#pragma unroll 1
    for (int j = 0; j < 3; ++j, ++round) {  // rounds 0..2
        poseidon::sbox7_x3(w[0], w[1], w[2]);
        mds_quad(w, q, rc + 12 * (round + 1), true);
    }
    poseidon::sbox7_x3(w[0], w[1], w[2]);  // round 3: the batched partial rounds add their scalar constants themselves
    mds_quad(w, q, rc, false);
    ++round;
#pragma unroll 1
    for (int j = 0; j < 7; ++j, round += 3)  // rounds 4..24 in seven batches of three
        partial_rounds3_quad(w, q, P2_POSEIDON_PUSHED_ROUND_CONSTANTS[12 * round], P2_POSEIDON_PUSHED_ROUND_CONSTANTS[12 * (round + 1)],
                             P2_POSEIDON_PUSHED_ROUND_CONSTANTS[12 * (round + 2)], k);
    {  // round 25, then the full constant vector of round 26 (which absorbed the pushed remainder)
        const u64 s0 = poseidon::sbox7_asm(gl::add_canon(w[0], P2_POSEIDON_PUSHED_ROUND_CONSTANTS[12 * round]));
        w[0] = q == 0 ? s0 : w[0];
        mds_quad(w, q, rc + 12 * (round + 1), true);
        ++round;
    }
#pragma unroll 1
    for (int j = 0; j < 3; ++j, ++round) {  // rounds 26..28
        poseidon::sbox7_x3(w[0], w[1], w[2]);
        mds_quad(w, q, rc + 12 * (round + 1), true);
    }
    poseidon::sbox7_x3(w[0], w[1], w[2]);
    mds_quad(w, q, rc, false);
}

}  // namespace poseidon4
