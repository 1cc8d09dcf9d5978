// poseidon.hpp -- width-12 Poseidon over Goldilocks, one permutation per lane (state in VGPRs,
// round constants wave-uniform from constant memory -> SGPRs).
//
// Replaces Poseidon::poseidon (plonky2/src/hash/poseidon.rs:767-777), the sponge
// hash_n_to_hash_no_pad (hashing.rs:118-145), compress/two_to_one (hashing.rs:97-114) and
// Hasher::hash_or_noop (plonk/config.rs:63-74).
//
// MI355X-first choices (not the reference's CPU schedule):
//  * The permutation is evaluated in its defining 30-round form (ARK, S-box, MDS; S-box on lane 0
//    only in rounds 4..25, poseidon.rs:781-801) instead of the reference's "fast partial round"
//    refactoring: on CDNA a 64x64 multiply costs four quarter-rate v_mad_u64_u32 while the small
//    circulant MDS (entries <= 41) runs on full-rate 24-bit multiply-adds, so the sparse-matrix
//    form (22 wide multiplies per partial round) does not win here.  Both forms are the same
//    function (poseidon.rs:944-957 checks that in the reference); the oracle pins both to the KATs.
//  * MDS: every state word is split into 22/21/21-bit limbs; y_r = sum_i c_i * x_{(i+r)%12}
//    (+ 8*x_0 for r = 0) is accumulated per limb in 32-bit registers (264 * 2^22 < 2^32) with
//    v_mad_u32_u24 and recombined/reduced once per output.
//  * Round loops stay rolled so the kernel body fits the instruction cache.
#pragma once
#include "gl.hpp"
#ifndef P2HOT_EMU
#define P2_CONST_QUAL __constant__  // device constant memory; indices are wave-uniform -> scalar loads
#endif
#include "poseidon_constants.h"

namespace poseidon {
using gl::u32;
using gl::u64;

#define RC P2_POSEIDON_ALL_ROUND_CONSTANTS  // poseidon.rs:59-157

__device__ __forceinline__ u64 sbox7(u64 x) {  // poseidon.rs:690-696
    u64 x2 = gl::sqr(x);
    u64 x4 = gl::sqr(x2);
    u64 x3 = gl::mul(x, x2);
    return gl::mul(x3, x4);
}

// circulant first row [17,15,41,16,2,28,13,13,39,18,34,20], diag [8,0,...] (poseidon_goldilocks.rs:24-25)
__device__ __forceinline__ void mds_layer(u64 s[12]) {
    constexpr u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    u32 l0[12], l1[12], l2[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        u32 lo = (u32)s[i], hi = (u32)(s[i] >> 32);
        l0[i] = lo & 0x3FFFFFu;                          // bits 0..21
        l1[i] = ((lo >> 22) | (hi << 10)) & 0x1FFFFFu;   // bits 22..42
        l2[i] = hi >> 11;                                // bits 43..63
    }
#pragma unroll
    for (int r = 0; r < 12; ++r) {
        u32 a0 = 0, a1 = 0, a2 = 0;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int j = (i + r) % 12;
            a0 += C[i] * l0[j];
            a1 += C[i] * l1[j];
            a2 += C[i] * l2[j];
        }
        if (r == 0) {
            a0 += 8u * l0[0];
            a1 += 8u * l1[0];
            a2 += 8u * l2[0];
        }
        // y = a0 + a1*2^22 + a2*2^43  (< 2^74), then fold bits >= 64 with 2^64 = EPS
        u64 lo64 = (u64)a0 + ((u64)a1 << 22);
        u64 t = (u64)a2 << 43;
        lo64 += t;
        u64 hi = (u64)(a2 >> 21) + (u64)(lo64 < t);
        u64 e = (hi << 32) - hi;
        u64 y = lo64 + e;
        if (y < e) y += gl::EPS;
        s[r] = y;
    }
}

// the permutation; output words are NOT canonicalised (callers canonicalise what they emit)
__device__ inline void permute(u64 s[12]) {
    int round = 0;
#pragma unroll 1
    for (int k = 0; k < 4; ++k, ++round) {
#pragma unroll
        for (int i = 0; i < 12; ++i) s[i] = sbox7(gl::add_canon(s[i], RC[12 * round + i]));
        mds_layer(s);
    }
#pragma unroll 1
    for (int k = 0; k < 22; ++k, ++round) {
#pragma unroll
        for (int i = 1; i < 12; ++i) s[i] = gl::add_canon(s[i], RC[12 * round + i]);
        s[0] = sbox7(gl::add_canon(s[0], RC[12 * round]));
        mds_layer(s);
    }
#pragma unroll 1
    for (int k = 0; k < 4; ++k, ++round) {
#pragma unroll
        for (int i = 0; i < 12; ++i) s[i] = sbox7(gl::add_canon(s[i], RC[12 * round + i]));
        mds_layer(s);
    }
}

// two_to_one (hashing.rs:97-114): state = [l, r, 0^4], permute, first 4 words
__device__ __forceinline__ void two_to_one(const u64 l[4], const u64 r[4], u64 out[4]) {
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        s[i] = l[i];
        s[4 + i] = r[i];
        s[8 + i] = 0;
    }
    permute(s);
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = gl::canon(s[i]);
}

#undef RC
}  // namespace poseidon
