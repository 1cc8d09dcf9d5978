// poseidon.hpp -- width-12 Poseidon over Goldilocks, one permutation per lane (state in VGPRs,
// round constants wave-uniform from constant memory -> SGPRs).
//
// Replaces Poseidon::poseidon (plonky2/src/hash/poseidon.rs:767-777), the sponge
// hash_n_to_hash_no_pad (hashing.rs:118-145), compress/two_to_one (hashing.rs:97-114) and
// Hasher::hash_or_noop (plonk/config.rs:63-74).
//
// MI355X-first choices (not the reference's CPU schedule):
//  * gfx950 issues v_mad_u64_u32 (32x32+64) at the ordinary VOP3 rate while the 24-bit multiplies
//    are half rate (measured, tools/ubench.hip), so everything is built from 32x32+64 multiply-adds:
//    a field multiply is 4 of them plus a two-fold reduction; an MDS output is two 12-term
//    multiply-add chains over the 32-bit halves of the state (small constants as inline operands).
//  * The permutation is evaluated in its defining 30-round form (ARK, S-box, MDS; S-box on word 0
//    only in rounds 4..25, poseidon.rs:781-801) rather than the reference's "fast partial round"
//    refactoring: with cheap small-constant MDS rows the sparse-matrix form (22 wide multiplies
//    per partial round plus a 121-multiply pre-matrix) costs more instructions here.  Both forms
//    are the same function (poseidon.rs:944-957); the oracle pins both to the reference KATs.
//  * The constant layer is fused into the preceding MDS: the round constants, pre-split into
//    {lo32, hi32} words, seed the two accumulators straight from SGPRs.  In the partial rounds the
//    constants of the 11 passive words are pushed forward through the (linear) MDS at table-generation
//    time, so those rounds add a single scalar to word 0 and round 26 absorbs the remainder.
//  * The partial rounds run three to a dense pass over integer products of the MDS matrix, written with the projection
//    D = diag(0, 1, .., 1) (a round REPLACES word 0) so that no field subtraction is needed; their scalars are chain addends too.
//  * Round loops stay rolled so the kernel body fits the instruction cache (one batch of partial rounds is peeled: it carries
//    round 26's constant vector).
#pragma once
#include "gl.hpp"
#include "gl_mul3.hpp"
#ifndef P2HOT_EMU
#define P2_CONST_QUAL __constant__  // device constant memory; indices are wave-uniform -> scalar loads
#endif
#define P2_LITERAL_QUAL static constexpr  // entries become instruction-stream immediates (s_mov_b32 literals); constexpr in both builds:
                                          // M^2 below is computed from them at compile time
#include "poseidon_constants.h"

namespace poseidon {
using gl::u32;
using gl::u64;

#define RC P2_POSEIDON_ALL_ROUND_CONSTANTS  // poseidon.rs:59-157
#define RC_SPLIT P2_POSEIDON_PUSHED_ROUND_CONSTANTS_SPLIT  // [r][i] -> {lo32, hi32} as two u64 words; partial rounds: word 0 only

__device__ __forceinline__ u64 sbox7(u64 x) {  // poseidon.rs:690-696
    u64 x2 = gl::sqr(x);
    u64 x4 = gl::sqr(x2);
    u64 x3 = gl::mul(x, x2);
    return gl::mul(x3, x4);
}

// P2HOT_SBOX_CF (default 2): the S-box products through the carry-free multiply streams of round 6 -- 2: gl::mul3cg / mul1cg (two
// partial products chained through the multiply-add's addend, the one real carry left to the multiply-add's own carry-out: 14
// instructions, 3 of them co-issued moves); 1: gl::mul3cf / mul1cf (all three chained: 16 instructions, 4 moves); 0: the round-3
// streams (gl::mul3 / mul1: 14 instructions, 3 carry adds)
#ifndef P2HOT_SBOX_CF
#define P2HOT_SBOX_CF 2
#endif
__device__ __forceinline__ u64 sbox_mul1(u64 a, u64 b) {
    return P2HOT_SBOX_CF >= 2 ? gl::mul1cg(a, b) : P2HOT_SBOX_CF ? gl::mul1cf(a, b) : gl::mul1(a, b);
}
__device__ __forceinline__ void sbox_mul3(const u64 a[3], const u64 b[3], u64 r[3]) {
    if (P2HOT_SBOX_CF == 3)
        gl::mul3ch(a, b, r);
    else if (P2HOT_SBOX_CF == 2)
        gl::mul3cg(a, b, r);
    else if (P2HOT_SBOX_CF)
        gl::mul3cf(a, b, r);
    else
        gl::mul3(a, b, r);
}

// a + c for a round constant c (canonical): the 64-bit add, its wrap as a compare, and the fold-back of 2^64 = 2^32 - 1 as ONE multiply-add
// on the select (hipcc's form of gl::add_canon is a second 64-bit add and two selects: 5 instructions instead of 4)
__device__ __forceinline__ u64 ark(u64 a, u64 c) {
    if (!P2_ASM_INTERPRETED()) return gl::add_canon(a, c);
    const u64 s = a + c;
    const u32 e = s < c ? 0xFFFFFFFFu : 0u;
    u64 r, d;
    P2_ASM_NC("v_mad_u64_u32 %0, %1, %2, 1, %3", (P2_O(, "=v", r), P2_O(, "=&s", d)), (P2_I(, "v", e), P2_I(, "v", s)));
    return r;
}
// A wave-uniform zero the optimiser cannot see through, made where it is used: a constant-table index built on it is loaded THERE
// (s_load) instead of being hoisted out of the leaf loop into SGPRs that are spilled to VGPR lanes and read back with v_readlane --
// VALU instructions -- at every use (24 per constant vector, measured on the ISA)
__device__ __forceinline__ u32 opaque_zero() {
#ifndef P2HOT_EMU
    u32 z;
    asm volatile("s_mov_b32 %0, 0" : "=s"(z));
    return z;
#else
    return 0;
#endif
}

// x^7 of one word with the hand-scheduled multiply (the partial rounds' single S-box)
__device__ __forceinline__ u64 sbox7_asm(u64 x) {
    u64 x2 = sbox_mul1(x, x);
    u64 x4 = sbox_mul1(x2, x2);
    u64 x3 = sbox_mul1(x, x2);
    return sbox_mul1(x3, x4);
}

// x^7 of three independent words at once: x2 = x*x; (x3 = x*x2 and x4 = x2*x2 are independent); x7 = x3*x4
__device__ __forceinline__ void sbox7_x3(u64 &a, u64 &b, u64 &c) {
    u64 x[3] = {a, b, c}, x2[3], x3[3], x4[3];
    sbox_mul3(x, x, x2);
    sbox_mul3(x, x2, x3);
    sbox_mul3(x2, x2, x4);
    sbox_mul3(x3, x4, x);
    a = x[0];
    b = x[1];
    c = x[2];
}

__device__ __forceinline__ void sbox_layer(u64 s[12]) {  // poseidon.rs:712-718
#pragma unroll
    for (int i = 0; i < 12; i += 3) sbox7_x3(s[i], s[i + 1], s[i + 2]);
}

// Wave-uniform small constant that the optimiser must not see through: keeps c * x as ONE
// v_mad_u64_u32 (SGPR operand) instead of a shift + 64-bit add on a zero-extended register pair.
__device__ __forceinline__ u32 opaque_const(u32 c) {
#ifndef P2HOT_EMU
    asm volatile("" : "+s"(c));
#endif
    return c;
}

// One term of three MDS rows: the six accumulators (lo / hi halves of rows r, r+1, r+2) each take x * C in ONE
// multiply-add.  Written as asm because hipcc re-associates the 12-term chains into two half chains plus a 64-bit add
// (24 extra instructions per layer) and adds the round constants with separate 64-bit adds; here a chain is exactly
// its 12 multiply-adds, the first of which reads the constant from its SGPR pair as the addend.
template <u32 C>
__device__ __forceinline__ void mds_term(u64 (&al)[3], u64 (&ah)[3], u32 l0, u32 h0, u32 l1, u32 h1, u32 l2, u32 h2) {
    if (!P2_ASM_INTERPRETED()) {  // emulator build with the interpreter off (constant-false in the product)
        al[0] += (u64)l0 * C;
        ah[0] += (u64)h0 * C;
        al[1] += (u64)l1 * C;
        ah[1] += (u64)h1 * C;
        al[2] += (u64)l2 * C;
        ah[2] += (u64)h2 * C;
        return;
    }
    u64 d;
    P2_ASM_NC("v_mad_u64_u32 %0, %6, %7, %13, %0\n\t"
              "v_mad_u64_u32 %1, %6, %8, %13, %1\n\t"
              "v_mad_u64_u32 %2, %6, %9, %13, %2\n\t"
              "v_mad_u64_u32 %3, %6, %10, %13, %3\n\t"
              "v_mad_u64_u32 %4, %6, %11, %13, %4\n\t"
              "v_mad_u64_u32 %5, %6, %12, %13, %5",
              (P2_O(, "+v", al[0]), P2_O(, "+v", ah[0]), P2_O(, "+v", al[1]), P2_O(, "+v", ah[1]), P2_O(, "+v", al[2]),
               P2_O(, "+v", ah[2]), P2_O(, "=&s", d)),
              (P2_I(, "v", l0), P2_I(, "v", h0), P2_I(, "v", l1), P2_I(, "v", h1), P2_I(, "v", l2), P2_I(, "v", h2),
               P2_I(, "n", C)));
}
// the first term (C[0] = 17): starts the six chains from the six wave-uniform 64-bit addends k (SGPR pairs) or from 0
template <bool HAS_K>
__device__ __forceinline__ void mds_first(u64 (&al)[3], u64 (&ah)[3], u32 l0, u32 h0, u32 l1, u32 h1, u32 l2, u32 h2,
                                          const u64 *k) {
    if (!P2_ASM_INTERPRETED()) {
        const u64 z[6] = {0, 0, 0, 0, 0, 0};
        const u64 *kk = HAS_K ? k : z;
        al[0] = (u64)l0 * 17u + kk[0];
        ah[0] = (u64)h0 * 17u + kk[1];
        al[1] = (u64)l1 * 17u + kk[2];
        ah[1] = (u64)h1 * 17u + kk[3];
        al[2] = (u64)l2 * 17u + kk[4];
        ah[2] = (u64)h2 * 17u + kk[5];
        return;
    }
    u64 d;
    if (HAS_K) {
        P2_ASM_NC("v_mad_u64_u32 %0, %6, %7, 17, %13\n\t"
                  "v_mad_u64_u32 %1, %6, %8, 17, %14\n\t"
                  "v_mad_u64_u32 %2, %6, %9, 17, %15\n\t"
                  "v_mad_u64_u32 %3, %6, %10, 17, %16\n\t"
                  "v_mad_u64_u32 %4, %6, %11, 17, %17\n\t"
                  "v_mad_u64_u32 %5, %6, %12, 17, %18",
                  (P2_O(, "=&v", al[0]), P2_O(, "=&v", ah[0]), P2_O(, "=&v", al[1]), P2_O(, "=&v", ah[1]),
                   P2_O(, "=&v", al[2]), P2_O(, "=&v", ah[2]), P2_O(, "=&s", d)),
                  (P2_I(, "v", l0), P2_I(, "v", h0), P2_I(, "v", l1), P2_I(, "v", h1), P2_I(, "v", l2), P2_I(, "v", h2),
                   P2_I(, "s", k[0]), P2_I(, "s", k[1]), P2_I(, "s", k[2]), P2_I(, "s", k[3]), P2_I(, "s", k[4]),
                   P2_I(, "s", k[5])));
    } else {
        P2_ASM_NC("v_mad_u64_u32 %0, %6, %7, 17, 0\n\t"
                  "v_mad_u64_u32 %1, %6, %8, 17, 0\n\t"
                  "v_mad_u64_u32 %2, %6, %9, 17, 0\n\t"
                  "v_mad_u64_u32 %3, %6, %10, 17, 0\n\t"
                  "v_mad_u64_u32 %4, %6, %11, 17, 0\n\t"
                  "v_mad_u64_u32 %5, %6, %12, 17, 0",
                  (P2_O(, "=&v", al[0]), P2_O(, "=&v", ah[0]), P2_O(, "=&v", al[1]), P2_O(, "=&v", ah[1]),
                   P2_O(, "=&v", al[2]), P2_O(, "=&v", ah[2]), P2_O(, "=&s", d)),
                  (P2_I(, "v", l0), P2_I(, "v", h0), P2_I(, "v", l1), P2_I(, "v", h1), P2_I(, "v", l2), P2_I(, "v", h2)));
    }
}

// MDS layer fused with the NEXT round's constant layer:
//   y_r = rc_r + sum_i C[i] * x_{(i+r)%12} (+ 8 * x_0 for r = 0)
// circulant first row C = [17,15,41,16,2,28,13,13,39,18,34,20], diag [8,0,...] (poseidon_goldilocks.rs:24-25).
// On gfx950 v_mad_u64_u32 issues at the plain VOP3 rate (tools/ubench), so each output is two
// 12-term multiply-add chains over the 32-bit halves (al, ah < 2^42) and one fold:
//   y = al + ah * 2^32 (mod P), gl::fold3.
// rc2 points at the round's constants split as {lo32, hi32} pairs (RC_SPLIT) or is null.
// `groups` (wave-uniform): bit g set = rows 3g..3g+2 are wanted; the other rows are left stale.
// one MDS row (no constant, not row 0): two 12-term chains and a single-stream fold -- for the LAST layer of a permutation, whose
// reader wants four words (a digest, or the capacity between two absorbs): one triple and this row instead of two triples
template <int R>
__device__ __forceinline__ u64 mds_row(const u32 (&xl)[12], const u32 (&xh)[12]) {
    static_assert(R > 0 && R < 12, "row 0 carries the diagonal term");
    constexpr u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    u64 al = (u64)xl[R] * opaque_const(C[0]), ah = (u64)xh[R] * opaque_const(C[0]);
#pragma unroll
    for (int i = 1; i < 12; ++i) {
        al += (u64)xl[(R + i) % 12] * opaque_const(C[i]);
        ah += (u64)xh[(R + i) % 12] * opaque_const(C[i]);
    }
    return gl::fold1(al, ah);
}

// `single` (wave-uniform; 3, 8 or none): one more row computed on its own (mds_row)
__device__ __forceinline__ void mds_layer(u64 s[12], const u64 *rc2, unsigned groups = 0xFu, int single = -1) {
    u32 xl[12], xh[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        xl[i] = (u32)s[i];
        xh[i] = (u32)(s[i] >> 32);
    }
#pragma unroll
    for (int g = 0; g < 12; g += 3) {  // three rows at a time: six independent chains, then one fold3 stream
        if (!(groups >> (g / 3) & 1u)) continue;
        u64 al[3], ah[3], y[3];
#define P2_X(i) xl[(g + (i)) % 12], xh[(g + (i)) % 12], xl[(g + (i) + 1) % 12], xh[(g + (i) + 1) % 12], xl[(g + (i) + 2) % 12], xh[(g + (i) + 2) % 12]
        if (rc2)
            mds_first<true>(al, ah, P2_X(0), rc2 + 2 * g);
        else
            mds_first<false>(al, ah, P2_X(0), nullptr);
        mds_term<15>(al, ah, P2_X(1));
        mds_term<41>(al, ah, P2_X(2));
        mds_term<16>(al, ah, P2_X(3));
        mds_term<2>(al, ah, P2_X(4));
        mds_term<28>(al, ah, P2_X(5));
        mds_term<13>(al, ah, P2_X(6));
        mds_term<13>(al, ah, P2_X(7));
        mds_term<39>(al, ah, P2_X(8));
        mds_term<18>(al, ah, P2_X(9));
        mds_term<34>(al, ah, P2_X(10));
        mds_term<20>(al, ah, P2_X(11));
#undef P2_X
        if (g == 0) {  // diag [8, 0, ...]: row 0 only
            al[0] += (u64)xl[0] * opaque_const(8);
            ah[0] += (u64)xh[0] * opaque_const(8);
        }
        gl::fold3(al, ah, y);
        s[g] = y[0];
        s[g + 1] = y[1];
        s[g + 2] = y[2];
    }
    if (single == 3)
        s[3] = mds_row<3>(xl, xh);
    else if (single == 8)
        s[8] = mds_row<8>(xl, xh);
}

// M^2 = the integer square of the MDS matrix M[r][c] = C[(c - r) mod 12] + 8 [r = c = 0] (entries < 2^15, rows < 2^17), built
// at compile time from the generated first row and checked against the generated tables it must agree with
struct Mat12 {
    u32 v[144];
};
constexpr Mat12 mds_matrix() {
    Mat12 m{};
    for (int r = 0; r < 12; ++r)
        for (int c = 0; c < 12; ++c) m.v[12 * r + c] = P2_POSEIDON_M1_ROW0[(c - r + 12) % 12] - ((c - r + 12) % 12 == 0 ? 8u : 0u) + (r == 0 && c == 0 ? 8u : 0u);
    return m;
}
constexpr Mat12 mat_mul(const Mat12 &a, const Mat12 &b) {
    Mat12 m{};
    for (int r = 0; r < 12; ++r)
        for (int c = 0; c < 12; ++c) {
            u64 acc = 0;
            for (int k = 0; k < 12; ++k) acc += (u64)a.v[12 * r + k] * b.v[12 * k + c];
            m.v[12 * r + c] = (u32)acc;
        }
    return m;
}
constexpr bool mat_checks() {
    const Mat12 m1 = mds_matrix(), m2 = mat_mul(m1, m1), m3 = mat_mul(m1, m2);
    for (int j = 0; j < 12; ++j) {
        if (m1.v[j] != P2_POSEIDON_M1_ROW0[j] || m2.v[j] != P2_POSEIDON_M2_ROW0[j]) return false;
        if (m1.v[12 * j] != P2_POSEIDON_MCOL0[j] || m2.v[12 * j] != P2_POSEIDON_MCOL0[12 + j]) return false;
    }
    for (int i = 0; i < 144; ++i)
        if (m3.v[i] != P2_POSEIDON_M3[i]) return false;
    return true;
}
static_assert(mat_checks(), "M, M^2, M^3: the compile-time products disagree with the generated tables");
// The partial rounds REPLACE word 0 (y <- M (D y + s e0), D = diag(0, 1, .., 1), s = sbox(y[0] + c)) -- written with D instead of
// with the difference s - y[0], a batch of rounds is linear in (z, s_1, s_2, ..) with NON-NEGATIVE integer coefficients, entry by entry
// at most those of M^2 and M^3, and needs no field subtraction (round 6; a gl::sub is 11 VALU instructions as hipcc writes it):
//   M (D (M z) + s1 e0)                   = (MD) M z + s1 (M e0)
//   M (D (M (D (M z) + s1 e0)) + s2 e0)   = (MD)(MD) M z + s1 (MD) M e0 + s2 (M e0)            MD = M with column 0 zeroed
constexpr Mat12 mat_drop_col0(const Mat12 &a) {
    Mat12 m = a;
    for (int r = 0; r < 12; ++r) m.v[12 * r] = 0;
    return m;
}
P2_LITERAL_QUAL Mat12 MDS1 = mds_matrix();
P2_LITERAL_QUAL Mat12 MDM = mat_mul(mat_drop_col0(mds_matrix()), mds_matrix());
P2_LITERAL_QUAL Mat12 MDMDM = mat_mul(mat_drop_col0(mds_matrix()), mat_mul(mat_drop_col0(mds_matrix()), mds_matrix()));

// A full round's MDS and the partial round after it in one dense pass (round 3's linear layer + round 4): with x the state after
// the full round's S-box layer, u = M x, s = sbox(u[0] + c), the state after the partial round is
//   y = M (D u + s e0) = (MD) M x + s (M e0)
// -- one 12-row pass over (MD) M (+1 term per row), one single row for u[0] and one S-box, instead of two 12-row passes.  With it the
// 1 + 22 applications of M between the two full-round halves take 8 dense passes (this one and seven batches of three) instead of 9.
// No pending constant on entry or exit (round 3 fuses none; c = the partial round's pushed scalar).
// c rides the u[0] chain as its starting addend (halves, like every constant here); cn = the NEXT partial round's scalar, fused into
// row 0 the same way, so that no partial round adds its constant with instructions of its own.
__device__ __forceinline__ void mds_partial_round(u64 s[12], u64 c, u64 cn) {
    u32 xl[12], xh[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        xl[i] = (u32)s[i];
        xh[i] = (u32)(s[i] >> 32);
    }
    u64 al = (u32)c, ah = c >> 32;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        al += (u64)xl[j] * P2_POSEIDON_M1_ROW0[j];
        ah += (u64)xh[j] * P2_POSEIDON_M1_ROW0[j];
    }
    const u64 s1 = sbox7_asm(gl::fold1(al, ah));
    const u32 sl = (u32)s1, sh = (u32)(s1 >> 32);
#pragma unroll
    for (int g = 0; g < 12; g += 3) {
        u64 bl[3], bh[3], y[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int i = g + t;
            bl[t] = (u64)sl * MDS1.v[12 * i] + (i == 0 ? (u64)(u32)cn : 0);
            bh[t] = (u64)sh * MDS1.v[12 * i] + (i == 0 ? cn >> 32 : 0);
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                bl[t] += (u64)xl[j] * MDM.v[12 * i + j];
                bh[t] += (u64)xh[j] * MDM.v[12 * i + j];
            }
        }
        gl::fold3(bl, bh, y);
        s[g] = y[0];
        s[g + 1] = y[1];
        s[g + 2] = y[2];
    }
}

// Three partial rounds in one dense pass.  A partial round is y <- M (D y + s e0) with s = sbox(y[0] + c); with z the state whose
// word 0 already went through round 0's S-box:
//   y1[0] = (M z)[0]
//   y2[0] = ((MD) M z)[0] + s1 M[0][0]
//   y3    = (MD)(MD) M z + s1 ((MD) M e0) + s2 (M e0)
// and the integer products of the MDS matrix stay small ((MD)(MD) M <= M^3 < 2^21 per entry, < 2^25 per row), so a row is still two
// 32x32+64 multiply-add chains (accumulators < 2^58): three rounds cost one 12-row pass (+2 terms per row), two single rows and three
// S-boxes instead of three 12-row passes.  The matrix entries are s_mov_b32 literals in the instruction stream (P2_LITERAL_QUAL).  A
// fourth power does not pay: the rows of M^4 sum to 1.04 * 2^32 (the chains would overflow); splitting off its near-constant part
// t * J costs a scalar t * sum(z) added to all 24 accumulators, and the measured gain was 1 %.  The rounds' scalar constants
// (P2_POSEIDON_PUSHED_ROUND_CONSTANTS[12 r]) ride the chains as starting addends: c1, c2 those of this batch's second and third round,
// cn that of the NEXT pass's first round (fused into row 0); this batch's first is already in s[0].
// TAIL (the last batch): cv = the constant VECTOR of the full round that follows, fused into all twelve rows (cn unused).
template <bool TAIL = false>
__device__ __forceinline__ void partial_rounds3(u64 s[12], u64 c1, u64 c2, u64 cn, const u64 *cv = nullptr) {
    u32 xl[12], xh[12];
    const u64 z0 = sbox7_asm(s[0]);  // round 0's scalar is already in s[0]: the previous pass fused it into its row 0
    xl[0] = (u32)z0;
    xh[0] = (u32)(z0 >> 32);
#pragma unroll
    for (int i = 1; i < 12; ++i) {
        xl[i] = (u32)s[i];
        xh[i] = (u32)(s[i] >> 32);
    }
    u64 al = (u32)c1, ah = c1 >> 32;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        al += (u64)xl[j] * P2_POSEIDON_M1_ROW0[j];
        ah += (u64)xh[j] * P2_POSEIDON_M1_ROW0[j];
    }
    const u64 s1 = sbox7_asm(gl::fold1(al, ah));
    const u32 s1l = (u32)s1, s1h = (u32)(s1 >> 32);
    al = (u64)s1l * MDS1.v[0] + (u32)c2;
    ah = (u64)s1h * MDS1.v[0] + (c2 >> 32);
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        al += (u64)xl[j] * MDM.v[j];
        ah += (u64)xh[j] * MDM.v[j];
    }
    const u64 s2 = sbox7_asm(gl::fold1(al, ah));
    const u32 s2l = (u32)s2, s2h = (u32)(s2 >> 32);
#pragma unroll
    for (int g = 0; g < 12; g += 3) {
        u64 bl[3], bh[3], y[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int i = g + t;
            const u64 ci = TAIL ? cv[i] : (i == 0 ? cn : 0);
            bl[t] = (u64)s1l * MDM.v[12 * i] + (u64)s2l * MDS1.v[12 * i] + (u32)ci;
            bh[t] = (u64)s1h * MDM.v[12 * i] + (u64)s2h * MDS1.v[12 * i] + (ci >> 32);
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                bl[t] += (u64)xl[j] * MDMDM.v[12 * i + j];
                bh[t] += (u64)xh[j] * MDMDM.v[12 * i + j];
            }
        }
        gl::fold3(bl, bh, y);
        s[g] = y[0];
        s[g + 1] = y[1];
        s[g + 2] = y[2];
    }
}

#ifndef P2HOT_TAIL_BATCH
#define P2HOT_TAIL_BATCH 1  // 1: the last batch of partial rounds is its own copy with round 26's constant vector fused into its rows (eleven
                            // 4-instruction additions less per permutation: -0.3 % cycles, profiles/r06_partial_rounds_ab.txt); 0: one rolled loop
#endif
// the permutation; output words are NOT canonicalised (callers canonicalise what they emit).
// Round r: ARK(r) was already added by the previous MDS (or up front for r = 0); S-box; MDS + ARK(r+1).
// `out_groups`: which output word triples the caller reads (bit g = words 3g..3g+2), `out_single`: one more word (3 or 8) it reads;
// the last MDS skips the rest.  A digest is words 0..3 = (triple 0, word 3); the capacity a full absorb keeps is words 8..11 =
// (word 8, triple 3): four rows instead of the six of two triples.
__device__ inline void permute(u64 s[12], unsigned out_groups = 0xFu, int out_single = -1) {
    const u32 z0 = opaque_zero();
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = ark(s[i], RC[i + z0]);
    int round = 0;
#pragma unroll 1
    for (int k = 0; k < 3; ++k, ++round) {
        sbox_layer(s);
        mds_layer(s, RC_SPLIT + 24 * (round + 1));
    }
    // round 3's S-box layer; its MDS and partial round 4 in ONE dense pass (M^2: mds_partial_round); no constant is fused into
    // round 3's MDS -- the batched partial rounds add their scalars themselves (the passive part of the partial-round constants
    // is pushed forward through the MDS at table-generation time, see the generator)
    sbox_layer(s);
    mds_partial_round(s, P2_POSEIDON_PUSHED_ROUND_CONSTANTS[12 * 4], P2_POSEIDON_PUSHED_ROUND_CONSTANTS[12 * 5]);
    round = 5;
    // partial rounds 5..25 in seven batches of three (each batch's first scalar was fused into row 0 of the pass before it).  The
    // constant vector of round 26 (which absorbed the pushed remainder) rides the LAST batch's rows; that batch is its own copy of the
    // code, outside the rolled loop: fused into the loop's accumulators the vector would be 44 more live SGPRs in EVERY batch
    // (measured: 96 v_readlane per batch of SGPR spill traffic, +4 % instructions)
#pragma unroll 1
    for (int k = 0; k < (P2HOT_TAIL_BATCH ? 6 : 7); ++k, round += 3)
        partial_rounds3(s, P2_POSEIDON_PUSHED_ROUND_CONSTANTS[12 * (round + 1)], P2_POSEIDON_PUSHED_ROUND_CONSTANTS[12 * (round + 2)],
                        P2_POSEIDON_PUSHED_ROUND_CONSTANTS[12 * (round + 3)]);  // the last batch's cn is round 26's word 0
    const u32 z26 = opaque_zero();
    if (P2HOT_TAIL_BATCH) {
        partial_rounds3<true>(s, P2_POSEIDON_PUSHED_ROUND_CONSTANTS[12 * 24], P2_POSEIDON_PUSHED_ROUND_CONSTANTS[12 * 25], 0,
                              P2_POSEIDON_PUSHED_ROUND_CONSTANTS + 12 * 26 + z26);
        round = 26;
    } else {
#pragma unroll
        for (int i = 1; i < 12; ++i) s[i] = ark(s[i], P2_POSEIDON_PUSHED_ROUND_CONSTANTS[12 * round + i + z26]);
    }
#pragma unroll 1
    for (int k = 0; k < 3; ++k, ++round) {
        sbox_layer(s);
        mds_layer(s, RC_SPLIT + 24 * (round + 1));
    }
    sbox_layer(s);
    mds_layer(s, nullptr, out_groups, out_single);
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
    permute(s, 0x1u, 3);  // words 0..3 only: triple 0 + word 3
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = gl::canon(s[i]);
}

#undef RC
#undef RC_SPLIT
}  // namespace poseidon
