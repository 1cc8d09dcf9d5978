// gl_mul3.hpp -- three independent Goldilocks multiplications as ONE hand-scheduled gfx950
// instruction stream (used by the Poseidon S-box, where independent products come in threes).
//
// Why asm: on gfx950 any VALU write of an SGPR/VCC (carry, compare) needs two wait states before
// a VALU reads it, and hipcc's 64-bit code wastes ~9 of 26 instructions per multiply on v_mov /
// v_cmp / v_cndmask glue (measured on the ISA).  Interleaving three products round-robin puts
// exactly two independent instructions between every carry producer and its consumer, so the
// 14-instruction multiply-reduce needs no s_nop:
//    1-4   four v_mad_u64_u32: P = a0*b0, M = a1*b0 + a0*b1 (carry cm), Q = a1*b1
//    5-7   128-bit assembly: lo = {P0, P1+M0}, hi = Q + M1 + carry; cm (weight 2^96 = -1 mod P) is NOT merged into hi
//    8-9   t = lo - hi.hi - cm (mod 2^64; cm enters as the borrow-in of the first subtract), borrow b
//    10    u = t + hi.lo * (2^32-1) (mod 2^64), carry c
//    11-14 the value is u + (c - b) * 2^64 = u + e * (2^32 - 1) with e = c - b in {-1, 0, 1} (never out of [0, 2^64):
//          see the bounds in gl::reduce128): e by a v_cndmask and a v_subb, then u + (e << 32) - e as a 32-bit add on
//          the high word and one SIGNED multiply-add (v_mad_i64_i32, e * -1), which writes the 64-bit result pair
//                                                                               (goldilocks_field.rs:402-415)
// Since round 6 the hot kernels (Poseidon S-boxes, the limb NTT's general multiplies) run mul3cg / mul1cg below: the same 14
// instructions with steps 5-7 (three carry adds) replaced by three co-issued v_mov_b32 and two chained addends; mul3 / mul2 / mad3
// stay for the plonk and FRI kernels.
// Temporaries are fixed VGPR/SGPR pairs (declared as clobbers) because inline-asm operands cannot
// name the halves of a 64-bit register pair.  The emulator build runs these blocks either through its instruction
// interpreter (asm_block.h: the same template strings, hazard- and clobber-checked) or, by default, as gl::mul.
#pragma once
#include "asm_block.h"
#include "gl.hpp"

namespace gl {

// stream register sets: P, M, Q pairs + two SGPR carry pairs
#define P2_SA "v[70:71]", "v70", "v71", "v[72:73]", "v72", "v73", "v[74:75]", "v74", "v75", "s[40:41]", "s[42:43]"
#define P2_SB "v[76:77]", "v76", "v77", "v[78:79]", "v78", "v79", "v[80:81]", "v80", "v81", "s[44:45]", "s[46:47]"
#define P2_SC "v[82:83]", "v82", "v83", "v[84:85]", "v84", "v85", "v[86:87]", "v86", "v87", "s[48:49]", "s[50:51]"

// one instruction of the multiply for one stream; X = register set, a0/a1/b0/b1 = operand names, r0 = the 64-bit result
#define P2_ST1(P, P0, P1, M, M0, M1, Q, Q0, Q1, C1, C2, a0, a1, b0, b1, r0, r1) "v_mad_u64_u32 " P ", " C1 ", %[" a0 "], %[" b0 "], 0\n\t"
#define P2_ST2(P, P0, P1, M, M0, M1, Q, Q0, Q1, C1, C2, a0, a1, b0, b1, r0, r1) "v_mad_u64_u32 " M ", " C1 ", %[" a1 "], %[" b0 "], 0\n\t"
#define P2_ST3(P, P0, P1, M, M0, M1, Q, Q0, Q1, C1, C2, a0, a1, b0, b1, r0, r1) "v_mad_u64_u32 " M ", " C2 ", %[" a0 "], %[" b1 "], " M "\n\t"
#define P2_ST4(P, P0, P1, M, M0, M1, Q, Q0, Q1, C1, C2, a0, a1, b0, b1, r0, r1) "v_mad_u64_u32 " Q ", " C1 ", %[" a1 "], %[" b1 "], 0\n\t"
#define P2_ST5(P, P0, P1, M, M0, M1, Q, Q0, Q1, C1, C2, a0, a1, b0, b1, r0, r1) "v_add_co_u32 " P1 ", " C1 ", " P1 ", " M0 "\n\t"
#define P2_ST6(P, P0, P1, M, M0, M1, Q, Q0, Q1, C1, C2, a0, a1, b0, b1, r0, r1) "v_addc_co_u32 " Q0 ", " C1 ", " Q0 ", " M1 ", " C1 "\n\t"
#define P2_ST7(P, P0, P1, M, M0, M1, Q, Q0, Q1, C1, C2, a0, a1, b0, b1, r0, r1) "v_addc_co_u32 " Q1 ", " C1 ", " Q1 ", 0, " C1 "\n\t"
#define P2_ST8(P, P0, P1, M, M0, M1, Q, Q0, Q1, C1, C2, a0, a1, b0, b1, r0, r1) "v_subb_co_u32 " P0 ", " C1 ", " P0 ", " Q1 ", " C2 "\n\t"
#define P2_ST9(P, P0, P1, M, M0, M1, Q, Q0, Q1, C1, C2, a0, a1, b0, b1, r0, r1) "v_subb_co_u32 " P1 ", " C1 ", " P1 ", 0, " C1 "\n\t"
#define P2_ST10(P, P0, P1, M, M0, M1, Q, Q0, Q1, C1, C2, a0, a1, b0, b1, r0, r1) "v_mad_u64_u32 " P ", " C2 ", " Q0 ", -1, " P "\n\t"
#define P2_ST11(P, P0, P1, M, M0, M1, Q, Q0, Q1, C1, C2, a0, a1, b0, b1, r0, r1) "v_cndmask_b32 " M0 ", 0, 1, " C2 "\n\t"
#define P2_ST12(P, P0, P1, M, M0, M1, Q, Q0, Q1, C1, C2, a0, a1, b0, b1, r0, r1) "v_subb_co_u32 " M0 ", " C1 ", " M0 ", 0, " C1 "\n\t"
#define P2_ST13(P, P0, P1, M, M0, M1, Q, Q0, Q1, C1, C2, a0, a1, b0, b1, r0, r1) "v_add_u32 " P1 ", " P1 ", " M0 "\n\t"
#define P2_ST14(P, P0, P1, M, M0, M1, Q, Q0, Q1, C1, C2, a0, a1, b0, b1, r0, r1) "v_mad_i64_i32 %[" r0 "], " C1 ", " M0 ", -1, " P "\n\t"

#define P2_APPLY(ST, ...) ST(__VA_ARGS__)
#define P2_ROW(ST)                                                   \
    P2_APPLY(ST, P2_SA, "xa0", "xa1", "ya0", "ya1", "ra0", "ra1")    \
    P2_APPLY(ST, P2_SB, "xb0", "xb1", "yb0", "yb1", "rb0", "rb1")    \
    P2_APPLY(ST, P2_SC, "xc0", "xc1", "yc0", "yc1", "rc0", "rc1")

// r[k] = a[k] * b[k] (mod P), k = 0..2; any representatives in, any representative out
__device__ __forceinline__ void mul3(const u64 a[3], const u64 b[3], u64 r[3]) {
    if (!P2_ASM_INTERPRETED()) {  // emulator build with the interpreter off (constant-false in the product)
        for (int k = 0; k < 3; ++k) r[k] = mul(a[k], b[k]);
        return;
    }
    u64 ra, rb, rc;
    P2_ASM(P2_ROW(P2_ST1) P2_ROW(P2_ST2) P2_ROW(P2_ST3) P2_ROW(P2_ST4) P2_ROW(P2_ST5) P2_ROW(P2_ST6) P2_ROW(P2_ST7)
               P2_ROW(P2_ST8) P2_ROW(P2_ST9) P2_ROW(P2_ST10) P2_ROW(P2_ST11) P2_ROW(P2_ST12) P2_ROW(P2_ST13)
                   P2_ROW(P2_ST14),
           (P2_O([ra0], "=&v", ra), P2_O([rb0], "=&v", rb), P2_O([rc0], "=&v", rc)),
           (P2_I([xa0], "v", (u32)a[0]), P2_I([xa1], "v", (u32)(a[0] >> 32)), P2_I([ya0], "v", (u32)b[0]),
            P2_I([ya1], "v", (u32)(b[0] >> 32)), P2_I([xb0], "v", (u32)a[1]), P2_I([xb1], "v", (u32)(a[1] >> 32)),
            P2_I([yb0], "v", (u32)b[1]), P2_I([yb1], "v", (u32)(b[1] >> 32)), P2_I([xc0], "v", (u32)a[2]),
            P2_I([xc1], "v", (u32)(a[2] >> 32)), P2_I([yc0], "v", (u32)b[2]), P2_I([yc1], "v", (u32)(b[2] >> 32))),
           ("v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85",
            "v86", "v87", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51"));
    r[0] = ra;
    r[1] = rb;
    r[2] = rc;
}

// ---- mad3 / mad2: r[k] = a[k] * b[k] + c[k] (mod P) with c ANY 64-bit word, at the price of the plain product.  The halves of c
// ride the first two multiply-adds of a stream: P = a0*b0 + c.lo and M = a1*b0 + c.hi still fit in 64 bits ((2^32-1)^2 + 2^32 - 1),
// M + a0*b1 carries out once as before, and the 128-bit value a*b + c <= 2^128 - 2^64 reduces like any product.  The callers
// pass the halves as 64-bit operands (high word zero): v_mad_u64_u32's addend is a register pair.
#define P2_MADROW1                                                       \
    "v_mad_u64_u32 v[70:71], s[40:41], %[xa0], %[ya0], %[ca0]\n\t"       \
    "v_mad_u64_u32 v[76:77], s[44:45], %[xb0], %[yb0], %[cb0]\n\t"       \
    "v_mad_u64_u32 v[82:83], s[48:49], %[xc0], %[yc0], %[cc0]\n\t"
#define P2_MADROW2                                                       \
    "v_mad_u64_u32 v[72:73], s[40:41], %[xa1], %[ya0], %[ca1]\n\t"       \
    "v_mad_u64_u32 v[78:79], s[44:45], %[xb1], %[yb0], %[cb1]\n\t"       \
    "v_mad_u64_u32 v[84:85], s[48:49], %[xc1], %[yc0], %[cc1]\n\t"
__host__ __device__ __forceinline__ u64 mul_add(u64 a, u64 b, u64 c);
__device__ __forceinline__ void mad3(const u64 a[3], const u64 b[3], const u64 c[3], u64 r[3]) {
    if (!P2_ASM_INTERPRETED()) {
        for (int k = 0; k < 3; ++k) r[k] = mul_add(a[k], b[k], c[k]);
        return;
    }
    u64 ra, rb, rc;
    P2_ASM(P2_MADROW1 P2_MADROW2 P2_ROW(P2_ST3) P2_ROW(P2_ST4) P2_ROW(P2_ST5) P2_ROW(P2_ST6) P2_ROW(P2_ST7)
               P2_ROW(P2_ST8) P2_ROW(P2_ST9) P2_ROW(P2_ST10) P2_ROW(P2_ST11) P2_ROW(P2_ST12) P2_ROW(P2_ST13)
                   P2_ROW(P2_ST14),
           (P2_O([ra0], "=&v", ra), P2_O([rb0], "=&v", rb), P2_O([rc0], "=&v", rc)),
           (P2_I([xa0], "v", (u32)a[0]), P2_I([xa1], "v", (u32)(a[0] >> 32)), P2_I([ya0], "v", (u32)b[0]),
            P2_I([ya1], "v", (u32)(b[0] >> 32)), P2_I([xb0], "v", (u32)a[1]), P2_I([xb1], "v", (u32)(a[1] >> 32)),
            P2_I([yb0], "v", (u32)b[1]), P2_I([yb1], "v", (u32)(b[1] >> 32)), P2_I([xc0], "v", (u32)a[2]),
            P2_I([xc1], "v", (u32)(a[2] >> 32)), P2_I([yc0], "v", (u32)b[2]), P2_I([yc1], "v", (u32)(b[2] >> 32)),
            P2_I([ca0], "v", (u64)(u32)c[0]), P2_I([ca1], "v", c[0] >> 32), P2_I([cb0], "v", (u64)(u32)c[1]), P2_I([cb1], "v", c[1] >> 32),
            P2_I([cc0], "v", (u64)(u32)c[2]), P2_I([cc1], "v", c[2] >> 32)),
           ("v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85",
            "v86", "v87", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51"));
    r[0] = ra;
    r[1] = rb;
    r[2] = rc;
}

// ---- mul2: two interleaved streams (what is left of a group of independent products after the threes) ----
// One foreign instruction sits between a carry producer and its consumer, so the four adjacent producer/consumer rows
// (5->6, 6->7, 8->9, 10->11) are followed by one more wait state each (`s_nop 0`): 28 instructions + 4 single wait states
// instead of two single streams' 28 + 10 double ones.
#define P2_NOP0 "s_nop 0\n\t"
#define P2_ROW2(ST)                                                  \
    P2_APPLY(ST, P2_SA, "xa0", "xa1", "ya0", "ya1", "ra0", "ra1")    \
    P2_APPLY(ST, P2_SB, "xb0", "xb1", "yb0", "yb1", "rb0", "rb1")
__device__ __forceinline__ void mul2(const u64 a[2], const u64 b[2], u64 r[2]) {
    if (!P2_ASM_INTERPRETED()) {
        for (int k = 0; k < 2; ++k) r[k] = mul(a[k], b[k]);
        return;
    }
    u64 ra, rb;
    P2_ASM(P2_ROW2(P2_ST1) P2_ROW2(P2_ST2) P2_ROW2(P2_ST3) P2_ROW2(P2_ST4) P2_ROW2(P2_ST5) P2_NOP0 P2_ROW2(P2_ST6) P2_NOP0
               P2_ROW2(P2_ST7) P2_ROW2(P2_ST8) P2_NOP0 P2_ROW2(P2_ST9) P2_ROW2(P2_ST10) P2_NOP0 P2_ROW2(P2_ST11)
                   P2_ROW2(P2_ST12) P2_ROW2(P2_ST13) P2_ROW2(P2_ST14),
           (P2_O([ra0], "=&v", ra), P2_O([rb0], "=&v", rb)),
           (P2_I([xa0], "v", (u32)a[0]), P2_I([xa1], "v", (u32)(a[0] >> 32)), P2_I([ya0], "v", (u32)b[0]),
            P2_I([ya1], "v", (u32)(b[0] >> 32)), P2_I([xb0], "v", (u32)a[1]), P2_I([xb1], "v", (u32)(a[1] >> 32)),
            P2_I([yb0], "v", (u32)b[1]), P2_I([yb1], "v", (u32)(b[1] >> 32))),
           ("v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "s40", "s41", "s42", "s43",
            "s44", "s45", "s46", "s47"));
    r[0] = ra;
    r[1] = rb;
}

// ---- mul1: one multiplication as a single stream (dependent S-box chains of the partial rounds) ----
// Same 14 instructions as a mul3 stream; the two wait states every carry consumer needs after its
// producer are explicit `s_nop 1` (they cost this wave latency, not the SIMD issue slots).
#define P2_NOP "s_nop 1\n\t"
#define P2_A1(ST, ...) P2_APPLY(ST, __VA_ARGS__, "xa0", "xa1", "ya0", "ya1", "ra0", "ra1")
#define P2_MUL1_BODY(SET)                                                                                     \
    P2_A1(P2_ST1, SET) P2_A1(P2_ST2, SET) P2_A1(P2_ST3, SET) P2_A1(P2_ST4, SET) P2_A1(P2_ST5, SET) P2_NOP      \
    P2_A1(P2_ST6, SET) P2_NOP P2_A1(P2_ST7, SET) P2_A1(P2_ST8, SET) P2_NOP P2_A1(P2_ST9, SET)                \
    P2_A1(P2_ST10, SET) P2_NOP P2_A1(P2_ST11, SET) P2_A1(P2_ST12, SET) P2_A1(P2_ST13, SET) P2_A1(P2_ST14, SET)

__device__ __forceinline__ u64 mul1(u64 a, u64 b) {
    if (!P2_ASM_INTERPRETED()) return mul(a, b);
    u64 ra;
    P2_ASM(P2_MUL1_BODY(P2_SA), (P2_O([ra0], "=&v", ra)),
           (P2_I([xa0], "v", (u32)a), P2_I([xa1], "v", (u32)(a >> 32)), P2_I([ya0], "v", (u32)b),
            P2_I([ya1], "v", (u32)(b >> 32))),
           ("v70", "v71", "v72", "v73", "v74", "v75", "s40", "s41", "s42", "s43"));
    return ra;
}

// the same stream on a low register set, for kernels that must stay within 64 VGPRs (the NTT passes)
#define P2_SN "v[58:59]", "v58", "v59", "v[60:61]", "v60", "v61", "v[62:63]", "v62", "v63", "s[60:61]", "s[62:63]"
__device__ __forceinline__ u64 mul1_lowregs(u64 a, u64 b) {
    if (!P2_ASM_INTERPRETED()) return mul(a, b);
    u64 ra;
    P2_ASM(P2_MUL1_BODY(P2_SN), (P2_O([ra0], "=&v", ra)),
           (P2_I([xa0], "v", (u32)a), P2_I([xa1], "v", (u32)(a >> 32)), P2_I([ya0], "v", (u32)b),
            P2_I([ya1], "v", (u32)(b >> 32))),
           ("v58", "v59", "v60", "v61", "v62", "v63", "s60", "s61", "s62", "s63"));
    return ra;
}

// ---- mul3cf / mul1cf: the same product with a CARRY-FREE 128-bit assembly (round 6; used by the Poseidon S-boxes) -------------------
// On gfx950 a v_mov_b32 next to a multiply-add costs ~1.5 cycles where a carry add costs 4.3 (profiles/r06_ubench_cheap.txt: cheap
// FP32-class instructions and v_mov overlap the 64-bit multiply-add, integer carry arithmetic does not).  So the partial products are
// chained through the multiply-add's 64-bit ADDEND instead of being summed with v_add_co / v_addc -- every step fits 64 bits:
//    T.hi = 0
//    P  = a0*b0                      T.lo = P.hi
//    M  = a1*b0 + T   (< 2^64)       T.lo = M.lo
//    N  = a0*b1 + T   (< 2^64)       T.lo = M.hi        N.lo is limb 1 of the product, P.lo limb 0
//    T  = a1*b1 + T                  T += N.hi * 1      (<= (2^32-1)^2 + 2 (2^32-1) = 2^64 - 1): T = the high 64 bits
// 5 multiply-adds + 4 moves instead of 4 multiply-adds + 3 carry adds; the reduction is the one above (steps 8-14), reading
// lo = {P.lo, N.lo}, hi = T.  16 instructions, measured as a dependency-free mix at 52.0 cycles per product against 56.6
// (tools/ubench.hip probes 170 / 171).  Register sets: P, M, N, T pairs + two SGPR carry pairs per stream.
#define P2_CA "v[70:71]", "v70", "v71", "v[72:73]", "v72", "v73", "v[74:75]", "v74", "v75", "v[76:77]", "v76", "v77", "s[40:41]", "s[42:43]"
#define P2_CB "v[78:79]", "v78", "v79", "v[80:81]", "v80", "v81", "v[82:83]", "v82", "v83", "v[84:85]", "v84", "v85", "s[44:45]", "s[46:47]"
#define P2_CC "v[86:87]", "v86", "v87", "v[88:89]", "v88", "v89", "v[90:91]", "v90", "v91", "v[92:93]", "v92", "v93", "s[48:49]", "s[50:51]"
#define P2_CF0(P, P0, P1, M, M0, M1, N, N0, N1, T, T0, T1, C1, C2, a0, a1, b0, b1, r0) "v_mov_b32 " T1 ", 0\n\t"
#define P2_CF1(P, P0, P1, M, M0, M1, N, N0, N1, T, T0, T1, C1, C2, a0, a1, b0, b1, r0) "v_mad_u64_u32 " P ", " C1 ", %[" a0 "], %[" b0 "], 0\n\t"
#define P2_CF2(P, P0, P1, M, M0, M1, N, N0, N1, T, T0, T1, C1, C2, a0, a1, b0, b1, r0) "v_mov_b32 " T0 ", " P1 "\n\t"
#define P2_CF3(P, P0, P1, M, M0, M1, N, N0, N1, T, T0, T1, C1, C2, a0, a1, b0, b1, r0) "v_mad_u64_u32 " M ", " C1 ", %[" a1 "], %[" b0 "], " T "\n\t"
#define P2_CF4(P, P0, P1, M, M0, M1, N, N0, N1, T, T0, T1, C1, C2, a0, a1, b0, b1, r0) "v_mov_b32 " T0 ", " M0 "\n\t"
#define P2_CF5(P, P0, P1, M, M0, M1, N, N0, N1, T, T0, T1, C1, C2, a0, a1, b0, b1, r0) "v_mad_u64_u32 " N ", " C1 ", %[" a0 "], %[" b1 "], " T "\n\t"
#define P2_CF6(P, P0, P1, M, M0, M1, N, N0, N1, T, T0, T1, C1, C2, a0, a1, b0, b1, r0) "v_mov_b32 " T0 ", " M1 "\n\t"
#define P2_CF7(P, P0, P1, M, M0, M1, N, N0, N1, T, T0, T1, C1, C2, a0, a1, b0, b1, r0) "v_mad_u64_u32 " T ", " C1 ", %[" a1 "], %[" b1 "], " T "\n\t"
#define P2_CF8(P, P0, P1, M, M0, M1, N, N0, N1, T, T0, T1, C1, C2, a0, a1, b0, b1, r0) "v_mad_u64_u32 " T ", " C1 ", " N1 ", 1, " T "\n\t"
#define P2_CF9(P, P0, P1, M, M0, M1, N, N0, N1, T, T0, T1, C1, C2, a0, a1, b0, b1, r0) "v_sub_co_u32 " M0 ", " C1 ", " P0 ", " T1 "\n\t"
#define P2_CF10(P, P0, P1, M, M0, M1, N, N0, N1, T, T0, T1, C1, C2, a0, a1, b0, b1, r0) "v_subb_co_u32 " M1 ", " C1 ", " N0 ", 0, " C1 "\n\t"
#define P2_CF11(P, P0, P1, M, M0, M1, N, N0, N1, T, T0, T1, C1, C2, a0, a1, b0, b1, r0) "v_mad_u64_u32 " M ", " C2 ", " T0 ", -1, " M "\n\t"
#define P2_CF12(P, P0, P1, M, M0, M1, N, N0, N1, T, T0, T1, C1, C2, a0, a1, b0, b1, r0) "v_cndmask_b32 " N0 ", 0, 1, " C2 "\n\t"
#define P2_CF13(P, P0, P1, M, M0, M1, N, N0, N1, T, T0, T1, C1, C2, a0, a1, b0, b1, r0) "v_subb_co_u32 " N0 ", " C1 ", " N0 ", 0, " C1 "\n\t"
#define P2_CF14(P, P0, P1, M, M0, M1, N, N0, N1, T, T0, T1, C1, C2, a0, a1, b0, b1, r0) "v_add_u32 " M1 ", " M1 ", " N0 "\n\t"
#define P2_CF15(P, P0, P1, M, M0, M1, N, N0, N1, T, T0, T1, C1, C2, a0, a1, b0, b1, r0) "v_mad_i64_i32 %[" r0 "], " C1 ", " N0 ", -1, " M "\n\t"
#define P2_CROW(ST)                                          \
    P2_APPLY(ST, P2_CA, "xa0", "xa1", "ya0", "ya1", "ra0")   \
    P2_APPLY(ST, P2_CB, "xb0", "xb1", "yb0", "yb1", "rb0")   \
    P2_APPLY(ST, P2_CC, "xc0", "xc1", "yc0", "yc1", "rc0")
#define P2_CF_CLOBBERS3                                                                                                      \
    "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", \
        "v88", "v89", "v90", "v91", "v92", "v93", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51"

// r[k] = a[k] * b[k] (mod P), k = 0..2: three carry-free streams round-robin (every SGPR producer / consumer pair three
// instructions apart: no s_nop)
__device__ __forceinline__ void mul3cf(const u64 a[3], const u64 b[3], u64 r[3]) {
    if (!P2_ASM_INTERPRETED()) {
        for (int k = 0; k < 3; ++k) r[k] = mul(a[k], b[k]);
        return;
    }
    u64 ra, rb, rc;
    P2_ASM(P2_CROW(P2_CF0) P2_CROW(P2_CF1) P2_CROW(P2_CF2) P2_CROW(P2_CF3) P2_CROW(P2_CF4) P2_CROW(P2_CF5) P2_CROW(P2_CF6) P2_CROW(P2_CF7)
               P2_CROW(P2_CF8) P2_CROW(P2_CF9) P2_CROW(P2_CF10) P2_CROW(P2_CF11) P2_CROW(P2_CF12) P2_CROW(P2_CF13) P2_CROW(P2_CF14)
                   P2_CROW(P2_CF15),
           (P2_O([ra0], "=&v", ra), P2_O([rb0], "=&v", rb), P2_O([rc0], "=&v", rc)),
           (P2_I([xa0], "v", (u32)a[0]), P2_I([xa1], "v", (u32)(a[0] >> 32)), P2_I([ya0], "v", (u32)b[0]),
            P2_I([ya1], "v", (u32)(b[0] >> 32)), P2_I([xb0], "v", (u32)a[1]), P2_I([xb1], "v", (u32)(a[1] >> 32)),
            P2_I([yb0], "v", (u32)b[1]), P2_I([yb1], "v", (u32)(b[1] >> 32)), P2_I([xc0], "v", (u32)a[2]),
            P2_I([xc1], "v", (u32)(a[2] >> 32)), P2_I([yc0], "v", (u32)b[2]), P2_I([yc1], "v", (u32)(b[2] >> 32))),
           (P2_CF_CLOBBERS3));
    r[0] = ra;
    r[1] = rb;
    r[2] = rc;
}

// one carry-free stream (the partial rounds' dependent S-box chains): the two adjacent SGPR producer / consumer rows (9 -> 10,
// 11 -> 12) get their two wait states as `s_nop 1`
#define P2_CA1(ST) P2_APPLY(ST, P2_CA, "xa0", "xa1", "ya0", "ya1", "ra0")
__device__ __forceinline__ u64 mul1cf(u64 a, u64 b) {
    if (!P2_ASM_INTERPRETED()) return mul(a, b);
    u64 ra;
    P2_ASM(P2_CA1(P2_CF0) P2_CA1(P2_CF1) P2_CA1(P2_CF2) P2_CA1(P2_CF3) P2_CA1(P2_CF4) P2_CA1(P2_CF5) P2_CA1(P2_CF6) P2_CA1(P2_CF7)
               P2_CA1(P2_CF8) P2_CA1(P2_CF9) P2_NOP P2_CA1(P2_CF10) P2_CA1(P2_CF11) P2_NOP P2_CA1(P2_CF12) P2_CA1(P2_CF13)
                   P2_CA1(P2_CF14) P2_CA1(P2_CF15),
           (P2_O([ra0], "=&v", ra)),
           (P2_I([xa0], "v", (u32)a), P2_I([xa1], "v", (u32)(a >> 32)), P2_I([ya0], "v", (u32)b), P2_I([ya1], "v", (u32)(b >> 32))),
           ("v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "s40", "s41", "s42", "s43"));
    return ra;
}

// ---- mul3cg / mul1cg: the 14-instruction form of the carry-free assembly (round 6, second step; what the Poseidon S-boxes run) -------
// Only the two chained addends that SAVE carry adds are kept; the one carry the 128-bit assembly really has stays a carry, because it is
// free on both ends -- the multiply-add writes it (VOP3B sdst) and the reduction's first subtract takes it as its borrow-in:
//    T.hi = 0
//    P  = a0*b0                      T.lo = P.hi
//    M  = a1*b0 + T   (< 2^64)
//    N  = a0*b1 + M   (carry cm, weight 2^96 = -1 mod P)          N.lo is limb 1 of the product, P.lo limb 0
//                                    T.lo = N.hi
//    T  = a1*b1 + T   (< 2^64)       T = the high 64 bits without cm: exactly the (lo, hi, cm) of mul3's steps 1-7, so the reduction
//                                    (steps 8-14, cm as the borrow-in of the first subtract) and its bounds are mul3's.
// 4 multiply-adds + 3 moves + the 7-instruction reduction: 14 instructions like mul3, of which 11 pay a full issue slot (mul3: 14,
// mul3cf: 12 of 16).
#define P2_CG0(P, P0, P1, M, M0, M1, N, N0, N1, T, T0, T1, C1, C2, a0, a1, b0, b1, r0) "v_mov_b32 " T1 ", 0\n\t"
#define P2_CG1(P, P0, P1, M, M0, M1, N, N0, N1, T, T0, T1, C1, C2, a0, a1, b0, b1, r0) "v_mad_u64_u32 " P ", " C1 ", %[" a0 "], %[" b0 "], 0\n\t"
#define P2_CG2(P, P0, P1, M, M0, M1, N, N0, N1, T, T0, T1, C1, C2, a0, a1, b0, b1, r0) "v_mov_b32 " T0 ", " P1 "\n\t"
#define P2_CG3(P, P0, P1, M, M0, M1, N, N0, N1, T, T0, T1, C1, C2, a0, a1, b0, b1, r0) "v_mad_u64_u32 " M ", " C1 ", %[" a1 "], %[" b0 "], " T "\n\t"
#define P2_CG4(P, P0, P1, M, M0, M1, N, N0, N1, T, T0, T1, C1, C2, a0, a1, b0, b1, r0) "v_mad_u64_u32 " N ", " C2 ", %[" a0 "], %[" b1 "], " M "\n\t"
#define P2_CG5(P, P0, P1, M, M0, M1, N, N0, N1, T, T0, T1, C1, C2, a0, a1, b0, b1, r0) "v_mov_b32 " T0 ", " N1 "\n\t"
#define P2_CG6(P, P0, P1, M, M0, M1, N, N0, N1, T, T0, T1, C1, C2, a0, a1, b0, b1, r0) "v_mad_u64_u32 " T ", " C1 ", %[" a1 "], %[" b1 "], " T "\n\t"
#define P2_CG7(P, P0, P1, M, M0, M1, N, N0, N1, T, T0, T1, C1, C2, a0, a1, b0, b1, r0) "v_subb_co_u32 " M0 ", " C1 ", " P0 ", " T1 ", " C2 "\n\t"
#define P2_CG8 P2_CF10
#define P2_CG9 P2_CF11
#define P2_CG10 P2_CF12
#define P2_CG11 P2_CF13
#define P2_CG12 P2_CF14
#define P2_CG13 P2_CF15

// r[k] = a[k] * b[k] (mod P), k = 0..2: three streams round-robin (every SGPR producer / consumer pair at least three instructions
// apart: no s_nop)
__device__ __forceinline__ void mul3cg(const u64 a[3], const u64 b[3], u64 r[3]) {
    if (!P2_ASM_INTERPRETED()) {
        for (int k = 0; k < 3; ++k) r[k] = mul(a[k], b[k]);
        return;
    }
    u64 ra, rb, rc;
    P2_ASM(P2_CROW(P2_CG0) P2_CROW(P2_CG1) P2_CROW(P2_CG2) P2_CROW(P2_CG3) P2_CROW(P2_CG4) P2_CROW(P2_CG5) P2_CROW(P2_CG6) P2_CROW(P2_CG7)
               P2_CROW(P2_CG8) P2_CROW(P2_CG9) P2_CROW(P2_CG10) P2_CROW(P2_CG11) P2_CROW(P2_CG12) P2_CROW(P2_CG13),
           (P2_O([ra0], "=&v", ra), P2_O([rb0], "=&v", rb), P2_O([rc0], "=&v", rc)),
           (P2_I([xa0], "v", (u32)a[0]), P2_I([xa1], "v", (u32)(a[0] >> 32)), P2_I([ya0], "v", (u32)b[0]),
            P2_I([ya1], "v", (u32)(b[0] >> 32)), P2_I([xb0], "v", (u32)a[1]), P2_I([xb1], "v", (u32)(a[1] >> 32)),
            P2_I([yb0], "v", (u32)b[1]), P2_I([yb1], "v", (u32)(b[1] >> 32)), P2_I([xc0], "v", (u32)a[2]),
            P2_I([xc1], "v", (u32)(a[2] >> 32)), P2_I([yc0], "v", (u32)b[2]), P2_I([yc1], "v", (u32)(b[2] >> 32))),
           (P2_CF_CLOBBERS3));
    r[0] = ra;
    r[1] = rb;
    r[2] = rc;
}

// mul3ch: mul3cg with ONE (x, 0) pair shared by the three streams (P2HOT_SBOX_CF=3) -- each move sits directly in front of the
// multiply-add that reads the pair, the high multiply-add writes the (dead) M pair instead of T, so T.hi is zeroed once per block
// instead of once per stream: 40 instructions per three products instead of 42, two VGPR pairs fewer.
#define P2_HT "v[76:77]", "v76", "v77"
#define P2_HA "v[70:71]", "v70", "v71", "v[72:73]", "v72", "v73", "v[74:75]", "v74", "v75", P2_HT, "s[40:41]", "s[42:43]"
#define P2_HB "v[78:79]", "v78", "v79", "v[80:81]", "v80", "v81", "v[82:83]", "v82", "v83", P2_HT, "s[44:45]", "s[46:47]"
#define P2_HC "v[86:87]", "v86", "v87", "v[88:89]", "v88", "v89", "v[90:91]", "v90", "v91", P2_HT, "s[48:49]", "s[50:51]"
#define P2_CH23(...) P2_CG2(__VA_ARGS__) P2_CG3(__VA_ARGS__)
#define P2_CH6(P, P0, P1, M, M0, M1, N, N0, N1, T, T0, T1, C1, C2, a0, a1, b0, b1, r0) "v_mad_u64_u32 " M ", " C1 ", %[" a1 "], %[" b1 "], " T "\n\t"
#define P2_CH56(...) P2_CG5(__VA_ARGS__) P2_CH6(__VA_ARGS__)
#define P2_CH7(P, P0, P1, M, M0, M1, N, N0, N1, T, T0, T1, C1, C2, a0, a1, b0, b1, r0) "v_subb_co_u32 " P0 ", " C1 ", " P0 ", " M1 ", " C2 "\n\t"
#define P2_CH8(P, P0, P1, M, M0, M1, N, N0, N1, T, T0, T1, C1, C2, a0, a1, b0, b1, r0) "v_subb_co_u32 " P1 ", " C1 ", " N0 ", 0, " C1 "\n\t"
#define P2_CH9(P, P0, P1, M, M0, M1, N, N0, N1, T, T0, T1, C1, C2, a0, a1, b0, b1, r0) "v_mad_u64_u32 " P ", " C2 ", " M0 ", -1, " P "\n\t"
#define P2_CH12(P, P0, P1, M, M0, M1, N, N0, N1, T, T0, T1, C1, C2, a0, a1, b0, b1, r0) "v_add_u32 " P1 ", " P1 ", " N0 "\n\t"
#define P2_CH13(P, P0, P1, M, M0, M1, N, N0, N1, T, T0, T1, C1, C2, a0, a1, b0, b1, r0) "v_mad_i64_i32 %[" r0 "], " C1 ", " N0 ", -1, " P "\n\t"
#define P2_HROW(ST)                                          \
    P2_APPLY(ST, P2_HA, "xa0", "xa1", "ya0", "ya1", "ra0")   \
    P2_APPLY(ST, P2_HB, "xb0", "xb1", "yb0", "yb1", "rb0")   \
    P2_APPLY(ST, P2_HC, "xc0", "xc1", "yc0", "yc1", "rc0")
__device__ __forceinline__ void mul3ch(const u64 a[3], const u64 b[3], u64 r[3]) {
    if (!P2_ASM_INTERPRETED()) {
        for (int k = 0; k < 3; ++k) r[k] = mul(a[k], b[k]);
        return;
    }
    u64 ra, rb, rc;
    P2_ASM("v_mov_b32 v77, 0\n\t" P2_HROW(P2_CG1) P2_HROW(P2_CH23) P2_HROW(P2_CG4) P2_HROW(P2_CH56) P2_HROW(P2_CH7) P2_HROW(P2_CH8)
               P2_HROW(P2_CH9) P2_HROW(P2_CG10) P2_HROW(P2_CG11) P2_HROW(P2_CH12) P2_HROW(P2_CH13),
           (P2_O([ra0], "=&v", ra), P2_O([rb0], "=&v", rb), P2_O([rc0], "=&v", rc)),
           (P2_I([xa0], "v", (u32)a[0]), P2_I([xa1], "v", (u32)(a[0] >> 32)), P2_I([ya0], "v", (u32)b[0]),
            P2_I([ya1], "v", (u32)(b[0] >> 32)), P2_I([xb0], "v", (u32)a[1]), P2_I([xb1], "v", (u32)(a[1] >> 32)),
            P2_I([yb0], "v", (u32)b[1]), P2_I([yb1], "v", (u32)(b[1] >> 32)), P2_I([xc0], "v", (u32)a[2]),
            P2_I([xc1], "v", (u32)(a[2] >> 32)), P2_I([yc0], "v", (u32)b[2]), P2_I([yc1], "v", (u32)(b[2] >> 32))),
           ("v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v86", "v87", "v88", "v89",
            "v90", "v91", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51"));
    r[0] = ra;
    r[1] = rb;
    r[2] = rc;
}

// one stream (the partial rounds' dependent S-box chains): cm has its two wait states from the move and the multiply-add between its
// producer and its consumer; the two adjacent SGPR producer / consumer rows (7 -> 8, 9 -> 10) get theirs as `s_nop 1`
__device__ __forceinline__ u64 mul1cg(u64 a, u64 b) {
    if (!P2_ASM_INTERPRETED()) return mul(a, b);
    u64 ra;
    P2_ASM(P2_CA1(P2_CG0) P2_CA1(P2_CG1) P2_CA1(P2_CG2) P2_CA1(P2_CG3) P2_CA1(P2_CG4) P2_CA1(P2_CG5) P2_CA1(P2_CG6) P2_CA1(P2_CG7) P2_NOP
               P2_CA1(P2_CG8) P2_CA1(P2_CG9) P2_NOP P2_CA1(P2_CG10) P2_CA1(P2_CG11) P2_CA1(P2_CG12) P2_CA1(P2_CG13),
           (P2_O([ra0], "=&v", ra)),
           (P2_I([xa0], "v", (u32)a), P2_I([xa1], "v", (u32)(a >> 32)), P2_I([ya0], "v", (u32)b), P2_I([ya1], "v", (u32)(b >> 32))),
           ("v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "s40", "s41", "s42", "s43"));
    return ra;
}

// ---- fold3: three MDS-row recombinations in one interleaved stream ------------------------------
// A row's two accumulators al = sum c*x.lo32, ah = sum c*x.hi32 (each < 2^63) stand for
//   y = al + ah*2^32 = al + ah.lo * 2^32 + ah.hi * 2^64 = al + ah.hi * (2^32 - 1) + ah.lo * 2^32   (mod P):
//   1 v_mad_u64_u32 T = ah.hi * -1 + al      (< 2^64: no carry)
//   2 v_add_co      T.hi += ah.lo            (carry c: weight 2^64 = 2^32 - 1)
//   3 v_cndmask     e = c ? -1 : 0           4 v_mad_u64_u32 r = e * 1 + T   (T < 2^63 after a wrap: no second carry)
// a0 names the 64-bit operand al, b0 / b1 the halves of ah, r0 the 64-bit result.
#define P2_FD1(P, P0, P1, M, M0, M1, Q, Q0, Q1, C1, C2, a0, a1, b0, b1, r0, r1) "v_mad_u64_u32 " P ", " C1 ", %[" b1 "], -1, %[" a0 "]\n\t"
#define P2_FD2(P, P0, P1, M, M0, M1, Q, Q0, Q1, C1, C2, a0, a1, b0, b1, r0, r1) "v_add_co_u32 " P1 ", " C1 ", " P1 ", %[" b0 "]\n\t"
#define P2_FD3(P, P0, P1, M, M0, M1, Q, Q0, Q1, C1, C2, a0, a1, b0, b1, r0, r1) "v_cndmask_b32 " M0 ", 0, -1, " C1 "\n\t"
#define P2_FD4(P, P0, P1, M, M0, M1, Q, Q0, Q1, C1, C2, a0, a1, b0, b1, r0, r1) "v_mad_u64_u32 %[" r0 "], " C1 ", " M0 ", 1, " P "\n\t"

// the C statement of one row recombination (the emulator's fast path)
__host__ __device__ __forceinline__ u64 fold_row_c(u64 al, u64 ah) {
    u32 k1;
    u32 w1 = addc32((u32)(al >> 32), (u32)ah, 0u, &k1);
    u32 w2 = (u32)(ah >> 32) + k1;
    u64 lo64 = ((u64)w1 << 32) | (u32)al;
    u64 t = (u64)w2 * 0xFFFFFFFFu + lo64;
    return fold_carry(t, t < lo64);
}

// y[k] = al[k] + ah[k] * 2^32 (mod P), k = 0..2, for al, ah < 2^63
__device__ __forceinline__ void fold3(const u64 al[3], const u64 ah[3], u64 y[3]) {
    if (!P2_ASM_INTERPRETED()) {
        for (int k = 0; k < 3; ++k) y[k] = fold_row_c(al[k], ah[k]);
        return;
    }
    u64 ra, rb, rc;  // 64-bit outputs: the last instruction of a stream writes the register pair
    P2_ASM(P2_ROW(P2_FD1) P2_ROW(P2_FD2) P2_ROW(P2_FD3) P2_ROW(P2_FD4),
           (P2_O([ra0], "=&v", ra), P2_O([rb0], "=&v", rb), P2_O([rc0], "=&v", rc)),
           (P2_I([xa0], "v", al[0]), P2_I([ya0], "v", (u32)ah[0]), P2_I([ya1], "v", (u32)(ah[0] >> 32)),
            P2_I([xb0], "v", al[1]), P2_I([yb0], "v", (u32)ah[1]), P2_I([yb1], "v", (u32)(ah[1] >> 32)),
            P2_I([xc0], "v", al[2]), P2_I([yc0], "v", (u32)ah[2]), P2_I([yc1], "v", (u32)(ah[2] >> 32))),
           ("v70", "v71", "v72", "v76", "v77", "v78", "v82", "v83", "v84", "s40", "s41", "s44", "s45", "s48", "s49"));
    y[0] = ra;
    y[1] = rb;
    y[2] = rc;
}
// two row recombinations, interleaved (one more wait state after the carry row)
__device__ __forceinline__ void fold2(const u64 al[2], const u64 ah[2], u64 y[2]) {
    if (!P2_ASM_INTERPRETED()) {
        for (int k = 0; k < 2; ++k) y[k] = fold_row_c(al[k], ah[k]);
        return;
    }
    u64 ra, rb;
    P2_ASM(P2_ROW2(P2_FD1) P2_ROW2(P2_FD2) P2_NOP0 P2_ROW2(P2_FD3) P2_ROW2(P2_FD4),
           (P2_O([ra0], "=&v", ra), P2_O([rb0], "=&v", rb)),
           (P2_I([xa0], "v", al[0]), P2_I([ya0], "v", (u32)ah[0]), P2_I([ya1], "v", (u32)(ah[0] >> 32)),
            P2_I([xb0], "v", al[1]), P2_I([yb0], "v", (u32)ah[1]), P2_I([yb1], "v", (u32)(ah[1] >> 32))),
           ("v70", "v71", "v72", "v76", "v77", "v78", "s40", "s41", "s44", "s45"));
    y[0] = ra;
    y[1] = rb;
}
// one row recombination as a single stream (the batched partial rounds' single rows), explicit wait states
__device__ __forceinline__ u64 fold1(u64 al, u64 ah) {
    if (!P2_ASM_INTERPRETED()) return fold_row_c(al, ah);
    u64 ra;
    P2_ASM(P2_A1(P2_FD1, P2_SA) P2_A1(P2_FD2, P2_SA) P2_NOP P2_A1(P2_FD3, P2_SA) P2_A1(P2_FD4, P2_SA),
           (P2_O([ra0], "=&v", ra)), (P2_I([xa0], "v", al), P2_I([ya0], "v", (u32)ah), P2_I([ya1], "v", (u32)(ah >> 32))),
           ("v70", "v71", "v72", "s40", "s41"));
    return ra;
}

}  // namespace gl
