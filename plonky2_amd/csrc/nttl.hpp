// nttl.hpp -- the NTT passes in a carry-free limb form (the production path for 4096-element tiles, round 3).
//
// Same pass algebra, tile, twiddle placement and output order as ntt.hpp's ntt_regpass_kernel (so it replaces the same
// reference code: field/src/fft.rs:165-202 fft_classic, :68-91 ifft_with_options, polynomial/mod.rs:280-293 coset LDE);
// what changes is the arithmetic inside a radix-8 unit.  Measured on the MI355X (profiles/r03_ubench.txt): every
// carry-writing / carry-reading / 64-bit / VOP3 instruction costs ~2x a plain 32-bit VOP2 add, and a 64-bit modular
// add/sub is 7-9 of them.  So inside a unit an element is held as four signed 32-bit limbs in base B = 2^24,
//      x = l0 + l1*B + l2*B^2 + l3*B^3      in the ring Z[B]/(B^4 + 1)   (P divides 2^96 + 1 = B^4 + 1),
// where
//   * add / sub are four plain v_add_u32 / v_sub_u32 -- no carries, no VCC, no hazards; three butterfly layers grow a
//     limb by three bits, far from the 32-bit edge;
//   * the radix-8 twiddles are powers of w_8 = 2^24 = B: a multiplication by B^m is a negacyclic ROTATION of the limbs
//     (register renaming plus swapping the operands of the subtraction that produced them) -- free;
//   * the one general twiddle per output, w, is applied together with the conversion back to a 64-bit word: with the
//     table holding W_i = w * B^i mod P (i = 0..3) the product is  sum_i L_i * W_i  = two 4-term v_mad_u64_u32 chains over
//     the 32-bit halves of the W_i (each < 2^63) and one 4-instruction fold -- 12 instructions instead of a 14-instruction
//     multiply plus a carry-propagating recombination.  The limbs are made non-negative first by adding a limb
//     pattern that represents a multiple of P to input 0 of the unit only (input 0 reaches every output with
//     coefficient +1).
// A unit of eight points then costs ~130 cheap + ~110 full-price instructions instead of ~430 full-price ones.
#pragma once
#include "gl.hpp"
#include "gl_mul3.hpp"
#include "ntt.hpp"

namespace nttl {
using gl::u32;
using gl::u64;

constexpr int TILE_LOG = 12;
constexpr int NT = 512;  // threads per TILE (a workgroup is NT * DUAL threads: DUAL = 2 transforms two tiles side by side)
// P2HOT_LIMB_DUAL: also build the two-tiles-per-workgroup form (1024 threads = 16 waves = the CU's four waves per SIMD in ONE barrier
// domain) and P2HOT_LIMB_PHASE: separate a round's butterfly phase (plain 32-bit adds) from its conversion phase (multiply-adds)
// by workgroup barriers, so that every wave of a SIMD is in the same phase at the same time.  Why: profiles/r06_ubench_cheap.txt --
// the cheap integer class issues two instructions per 4-cycle slot only while EVERY resident wave offers one.
#ifndef P2HOT_LIMB_DUAL
#define P2HOT_LIMB_DUAL 0
#endif
#ifndef P2HOT_LIMB_PHASE
#define P2HOT_LIMB_PHASE 0
#endif
constexpr bool LIMB_PHASE = P2HOT_LIMB_PHASE != 0;
#ifndef P2HOT_LIMB_MIN_WAVES
#define P2HOT_LIMB_MIN_WAVES 4
#endif
#ifndef P2HOT_LIMB_DIRECT_CONTIG
#define P2HOT_LIMB_DIRECT_CONTIG 0
#endif
#ifndef P2HOT_LIMB_DEFER
#define P2HOT_LIMB_DEFER 1
#endif
#ifndef P2HOT_LIMB_MUL3
#define P2HOT_LIMB_MUL3 1
#endif
constexpr bool LIMB_DEFER = P2HOT_LIMB_DEFER != 0;  // a borrowed first-round table defers its small factor to a later round's table
#ifndef P2HOT_LIMB_MULCF
#define P2HOT_LIMB_MULCF 2  // the passes' general multiplies (coset scale, inter-pass twiddle): 2 = gl::mul3cg / mul1cg, the 14-instruction stream with 3
#endif                      // co-issued moves the Poseidon S-boxes run (same count as gl::mul3; LDE strided pass -2.8 % cycles under PMC, 4.31 - 4.34
                            // against 4.35 ms: profiles/r06_sbox_hybrid_ab.txt); 1 = gl::mul3cf (16 instructions: measured no faster here); 0 = gl::mul3
#ifndef P2HOT_LIMB_FOLD3
#define P2HOT_LIMB_FOLD3 0
#endif
constexpr bool LIMB_FOLD3 = P2HOT_LIMB_FOLD3 != 0;  // a unit's conversions fold their accumulator pairs three at a time (gl::fold3): measured
                                                    // SLOWER (LDE contiguous pass 4.69 vs 4.56 ms: 116 VGPRs instead of 101), and so did
                                                    // pairs (= 2: 4.83 vs 4.71 ms, 114 VGPRs); off
constexpr bool LIMB_MUL3 = P2HOT_LIMB_MUL3 != 0;    // independent general multiplies of a unit go through gl::mul3 streams
constexpr bool LIMB_DIRECT_CONTIG = P2HOT_LIMB_DIRECT_CONTIG != 0;  // contiguous pass: store the last round's outputs from registers
constexpr int LIMB_MIN_WAVES = P2HOT_LIMB_MIN_WAVES;  // waves per SIMD the register allocation must allow (4: <= 128 VGPRs)

// The tile's place in LDS.  ntt.hpp pads one word per 16 (pad_idx); under the per-instruction banking of gfx950 -- ds_read_b64: two
// groups of 32 lanes over 32 eight-byte banks, ds_write_b64: four groups of 16 lanes over 16 -- that padding itself costs a second
// LDS cycle whenever 32 consecutive words straddle a pad (the contiguous pass ran 17.7 % of its LDS-active cycles in conflicts,
// profiles/r04_x_pmc_sq.txt).  The limb passes use an XOR swizzle instead: word i lives at
//     swz(i) = i ^ (((i >> 2) ^ (i >> 3)) & 31)        (bit k ^= bit k+2 ^ bit k+3, k < 5: GF(2)-linear, unit upper triangular)
// which is conflict-free for every access pattern of every tile shape but one (2^10 x 2^2: 1.11 cycles per group), no padding
// words, and -- being linear -- lets a unit's 2^P addresses be ONE swizzled base XOR compile-time constants: the base and the
// q-offsets occupy disjoint bits, so swz(base + off_q) = swz(base) ^ swz(off_q).  (tools/lds_conflicts.py is the model.)
#ifndef P2HOT_LIMB_SWZ
#define P2HOT_LIMB_SWZ 1
#endif
constexpr bool LIMB_SWZ = P2HOT_LIMB_SWZ != 0;
__host__ __device__ constexpr unsigned swz(unsigned i) { return LIMB_SWZ ? i ^ (((i >> 2) ^ (i >> 3)) & 31u) : i + (i >> 4); }
// the tile word of base + off for DISJOINT bit sets base / off (off a compile-time constant at every call site)
__device__ __forceinline__ unsigned tix(unsigned sbase, unsigned base, unsigned off) {
    if constexpr (LIMB_SWZ) return sbase ^ swz(off);
    return swz(base + off);
}
constexpr unsigned TILE_WORDS = LIMB_SWZ ? (1u << TILE_LOG) : ntt::TILE_WORDS_PADDED;

struct L4 {
    u32 l[4];  // signed limbs, two's complement
};
struct alignas(16) W2 {
    u64 a, b;
};

// B^i mod P
constexpr u64 B1 = 1ull << 24, B2 = 1ull << 48, B3 = (1ull << 40) - (1ull << 8);  // 2^72 = 2^64 * 2^8 = (2^32 - 1) * 2^8

// Bias: o_i = 2^27 + d_i with sum o_i B^i = 0 (mod P).  2^27 (1 + B + B^2 + B^3) = 2^27 + 2^51 + (2^43 - 2^11) - 8 (mod P).
constexpr u64 BIAS_S = (1ull << 27) + (1ull << 51) + (1ull << 43) - (1ull << 11) - 8;
constexpr u64 BIAS_R = gl::P - BIAS_S;
constexpr u32 O0 = (1u << 27) + (u32)(BIAS_R & 0xFFFFFFu), O1 = (1u << 27) + (u32)((BIAS_R >> 24) & 0xFFFFFFu),
              O2 = (1u << 27) + (u32)(BIAS_R >> 48), O3 = 1u << 27;
static_assert((BIAS_R >> 48) < (1u << 16), "bias digits");

__host__ __device__ __forceinline__ L4 split(u64 x) {
    L4 r;
    r.l[0] = (u32)x & 0xFFFFFFu;
    r.l[1] = (u32)(x >> 24) & 0xFFFFFFu;
    r.l[2] = (u32)(x >> 48);
    r.l[3] = 0;
    return r;
}

__host__ __device__ __forceinline__ L4 add(const L4 &a, const L4 &b) {
    L4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.l[i] = a.l[i] + b.l[i];
    return r;
}

// (a - b) * B^R in Z[B]/(B^4 + 1), R in 0..7 (B^4 = -1): limb i lands in slot (i + R) mod 4, negated once per wrap
template <int R>
__host__ __device__ __forceinline__ L4 subrot(const L4 &a, const L4 &b) {
    L4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = (i + R) & 3;
        const bool neg = ((i + R) >> 2) & 1;
        r.l[j] = neg ? b.l[i] - a.l[i] : a.l[i] - b.l[i];
    }
    return r;
}

// 2^P-point DFT in place (DIF, bit-reversed output); twiddles are powers of w_8 = B (forward) or B^-1 (inverse):
// the same butterflies as ntt::dft_pow2.
template <int P, bool INV>
__host__ __device__ __forceinline__ void dft_limbs(L4 (&x)[1 << P]) {
    if constexpr (P == 3) {
        // layer 1: distance 4, rotation j = q
        {
            L4 a, b;
#define P2_BF(q, R)                      \
    a = x[q];                            \
    b = x[(q) + 4];                      \
    x[q] = add(a, b);                    \
    x[(q) + 4] = subrot<INV ? ((8 - (R)) & 7) : (R)>(a, b);
            P2_BF(0, 0) P2_BF(1, 1) P2_BF(2, 2) P2_BF(3, 3)
#undef P2_BF
        }
        // layer 2: distance 2, rotation 2j
        {
            L4 a, b;
#define P2_BF(q, R)                      \
    a = x[q];                            \
    b = x[(q) + 2];                      \
    x[q] = add(a, b);                    \
    x[(q) + 2] = subrot<INV ? ((8 - (R)) & 7) : (R)>(a, b);
            P2_BF(0, 0) P2_BF(1, 2) P2_BF(4, 0) P2_BF(5, 2)
#undef P2_BF
        }
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
            L4 a = x[q], b = x[q + 1];
            x[q] = add(a, b);
            x[q + 1] = subrot<0>(a, b);
        }
    } else if constexpr (P == 2) {
        {
            L4 a = x[0], b = x[2];
            x[0] = add(a, b);
            x[2] = subrot<0>(a, b);
            a = x[1];
            b = x[3];
            x[1] = add(a, b);
            x[3] = subrot<INV ? 6 : 2>(a, b);
        }
#pragma unroll
        for (int q = 0; q < 4; q += 2) {
            L4 a = x[q], b = x[q + 1];
            x[q] = add(a, b);
            x[q + 1] = subrot<0>(a, b);
        }
    } else {
        L4 a = x[0], b = x[1];
        x[0] = add(a, b);
        x[1] = subrot<0>(a, b);
    }
}

// sum_i L_i * W_i (mod P) for non-negative limbs L_i < 2^29 and W_i < 2^64: two 4-term multiply-add chains (the value is
// al + ah * 2^32, both below 2^63) + one fold
struct Acc {
    u64 al, ah;
};
__device__ __forceinline__ Acc convmul_acc(const L4 &v, u64 w0, u64 w1, u64 w2, u64 w3) {
    Acc r;
    r.al = (u64)v.l[0] * (u32)w0;
    r.ah = (u64)v.l[0] * (u32)(w0 >> 32);
    r.al += (u64)v.l[1] * (u32)w1;
    r.ah += (u64)v.l[1] * (u32)(w1 >> 32);
    r.al += (u64)v.l[2] * (u32)w2;
    r.ah += (u64)v.l[2] * (u32)(w2 >> 32);
    r.al += (u64)v.l[3] * (u32)w3;
    r.ah += (u64)v.l[3] * (u32)(w3 >> 32);
    return r;
}
__device__ __forceinline__ u64 convmul(const L4 &v, u64 w0, u64 w1, u64 w2, u64 w3) {
    const Acc r = convmul_acc(v, w0, w1, w2, w3);
    return gl::fold1(r.al, r.ah);
}

// the same with W_i = B^i (no twiddle: frequency 0, or the last round of a tile)
__device__ __forceinline__ Acc conv_unit_acc(const L4 &v) {
    Acc r;
    r.al = (u64)v.l[3] * (u32)B3 + v.l[0];
    r.al += (u64)v.l[1] << 24;
    r.ah = (u64)v.l[3] * (u32)(B3 >> 32) + ((u64)v.l[2] << 16);
    return r;
}
__device__ __forceinline__ u64 conv_unit(const L4 &v) {
    const Acc r = conv_unit_acc(v);
    return gl::fold1(r.al, r.ah);
}

// y[q] = fold(acc_of(q)), q < N, three rows per interleaved gl::fold3 stream (then two, then one): the single-stream fold
// stalls its wave two wait states per row, and a unit has 2^P independent rows
template <int N, class AccF>
__device__ __forceinline__ void fold_groups(AccF acc_of, u64 (&y)[N]) {
    if constexpr (P2HOT_LIMB_FOLD3 == 2) {  // pairs: two rows per gl::fold2 stream (two accumulator pairs alive instead of three)
#pragma unroll
        for (int q = 0; q + 2 <= N; q += 2) {
            const Acc a0 = acc_of(q), a1 = acc_of(q + 1);
            const u64 al[2] = {a0.al, a1.al}, ah[2] = {a0.ah, a1.ah};
            u64 r[2];
            gl::fold2(al, ah, r);
            y[q] = r[0], y[q + 1] = r[1];
        }
        if constexpr (N & 1) {
            const Acc a0 = acc_of(N - 1);
            y[N - 1] = gl::fold1(a0.al, a0.ah);
        }
        return;
    }
    constexpr int N3 = N / 3 * 3;
#pragma unroll
    for (int q = 0; q < N3; q += 3) {
        const Acc a0 = acc_of(q), a1 = acc_of(q + 1), a2 = acc_of(q + 2);
        const u64 al[3] = {a0.al, a1.al, a2.al}, ah[3] = {a0.ah, a1.ah, a2.ah};
        u64 r[3];
        gl::fold3(al, ah, r);
        y[q] = r[0], y[q + 1] = r[1], y[q + 2] = r[2];
    }
    if constexpr (N - N3 == 2) {
        const Acc a0 = acc_of(N3), a1 = acc_of(N3 + 1);
        const u64 al[2] = {a0.al, a1.al}, ah[2] = {a0.ah, a1.ah};
        u64 r[2];
        gl::fold2(al, ah, r);
        y[N3] = r[0], y[N3 + 1] = r[1];
    } else if constexpr (N - N3 == 1) {
        const Acc a0 = acc_of(N3);
        y[N3] = gl::fold1(a0.al, a0.ah);
    }
}

struct LimbPassArgs {
    ntt::PassArgs a;
    const W2 *tw_all;    // the round tables of this tile shape, concatenated in their LDS layout (limb_tables_w2 entries)
    const u64 *ufac;     // tiles whose first round borrows the second round's table: ufac[a * 8 + k] = w_{2^LOG_R}^(a * k)
    const u64 *srow2;    // strided LDE first pass: srow2[z * 4096 + e] = s_z^((e >> LOG_C) * stride + (e & (C-1))), tile-shaped
    const W2 *sbase;     // ... and sbase[(z * tiles + tile) * 2 + h] = 4-form of s_z^(tile * C): the last conversion's constant
    u64 wlast[4];        // WLAST kernels: the forms the round without table twiddles multiplies by, c * B^i (c = 1/n for the iNTT)
    const u64 *twid;     // inter-pass twiddles as in ntt::RegPassArgs
    unsigned xcd_remap;  // > 0 (= log2 of gridDim.x): workgroup b takes slot (b % 8) * gridDim.x / 8 + b / 8 (see ntt::RegPassArgs)
    unsigned zloop;      // > 0: this workgroup produces z = 0..zloop-1 itself
    unsigned tiles_log;  // a workgroup processes 2^tiles_log consecutive tiles (the staged tables are loaded once)
    // BRIN kernels (the first, strided pass of a two-pass coset LDE that follows an inverse transform): the coefficients are read from
    // the inverse transform's BIT-REVERSED output and written once more in natural order -- the stand-alone bit reversal between
    // the two transforms (fri/oracle.rs:65-69 then :91-98) disappears into this pass
    const u64 *brin_src;     // [polynomials][brin_stride]: natural coefficient t of a polynomial lives at reverse_bits(t, LOG_R + 12)
    size_t brin_stride;
    u64 *nat_out;            // [polynomials][nat_stride]: canonical natural-order coefficients (`polynomials`, oracle.rs:32), or null
    size_t nat_stride;
};

// radix bits of round r of a LOG_R-bit tile: as even as possible, larger parts first (the host builds the tables the same way)
constexpr int n_rounds(int log_r) { return (log_r + 2) / 3; }
constexpr int round_bits(int log_r, int r) {
    int rem = log_r, nr = n_rounds(log_r), part = 0;
    for (int q = 0; q <= r; ++q) {
        part = (rem + (nr - q) - 1) / (nr - q);
        rem -= part;
    }
    return part;
}
constexpr int round_log_rb(int log_r, int r) {
    int rb = log_r;
    for (int q = 0; q < r; ++q) rb -= round_bits(log_r, q);
    return rb;
}
// Round tables live in LDS: the 4-forms of w_{Rb}^(lo * k) for lo < S = Rb / 2^p, k = 1..2^p - 1, as
// t[((k-1)*2 + h) << s_log | lo] = {W_{2h}, W_{2h+1}}.  A round with more than 64 twiddle columns (the first round of the
// contiguous pass: 512) has no table of its own: with lo = a + 2^p b its twiddle is w_{Rb}^(a k) * w_{Rb/2^p}^(b k), i.e. the
// NEXT round's table entry (same radix) times one of 2^p (2^p - 1) small factors u[a * 8 + k] = w_{Rb}^(a * k), also in LDS:
// 3584 distinct twiddles from 448 + 56 table entries, and 51 KB of LDS per workgroup (three workgroups per CU).
constexpr int TW_S_MAX_LOG = 6;
constexpr bool round_borrows(int log_r, int r) { return round_log_rb(log_r, r) - round_bits(log_r, r) > TW_S_MAX_LOG; }
// The small factor of a borrowed round-0 twiddle, w_{R}^(a * k0) (a = the low p0 bits of the unit's lo, k0 = the output's
// frequency), is the same for all inputs of every later unit as long as that unit's inputs differ neither in their low p0
// position bits (= a) nor in their top p0 position bits (= the block k0 went to): it commutes with those rounds and is
// ABSORBED by the table of the round whose own twiddle index is exactly a (S_LOG == p0) -- that table becomes
// w_{Rb}^(a * k) * w_{R}^(a * k0), one slab per k0 (wave-uniform: a wave owns one first-round block), frequency 0 included.
// For the 4096-point tile (rounds 3+3+3+3): round 2, 8 * 8 * 8 entries = 16 KiB, and round 0 loses its second multiply.
constexpr int absorb_round(int log_r) {  // the round that absorbs, or -1: then round 0 keeps its second multiply
    if (!LIMB_DEFER || !round_borrows(log_r, 0)) return -1;
    for (int r = 1; r < n_rounds(log_r); ++r)
        if (round_log_rb(log_r, r) - round_bits(log_r, r) == round_bits(log_r, 0)) return r;
    return -1;
}
constexpr bool round_absorbs(int log_r, int r) { return absorb_round(log_r) == r; }
constexpr bool defers(int log_r) { return absorb_round(log_r) >= 0; }
constexpr int round_table_w2(int log_r, int r) {  // W2 entries of round r's own table
    const int s = round_log_rb(log_r, r) - round_bits(log_r, r);
    if (round_absorbs(log_r, r)) return ((1 << round_bits(log_r, 0)) * (1 << round_bits(log_r, r)) * 2) << s;
    return (s == 0 || round_borrows(log_r, r)) ? 0 : (((1 << round_bits(log_r, r)) - 1) * 2) << s;
}
constexpr int round_table_off(int log_r, int r) {
    int o = 0;
    for (int q = 0; q < r; ++q) o += round_table_w2(log_r, q);
    return o;
}
constexpr int limb_tables_w2(int log_r) { return round_table_off(log_r, n_rounds(log_r)); }
constexpr int UFAC_WORDS = 64;  // u[a * 8 + k], a < 8, k < 8 (only round 0 ever borrows)
constexpr bool uses_ufac(int log_r) { return round_borrows(log_r, 0) && !defers(log_r); }
constexpr size_t limb_shmem_bytes(int log_r, int dual = 1) {
    return (size_t)8 * TILE_WORDS * dual + (size_t)16 * limb_tables_w2(log_r) + (uses_ufac(log_r) ? 8 * UFAC_WORDS : 0);
}

__device__ __forceinline__ u64 limb_mul(u64 a, u64 b) { return P2HOT_LIMB_MULCF == 2 ? gl::mul1cg(a, b) : gl::mul1(a, b); }
// v[q] *= w[q] for N independent pairs: three-stream gl::mul3 blocks (no wait states, three chains in flight) where the
// build asks for them, single streams otherwise
template <int N>
__device__ __forceinline__ void limb_mul_n(u64 (&v)[N], const u64 (&w)[N]) {
    if constexpr (LIMB_MUL3) {
        int q = 0;
#pragma unroll
        for (; q + 3 <= N; q += 3) {
            const u64 a3[3] = {v[q], v[q + 1], v[q + 2]}, b3[3] = {w[q], w[q + 1], w[q + 2]};
            u64 r3[3];
            if constexpr (P2HOT_LIMB_MULCF == 2)
                gl::mul3cg(a3, b3, r3);  // the 14-instruction form with 3 co-issued moves (what the Poseidon S-boxes run)
            else if constexpr (P2HOT_LIMB_MULCF != 0)
                gl::mul3cf(a3, b3, r3);  // carry-free streams (gl_mul3.hpp, round 6): moves ride the multiply-adds, carry adds do not
            else
                gl::mul3(a3, b3, r3);
            v[q] = r3[0], v[q + 1] = r3[1], v[q + 2] = r3[2];
        }
        if (N - q == 2) {
            const u64 a2[2] = {v[q], v[q + 1]}, b2[2] = {w[q], w[q + 1]};
            u64 r2[2];
            gl::mul2(a2, b2, r2);
            v[q] = r2[0], v[q + 1] = r2[1];
        } else if (N - q == 1) {
            v[q] = limb_mul(v[q], w[q]);
        }
    } else {
#pragma unroll
        for (int q = 0; q < N; ++q) v[q] = limb_mul(v[q], w[q]);
    }
}
// A wave-uniform constant the optimiser must not see through.  The bias limb O3 is the ONLY contribution to some limbs of a
// unit's outputs (a split word has l3 = 0), and hipcc turns "known 2^27 times a table word" into two 64-bit shifts, four
// masks and two 64-bit adds where the general path is two multiply-adds.
__device__ __forceinline__ u32 opaque_u32(u32 c) {
#ifndef P2HOT_EMU
    asm("" : "+s"(c));
#endif
    return c;
}
__device__ __forceinline__ void opaque_branch() {
#ifndef P2HOT_EMU
    asm volatile("");
#endif
}
__device__ __forceinline__ unsigned wave_uniform(unsigned v) {
#ifdef P2HOT_EMU
    return v;
#else
    return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
#endif
}

// P2HOT_LIMB_PHASE: a workgroup barrier the instruction scheduler does not move arithmetic across
__device__ __forceinline__ void phase_sync() {
#ifdef P2HOT_EMU
    __syncthreads();
#else
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
#endif
}

// the workgroup barrier, or -- after a round whose sub-blocks a wave owns entirely -- nothing but program order
// (the LDS executes one wave's accesses in order; the emulator's lanes are fibers, so there it stays a barrier)
template <bool WAVE_PRIVATE>
__device__ __forceinline__ void round_sync() {
#ifdef P2HOT_EMU
    __syncthreads();
#else
    if constexpr (WAVE_PRIVATE && !LIMB_PHASE) {
        // the wave barrier alone is declared IntrNoMem: the wavefront-scope fences are what formally order this lane's LDS
        // stores before the other lanes' LDS loads (no instruction is emitted for them; the LDS runs a wave's accesses in order)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else if constexpr (LIMB_PHASE) {
        phase_sync();
    } else {
        __syncthreads();
    }
#endif
}

// how the tile's last round (no table twiddles) converts: by 1, by ra.wlast (a constant: the 1/n of the inverse transform),
// or by the tile's entry of ra.sbase (the column part of the coset scale, see below)
enum { LAST_UNIT = 0, LAST_CONST = 1, LAST_TILE = 2 };

// the eight inputs of a thread's first-round units: element q of unit uu is raw[uu * 2^P + q]
template <int LOG_R, int LOG_C>
__device__ __forceinline__ void load_inputs(const u64 *gin, unsigned log_stride, u64 (&raw)[8]) {
    constexpr int P = round_bits(LOG_R, 0);
    constexpr int S_LOG = LOG_R - P;
    constexpr unsigned C = 1u << LOG_C;
    constexpr int UPT = 8 >> P;
    constexpr unsigned UW = 512u >> P;
    const unsigned wave = (threadIdx.x >> 6) & 7u, lane = threadIdx.x & 63;  // (the tile's eight waves; a DUAL workgroup has two tiles)
#pragma unroll
    for (int uu = 0; uu < UPT; ++uu) {
        const unsigned u = wave * UW + lane + 64u * (unsigned)uu;
        const unsigned c = u & (C - 1), lo = u >> LOG_C;  // first round: hi = 0
        // one 32-bit per-thread byte offset, wave-uniform row steps (a block of a strided limb pass is <= 2^24 elements)
        const u32 off0 = ((lo << log_stride) + c) * 8u;
#pragma unroll
        for (int q = 0; q < (1 << P); ++q)
            raw[(uu << P) + q] =
                *reinterpret_cast<const u64 *>(reinterpret_cast<const char *>(gin + ((size_t)q << (S_LOG + log_stride))) + off0);
    }
}

// BRIN: the tile's 4096 inputs out of the bit-reversed coefficient array, through LDS.  Natural coefficient t = row * 2^12 + col
// (row < 2^LOG_R, col = tau * 2^LOG_C + c) sits at reverse_bits(t, LOG_R + 12) = ((rev(c) << (12 - LOG_C) | rev(tau)) << LOG_R) | rev(row):
// per column c one run of 2^LOG_R consecutive words.  The 512 threads read the 2^LOG_C runs as 16-byte pieces (a wave: one
// contiguous KiB per load), park every word at its NATURAL tile position, and after one barrier each thread takes its first-round
// units' inputs from there (the mapping of load_inputs) -- and stores them, canonical, to the natural-order array (128-byte runs).
template <int LOG_R, int LOG_C>
__device__ __forceinline__ void load_inputs_bitrev(const u64 *src, unsigned tau, u64 *tile, u64 (&raw)[8], u64 *nat, unsigned tid) {
    static_assert(LOG_R >= 1 && LOG_C >= 1, "a strided pass");
    constexpr int P = round_bits(LOG_R, 0);
    constexpr int S_LOG = LOG_R - P;
    constexpr unsigned C = 1u << LOG_C;
    constexpr int UPT = 8 >> P;
    constexpr unsigned UW = 512u >> P;
    const unsigned taurev = LOG_C < 12 ? (unsigned)(__brev(tau) >> (32 - (12 - LOG_C))) : 0u;
    W2 st[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned m = 2u * ((unsigned)j * 512u + tid);  // word index inside the tile's bit-reversed order: run (m >> LOG_R), place (m & (R - 1))
        const unsigned cp = m >> LOG_R, rp = m & ((1u << LOG_R) - 1);
        st[j] = *reinterpret_cast<const W2 *>(src + ((((size_t)cp << (12 - LOG_C)) | taurev) << LOG_R) + rp);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned m = 2u * ((unsigned)j * 512u + tid);
        const unsigned cp = m >> LOG_R, rp = m & ((1u << LOG_R) - 1);
        const unsigned c = (unsigned)(__brev(cp) >> (32 - LOG_C));
        const unsigned row0 = (unsigned)(__brev(rp) >> (32 - LOG_R)), row1 = row0 | (1u << (LOG_R - 1));  // rp is even: rp + 1 flips the top row bit
        tile[swz((row0 << LOG_C) | c)] = st[j].a;
        tile[swz((row1 << LOG_C) | c)] = st[j].b;
    }
    __syncthreads();
    const unsigned wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int uu = 0; uu < UPT; ++uu) {
        const unsigned u = wave * UW + lane + 64u * (unsigned)uu;
        const unsigned c = u & (C - 1), lo = u >> LOG_C;
        const unsigned eb = (lo << LOG_C) + c, seb = swz(eb);
#pragma unroll
        for (int q = 0; q < (1 << P); ++q) {
            const u64 v = tile[tix(seb, eb, (unsigned)q << (S_LOG + LOG_C))];
            raw[(uu << P) + q] = v;
            if (nat) nat[((size_t)(lo + ((unsigned)q << S_LOG)) << 12) + ((size_t)tau << LOG_C) + c] = gl::canon(v);
        }
    }
    // (no second barrier: the first round overwrites exactly the eight slots this thread just read)
}

// Round RI of the tile.  FIRST round: the inputs are `raw` (already loaded); once they are split, the inputs of the next tile
// are fetched from `next_gin` (if not null) so that their latency hides behind the rest of this tile.
template <bool INV, int LOG_R, int LOG_C, int SCALE, int LAST, int RI>
__device__ __forceinline__ void limb_round(const LimbPassArgs &ra, u64 *tile, const W2 *ltw, const u64 *lu, u64 (&raw)[8],
                                           const u64 *next_gin, u64 *gout, unsigned log_stride, size_t z, size_t base0) {
    constexpr int P = round_bits(LOG_R, RI);
    constexpr int LOG_RB = round_log_rb(LOG_R, RI);
    constexpr int S_LOG = LOG_RB - P;
    constexpr bool BORROW = round_borrows(LOG_R, RI);
    constexpr bool ABSORB = round_absorbs(LOG_R, RI);
    static_assert(!BORROW || (RI == 0 && round_bits(LOG_R, 1) == P && S_LOG - P <= TW_S_MAX_LOG), "borrowed table shape");
    constexpr int T_LOG = BORROW ? S_LOG - P : S_LOG;  // log2 of the table's columns
    constexpr int T_OFF = round_table_off(LOG_R, BORROW ? RI + 1 : RI);
    constexpr bool FIRST = RI == 0;
    constexpr unsigned C = 1u << LOG_C;
    constexpr int UPT = (1 << (TILE_LOG - P)) / NT;  // units per thread
    constexpr unsigned UW = 512u >> P;               // units per wave: a wave's units cover 512 consecutive tile elements
    [[maybe_unused]] constexpr int UNROLL = FIRST ? UPT : 1;  // `raw` is indexed by uu: registers only when unrolled
    const ntt::PassArgs &a = ra.a;
    const unsigned wave = (threadIdx.x >> 6) & 7u, lane = threadIdx.x & 63;
#pragma unroll UNROLL
    for (int uu = 0; uu < UPT; ++uu) {
        const unsigned u = wave * UW + lane + 64u * (unsigned)uu;
        const unsigned c = u & (C - 1), rest = u >> LOG_C;
        const unsigned lo = rest & ((1u << S_LOG) - 1), hi = rest >> S_LOG;
        const unsigned i0 = (hi << LOG_RB) + lo;
        const unsigned eb = (i0 << LOG_C) + c, seb = swz(eb);  // the unit's element q is tile word tix(seb, eb, q << (S_LOG + LOG_C))
        u64 v[1 << P];
        if constexpr (FIRST) {
#pragma unroll
            for (int q = 0; q < (1 << P); ++q) v[q] = raw[(uu << P) + q];
        } else {
#pragma unroll
            for (int q = 0; q < (1 << P); ++q) v[q] = tile[tix(seb, eb, (unsigned)q << (S_LOG + LOG_C))];
        }
        if constexpr (FIRST && SCALE == ntt::SCALE_CONST) {
#pragma unroll
            for (int q = 0; q < (1 << P); ++q) v[q] = limb_mul(v[q], a.scale_const);
        } else if constexpr (FIRST && SCALE == ntt::SCALE_TABLE && LOG_C > 0) {
            // coset scale s_z^t, t = i * stride + base0 + c: the (i, c) part from a tile-shaped table, the base0 part (uniform)
            // in the last round's conversion (LAST_TILE)
            const u64 *sr = ra.srow2 + (z << TILE_LOG) + (i0 << LOG_C) + c;
            u64 sw[1 << P];
#pragma unroll
            for (int q = 0; q < (1 << P); ++q) sw[q] = sr[(size_t)q << (S_LOG + LOG_C)];
            limb_mul_n<(1 << P)>(v, sw);
        } else if constexpr (FIRST && SCALE == ntt::SCALE_TABLE) {  // single-pass transform: the row table is everything
#pragma unroll
            for (int q = 0; q < (1 << P); ++q) v[q] = limb_mul(v[q], a.srow[z * (1u << LOG_R) + i0 + ((unsigned)q << S_LOG)]);
        }
        L4 x[1 << P];
#pragma unroll
        for (int q = 0; q < (1 << P); ++q) x[q] = split(v[q]);
        if constexpr (FIRST)
            if (uu == UPT - 1 && next_gin) load_inputs<LOG_R, LOG_C>(next_gin, log_stride, raw);
        x[0].l[0] += O0;
        x[0].l[1] += O1;
        x[0].l[2] += O2;
        x[0].l[3] += opaque_u32(O3);
        dft_limbs<P, INV>(x);
        if constexpr (LIMB_PHASE) phase_sync();  // butterflies (plain adds) | conversions (multiply-adds): every wave switches together
        if constexpr (S_LOG > 0) {
            if constexpr (LIMB_FOLD3 && !(BORROW && !defers(LOG_R))) {
                // all 2^P accumulator pairs of the unit, folded three rows per stream
                constexpr int P0 = round_bits(LOG_R, 0);
                [[maybe_unused]] const unsigned k0 = ABSORB ? (unsigned)(__brev(hi >> (LOG_R - P0 - LOG_RB)) >> (32 - P0)) : 0u;
                const W2 *tw = ABSORB ? ltw + T_OFF + ((k0 << (P + 1)) << T_LOG) + lo : ltw + T_OFF + (BORROW ? lo >> P : lo);
                u64 y[1 << P];
                fold_groups<(1 << P)>(
                    [&](int q) {
                        if (!ABSORB && q == 0) return conv_unit_acc(x[0]);
                        const unsigned k = q ? (unsigned)(__brev((unsigned)q) >> (32 - P)) : 0u;
                        const unsigned e = ABSORB ? k * 2 : (k - 1) * 2;
                        const W2 wa = tw[e << T_LOG], wb = tw[(e + 1) << T_LOG];
                        return convmul_acc(x[q], wa.a, wa.b, wb.a, wb.b);
                    },
                    y);
#pragma unroll
                for (int q = 0; q < (1 << P); ++q) tile[tix(seb, eb, (unsigned)q << (S_LOG + LOG_C))] = y[q];
            } else if constexpr (ABSORB) {
                // this round's twiddle times the factor round 0 deferred: slab k0 = the frequency whose first-round block
                // this unit lies in (position bits LOG_R-P0 .. LOG_R-1, bit-reversed), every output converts through the table
                constexpr int P0 = round_bits(LOG_R, 0);
                const unsigned k0 = (unsigned)(__brev(hi >> (LOG_R - P0 - LOG_RB)) >> (32 - P0));
                const W2 *tw = ltw + T_OFF + ((k0 << (P + 1)) << T_LOG) + lo;
#pragma unroll
                for (int q = 0; q < (1 << P); ++q) {
                    const unsigned k = q ? (unsigned)(__brev((unsigned)q) >> (32 - P)) : 0u;
                    const W2 wa = tw[(k * 2) << T_LOG], wb = tw[(k * 2 + 1) << T_LOG];
                    tile[tix(seb, eb, (unsigned)q << (S_LOG + LOG_C))] = convmul(x[q], wa.a, wa.b, wb.a, wb.b);
                }
            } else {
                const W2 *tw = ltw + T_OFF + (BORROW ? lo >> P : lo);
                [[maybe_unused]] const u64 *uf = lu + (lo & ((1u << P) - 1)) * 8;
                tile[seb] = conv_unit(x[0]);
                [[maybe_unused]] u64 yb[(1 << P) - 1], ub[(1 << P) - 1];
#pragma unroll
                for (int q = 1; q < (1 << P); ++q) {
                    const unsigned k = (unsigned)(__brev((unsigned)q) >> (32 - P));
                    const W2 wa = tw[((k - 1) * 2) << T_LOG], wb = tw[((k - 1) * 2 + 1) << T_LOG];
                    u64 y = convmul(x[q], wa.a, wa.b, wb.a, wb.b);
                    if constexpr (BORROW && !defers(LOG_R) && !LIMB_MUL3) y = limb_mul(y, uf[k]);  // (deferred: absorbed by a later round's table)
                    if constexpr (BORROW && !defers(LOG_R) && LIMB_MUL3) {
                        yb[q - 1] = y, ub[q - 1] = uf[k];
                    } else {
                        tile[tix(seb, eb, (unsigned)q << (S_LOG + LOG_C))] = y;
                    }
                }
                if constexpr (BORROW && !defers(LOG_R) && LIMB_MUL3) {  // the second factors of the unit's outputs, three streams at a time
                    limb_mul_n<(1 << P) - 1>(yb, ub);
#pragma unroll
                    for (int q = 1; q < (1 << P); ++q) tile[tix(seb, eb, (unsigned)q << (S_LOG + LOG_C))] = yb[q - 1];
                }
            }
        } else {
            // the tile's last round.  A strided pass finishes here: inter-pass twiddle w_{n'}^(col * k1) from the per-pass table (laid
            // out like a block; the host always supplies it) and the store, straight from registers -- a unit's rows are
            // consecutive, so 2^LOG_C lanes still write 2^LOG_C * 8 contiguous bytes per row.  The contiguous pass goes through
            // LDS once more for coalescing.
            u64 w0 = 0, w1 = 0, w2 = 0, w3 = 0;
            if constexpr (LAST == LAST_CONST) {
                w0 = ra.wlast[0], w1 = ra.wlast[1], w2 = ra.wlast[2], w3 = ra.wlast[3];
            } else if constexpr (LAST == LAST_TILE) {  // wave-uniform: scalar loads
                const W2 *sb = ra.sbase + (((z << (log_stride - LOG_C)) + (base0 >> LOG_C)) << 1);
                w0 = sb[0].a, w1 = sb[0].b, w2 = sb[1].a, w3 = sb[1].b;
            }
            auto conv_last = [&](const L4 &v_) { return LAST == LAST_UNIT ? conv_unit(v_) : convmul(v_, w0, w1, w2, w3); };
            auto conv_last_all = [&](u64 (&y_)[1 << P]) {  // y_[q] = conv_last(x[q])
                if constexpr (LIMB_FOLD3) {
                    fold_groups<(1 << P)>(
                        [&](int q) { return LAST == LAST_UNIT ? conv_unit_acc(x[q]) : convmul_acc(x[q], w0, w1, w2, w3); }, y_);
                } else {
#pragma unroll
                    for (int q = 0; q < (1 << P); ++q) y_[q] = conv_last(x[q]);
                }
            };
            if constexpr (LOG_C > 0 && P <= 2) {  // (a radix-8 last round keeps too much alive: it goes through LDS like the contiguous pass)
                const u32 off0 = ((i0 << log_stride) + c) * 8u;  // bytes; the row steps q << log_stride are wave-uniform
                const char *tw = reinterpret_cast<const char *>(ra.twid + base0);
                char *go = reinterpret_cast<char *>(gout);
                if constexpr (LIMB_MUL3) {
                    u64 y[1 << P], t[1 << P];
#pragma unroll
                    for (int q = 0; q < (1 << P); ++q) t[q] = *reinterpret_cast<const u64 *>(tw + ((size_t)q << log_stride) * 8 + off0);
                    conv_last_all(y);
                    limb_mul_n<(1 << P)>(y, t);
#pragma unroll
                    for (int q = 0; q < (1 << P); ++q)
                        *reinterpret_cast<u64 *>(go + ((size_t)q << log_stride) * 8 + off0) = y[q];  // (a strided pass is never the last: no canon)
                } else {
#pragma unroll
                    for (int q = 0; q < (1 << P); ++q) {
                        const size_t step = ((size_t)q << log_stride) * 8;
                        const u64 v_ = limb_mul(conv_last(x[q]), *reinterpret_cast<const u64 *>(tw + step + off0));
                        *reinterpret_cast<u64 *>(go + step + off0) = v_;
#ifndef P2HOT_EMU
                        if ((q & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // four outputs at a time: the twiddle loads are not hoisted past this
#endif
                    }
                }
            } else if constexpr (P == 3 && LIMB_DIRECT_CONTIG) {
                // the contiguous pass: a unit's eight outputs are 64 consecutive bytes of the block -- four 16-byte stores
                // (measured slower than the LDS round trip: 5.01 vs 4.87 ms for the LDE's pass; off by default)
                W2 *go = reinterpret_cast<W2 *>(gout + i0);
#pragma unroll
                for (int q = 0; q < 8; q += 2) {
                    W2 pr;
                    pr.a = conv_last(x[q]);
                    pr.b = conv_last(x[q + 1]);
                    if (a.canon_out) pr.a = gl::canon(pr.a), pr.b = gl::canon(pr.b);
                    go[q >> 1] = pr;
                }
            } else {
                u64 y[1 << P];
                conv_last_all(y);
#pragma unroll
                for (int q = 0; q < (1 << P); ++q) tile[tix(seb, eb, (unsigned)q << LOG_C)] = y[q];
            }
        }
    }
    // the next round (and the store phase) reads what this one wrote: a wave's own 512 elements once the sub-blocks are that small
    if constexpr (RI + 1 < n_rounds(LOG_R)) {
        round_sync<(LOG_RB + LOG_C <= 9)>();
        limb_round<INV, LOG_R, LOG_C, SCALE, LAST, RI + 1>(ra, tile, ltw, lu, raw, next_gin, gout, log_stride, z, base0);
    } else if constexpr (LOG_C == 0 ? !(P == 3 && LIMB_DIRECT_CONTIG) : P == 3) {
        round_sync<(LOG_RB + LOG_C <= 9)>();  // the store phase reads what this round wrote
    }
}

// One pass over 2^LOG_R x 2^LOG_C tiles (LOG_R + LOG_C = 12); LOG_C = 0 is the contiguous (last) pass.
// grid = (tiles per polynomial >> tiles_log >> (DUAL - 1), polynomials, z), 512 * DUAL threads, limb_shmem_bytes(LOG_R, DUAL) of
// dynamic LDS.  DUAL = 2: threads 512..1023 transform the NEXT tile group side by side with threads 0..511 (their own tile in LDS,
// the round tables shared): one workgroup then holds all four waves of every SIMD, so a workgroup barrier is a SIMD-wide one.
// Workgroup barriers: one after the first round and one per tile; the later rounds and the store phase are wave-private.
template <bool INV, int LOG_R, int LOG_C, int SCALE, int LAST = LAST_UNIT, int DUAL = 1, bool BRIN = false>
__global__ void __launch_bounds__(NT * DUAL, LIMB_MIN_WAVES) ntt_limbpass_kernel(LimbPassArgs ra) {
    static_assert(LOG_R + LOG_C == TILE_LOG, "a tile is 4096 elements");
    static_assert(DUAL == 1 || DUAL == 2, "one or two tiles side by side");
    static_assert(!BRIN || (LOG_C > 0 && DUAL == 1 && !INV), "the bit-reversed source is the first pass of a two-pass coset LDE");
    P2HOT_DYN_SHARED(u64, tile_all);
    const ntt::PassArgs &a = ra.a;
    const unsigned tid = threadIdx.x & (NT - 1);                          // the thread's place among its tile's 512
    const unsigned half = DUAL == 2 ? wave_uniform(threadIdx.x >> 9) : 0u;  // which of the workgroup's tiles
    u64 *tile = tile_all + (size_t)half * TILE_WORDS;
    W2 *ltw = reinterpret_cast<W2 *>(tile_all + (size_t)DUAL * TILE_WORDS);
    u64 *lu = reinterpret_cast<u64 *>(ltw + limb_tables_w2(LOG_R));
    for (unsigned e = threadIdx.x; e < (unsigned)limb_tables_w2(LOG_R); e += NT * DUAL) ltw[e] = ra.tw_all[e];
    if constexpr (uses_ufac(LOG_R))
        if (threadIdx.x < (unsigned)UFAC_WORDS) lu[threadIdx.x] = ra.ufac[threadIdx.x];
    __syncthreads();
    const unsigned log_stride = LOG_C ? a.log_nblk - LOG_R : 0u;  // the contiguous pass is the last one: blocks of one tile
    const unsigned tiles_per_blk_log = log_stride - LOG_C;
    // (xcd_remap counts WORKGROUPS: the host passes log2 of gridDim.x; a DUAL workgroup takes two consecutive tile groups)
    const size_t wg0 = (LOG_C && ra.xcd_remap) ? (((size_t)(blockIdx.x & 7u) << (ra.xcd_remap - 3)) | (blockIdx.x >> 3)) : blockIdx.x;
    const size_t wg = wg0 * DUAL + half;
    const size_t z_begin = ra.zloop ? 0 : blockIdx.z, z_end = ra.zloop ? ra.zloop : blockIdx.z + 1;
    const bool same_input = a.in_z_stride == 0;  // every z slice transforms the same polynomials (coset LDE): fetch them once
    const unsigned e0 = (tid >> 6) * 512u + (tid & 63u);  // store phase: wave w moves tile elements [512 w, 512 w + 512)
    const unsigned se0 = swz(e0);
    const size_t n_tiles = (size_t)1 << ra.tiles_log;
    auto tile_in = [&](size_t t, size_t z) {
        const size_t tau = (wg << ra.tiles_log) + t;
        const size_t blk = tau >> tiles_per_blk_log;
        const size_t base0 = (tau & (((size_t)1 << tiles_per_blk_log) - 1)) << LOG_C;
        return a.in + (size_t)blockIdx.y * a.in_poly_stride + z * a.in_z_stride + (blk << a.log_nblk) + base0;
    };
    u64 raw[8];
    if constexpr (BRIN) {  // one tile per workgroup (host: tiles_log = 0, one block); the z = 0 workgroup also writes the natural copy
        u64 *nat = (ra.nat_out && z_begin == 0) ? ra.nat_out + (size_t)blockIdx.y * ra.nat_stride : nullptr;
        load_inputs_bitrev<LOG_R, LOG_C>(ra.brin_src + (size_t)blockIdx.y * ra.brin_stride, (unsigned)wg, tile, raw, nat, tid);
    } else {
        load_inputs<LOG_R, LOG_C>(tile_in(0, z_begin), log_stride, raw);
    }
#pragma unroll 1
    for (size_t t = 0; t < n_tiles; ++t) {
        const size_t tau = (wg << ra.tiles_log) + t;
        const size_t blk = tau >> tiles_per_blk_log;
        const size_t base0 = (tau & (((size_t)1 << tiles_per_blk_log) - 1)) << LOG_C;
#pragma unroll 1
        for (size_t z = z_begin; z < z_end; ++z) {
            u64 *out = a.out + (size_t)blockIdx.y * a.out_poly_stride + z * a.out_z_stride + (blk << a.log_nblk) + base0;
            // what to fetch while this tile is transformed: the same tile's next z slice (unless it is the same data), else the
            // next tile's first slice
            const u64 *next = nullptr;
            if (z + 1 < z_end) {
                if (!same_input) next = tile_in(t, z + 1);
            } else if (t + 1 < n_tiles) {
                next = tile_in(t + 1, z_begin);
            }
            limb_round<INV, LOG_R, LOG_C, SCALE, LAST, 0>(ra, tile, ltw, lu, raw, next, out, log_stride, z, base0);
            constexpr int LAST_P = round_bits(LOG_R, n_rounds(LOG_R) - 1);
            if constexpr (LOG_C == 0 && !(LAST_P == 3 && LIMB_DIRECT_CONTIG)) {
                // canonical representatives or not is the launch's choice: ONE wave-uniform branch around the eight stores
                // (as `canon_out ? canon(v) : v` it was a 64-bit compare, an add and four selects per word; the empty asm keeps
                // the optimiser from merging the two arms back into selects)
                if (a.canon_out) {
                    opaque_branch();
#pragma unroll
                    for (unsigned j = 0; j < 8; ++j) out[e0 + 64 * j] = gl::canon(tile[tix(se0, e0, 64 * j)]);
                } else {
#pragma unroll
                    for (unsigned j = 0; j < 8; ++j) out[e0 + 64 * j] = tile[tix(se0, e0, 64 * j)];
                }
            } else if constexpr (LOG_C > 0 && LAST_P == 3) {
                // a strided pass whose last round is radix 8 (2^9 and 2^6 rows): inter-pass twiddle + store from LDS.
                // element e = U + lane with U = 512 w + 64 j wave-uniform: row (U >> LOG_C) + (lane >> LOG_C), column (U + lane) mod C
                constexpr unsigned C = 1u << LOG_C;
                const unsigned lane = tid & 63u;
                const u32 off0 = (((lane >> LOG_C) << log_stride) + (lane & (C - 1))) * 8u;  // bytes, per lane
                const unsigned U0 = wave_uniform(tid >> 6) * 512u;
                if constexpr (LIMB_MUL3) {  // the thread's eight products as three-stream blocks
                    u64 v8[8], w8[8];
#pragma unroll
                    for (unsigned j = 0; j < 8; ++j) {
                        const unsigned U = U0 + 64 * j;
                        const size_t step = ((size_t)(U >> LOG_C) << log_stride) + (U & (C - 1));
                        w8[j] = *reinterpret_cast<const u64 *>(reinterpret_cast<const char *>(ra.twid + base0 + step) + off0);
                        v8[j] = tile[tix(se0, e0, 64 * j)];
                    }
                    limb_mul_n<8>(v8, w8);
#pragma unroll
                    for (unsigned j = 0; j < 8; ++j) {
                        const unsigned U = U0 + 64 * j;
                        const size_t step = ((size_t)(U >> LOG_C) << log_stride) + (U & (C - 1));
                        *reinterpret_cast<u64 *>(reinterpret_cast<char *>(out + step) + off0) = v8[j];
                    }
                } else {
#pragma unroll 2
                    for (unsigned j = 0; j < 8; ++j) {
                        const unsigned U = U0 + 64 * j;
                        const size_t step = ((size_t)(U >> LOG_C) << log_stride) + (U & (C - 1));
                        const u64 w = *reinterpret_cast<const u64 *>(reinterpret_cast<const char *>(ra.twid + base0 + step) + off0);
                        u64 v = limb_mul(tile[tix(se0, e0, 64 * j)], w);
                        *reinterpret_cast<u64 *>(reinterpret_cast<char *>(out + step) + off0) = v;
                    }
                }
            }
            __syncthreads();  // the next tile's first round overwrites what other waves may still be reading
        }
    }
}

// Fills one round's table in its LDS layout: t[((k-1)*2 + h) << s_log | lo] = 4-form of w_{2^log_rb}^(lo*k), s_log = log_rb - p
__global__ void limb_twiddle_kernel(W2 *t, unsigned log_rb, unsigned p, ntt::RootTable roots) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned s_log = log_rb - p;
    const size_t S = (size_t)1 << s_log;
    if (idx >= (((size_t)1 << p) - 1) * S) return;
    const u32 lo = (u32)(idx & (S - 1)), k = (u32)(idx >> s_log) + 1;
    const u64 w = ntt::root_pow(roots, (u32)(((u64)lo * k) << (32 - log_rb)));
    W2 *o = t + ((size_t)(k - 1) * 2 << s_log) + lo;
    o[0] = W2{gl::canon(w), gl::canon(gl::mul(w, B1))};
    o[S] = W2{gl::canon(gl::mul(w, B2)), gl::canon(gl::mul(w, B3))};
}
// The table of a round that absorbs the first round's deferred factor (round_absorbs): slab k0, frequency k, column a:
// t[(((k0 << p) + k) * 2 + h) << s_log | a] = 4-form of w_{2^log_rb}^(a * k) * w_{2^log_r}^(a * k0), s_log = log_rb - p
__global__ void limb_twiddle_absorb_kernel(W2 *t, unsigned log_r, unsigned log_rb, unsigned p, ntt::RootTable roots) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned s_log = log_rb - p;  // = the first round's radix bits
    const size_t S = (size_t)1 << s_log;
    if (idx >= (S << p) * S) return;
    const u32 a = (u32)(idx & (S - 1)), k = (u32)(idx >> s_log) & ((1u << p) - 1), k0 = (u32)(idx >> (s_log + p));
    const u64 w = ntt::root_pow(roots, (u32)(((u64)a * k) << (32 - log_rb)) + (u32)(((u64)a * k0) << (32 - log_r)));
    W2 *o = t + ((size_t)(((k0 << p) + k) * 2) << s_log) + a;
    o[0] = W2{gl::canon(w), gl::canon(gl::mul(w, B1))};
    o[S] = W2{gl::canon(gl::mul(w, B2)), gl::canon(gl::mul(w, B3))};
}
// u[a * 8 + k] = w_{2^log_rb}^(a * k), a < 8, k < 8
__global__ void limb_ufac_kernel(u64 *u, unsigned log_rb, ntt::RootTable roots) {
    const unsigned idx = threadIdx.x;
    if (idx >= (unsigned)UFAC_WORDS) return;
    u[idx] = gl::canon(ntt::root_pow(roots, (u32)((u64)((idx >> 3) * (idx & 7)) << (32 - log_rb))));
}
// the coset scale of a strided first pass, tile-shaped: srow2[e] = s^((e >> log_c) << log_stride | (e & (C-1))), e < 4096, and
// sbase[tile * 2 + h] = 4-form of s^(tile << log_c), tile < 2^(log_stride - log_c)   (one launch per coset)
__global__ void limb_scale_kernel(u64 *srow2, W2 *sbase, unsigned log_c, unsigned log_stride, u64 s) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < ((size_t)1 << TILE_LOG))
        srow2[idx] = gl::canon(gl::pow(s, ((idx >> log_c) << log_stride) + (idx & (((size_t)1 << log_c) - 1))));
    if (idx < ((size_t)1 << (log_stride - log_c))) {
        const u64 w = gl::pow(s, idx << log_c);
        sbase[idx * 2] = W2{gl::canon(w), gl::canon(gl::mul(w, B1))};
        sbase[idx * 2 + 1] = W2{gl::canon(gl::mul(w, B2)), gl::canon(gl::mul(w, B3))};
    }
}

}  // namespace nttl
