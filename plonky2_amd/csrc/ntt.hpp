// ntt.hpp -- batched radix-2 NTT passes over Goldilocks for gfx950, LDS-staged.
//
// Replaces field/src/fft.rs (fft_classic :165-202, ifft_with_options :68-91, root tables :14-33),
// PolynomialCoeffs::lde + coset_fft_with_options (field/src/polynomial/mod.rs:199-201, :280-293),
// PolynomialBatch::lde_values (plonky2/src/fri/oracle.rs:114-139) and the two bit-reversal /
// transpose steps of from_coeffs (oracle.rs:97-98; util/src/lib.rs:185-234; plonky2/src/util/mod.rs:25-31).
//
// Design (not the reference's per-column in-cache DIT):
//  * A size-n transform is a chain of "passes".  A pass takes the current blocks of size n' and
//    performs LOGR decimation-in-frequency layers on the R = 2^LOGR points {blk*n' + i*stride + base}
//    (stride = n'/R) inside LDS, then multiplies by the inter-pass twiddle w_{n'}^(base * k1)
//    (four-step factorisation).  A workgroup stages an R x C tile (C consecutive `base` values, so
//    every global access is a run of C*8 >= 128 bytes); the last pass (stride 1) is a contiguous tile.
//    Natural-order input gives bit-reversed output with no separate permutation.
//  * The rate-1/B coset LDE is B independent size-n transforms of coeff[t] * (g * w_N^j)^t (one per
//    coset j of H_n in H_N).  In the committed, bit-reversed leaf order coset j is exactly the
//    contiguous row block bitrev_rb(j), already in the order the DIF passes emit: the reference's
//    zero-padding, the first rb butterfly layers, the transpose and reverse_index_bits all vanish
//    into addressing.  Cosets are also the multi-GPU sharding unit.
//  * Twiddles: one two-level table of w_{2^32}^e (2 x 65536 entries, L2 resident) serves every
//    size: w_{2^k}^e = HI[E >> 16] * LO[E & 0xffff] with E = e << (32 - k); in-tile twiddles have
//    E & 0xffff == 0 and cost one load.
#pragma once
#include "gl.hpp"
#include "gl_mul3.hpp"

namespace ntt {
using gl::u32;
using gl::u64;

constexpr unsigned TILE_LOG = 12;  // elements per workgroup tile (32 KiB of LDS)
constexpr unsigned THREADS = 256;

struct RootTable {  // device pointers, 65536 entries each
    const u64 *lo;  // w^e,        e < 2^16
    const u64 *hi;  // w^(e<<16),  e < 2^16
};

__device__ __forceinline__ u64 root_pow(const RootTable &t, u32 E) {
    u64 h = t.hi[E >> 16];
    u32 l = E & 0xFFFFu;
    return l ? gl::mul(h, t.lo[l]) : h;
}

// out[k] = base^(k * step + first), used for the root tables and the coset scale tables
__global__ void pow_table_kernel(u64 *out, size_t count, u64 base, u64 step, u64 first) {
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    out[k] = gl::canon(gl::pow(base, k * step + first));
}

enum { SCALE_NONE = 0, SCALE_CONST = 1, SCALE_TABLE = 2 };

struct PassArgs {
    const u64 *in;
    u64 *out;
    size_t in_poly_stride, out_poly_stride;  // elements between polynomials
    size_t in_z_stride, out_z_stride;        // elements between grid.z slices (coset blocks)
    unsigned log_n;                          // polynomial size
    unsigned log_nblk;                       // block size n' at this pass
    unsigned log_r;                          // layers done by this pass
    unsigned log_c;                          // tile columns (strided pass); 0 for the contiguous pass
    RootTable roots;
    int scale_mode;
    u64 scale_const;
    const u64 *srow, *scol;  // SCALE_TABLE: [z][R] and [z][stride]; first pass only (n' == n)
    int canon_out;           // store canonical representatives (last pass of an LDE)
};

// One pass on an R x C tile.  grid = (tiles per polynomial, polynomials, z).
__global__ void __launch_bounds__(THREADS) ntt_pass_kernel(PassArgs a) {
    P2HOT_DYN_SHARED(u64, tile);
    const unsigned tid = threadIdx.x;
    const unsigned R = 1u << a.log_r, logC = a.log_c, C = 1u << logC;
    const unsigned log_stride = a.log_nblk - a.log_r;
    const size_t stride = (size_t)1 << log_stride;
    const unsigned tiles_per_blk_log = log_stride - logC;
    const size_t tau = blockIdx.x;
    const size_t blk = tau >> tiles_per_blk_log;
    const size_t base0 = (tau & (((size_t)1 << tiles_per_blk_log) - 1)) << logC;
    const size_t z = blockIdx.z;
    const u64 *in = a.in + (size_t)blockIdx.y * a.in_poly_stride + z * a.in_z_stride + (blk << a.log_nblk) + base0;
    u64 *out = a.out + (size_t)blockIdx.y * a.out_poly_stride + z * a.out_z_stride + (blk << a.log_nblk) + base0;
    const unsigned elems = R << logC;

    // load (+ optional scaling): element e -> row i = e >> logC, column c = e & (C-1)
    for (unsigned e = tid; e < elems; e += THREADS) {
        unsigned i = e >> logC, c = e & (C - 1);
        u64 v = in[((size_t)i << log_stride) + c];
        if (a.scale_mode == SCALE_CONST) {
            v = gl::mul(v, a.scale_const);
        } else if (a.scale_mode == SCALE_TABLE) {
            u64 s = a.srow[z * R + i];
            if (log_stride) s = gl::mul(s, a.scol[z * stride + base0 + c]);
            v = gl::mul(v, s);
        }
        tile[e] = v;
    }
    __syncthreads();

    // LOGR decimation-in-frequency layers along i; twiddle w_R^(j << s) = HI[j << (16 - log_r + s)]
    for (unsigned s = 0; s < a.log_r; ++s) {
        const unsigned log_half = a.log_r - s - 1, half = 1u << log_half;
        for (unsigned e = tid; e < (elems >> 1); e += THREADS) {
            unsigned p = e >> logC, c = e & (C - 1);
            unsigned j = p & (half - 1);
            unsigned i0 = ((p >> log_half) << (log_half + 1)) + j;
            unsigned x0 = (i0 << logC) + c, x1 = x0 + (half << logC);
            u64 u = tile[x0], v = tile[x1];
            u64 w = a.roots.hi[j << (16 - a.log_r + s)];
            tile[x0] = gl::add(u, v);
            tile[x1] = gl::mul(gl::sub(u, v), w);
        }
        __syncthreads();
    }

    // inter-pass twiddle w_{n'}^(base * k1), k1 = bitrev_logr(i), then store
    for (unsigned e = tid; e < elems; e += THREADS) {
        unsigned i = e >> logC, c = e & (C - 1);
        u64 v = tile[e];
        if (log_stride) {
            u32 k1 = a.log_r ? (__brev(i) >> (32 - a.log_r)) : 0;
            u64 ex = (u64)(base0 + c) * k1;  // < n'
            v = gl::mul(v, root_pow(a.roots, (u32)(ex << (32 - a.log_nblk))));
        }
        out[((size_t)i << log_stride) + c] = a.canon_out ? gl::canon(v) : v;
    }
}

// ---------------------------------------------------------------------------------------------
// Register-radix pass (the production path).  Same tile, same pass algebra as ntt_pass_kernel, but
// the LOGR layers run as rounds of radix 2^p (p <= 4) butterflies held in registers:
//   * a round at sub-block size Rb takes the 2^p points {hi*Rb + q*(Rb>>p) + lo} of a unit, does a
//     2^p-point DFT whose twiddles are powers of w_16 = 2^12 (so w_8 = 2^24, w_4 = 2^48: shifts and one
//     reduction in Goldilocks, no multiplier), then one table twiddle w_Rb^(lo*k) per output
//     (four-step inside the tile);
//   * the tile touches LDS once per round (not once per layer) and the first round loads straight
//     from global memory with the pre-scaling fused;
//   * LDS indices are padded by one word per 16 so the stride-16 / stride-256 rounds stay conflict-free.
// The inverse transform uses w^-j = -2^(96 - e): the butterfly subtracts the other way round.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 mul_pow2_le32(u64 x, int s) {  // x * 2^s (mod P), 0 < s <= 32
    u64 lo = x << (s & 63), hi = x >> ((64 - s) & 63);  // hi < 2^32
    u64 t = (u64)(u32)hi * 0xFFFFFFFFu + lo;
    return gl::fold_carry(t, t < lo);
}
__device__ __forceinline__ u64 mul_pow2_c(u64 x, int s) {  // x * 2^s (mod P), 0 <= s < 96 (compiler-scheduled form)
    if (s == 0) return x;
    if (s <= 32) return mul_pow2_le32(x, s);
    if (s < 64) return gl::reduce128(x << (s & 63), x >> ((64 - s) & 63));
    if (s == 64) return gl::reduce128(0, x);
    return gl::reduce128(0, mul_pow2_le32(x, s - 64));  // 2^64 = 2^32 - 1: the 128-bit value {0, y}
}

// x * 2^S (mod P) for a compile-time S in (0, 96), hand-written: the shifted value is split at the word boundaries and
// folded with 2^64 = 2^32 - 1 and 2^96 = -1:
//   S <= 32:      (x << S) + (x >> (64-S)) * (2^32-1)                              5 instructions
//   32 < S < 64:  {0, x.lo << (S-32)} - (x >> (96-S)) + ((x >> (64-S)) mod 2^32) * (2^32-1)   10
//   64 < S < 96:  ((x.lo << (S-64)) mod 2^32) * (2^32-1) - (x >> (96-S))           8   (x*2^S = l0*2^64 + l1*2^96 + top*2^128)
// with the carry / borrow folds done as in gl_mul3.hpp (e = carry - borrow, one signed multiply-add).
template <int S>
__device__ __forceinline__ u64 mul_pow2_asm(u64 x) {
    if (!P2_ASM_INTERPRETED()) return mul_pow2_c(x, S);  // emulator build with the interpreter off (constant-false in the product)
    static_assert(S > 0 && S < 96 && S != 32 && S != 64, "shift handled elsewhere");
    const u32 x0 = (u32)x, x1 = (u32)(x >> 32);
    u64 r, t, c1;
    u32 m;
    if (S < 32) {
        u32 h;
        P2_ASM_NC("v_lshlrev_b64 %[t], %[s], %[x]\n\t"
                  "v_lshrrev_b32 %[h], %[rs], %[x1]\n\t"
                  "v_mad_u64_u32 %[t], %[c1], %[h], -1, %[t]\n\t"
                  "s_nop 1\n\t"
                  "v_cndmask_b32 %[m], 0, -1, %[c1]\n\t"
                  "v_mad_u64_u32 %[r], %[c1], %[m], 1, %[t]",
                  (P2_O([r], "=&v", r), P2_O([t], "=&v", t), P2_O([h], "=&v", h), P2_O([m], "=&v", m), P2_O([c1], "=&s", c1)),
                  (P2_I([x], "v", x), P2_I([x1], "v", x1), P2_I([s], "n", S), P2_I([rs], "n", 32 - S)));
    } else if (S < 64) {
        u64 c2;
        u32 a, b, cc, t0, t1;
        P2_ASM_NC("v_lshlrev_b32 %[a], %[s], %[x0]\n\t"             // hi word of the low 64 bits (the low word is 0)
                  "v_alignbit_b32 %[b], %[x1], %[x0], %[rs]\n\t"    // bits 64..95 of x << S
                  "v_lshrrev_b32 %[cc], %[rs], %[x1]\n\t"           // bits 96.. of x << S
                  "v_sub_co_u32 %[t0], %[c1], 0, %[cc]\n\t"
                  "s_nop 1\n\t"
                  "v_subb_co_u32 %[t1], %[c1], %[a], 0, %[c1]\n\t",  // {t0, t1} = lo64 - hi.hi, borrow c1
                  (P2_O([a], "=&v", a), P2_O([b], "=&v", b), P2_O([cc], "=&v", cc), P2_O([t0], "=&v", t0),
                   P2_O([t1], "=&v", t1), P2_O([c1], "=&s", c1)),
                  (P2_I([x0], "v", x0), P2_I([x1], "v", x1), P2_I([s], "n", S - 32), P2_I([rs], "n", 64 - S)));
        t = ((u64)t1 << 32) | t0;
        P2_ASM_NC("v_mad_u64_u32 %[t], %[c2], %[b], -1, %[t]\n\t"   // + hi.lo * (2^32-1), carry c2
                  "s_nop 1\n\t"
                  "v_cndmask_b32 %[m], 0, 1, %[c2]\n\t"
                  "v_subb_co_u32 %[m], %[c1], %[m], 0, %[c1]\n\t",   // e = carry - borrow
                  (P2_O([t], "+v", t), P2_O([m], "=&v", m), P2_O([c2], "=&s", c2), P2_O([c1], "+s", c1)), (P2_I([b], "v", b)));
        t += (u64)m << 32;                                          // u + (e << 32) - e
        P2_ASM_NC("v_mad_i64_i32 %[r], %[c2], %[m], -1, %[t]", (P2_O([r], "=&v", r), P2_O([c2], "=&s", c2)),
                  (P2_I([m], "v", m), P2_I([t], "v", t)));
    } else {
        u64 y;
        u32 l0;
        P2_ASM_NC("v_lshlrev_b32 %[l0], %[s], %[x0]\n\t"
                  "v_lshrrev_b64 %[y], %[rs], %[x]\n\t"             // x >> (96 - S)
                  "v_mad_u64_u32 %[t], %[c1], %[l0], -1, 0",            // l0 * (2^32-1)
                  (P2_O([l0], "=&v", l0), P2_O([y], "=&v", y), P2_O([t], "=&v", t), P2_O([c1], "=&s", c1)),
                  (P2_I([x0], "v", x0), P2_I([x], "v", x), P2_I([s], "n", S - 64), P2_I([rs], "n", 96 - S)));
        u32 d0, d1;
        P2_ASM_NC("v_sub_co_u32 %[d0], %[c1], %[t0], %[y0]\n\t"
                  "s_nop 1\n\t"
                  "v_subb_co_u32 %[d1], %[c1], %[t1], %[y1], %[c1]\n\t"
                  "s_nop 1\n\t"
                  "v_cndmask_b32 %[m], 0, -1, %[c1]\n\t"            // e = -borrow
                  "v_add_u32 %[d1], %[d1], %[m]",
                  (P2_O([d0], "=&v", d0), P2_O([d1], "=&v", d1), P2_O([m], "=&v", m), P2_O([c1], "=&s", c1)),
                  (P2_I([t0], "v", (u32)t), P2_I([t1], "v", (u32)(t >> 32)), P2_I([y0], "v", (u32)y),
                   P2_I([y1], "v", (u32)(y >> 32))));
        t = ((u64)d1 << 32) | d0;
        P2_ASM_NC("v_mad_i64_i32 %[r], %[c1], %[m], -1, %[t]", (P2_O([r], "=&v", r), P2_O([c1], "=&s", c1)),
                  (P2_I([m], "v", m), P2_I([t], "v", t)));
    }
    return r;
}

__device__ __forceinline__ u64 mul_pow2(u64 x, int s) {  // x * 2^s (mod P), 0 <= s < 96, s constant after unrolling
    switch (s) {  // the shifts the radix-8 / radix-16 butterflies use
        case 12: return mul_pow2_asm<12>(x);
        case 24: return mul_pow2_asm<24>(x);
        case 36: return mul_pow2_asm<36>(x);
        case 48: return mul_pow2_asm<48>(x);
        case 60: return mul_pow2_asm<60>(x);
        case 72: return mul_pow2_asm<72>(x);
        case 84: return mul_pow2_asm<84>(x);
        default: return mul_pow2_c(x, s);
    }
}

__device__ __forceinline__ unsigned pad_idx(unsigned i) { return i + (i >> 4); }
constexpr unsigned TILE_WORDS_PADDED = (1u << TILE_LOG) + (1u << (TILE_LOG - 4));

struct RegPassArgs {
    PassArgs a;
    const u64 *local;        // local[2^m + e] = w_{2^m}^e (forward or inverse), m <= TILE_LOG
    unsigned rounds[4];      // radix bits of each round, sum = log_r, zero-terminated
    int inverse;
    const u64 *twid;         // inter-pass twiddles of this (strided) pass as a table laid out like one block of the data,
                             // twid[(i << log_stride) + col] = w_{n'}^(col * bitrev(i)); null = build them from the root tables
    unsigned xcd_remap;      // > 0 (= log2 of gridDim.x): workgroup b works on tile (b % 8) * gridDim.x / 8 + b / 8, so the
                             // workgroups one XCD runs together are neighbouring column groups (they share 128-byte lines)
    unsigned zloop;          // > 0: this workgroup produces z = 0..zloop-1 itself (coset LDE first pass: the
                             // coefficient tile is fetched from HBM once and re-read from L2 for the other cosets)
};

// 2^P-point DFT of x[] in place (DIF, bit-reversed output) with power-of-two twiddles
template <int P, bool INV>
__device__ __forceinline__ void dft_pow2(u64 (&x)[1 << P]) {
#pragma unroll
    for (int u = 0; u < P; ++u) {
        const int d = 1 << (P - 1 - u);     // butterfly distance
        const int step = 96 >> (P - 1 - u); // exponent step: w_{2d}^j = 2^(96/d * j)
#pragma unroll
        for (int q = 0; q < (1 << P); ++q) {
            if (q & d) continue;
            const int j = q & (d - 1);
            u64 a = x[q], b = x[q + d];
            x[q] = gl::add(a, b);
            if (j == 0) {
                x[q + d] = gl::sub(a, b);
            } else if (!INV) {
                x[q + d] = mul_pow2(gl::sub(a, b), step * j);
            } else {
                x[q + d] = mul_pow2(gl::sub(b, a), 96 - step * j);
            }
        }
    }
}

template <int NT>
__device__ __forceinline__ u64 ntt_mul(u64 a, u64 b) {  // hand-scheduled multiply on the register set that fits the variant
    return NT == 512 ? gl::mul1_lowregs(a, b) : gl::mul1(a, b);
}

// SCALE: the pass's scale mode when known at compile time (the per-point mode tests disappear), -1 = read a.scale_mode
template <int P, bool INV, int NT, int SCALE, bool from_global, bool CONTIG>
__device__ __forceinline__ void reg_round(const RegPassArgs &ra, u64 *tile, const u64 *gin, unsigned log_rb,
                                          unsigned log_stride, size_t z, size_t base0) {
    const PassArgs &a = ra.a;
    const unsigned logC = CONTIG ? 0u : a.log_c, C = 1u << logC;  // CONTIG: the last pass, one column, unit stride
    const unsigned elems_log = a.log_r + logC;
    const unsigned s_log = log_rb - P;  // S = Rb >> P
    const unsigned R = 1u << a.log_r;
    const size_t stride = (size_t)1 << log_stride;
    for (unsigned u = threadIdx.x; u < (1u << (elems_log - P)); u += blockDim.x) {
        const unsigned c = u & (C - 1), rest = u >> logC;
        const unsigned lo = rest & ((1u << s_log) - 1), hi = rest >> s_log;
        const unsigned i0 = (hi << log_rb) + lo;
        u64 x[1 << P];
#pragma unroll
        for (int q = 0; q < (1 << P); ++q) {
            const unsigned i = i0 + ((unsigned)q << s_log);
            if (from_global) {
                u64 v = gin[((size_t)i << log_stride) + c];
                const int mode = SCALE >= 0 ? SCALE : a.scale_mode;
                if (mode == SCALE_CONST) {
                    v = ntt_mul<NT>(v, a.scale_const);
                } else if (mode == SCALE_TABLE) {
                    u64 sc = a.srow[z * R + i];
                    if (log_stride) sc = ntt_mul<NT>(sc, a.scol[z * stride + base0 + c]);
                    v = ntt_mul<NT>(v, sc);
                }
                x[q] = v;
            } else {
                x[q] = tile[pad_idx((i << logC) + c)];
            }
        }
        dft_pow2<P, INV>(x);
        if (s_log) {  // four-step twiddle inside the tile: position q holds frequency k = bitrev_P(q)
            const u64 *tw = ra.local + ((size_t)1 << log_rb);
#pragma unroll
            for (int q = 1; q < (1 << P); ++q) {
                const unsigned k = __brev((unsigned)q) >> (32 - P);
                x[q] = ntt_mul<NT>(x[q], tw[lo * k]);
            }
        }
#pragma unroll
        for (int q = 0; q < (1 << P); ++q) tile[pad_idx(((i0 + ((unsigned)q << s_log)) << logC) + c)] = x[q];
    }
    __syncthreads();
}

// CONTIG: the contiguous (last) pass of a transform -- log_c = 0 and unit stride known at compile time
template <bool INV, int NT, int MINW, int SCALE = -1, bool CONTIG = false>
__global__ void __launch_bounds__(NT, MINW) ntt_regpass_kernel(RegPassArgs ra) {
    P2HOT_DYN_SHARED(u64, tile);
    const PassArgs &a = ra.a;
    const unsigned tid = threadIdx.x;
    const unsigned logC = CONTIG ? 0u : a.log_c, C = 1u << logC;
    const unsigned log_stride = CONTIG ? 0u : a.log_nblk - a.log_r;
    const unsigned tiles_per_blk_log = log_stride - logC;
    const size_t tau = (!CONTIG && ra.xcd_remap) ? (((size_t)(blockIdx.x & 7u) << (ra.xcd_remap - 3)) | (blockIdx.x >> 3)) : blockIdx.x;
    const size_t blk = tau >> tiles_per_blk_log;
    const size_t base0 = (tau & (((size_t)1 << tiles_per_blk_log) - 1)) << logC;
    const unsigned elems = 1u << (a.log_r + logC);
    const size_t z_begin = ra.zloop ? 0 : blockIdx.z, z_end = ra.zloop ? ra.zloop : blockIdx.z + 1;
#pragma unroll 1
    for (size_t z = z_begin; z < z_end; ++z) {
        const u64 *in = a.in + (size_t)blockIdx.y * a.in_poly_stride + z * a.in_z_stride + (blk << a.log_nblk) + base0;
        u64 *out = a.out + (size_t)blockIdx.y * a.out_poly_stride + z * a.out_z_stride + (blk << a.log_nblk) + base0;

        if (ra.rounds[0] == 0) {  // log_r == 0: a pure scale / copy pass
            for (unsigned e = tid; e < elems; e += NT) {
                u64 v = in[e];
                if (a.scale_mode == SCALE_CONST) v = gl::mul(v, a.scale_const);
                if (a.scale_mode == SCALE_TABLE) v = gl::mul(v, a.srow[z]);
                out[e] = a.canon_out ? gl::canon(v) : v;
            }
            continue;
        }
        unsigned log_rb = a.log_r;
        bool first = true;
#pragma unroll 1
        for (int r = 0; r < 4 && ra.rounds[r]; ++r) {
#define P2_ROUND(PP, FG) reg_round<PP, INV, NT, SCALE, FG, CONTIG>(ra, tile, in, log_rb, log_stride, z, base0)
            if (first) {  // the first round loads from global memory (with the pass's scaling), the others from the tile
                switch (ra.rounds[r]) {
                    case 4: if (NT == 256) P2_ROUND(4, true); break;
                    case 3: P2_ROUND(3, true); break;
                    case 2: P2_ROUND(2, true); break;
                    default: P2_ROUND(1, true); break;
                }
            } else {
                switch (ra.rounds[r]) {
                    case 4: if (NT == 256) P2_ROUND(4, false); break;
                    case 3: P2_ROUND(3, false); break;
                    case 2: P2_ROUND(2, false); break;
                    default: P2_ROUND(1, false); break;
                }
            }
#undef P2_ROUND
            log_rb -= ra.rounds[r];
            first = false;
        }
        // inter-pass twiddle w_{n'}^(base * k1), k1 = bitrev_logr(i), then coalesced store
        if (!CONTIG && log_stride && ra.twid) {  // from the per-pass table: one coalesced load instead of two gathers + a multiply
            const u64 *tw = ra.twid + base0;
            for (unsigned e = tid; e < elems; e += NT) {
                unsigned i = e >> logC, c = e & (C - 1);
                const size_t off = ((size_t)i << log_stride) + c;
                u64 v = ntt_mul<NT>(tile[pad_idx(e)], tw[off]);
                out[off] = a.canon_out ? gl::canon(v) : v;
            }
        } else {
            for (unsigned e = tid; e < elems; e += NT) {
                unsigned i = e >> logC, c = e & (C - 1);
                u64 v = tile[pad_idx(e)];
                if (log_stride) {
                    u32 k1 = __brev(i) >> (32 - a.log_r);
                    u64 ex = (u64)(base0 + c) * k1;
                    const u32 E = (u32)(ex << (32 - a.log_nblk));
                    u64 w = a.roots.hi[E >> 16];
                    if (E & 0xFFFFu) w = ntt_mul<NT>(w, a.roots.lo[E & 0xFFFFu]);
                    v = ntt_mul<NT>(v, w);
                }
                out[((size_t)i << log_stride) + c] = a.canon_out ? gl::canon(v) : v;
            }
        }
        if (z + 1 < z_end) __syncthreads();  // the tile is reused by the next coset
    }
}

// the inter-pass twiddles of a strided pass over blocks of 2^log_nblk elements with 2^log_r rows per tile column
__global__ void interpass_twiddle_kernel(u64 *t, unsigned log_nblk, unsigned log_r, RootTable roots) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >> log_nblk) return;
    const unsigned log_stride = log_nblk - log_r;
    const u32 i = (u32)(idx >> log_stride), col = (u32)(idx & (((size_t)1 << log_stride) - 1));
    const u32 k1 = log_r ? __brev(i) >> (32 - log_r) : 0;
    t[idx] = gl::canon(root_pow(roots, (u32)(((u64)col * k1) << (32 - log_nblk))));
}

// out[bitrev_log_n(i)] = canon(in[i])   (util/src/lib.rs:53-62 semantics)
__global__ void bitrev_permute_kernel(const u64 *in, u64 *out, size_t in_poly_stride, size_t out_poly_stride,
                                      unsigned log_n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >> log_n) return;
    size_t j = log_n ? (size_t)(__brevll((unsigned long long)i) >> (64 - log_n)) : 0;
    out[(size_t)blockIdx.y * out_poly_stride + j] = gl::canon(in[(size_t)blockIdx.y * in_poly_stride + i]);
}

// the same permutation, LDS-tiled for large n: i = (hi:5 | mid | lo:5) -> bitrev(i) = (rev lo | rev mid | rev hi).
// A workgroup moves the 32 x 32 tile of one `mid`: 256-byte runs on both the read and the write side.
__global__ void __launch_bounds__(256) bitrev_tiled_kernel(const u64 *in, u64 *out, size_t in_poly_stride,
                                                          size_t out_poly_stride, unsigned log_n) {
    __shared__ u64 t[32][33];
    const unsigned mid_bits = log_n - 10;
    const size_t mid = blockIdx.x;
    const size_t rmid = mid_bits ? (size_t)(__brevll((unsigned long long)mid) >> (64 - mid_bits)) : 0;
    const u64 *src = in + (size_t)blockIdx.y * in_poly_stride;
    u64 *dst = out + (size_t)blockIdx.y * out_poly_stride;
    for (unsigned e = threadIdx.x; e < 1024; e += 256) {
        unsigned hi = e >> 5, lo = e & 31;
        t[hi][lo] = src[((size_t)hi << (log_n - 5)) | (mid << 5) | lo];
    }
    __syncthreads();
    for (unsigned e = threadIdx.x; e < 1024; e += 256) {
        unsigned a = e >> 5, b = e & 31;  // output row a = rev5(lo), column b = rev5(hi)
        unsigned lo = __brev(a) >> 27, hi = __brev(b) >> 27;
        dst[((size_t)a << (log_n - 5)) | (rmid << 5) | b] = gl::canon(t[hi][lo]);
    }
}

// data[p][i] *= lo[i & mask] * hi[i >> lo_bits]  (two-level power table): the coefficient rescale of
// PolynomialValues::coset_ifft (field/src/polynomial/mod.rs:63-73), canonical output
__global__ void scale_by_powers_kernel(u64 *data, size_t poly_stride, unsigned log_n, const u64 *lo, const u64 *hi,
                                       unsigned lo_bits) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >> log_n) return;
    u64 s = gl::mul(lo[i & (((size_t)1 << lo_bits) - 1)], hi[i >> lo_bits]);
    u64 *p = data + (size_t)blockIdx.y * poly_stride + i;
    *p = gl::canon(gl::mul(*p, s));
}

// Sub-coset sharding (more GPUs than LDE cosets).  On the coset s*H_{n'} of size n' = n / m every point has x^n' = c =
// s^n', so p(x) = (p mod (x^n' - c))(x): out[col][t] = sum_u in[col][t + u*n'] * c^u, Horner from the top block.  The size-n' coset
// NTT of `out` with shift s is then that block of the LDE (field/src/polynomial/mod.rs:280-293 evaluated on a sub-coset).
__global__ void fold_mod_kernel(const u64 *in, size_t in_stride, u64 *out, size_t out_stride, unsigned log_np, unsigned m, u64 c) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, np = (size_t)1 << log_np;
    if (t >= np) return;
    const u64 *p = in + (size_t)blockIdx.y * in_stride;
    u64 acc = p[t + (size_t)(m - 1) * np];
    for (unsigned u = m - 1; u-- > 0;) acc = gl::add(gl::mul(acc, c), p[t + (size_t)u * np]);
    out[(size_t)blockIdx.y * out_stride + t] = acc;
}

__global__ void canon_kernel(u64 *data, size_t count) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) data[i] = gl::canon(data[i]);
}

// column-major [W][rows] (stride between columns) -> row-major [rows][W]
// (plonky2/src/util/mod.rs:25-31 transpose).  64-row x 32-column LDS tile.
// rev_bits > 0: the destination row of source row r is reverse_bits(r, rev_bits) (a row is still written as 256-byte pieces)
__global__ void __launch_bounds__(256) transpose_kernel(const u64 *in, size_t stride, unsigned W, size_t rows, u64 *out, unsigned rev_bits) {
    __shared__ u64 t[32][65];
    const size_t r0 = (size_t)blockIdx.x * 64;
    const unsigned c0 = blockIdx.y * 32;
    const unsigned tid = threadIdx.x;
    for (unsigned e = tid; e < 32 * 64; e += 256) {
        unsigned c = e >> 6, r = e & 63;
        if (c0 + c < W && r0 + r < rows) t[c][r] = in[(size_t)(c0 + c) * stride + r0 + r];
    }
    __syncthreads();
    for (unsigned e = tid; e < 32 * 64; e += 256) {
        unsigned r = e >> 5, c = e & 31;
        if (c0 + c < W && r0 + r < rows) {
            const size_t src = r0 + r;
            const size_t dst = rev_bits ? (size_t)(__brevll((unsigned long long)src) >> (64 - rev_bits)) : src;
            out[dst * W + c0 + c] = gl::canon(t[c][r]);
        }
    }
}

// out[m][W] = in[c][idx[m]]  (lazy leaf fetch for query openings, oracle.rs:142-147 / merkle_tree.rs:227)
// An index >= rows (the reference would panic on the slice index) reads nothing: the row is zeroed and *oob is raised;
// the context reports it at the next synchronisation point.
__global__ void gather_rows_kernel(const u64 *in, size_t stride, size_t rows, unsigned W, const u64 *idx, size_t m, u64 *out,
                                   unsigned *oob) {
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m * W) return;
    size_t q = e / W;
    unsigned c = (unsigned)(e % W);
    const u64 r = idx[q];
    if (r >= rows) {
        out[e] = 0;
        if (c == 0) atomicOr(oob, 1u);
        return;
    }
    out[e] = gl::canon(in[(size_t)c * stride + r]);
}

}  // namespace ntt
