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
constexpr int NT = 512;
#ifndef P2HOT_LIMB_MIN_WAVES
#define P2HOT_LIMB_MIN_WAVES 4
#endif
constexpr int LIMB_MIN_WAVES = P2HOT_LIMB_MIN_WAVES;  // waves per SIMD the register allocation must allow (4: <= 128 VGPRs)

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

// sum_i L_i * W_i (mod P) for non-negative limbs L_i < 2^29 and W_i < 2^64: two 4-term multiply-add chains + one fold
__device__ __forceinline__ u64 convmul(const L4 &v, u64 w0, u64 w1, u64 w2, u64 w3) {
    u64 al = (u64)v.l[0] * (u32)w0;
    u64 ah = (u64)v.l[0] * (u32)(w0 >> 32);
    al += (u64)v.l[1] * (u32)w1;
    ah += (u64)v.l[1] * (u32)(w1 >> 32);
    al += (u64)v.l[2] * (u32)w2;
    ah += (u64)v.l[2] * (u32)(w2 >> 32);
    al += (u64)v.l[3] * (u32)w3;
    ah += (u64)v.l[3] * (u32)(w3 >> 32);
    return gl::fold1(al, ah);
}

// the same with W_i = B^i (no twiddle: frequency 0, or the last round of a tile)
__device__ __forceinline__ u64 conv_unit(const L4 &v) {
    u64 al = (u64)v.l[3] * (u32)B3 + v.l[0];
    al += (u64)v.l[1] << 24;
    u64 ah = (u64)v.l[3] * (u32)(B3 >> 32) + ((u64)v.l[2] << 16);
    return gl::fold1(al, ah);
}

struct LimbPassArgs {
    ntt::PassArgs a;
    const W2 *tw_all;    // the round tables of this tile shape, concatenated in their LDS layout (limb_tables_w2 entries)
    const u64 *ufac[4];  // rounds whose sub-block has more than 64 twiddle columns: ufac[r][a * 8 + k] = w_{Rb}^(64 * a * k)
    u64 wlast[4];        // WLAST kernels: the forms the round without table twiddles multiplies by, c * B^i (c = 1/n for the iNTT)
    const u64 *twid;     // inter-pass twiddles as in ntt::RegPassArgs
    unsigned xcd_remap;  // > 0 (= log2 of gridDim.x): workgroup b takes slot (b % 8) * gridDim.x / 8 + b / 8 (see ntt::RegPassArgs)
    unsigned zloop;      // > 0: this workgroup produces z = 0..zloop-1 itself
    unsigned tiles_log;  // a workgroup processes 2^tiles_log consecutive tiles (the staged tables are loaded once)
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
// Round tables live in LDS.  A table holds the 4-forms of w_{Rb}^(b * k) for b < 2^min(s_log, 6), k = 1..2^p - 1, as
// t[((k-1)*2 + h) << s_eff | b] = {W_{2h}, W_{2h+1}}; a round with more twiddle columns (lo = 64 a + b: the first round of the
// contiguous pass) multiplies by the wave-uniform w_{Rb}^(64 a k) afterwards -- 3584 distinct twiddles from 448 + 56.
constexpr int TW_S_MAX_LOG = 6;
constexpr int round_s_eff(int log_r, int r) {
    const int s = round_log_rb(log_r, r) - round_bits(log_r, r);
    return s > TW_S_MAX_LOG ? TW_S_MAX_LOG : s;
}
constexpr int round_table_w2(int log_r, int r) {  // W2 entries of round r's table
    const int s = round_log_rb(log_r, r) - round_bits(log_r, r);
    return s == 0 ? 0 : (((1 << round_bits(log_r, r)) - 1) * 2) << round_s_eff(log_r, r);
}
constexpr int round_table_off(int log_r, int r) {
    int o = 0;
    for (int q = 0; q < r; ++q) o += round_table_w2(log_r, q);
    return o;
}
constexpr int limb_tables_w2(int log_r) { return round_table_off(log_r, n_rounds(log_r)); }
constexpr size_t limb_shmem_bytes(int log_r) { return (size_t)8 * ntt::TILE_WORDS_PADDED + (size_t)16 * limb_tables_w2(log_r); }

__device__ __forceinline__ u64 limb_mul(u64 a, u64 b) { return gl::mul1(a, b); }
__device__ __forceinline__ unsigned wave_uniform(unsigned v) {
#ifdef P2HOT_EMU
    return v;
#else
    return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
#endif
}

template <bool INV, int LOG_R, int LOG_C, int SCALE, bool WLAST, int RI>
__device__ __forceinline__ void limb_round(const LimbPassArgs &ra, u64 *tile, const W2 *ltw, const u64 *gin, unsigned log_stride,
                                           size_t z, size_t base0) {
    constexpr int P = round_bits(LOG_R, RI);
    constexpr int LOG_RB = round_log_rb(LOG_R, RI);
    constexpr int S_LOG = LOG_RB - P;
    constexpr int S_EFF = round_s_eff(LOG_R, RI);
    constexpr bool FIRST = RI == 0;
    constexpr unsigned C = 1u << LOG_C;
    constexpr int UPT = (1 << (TILE_LOG - P)) / NT;  // units per thread
    const ntt::PassArgs &a = ra.a;
#pragma unroll 1
    for (int uu = 0; uu < UPT; ++uu) {
        const unsigned u = threadIdx.x + (unsigned)uu * NT;
        const unsigned c = u & (C - 1), rest = u >> LOG_C;
        const unsigned lo = rest & ((1u << S_LOG) - 1), hi = rest >> S_LOG;
        const unsigned i0 = (hi << LOG_RB) + lo;
        u64 v[1 << P];
        if constexpr (FIRST) {
            // one 32-bit per-thread offset, wave-uniform row steps (a block of one pass is < 2^32 elements)
            const u32 off0 = ((i0 << log_stride) + c) * 8u;  // bytes
#pragma unroll
            for (int q = 0; q < (1 << P); ++q)
                v[q] = *reinterpret_cast<const u64 *>(reinterpret_cast<const char *>(gin + ((size_t)q << (S_LOG + log_stride))) + off0);
        } else {
#pragma unroll
            for (int q = 0; q < (1 << P); ++q) v[q] = tile[ntt::pad_idx(((i0 + ((unsigned)q << S_LOG)) << LOG_C) + c)];
        }
        if constexpr (FIRST && SCALE == ntt::SCALE_CONST) {
#pragma unroll
            for (int q = 0; q < (1 << P); ++q) v[q] = limb_mul(v[q], a.scale_const);
        } else if constexpr (FIRST && SCALE == ntt::SCALE_TABLE) {
            const u64 scol = log_stride ? a.scol[(z << log_stride) + base0 + c] : 1;
#pragma unroll
            for (int q = 0; q < (1 << P); ++q) {
                u64 sc = a.srow[z * (1u << LOG_R) + i0 + ((unsigned)q << S_LOG)];
                if (log_stride) sc = limb_mul(sc, scol);
                v[q] = limb_mul(v[q], sc);
            }
        }
        L4 x[1 << P];
#pragma unroll
        for (int q = 0; q < (1 << P); ++q) x[q] = split(v[q]);
        x[0].l[0] += O0;
        x[0].l[1] += O1;
        x[0].l[2] += O2;
        x[0].l[3] += O3;
        dft_limbs<P, INV>(x);
        if constexpr (S_LOG > 0) {
            const W2 *tw = ltw + round_table_off(LOG_R, RI) + (lo & ((1u << S_EFF) - 1));
            const u64 *uf = nullptr;
            if constexpr (S_LOG > S_EFF) uf = ra.ufac[RI] + wave_uniform(lo >> S_EFF) * 8;  // lanes of a wave share u >> 6
            tile[ntt::pad_idx((i0 << LOG_C) + c)] = conv_unit(x[0]);
#pragma unroll
            for (int q = 1; q < (1 << P); ++q) {
                const unsigned k = (unsigned)(__brev((unsigned)q) >> (32 - P));
                const W2 wa = tw[((k - 1) * 2) << S_EFF], wb = tw[((k - 1) * 2 + 1) << S_EFF];
                u64 y = convmul(x[q], wa.a, wa.b, wb.a, wb.b);
                if constexpr (S_LOG > S_EFF) y = limb_mul(y, uf[k]);
                tile[ntt::pad_idx(((i0 + ((unsigned)q << S_LOG)) << LOG_C) + c)] = y;
            }
        } else {
            if constexpr (WLAST) {
#pragma unroll
                for (int q = 0; q < (1 << P); ++q)
                    tile[ntt::pad_idx(((i0 + (unsigned)q) << LOG_C) + c)] =
                        convmul(x[q], ra.wlast[0], ra.wlast[1], ra.wlast[2], ra.wlast[3]);
            } else {
#pragma unroll
                for (int q = 0; q < (1 << P); ++q) tile[ntt::pad_idx(((i0 + (unsigned)q) << LOG_C) + c)] = conv_unit(x[q]);
            }
        }
    }
    __syncthreads();
    if constexpr (RI + 1 < n_rounds(LOG_R))
        limb_round<INV, LOG_R, LOG_C, SCALE, WLAST, RI + 1>(ra, tile, ltw, gin, log_stride, z, base0);
}

// One pass over 2^LOG_R x 2^LOG_C tiles (LOG_R + LOG_C = 12); LOG_C = 0 is the contiguous (last) pass.
// grid = (tiles per polynomial >> tiles_log, polynomials, z), 512 threads, limb_shmem_bytes(LOG_R) of dynamic LDS.
// WLAST: the conversions of the tile's last round multiply by ra.wlast (the 1/n of an inverse transform) instead of 1.
template <bool INV, int LOG_R, int LOG_C, int SCALE, bool WLAST = false>
__global__ void __launch_bounds__(NT, LIMB_MIN_WAVES) ntt_limbpass_kernel(LimbPassArgs ra) {
    static_assert(LOG_R + LOG_C == TILE_LOG, "a tile is 4096 elements");
    P2HOT_DYN_SHARED(u64, tile);
    const ntt::PassArgs &a = ra.a;
    const unsigned tid = threadIdx.x;
    constexpr unsigned C = 1u << LOG_C;
    W2 *ltw = reinterpret_cast<W2 *>(tile + ntt::TILE_WORDS_PADDED);
    for (unsigned e = tid; e < (unsigned)limb_tables_w2(LOG_R); e += NT) ltw[e] = ra.tw_all[e];
    __syncthreads();
    const unsigned log_stride = LOG_C ? a.log_nblk - LOG_R : 0u;  // the contiguous pass is the last one: blocks of one tile
    const unsigned tiles_per_blk_log = log_stride - LOG_C;
    const size_t wg = (LOG_C && ra.xcd_remap) ? (((size_t)(blockIdx.x & 7u) << (ra.xcd_remap - 3)) | (blockIdx.x >> 3)) : blockIdx.x;
    const size_t z_begin = ra.zloop ? 0 : blockIdx.z, z_end = ra.zloop ? ra.zloop : blockIdx.z + 1;
#pragma unroll 1
    for (size_t t = 0; t < ((size_t)1 << ra.tiles_log); ++t) {
        const size_t tau = (wg << ra.tiles_log) + t;
        const size_t blk = tau >> tiles_per_blk_log;
        const size_t base0 = (tau & (((size_t)1 << tiles_per_blk_log) - 1)) << LOG_C;
#pragma unroll 1
        for (size_t z = z_begin; z < z_end; ++z) {
            const u64 *in = a.in + (size_t)blockIdx.y * a.in_poly_stride + z * a.in_z_stride + (blk << a.log_nblk) + base0;
            u64 *out = a.out + (size_t)blockIdx.y * a.out_poly_stride + z * a.out_z_stride + (blk << a.log_nblk) + base0;
            limb_round<INV, LOG_R, LOG_C, SCALE, WLAST, 0>(ra, tile, ltw, in, log_stride, z, base0);
            // inter-pass twiddle w_{n'}^(base * k1) from the per-pass table (laid out like a block), then the coalesced store
            if constexpr (LOG_C == 0) {
#pragma unroll
                for (unsigned e = tid; e < (1u << TILE_LOG); e += NT) {
                    const u64 v = tile[ntt::pad_idx(e)];
                    out[e] = a.canon_out ? gl::canon(v) : v;
                }
            } else {  // strided pass: the host always supplies the inter-pass table (blocks of <= 2^24 elements)
                const u32 off0 = (((tid >> LOG_C) << log_stride) + (tid & (C - 1))) * 8u;  // bytes, per thread; row steps are uniform
#pragma unroll 2
                for (unsigned j = 0; j < (1u << TILE_LOG) / NT; ++j) {
                    const size_t step = (size_t)(j * (NT >> LOG_C)) << log_stride;
                    const u64 w = *reinterpret_cast<const u64 *>(reinterpret_cast<const char *>(ra.twid + base0 + step) + off0);
                    u64 v = limb_mul(tile[ntt::pad_idx(tid + j * NT)], w);
                    *reinterpret_cast<u64 *>(reinterpret_cast<char *>(out + step) + off0) = a.canon_out ? gl::canon(v) : v;
                }
            }
            __syncthreads();  // the tile is reused by the next coset / tile
        }
    }
}

// Fills one round's table in its LDS layout (see round_s_eff) and, for a round with more than 64 twiddle columns, its
// wave-uniform factors: t[((k-1)*2 + h) << s_eff | b] = 4-form of w_{2^log_rb}^(b*k); u[a*8 + k] = w_{2^log_rb}^(64*a*k)
__global__ void limb_twiddle_kernel(W2 *t, u64 *u, unsigned log_rb, unsigned p, ntt::RootTable roots) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned s_log = log_rb - p, s_eff = s_log > (unsigned)TW_S_MAX_LOG ? (unsigned)TW_S_MAX_LOG : s_log;
    const size_t S = (size_t)1 << s_eff;
    if (idx < (((size_t)1 << p) - 1) * S) {
        const u32 b = (u32)(idx & (S - 1)), k = (u32)(idx >> s_eff) + 1;
        const u64 w = ntt::root_pow(roots, (u32)(((u64)b * k) << (32 - log_rb)));
        W2 *o = t + ((size_t)(k - 1) * 2 << s_eff) + b;
        o[0] = W2{gl::canon(w), gl::canon(gl::mul(w, B1))};
        o[S] = W2{gl::canon(gl::mul(w, B2)), gl::canon(gl::mul(w, B3))};
    }
    if (u && idx < ((size_t)8 << (s_log - s_eff))) {
        const u32 a = (u32)(idx >> 3), k = (u32)(idx & 7);
        u[idx] = gl::canon(ntt::root_pow(roots, (u32)((((u64)a << s_eff) * k) << (32 - log_rb))));
    }
}

}  // namespace nttl
