/*
 * p2fast.c -- tuned CPU implementation of PolynomialBatch::from_values / from_coeffs
 * (plonky2/src/fri/oracle.rs:57-112): the "port-tuned" CPU baseline of bench.py and the fast
 * checker of the full-size GPU parity tests.
 *
 * TEST INFRASTRUCTURE ONLY, like everything under oracle/ (see p2oracle.h): never loaded by the
 * product path.  It is NOT the thing parity is anchored on -- that is p2oracle.c, the faithful
 * restatement pinned by the reference's KATs.  This file computes the same function faster and is
 * itself pinned bit-for-bit to p2oracle.c in the CPU test tier (tests/test_fast_oracle.py: the
 * permutation on the reference KATs and on boundary states, whole commits over a grid of shapes).
 *
 * What is tuned, and after which part of the reference:
 *  - field multiply: branch-free reduce128 (field/src/goldilocks_field.rs:402-415; the reference's
 *    x86 build gets the same effect from `add_no_canonicalize_trashing_input`'s asm, :357-380).
 *    The faithful oracle's data-dependent borrow/carry branches mispredict half the time: 7-9 us per
 *    permutation there, ~10x less here.
 *  - Poseidon: the reference's own fast form (hash/poseidon.rs:742-777): full rounds with the
 *    32-bit-halves MDS (:180-199, :271-290), the 11x11 pre-matrix and 22 sparse partial rounds with
 *    lazily reduced u128 accumulators (:415-441, :516-542), non-canonical state throughout.
 *  - transforms: per-column radix-2 NTT as in field/src/fft.rs:165-202, but decimation in frequency
 *    with ONE cached root table per size (the reference rebuilds its table on every ifft call,
 *    fft.rs:41) and no bit-reversal pass where the committed order is bit-reversed anyway; the LDE
 *    is computed coset by coset (SURVEY 8e identity: the size-N coset LDE is 2^rate_bits size-n coset
 *    NTTs with shifts g*w_N^j, and coset j is the contiguous row block bitrev(j) of the committed order),
 *    so the working set is one [W][n] block instead of the [W][N] matrix.
 *  - "transpose LDEs": cache-blocked, per coset block; "build Merkle tree": flat-parallel leaf sponge,
 *    then level-parallel inner nodes.
 *  - AVX-512 (F + DQ, checked at run time, P2FAST_SCALAR=1 disables it): the leaf sponge and the tree levels hash eight
 *    rows / nodes per instruction stream (one 64-bit lane per state, 64 x 64 products from four vpmuludq), the NTT layers
 *    run eight butterflies per stream like the reference's packed fft_classic_simd (fft.rs:95-157).  The reference's own
 *    Poseidon is scalar on x86, so this baseline is, per core, faster than what the Rust prover would show.
 * OpenMP over columns / rows / nodes mirrors the reference's rayon fork-join points.
 * seconds[] reports the reference's TimingTree scopes (fri/oracle.rs:65-103): "IFFT",
 * "FFT + blinding", "transpose LDEs", "build Merkle tree".
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "poseidon_constants.h"

typedef unsigned __int128 u128;
typedef uint64_t u64;
#define P 0xFFFFFFFF00000001ULL
#define EPS 0xFFFFFFFFULL
#define COSET_SHIFT 14293326489335486720ULL
#define INL static inline __attribute__((always_inline))

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ------------------------------------------------------------------ field (any representative < 2^64) */
INL u64 canon(u64 x) { return x >= P ? x - P : x; }

INL u64 red128(u64 lo, u64 hi) { /* goldilocks_field.rs:402-415, branch-free */
    u64 hh = hi >> 32, hl = hi & EPS;
    u64 t0 = lo - hh;
    t0 -= (0 - (u64)(lo < hh)) & EPS;
    u64 t1 = (hl << 32) - hl;
    u64 t2 = t0 + t1;
    t2 += (0 - (u64)(t2 < t1)) & EPS;
    return t2;
}
INL u64 fmul(u64 a, u64 b) {
    u128 p = (u128)a * b;
    return red128((u64)p, (u64)(p >> 64));
}
INL u64 fadd(u64 a, u64 b) { /* goldilocks_field.rs:245-262: two wrap corrections cover non-canonical inputs */
    u64 s = a + b;
    u64 adj = (0 - (u64)(s < a)) & EPS;
    s += adj;
    s += (0 - (u64)(s < adj)) & EPS;
    return s;
}
INL u64 fsub(u64 a, u64 b) {
    u64 d = a - b;
    u64 adj = (0 - (u64)(a < b)) & EPS;
    u64 e = d - adj;
    e -= (0 - (u64)(d < adj)) & EPS;
    return e;
}
static u64 fpow(u64 a, u64 e) {
    u64 r = 1;
    while (e) {
        if (e & 1) r = fmul(r, a);
        a = fmul(a, a);
        e >>= 1;
    }
    return r;
}
static u64 root_of_unity(unsigned log_n) { /* field/src/types.rs:268-272 */
    u64 w = 7277203076849721926ULL;
    for (unsigned i = log_n; i < 32; ++i) w = fmul(w, w);
    return w;
}

/* ------------------------------------------------------------------ Poseidon (hash/poseidon.rs) */
INL u64 sbox(u64 x) { /* poseidon.rs:690-696 */
    u64 x2 = fmul(x, x), x4 = fmul(x2, x2), x3 = fmul(x, x2);
    return fmul(x3, x4);
}

INL void mds_layer(u64 s[12]) { /* poseidon.rs:180-199, :271-290 on the 32-bit halves of the state */
    /* 32 x 32 -> 64 widening multiplies of u32 arrays: the form gcc turns into vpmuludq (4 rows per instruction) */
    uint32_t lo[24], hi[24];
    u64 al[12], ah[12];
    for (int i = 0; i < 12; ++i) {
        lo[i] = lo[i + 12] = (uint32_t)s[i];
        hi[i] = hi[i + 12] = (uint32_t)(s[i] >> 32);
        al[i] = ah[i] = 0;
    }
    for (int i = 0; i < 12; ++i) {
        const uint32_t c = (uint32_t)P2_POSEIDON_MDS_CIRC[i];
        for (int r = 0; r < 12; ++r) {
            al[r] += (u64)lo[r + i] * c;
            ah[r] += (u64)hi[r + i] * c;
        }
    }
    al[0] += (u64)lo[0] * (uint32_t)P2_POSEIDON_MDS_DIAG[0]; /* the only nonzero diagonal entry (poseidon_goldilocks.rs:24-25) */
    ah[0] += (u64)hi[0] * (uint32_t)P2_POSEIDON_MDS_DIAG[0];
    for (int r = 0; r < 12; ++r) {
        u128 v = (u128)al[r] + ((u128)ah[r] << 32); /* < 2^75 */
        s[r] = red128((u64)v, (u64)(v >> 64));
    }
}

INL void full_round(u64 s[12], unsigned round) { /* poseidon.rs:742-749 */
    for (int i = 0; i < 12; ++i) s[i] = sbox(fadd(s[i], P2_POSEIDON_ALL_ROUND_CONSTANTS[12 * round + i]));
    mds_layer(s);
}

/* lo + 2^64 * hi with lo, hi < 2^68 (sums of at most 12 product halves); 2^64 = EPS, 2^128 = -2^32 (mod P) */
INL u64 red_acc(u128 lo, u128 hi) {
    u64 t = red128((u64)lo, (u64)hi);
    t = fadd(t, (u64)(lo >> 64) * EPS);
    return fsub(t, (u64)(hi >> 64) << 32);
}

static void poseidon(u64 s[12]) { /* poseidon.rs:767-777 */
    unsigned round = 0;
    for (int k = 0; k < 4; ++k) full_round(s, round++);
    /* partial_first_constant_layer + mds_partial_layer_init (poseidon.rs:365-375, :415-441) */
    for (int i = 0; i < 12; ++i) s[i] = fadd(s[i], P2_POSEIDON_FAST_PARTIAL_FIRST_ROUND_CONSTANT[i]);
    {
        u128 lo[11], hi[11];
        for (int c = 0; c < 11; ++c) lo[c] = hi[c] = 0;
        for (int r = 1; r < 12; ++r) {
            const u64 x = s[r];
            const u64 *row = P2_POSEIDON_FAST_PARTIAL_ROUND_INITIAL_MATRIX + (r - 1) * 11;
            for (int c = 0; c < 11; ++c) {
                u128 p = (u128)x * row[c];
                lo[c] += (u64)p;
                hi[c] += (u64)(p >> 64);
            }
        }
        for (int c = 0; c < 11; ++c) s[c + 1] = red_acc(lo[c], hi[c]);
    }
    /* 22 x { S-box on word 0, scalar constant, sparse matrix } (poseidon.rs:752-764, :516-542) */
    const u64 m00 = P2_POSEIDON_MDS_CIRC[0] + P2_POSEIDON_MDS_DIAG[0];
    for (int i = 0; i < 22; ++i) {
        const u64 s0 = fadd(sbox(s[0]), P2_POSEIDON_FAST_PARTIAL_ROUND_CONSTANTS[i]);
        const u64 *wh = P2_POSEIDON_FAST_PARTIAL_ROUND_W_HATS + 11 * i, *vs = P2_POSEIDON_FAST_PARTIAL_ROUND_VS + 11 * i;
        u128 p = (u128)s0 * m00;
        u128 lo = (u64)p, hi = (u64)(p >> 64);
        for (int j = 1; j < 12; ++j) {
            p = (u128)s[j] * wh[j - 1];
            lo += (u64)p;
            hi += (u64)(p >> 64);
        }
        for (int j = 1; j < 12; ++j) { /* s_j + s_0 * v_j: no overflow, (2^64-1)^2 + 2^64 - 1 < 2^128 */
            p = (u128)s0 * vs[j - 1] + s[j];
            s[j] = red128((u64)p, (u64)(p >> 64));
        }
        s[0] = red_acc(lo, hi);
    }
    round += 22;
    for (int k = 0; k < 4; ++k) full_round(s, round++);
}

static int have_avx512(void);
static void permute8(u64 *states);
void fast_poseidon(u64 *states, size_t count) {
    const size_t groups = have_avx512() ? count / 8 : 0;
#pragma omp parallel for schedule(static)
    for (size_t g = 0; g < groups; ++g) permute8(states + 96 * g);
#pragma omp parallel for schedule(static)
    for (size_t i = 8 * groups; i < count; ++i) {
        u64 *s = states + 12 * i;
        poseidon(s);
        for (int k = 0; k < 12; ++k) s[k] = canon(s[k]);
    }
}

/* hash_or_noop (plonk/config.rs:63-74) over hash_n_to_hash_no_pad (hash/hashing.rs:118-145) */
INL void hash_or_noop(const u64 *in, size_t len, u64 out[4]) {
    if (len <= 4) {
        for (size_t i = 0; i < 4; ++i) out[i] = i < len ? canon(in[i]) : 0;
        return;
    }
    u64 s[12] = {0};
    for (size_t off = 0; off < len; off += 8) {
        const size_t c = len - off < 8 ? len - off : 8;
        for (size_t i = 0; i < c; ++i) s[i] = in[off + i];
        poseidon(s);
    }
    for (int i = 0; i < 4; ++i) out[i] = canon(s[i]);
}
INL void two_to_one(const u64 *l, const u64 *r, u64 out[4]) { /* hashing.rs:97-114 */
    u64 s[12] = {l[0], l[1], l[2], l[3], r[0], r[1], r[2], r[3], 0, 0, 0, 0};
    poseidon(s);
    for (int i = 0; i < 4; ++i) out[i] = canon(s[i]);
}

static void hash_rows_any(const u64 *rows, size_t n_rows, size_t w, u64 *out);
void fast_hash_rows(const u64 *rows, size_t n_rows, size_t w, u64 *out) { hash_rows_any(rows, n_rows, w, out); }

/* ------------------------------------------------------------------ AVX-512: eight permutations per instruction stream
 * The reference's Poseidon is scalar on x86 (its AVX2 variant is commented out, hash/poseidon_goldilocks.rs:250-262), but
 * a leaf sponge over millions of independent rows vectorises across rows: one 64-bit lane per row, eight rows per zmm
 * register, the 12-word state in 12 registers.  A 64 x 64 -> 128 product is four vpmuludq (32 x 32 -> 64) partial
 * products; the reduction is the same reduce128 with mask registers for the borrow / carry corrections.  Used when the CPU
 * has AVX-512 F + DQ (checked at run time; the scalar path above is the fallback and the bit-exact cross-check). */
#include <immintrin.h>
typedef __m512i V;
#define VT __attribute__((target("avx512f,avx512dq"), always_inline)) static inline
#define VK(x) _mm512_set1_epi64((long long)(x))

VT V vred(V lo, V hi) { /* reduce128, goldilocks_field.rs:402-415 */
    const V E = VK(EPS);
    V hh = _mm512_srli_epi64(hi, 32), hl = _mm512_and_si512(hi, E);
    V t0 = _mm512_sub_epi64(lo, hh);
    __mmask8 b = _mm512_cmplt_epu64_mask(lo, hh);
    t0 = _mm512_mask_sub_epi64(t0, b, t0, E);
    V t1 = _mm512_sub_epi64(_mm512_slli_epi64(hl, 32), hl);
    V t2 = _mm512_add_epi64(t0, t1);
    __mmask8 c = _mm512_cmplt_epu64_mask(t2, t1);
    return _mm512_mask_add_epi64(t2, c, t2, E);
}
VT V vmul(V a, V b) {
    const V M = VK(EPS);
    V ah = _mm512_srli_epi64(a, 32), bh = _mm512_srli_epi64(b, 32);
    V ll = _mm512_mul_epu32(a, b), lh = _mm512_mul_epu32(a, bh), hl = _mm512_mul_epu32(ah, b), hh = _mm512_mul_epu32(ah, bh);
    V mid = _mm512_add_epi64(lh, _mm512_srli_epi64(ll, 32));   /* < 2^64: (2^32-1)^2 + 2^32 - 1 */
    V mid2 = _mm512_add_epi64(hl, _mm512_and_si512(mid, M));
    V lo = _mm512_or_si512(_mm512_slli_epi64(mid2, 32), _mm512_and_si512(ll, M));
    V hi = _mm512_add_epi64(hh, _mm512_add_epi64(_mm512_srli_epi64(mid, 32), _mm512_srli_epi64(mid2, 32)));
    return vred(lo, hi);
}
VT V vadd(V a, V b) { /* any representatives: two wrap corrections */
    const V E = VK(EPS);
    V s = _mm512_add_epi64(a, b);
    __mmask8 c = _mm512_cmplt_epu64_mask(s, a);
    V s2 = _mm512_mask_add_epi64(s, c, s, E);
    __mmask8 c2 = _mm512_mask_cmplt_epu64_mask(c, s2, E);
    return _mm512_mask_add_epi64(s2, c2, s2, E);
}
VT V vadd_canon(V a, V k) { /* k canonical (< P): one correction is enough */
    V s = _mm512_add_epi64(a, k);
    __mmask8 c = _mm512_cmplt_epu64_mask(s, a);
    return _mm512_mask_add_epi64(s, c, s, VK(EPS));
}
VT V vsbox(V x) {
    V x2 = vmul(x, x), x4 = vmul(x2, x2), x3 = vmul(x, x2);
    return vmul(x3, x4);
}
VT void vmds(V s[12]) { /* poseidon.rs:180-199, :271-290 on the 32-bit halves, one row at a time */
    V hi[12], out[12];
    const V E = VK(EPS);
    for (int i = 0; i < 12; ++i) hi[i] = _mm512_srli_epi64(s[i], 32);
    for (int r = 0; r < 12; ++r) {
        V al = _mm512_setzero_si512(), ah = _mm512_setzero_si512();
        for (int i = 0; i < 12; ++i) {
            const V c = VK(P2_POSEIDON_MDS_CIRC[i]);
            const int j = r + i >= 12 ? r + i - 12 : r + i;
            al = _mm512_add_epi64(al, _mm512_mul_epu32(s[j], c)); /* vpmuludq reads the low 32 bits of s[j] */
            ah = _mm512_add_epi64(ah, _mm512_mul_epu32(hi[j], c));
        }
        if (r == 0) {
            const V d = VK(P2_POSEIDON_MDS_DIAG[0]);
            al = _mm512_add_epi64(al, _mm512_mul_epu32(s[0], d));
            ah = _mm512_add_epi64(ah, _mm512_mul_epu32(hi[0], d));
        }
        /* al + ah * 2^32 (< 2^75): lo64 with its carry, the bits above 2^64 fold with 2^64 = EPS */
        V sh = _mm512_slli_epi64(ah, 32);
        V lo = _mm512_add_epi64(al, sh);
        __mmask8 c = _mm512_cmplt_epu64_mask(lo, sh);
        V top = _mm512_mask_add_epi64(_mm512_srli_epi64(ah, 32), c, _mm512_srli_epi64(ah, 32), VK(1)); /* < 2^11 */
        V t = _mm512_sub_epi64(_mm512_slli_epi64(top, 32), top);                                    /* top * EPS */
        V y = _mm512_add_epi64(lo, t);
        __mmask8 c2 = _mm512_cmplt_epu64_mask(y, t);
        out[r] = _mm512_mask_add_epi64(y, c2, y, E);
    }
    for (int r = 0; r < 12; ++r) s[r] = out[r];
}
VT void vfull_round(V s[12], unsigned round) {
    for (int i = 0; i < 12; ++i) s[i] = vsbox(vadd_canon(s[i], VK(P2_POSEIDON_ALL_ROUND_CONSTANTS[12 * round + i])));
    vmds(s);
}
__attribute__((target("avx512f,avx512dq"))) static void poseidon8(V s[12]) { /* poseidon.rs:767-777, eight states */
    unsigned round = 0;
    for (int k = 0; k < 4; ++k) vfull_round(s, round++);
    for (int i = 0; i < 12; ++i) s[i] = vadd_canon(s[i], VK(P2_POSEIDON_FAST_PARTIAL_FIRST_ROUND_CONSTANT[i]));
    {   /* mds_partial_layer_init (poseidon.rs:415-441) */
        V t[11];
        for (int c = 0; c < 11; ++c) {
            V acc = vmul(s[1], VK(P2_POSEIDON_FAST_PARTIAL_ROUND_INITIAL_MATRIX[c]));
            for (int r = 2; r < 12; ++r) acc = vadd(acc, vmul(s[r], VK(P2_POSEIDON_FAST_PARTIAL_ROUND_INITIAL_MATRIX[(r - 1) * 11 + c])));
            t[c] = acc;
        }
        for (int c = 0; c < 11; ++c) s[c + 1] = t[c];
    }
    const V m00 = VK(P2_POSEIDON_MDS_CIRC[0] + P2_POSEIDON_MDS_DIAG[0]);
    for (int i = 0; i < 22; ++i) { /* poseidon.rs:752-764, :516-542 */
        const u64 *wh = P2_POSEIDON_FAST_PARTIAL_ROUND_W_HATS + 11 * i, *vs = P2_POSEIDON_FAST_PARTIAL_ROUND_VS + 11 * i;
        const V s0 = vadd_canon(vsbox(s[0]), VK(P2_POSEIDON_FAST_PARTIAL_ROUND_CONSTANTS[i]));
        V d = vmul(s0, m00);
        for (int j = 1; j < 12; ++j) d = vadd(d, vmul(s[j], VK(wh[j - 1])));
        for (int j = 1; j < 12; ++j) s[j] = vadd(s[j], vmul(s0, VK(vs[j - 1])));
        s[0] = d;
    }
    round += 22;
    for (int k = 0; k < 4; ++k) vfull_round(s, round++);
}
VT V vcanon(V x) {
    __mmask8 ge = _mm512_cmpge_epu64_mask(x, VK(P));
    return _mm512_mask_sub_epi64(x, ge, x, VK(P));
}

static int have_avx512(void) {
    static int v = -1;
    if (v < 0) v = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") && !getenv("P2FAST_SCALAR");
    return v;
}

/* hash_or_noop of 8 rows at once (w > 4): rows r0 .. r0+7 of a row-major matrix with row stride `stride` words */
__attribute__((target("avx512f,avx512dq"))) static void hash8(const u64 *rows, size_t stride, size_t w, u64 *out /* [8][4] */) {
    const V idx = _mm512_mullo_epi64(_mm512_set_epi64(7, 6, 5, 4, 3, 2, 1, 0), VK(stride));
    V s[12];
    for (int i = 0; i < 12; ++i) s[i] = _mm512_setzero_si512();
    for (size_t off = 0; off < w; off += 8) {
        const size_t c = w - off < 8 ? w - off : 8;
        for (size_t i = 0; i < c; ++i) s[i] = _mm512_i64gather_epi64(idx, (const long long *)(rows + off + i), 8);
        poseidon8(s);
    }
    const V oidx = _mm512_set_epi64(28, 24, 20, 16, 12, 8, 4, 0);
    for (int i = 0; i < 4; ++i) _mm512_i64scatter_epi64((long long *)(out + i), oidx, vcanon(s[i]), 8);
}
/* two_to_one of 8 sibling pairs: pairs[k] = {l[4], r[4]} contiguous (8 words per node), out [8][4] */
__attribute__((target("avx512f,avx512dq"))) static void two_to_one8(const u64 *pairs, u64 *out) {
    const V idx = _mm512_set_epi64(56, 48, 40, 32, 24, 16, 8, 0);
    V s[12];
    for (int i = 0; i < 8; ++i) s[i] = _mm512_i64gather_epi64(idx, (const long long *)(pairs + i), 8);
    for (int i = 8; i < 12; ++i) s[i] = _mm512_setzero_si512();
    poseidon8(s);
    const V oidx = _mm512_set_epi64(28, 24, 20, 16, 12, 8, 4, 0);
    for (int i = 0; i < 4; ++i) _mm512_i64scatter_epi64((long long *)(out + i), oidx, vcanon(s[i]), 8);
}
__attribute__((target("avx512f,avx512dq"))) static void permute8(u64 *states /* [8][12] */) {
    const V idx = _mm512_set_epi64(84, 72, 60, 48, 36, 24, 12, 0);
    V s[12];
    for (int i = 0; i < 12; ++i) s[i] = _mm512_i64gather_epi64(idx, (const long long *)(states + i), 8);
    poseidon8(s);
    for (int i = 0; i < 12; ++i) _mm512_i64scatter_epi64((long long *)(states + i), idx, vcanon(s[i]), 8);
}

/* leaf digests of n_rows rows (row-major, width w): the AVX-512 path on groups of 8 rows, scalar for the rest */
static void hash_rows_any(const u64 *rows, size_t n_rows, size_t w, u64 *out) {
    const size_t groups = (have_avx512() && w > 4) ? n_rows / 8 : 0;
#pragma omp parallel for schedule(static)
    for (size_t g = 0; g < groups; ++g) hash8(rows + 8 * g * w, w, w, out + 32 * g);
#pragma omp parallel for schedule(static)
    for (size_t i = 8 * groups; i < n_rows; ++i) hash_or_noop(rows + i * w, w, out + 4 * i);
}

VT V vsub(V a, V b) { /* any representatives */
    const V E = VK(EPS);
    V d = _mm512_sub_epi64(a, b);
    __mmask8 c = _mm512_cmplt_epu64_mask(a, b);
    V d2 = _mm512_mask_sub_epi64(d, c, d, E);
    __mmask8 c2 = _mm512_mask_cmplt_epu64_mask(c, d, E);
    return _mm512_mask_sub_epi64(d2, c2, d2, E);
}
/* one DIF layer on blocks of m = 2 * half >= 16 points, eight butterflies per instruction stream (the reference's
 * fft_classic_simd, field/src/fft.rs:95-157, does the same with its packed field type) */
__attribute__((target("avx512f,avx512dq"))) static void dif_layer_avx512(u64 *v, size_t n, size_t half, size_t stride, const u64 *tw) {
    const V step = _mm512_mullo_epi64(_mm512_set_epi64(7, 6, 5, 4, 3, 2, 1, 0), VK(stride));
    for (size_t k = 0; k < n; k += 2 * half) {
        u64 *a = v + k, *b = v + k + half;
        for (size_t j = 0; j < half; j += 8) {
            const V x = _mm512_loadu_si512((const void *)(a + j)), y = _mm512_loadu_si512((const void *)(b + j));
            const V w = stride == 1 ? _mm512_loadu_si512((const void *)(tw + j)) : _mm512_i64gather_epi64(step, (const long long *)(tw + j * stride), 8);
            _mm512_storeu_si512((void *)(a + j), vadd(x, y));
            _mm512_storeu_si512((void *)(b + j), vmul(vsub(x, y), w));
        }
    }
}

/* ------------------------------------------------------------------ NTT (field/src/fft.rs) */
static size_t bitrev(size_t x, unsigned bits) {
    size_t r = 0;
    for (unsigned i = 0; i < bits; ++i) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

/* tw[i] = w^i, i < n/2 */
static u64 *root_table(unsigned log_n, int inverse) {
    const size_t half = log_n ? (size_t)1 << (log_n - 1) : 1;
    u64 *tw = (u64 *)malloc(half * 8);
    u64 w = root_of_unity(log_n);
    if (inverse) w = fpow(w, P - 2);
    /* independent chunks so that the table build itself is parallel: tw[c*B + k] = w^(c*B) * w^k */
    const size_t B = half < 4096 ? half : 4096;
#pragma omp parallel for schedule(static)
    for (size_t c = 0; c < half / B; ++c) {
        u64 x = fpow(w, c * B);
        for (size_t k = 0; k < B; ++k) {
            tw[c * B + k] = x;
            x = fmul(x, w);
        }
    }
    return tw;
}

/* decimation in frequency, in place: natural-order input -> bit-reversed-order output,
 * out[bitrev(i)] = sum_t in[t] w^(i t).  Same butterflies as fft_classic (fft.rs:165-202) run from the
 * widest layer down; twiddles from the one table by stride. */
static void ntt_dif(u64 *v, unsigned log_n, const u64 *tw) {
    const size_t n = (size_t)1 << log_n;
    for (unsigned lg_m = log_n; lg_m >= 1; --lg_m) {
        const size_t m = (size_t)1 << lg_m, half = m >> 1, stride = n >> lg_m;
        if (half >= 8 && have_avx512()) {
            dif_layer_avx512(v, n, half, stride, tw);
        } else if (half >= 4) {
            for (size_t k = 0; k < n; k += m) {
                u64 *a = v + k, *b = v + k + half;
                for (size_t j = 0; j < half; ++j) {
                    const u64 x = a[j], y = b[j];
                    a[j] = fadd(x, y);
                    b[j] = fmul(fsub(x, y), tw[j * stride]);
                }
            }
        } else { /* last two layers: few distinct twiddles, walk the blocks */
            for (size_t k = 0; k < n; k += m)
                for (size_t j = 0; j < half; ++j) {
                    const u64 x = v[k + j], y = v[k + half + j];
                    v[k + j] = fadd(x, y);
                    v[k + half + j] = fmul(fsub(x, y), tw[j * stride]);
                }
        }
    }
}

/* ------------------------------------------------------------------ PolynomialBatch */
/* cols [W][n] column-major.  coeffs_out [W][n] or NULL; leaves_out [N][W] (bit-reversed rows) or NULL;
 * digests_out 2*(N - 2^cap)*4 words or NULL; cap_out (4 << cap_height) words.  seconds[4] accumulates. */
int fast_commit(const u64 *cols, size_t W, unsigned log_n, unsigned rate_bits, unsigned cap_height, int is_values,
                u64 *coeffs_out, u64 *leaves_out, u64 *digests_out, u64 *cap_out, double seconds[4]) {
    const size_t n = (size_t)1 << log_n, N = n << rate_bits, n_cap = (size_t)1 << cap_height;
    const unsigned log_N = log_n + rate_bits;
    if (cap_height > log_N) return -1;
    double t_ifft = 0, t_fft = 0, t_tr = 0, t_tree = 0, t0 = now_s();
    u64 *coeffs = coeffs_out ? coeffs_out : (u64 *)malloc((W ? W : 1) * n * 8);
    u64 *tw = root_table(log_n, 0);
    if (!coeffs || !tw) return -2;
    /* "IFFT" (oracle.rs:65-69; fft.rs:68-91): DIF with inverse roots, then coeffs[i] = n^-1 * out[bitrev(i)] */
    if (is_values) {
        u64 *twi = root_table(log_n, 1);
        const u64 n_inv = fpow(n % P, P - 2);
#pragma omp parallel
        {
            u64 *tmp = (u64 *)malloc(n * 8);
#pragma omp for schedule(dynamic)
            for (size_t c = 0; c < W; ++c) {
                memcpy(tmp, cols + c * n, n * 8);
                ntt_dif(tmp, log_n, twi);
                u64 *dst = coeffs + c * n;
                for (size_t i = 0; i < n; ++i) dst[bitrev(i, log_n)] = canon(fmul(tmp[i], n_inv));
            }
            free(tmp);
        }
        free(twi);
    } else if (coeffs != cols) {
        for (size_t i = 0; i < W * n; ++i) coeffs[i] = canon(cols[i]);
    }
    t_ifft = now_s() - t0;
    /* leaf digests of the whole tree (32 B per row), then the levels */
    u64 *leaf_dig = (u64 *)malloc(N * 32);
    u64 *blk = (u64 *)malloc((W ? W : 1) * n * 8);      /* one coset block, column-major [W][n] */
    u64 *rows = leaves_out ? NULL : (u64 *)malloc((W ? W : 1) * n * 8); /* the same block, row-major */
    if (!leaf_dig || !blk || (!leaves_out && !rows)) return -2;
    const u64 wN = root_of_unity(log_N);
    for (size_t b = 0; b < ((size_t)1 << rate_bits); ++b) {
        /* "FFT + blinding" (oracle.rs:114-139), coset j = bitrev(b): p(g w_N^j w_n^q) = NTT_n(c_t (g w_N^j)^t)[q] */
        double t1 = now_s();
        const u64 s = fmul(COSET_SHIFT, fpow(wN, bitrev(b, rate_bits)));
#pragma omp parallel for schedule(dynamic)
        for (size_t c = 0; c < W; ++c) {
            u64 *dst = blk + c * n;
            const u64 *src = coeffs + c * n;
            /* powers of s in independent chunks (the reference's serial powers() chain, types.rs:580-584) */
            const size_t B = n < 1024 ? n : 1024;
            for (size_t k0 = 0; k0 < n; k0 += B) {
                u64 pw = fpow(s, k0);
                for (size_t k = k0; k < k0 + B; ++k) {
                    dst[k] = fmul(src[k], pw);
                    pw = fmul(pw, s);
                }
            }
            ntt_dif(dst, log_n, tw);
        }
        double t2 = now_s();
        /* "transpose LDEs" + reverse_index_bits (oracle.rs:97-98): block rows are already in committed order */
        u64 *r = leaves_out ? leaves_out + b * n * W : rows;
        const size_t TB = 64;
#pragma omp parallel for schedule(static)
        for (size_t q0 = 0; q0 < n; q0 += TB)
            for (size_t c = 0; c < W; ++c) {
                const u64 *src = blk + c * n + q0;
                const size_t lim = n - q0 < TB ? n - q0 : TB;
                for (size_t q = 0; q < lim; ++q) r[(q0 + q) * W + c] = canon(src[q]);
            }
        double t3 = now_s();
        /* "build Merkle tree", leaves (merkle_tree.rs:86-113 fill_subtree's leaf case) */
        hash_rows_any(r, n, W, leaf_dig + 4 * b * n);
        double t4 = now_s();
        t_fft += t2 - t1;
        t_tr += t3 - t2;
        t_tree += t4 - t3;
    }
    double t5 = now_s();
    /* inner nodes, level by level, written into the reference layout (merkle_tree.rs:50-57): node j of level i of a
     * cap subtree lives at 2*(((j>>1) << (i+1)) + 2^i - 1) + (j&1) of that subtree's slice; roots go to the cap */
    const unsigned h = log_N - cap_height;
    const size_t sub_leaves = (size_t)1 << h, sub_digests = 2 * (sub_leaves - 1);
    if (h == 0) {
        memcpy(cap_out, leaf_dig, N * 32);
    } else {
        u64 *cur = leaf_dig, *nxt = (u64 *)malloc(N * 16);
        u64 *dig = digests_out ? digests_out : (u64 *)malloc(n_cap * sub_digests * 32);
        if (!nxt || !dig) return -2;
        for (unsigned lvl = 0; lvl < h; ++lvl) {
            const size_t per = sub_leaves >> lvl; /* nodes of this level per subtree */
            const size_t parents = n_cap * per / 2;
            u64 *dst = (lvl + 1 == h) ? cap_out : nxt; /* parent g's digest goes to dst + 4 g (at the top level g = subtree) */
#pragma omp parallel for schedule(static)
            for (size_t g = 0; g < parents; ++g) { /* g = parent index over the whole forest; its children are cur[2g], cur[2g+1] */
                const size_t sidx = g / (per / 2), jp = g % (per / 2);
                u64 *slot = dig + 4 * (sidx * sub_digests + 2 * ((jp << (lvl + 1)) + ((size_t)1 << lvl) - 1));
                memcpy(slot, cur + 8 * g, 64);
            }
            const size_t groups = have_avx512() ? parents / 8 : 0;
#pragma omp parallel for schedule(static)
            for (size_t q = 0; q < groups; ++q) two_to_one8(cur + 64 * q, dst + 32 * q);
#pragma omp parallel for schedule(static)
            for (size_t g = 8 * groups; g < parents; ++g) two_to_one(cur + 8 * g, cur + 8 * g + 4, dst + 4 * g);
            if (lvl == 0) {
                cur = nxt;
                nxt = (u64 *)malloc(N * 8 > 32 ? N * 8 : 32);
                if (!nxt) return -2;
            } else {
                u64 *t = cur;
                cur = nxt;
                nxt = t;
            }
        }
        if (cur != leaf_dig) free(cur);
        free(nxt);
        if (!digests_out) free(dig);
    }
    t_tree += now_s() - t5;
    free(leaf_dig);
    free(blk);
    free(rows);
    free(tw);
    if (!coeffs_out) free(coeffs);
    if (seconds) {
        seconds[0] += t_ifft;
        seconds[1] += t_fft;
        seconds[2] += t_tr;
        seconds[3] += t_tree;
    }
    return 0;
}

int fast_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void fast_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
