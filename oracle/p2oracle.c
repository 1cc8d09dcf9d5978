/*
 * p2oracle.c -- CPU restatement of the plonky2 LDE + Poseidon-commit + FRI-commit path.
 * TEST INFRASTRUCTURE ONLY (see p2oracle.h).  Citations are into /root/reference.
 *
 * The structure deliberately follows the reference (per-column radix-2 DIT NTT after an
 * explicit bit reversal, gather transpose, recursive Merkle fill above a flat-parallel leaf hash,
 * a fresh extension NTT per FRI round) -- it is the thing the HIP path is checked against and the "port" CPU baseline,
 * not an optimised CPU prover.  OpenMP mirrors the reference's rayon fork-join points.
 */
#include "p2oracle.h"

#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "poseidon_constants.h"

typedef unsigned __int128 u128;
#define P ORA_P
#define EPS 0xFFFFFFFFULL

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

void ora_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ---- permutation argument ----
 * wires_permutation_partial_products_and_zs (plonky2/src/plonk/prover.rs:392-449) for one (beta, gamma):
 * per row the numerators / denominators (:410-424), their element-wise quotients through inverses
 * (batch_multiplicative_inverse == the element inverses, field/src/types.rs:133), quotient_chunk_products
 * (util/partial_products.rs:13-24), then the SEQUENTIAL row walk with partial_products_and_z_gx (:28-37) and the
 * Z(x) / Z(gx) swap (prover.rs:437-441).  out is the transposed result (:444-447): [num_prods + 1][n], the last
 * polynomial being Z.  Returns -1 if a denominator is zero (the reference panics). */
int ora_partial_products(const uint64_t *wires, const uint64_t *sigmas, const uint64_t *k_is, size_t num_routed,
                         unsigned log_n, size_t degree, uint64_t beta, uint64_t gamma, uint64_t *out) {
    const size_t n = (size_t)1 << log_n;
    const size_t num_chunks = (num_routed + degree - 1) / degree, num_prods = num_chunks - 1;
    const uint64_t g = ora_gl_root_of_unity(log_n);
    uint64_t x = 1, z_x = 1; /* subgroup[i] = g^i (two_adic_subgroup, field/src/types.rs:274-283) */
    uint64_t *q = (uint64_t *)malloc(sizeof(uint64_t) * (num_chunks ? num_chunks : 1));
    int rc = 0;
    for (size_t i = 0; i < n; ++i) {
        for (size_t c = 0; c < num_chunks; ++c) {
            uint64_t prod = 1;
            for (size_t j = c * degree; j < (c + 1) * degree && j < num_routed; ++j) {
                const uint64_t wire = wires[j * n + i];
                const uint64_t s_id = ora_gl_mul(k_is[j], x);
                const uint64_t num = ora_gl_add(ora_gl_add(wire, ora_gl_mul(beta, s_id)), gamma);
                const uint64_t den = ora_gl_add(ora_gl_add(wire, ora_gl_mul(beta, sigmas[j * n + i])), gamma);
                if (ora_gl_canon(den) == 0) rc = -1;
                prod = ora_gl_mul(prod, ora_gl_mul(num, ora_gl_inv(den)));
            }
            q[c] = prod;
        }
        uint64_t acc = z_x;
        for (size_t c = 0; c < num_chunks; ++c) {
            acc = ora_gl_mul(acc, q[c]);
            if (c < num_prods) out[c * n + i] = ora_gl_canon(acc);
        }
        out[num_prods * n + i] = ora_gl_canon(z_x); /* the swap: store Z(x), carry Z(gx) */
        z_x = acc;
        x = ora_gl_mul(x, g);
    }
    free(q);
    return rc;
}

int ora_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------ field */
/* goldilocks_field.rs:217-224 to_canonical_u64 */
uint64_t ora_gl_canon(uint64_t x) { return x >= P ? x - P : x; }

/* goldilocks_field.rs:245-262 (Add): canonical output here; equality in the reference is
 * canonical-value equality (goldilocks_field.rs:33-37) so any representative is equivalent. */
uint64_t ora_gl_add(uint64_t a, uint64_t b) {
    uint64_t s = a + b;
    if (s < a) {            /* wrapped: 2^64 = EPS (mod P) */
        s += EPS;
        if (s < EPS) s += EPS;
    }
    return ora_gl_canon(s);
}

uint64_t ora_gl_sub(uint64_t a, uint64_t b) {
    uint64_t d = a - b;
    if (a < b) {
        uint64_t e = d;
        d -= EPS;
        if (e < EPS) d -= EPS;
    }
    return ora_gl_canon(d);
}

/* goldilocks_field.rs:402-415 reduce128: x_lo - x_hi_hi + x_hi_lo * EPSILON */
static uint64_t reduce128(u128 x) {
    uint64_t x_lo = (uint64_t)x, x_hi = (uint64_t)(x >> 64);
    uint64_t x_hi_hi = x_hi >> 32, x_hi_lo = x_hi & EPS;
    uint64_t t0 = x_lo - x_hi_hi;
    if (x_lo < x_hi_hi) t0 -= EPS; /* borrow */
    uint64_t t1 = x_hi_lo * EPS;
    uint64_t t2 = t0 + t1;
    if (t2 < t1) t2 += EPS; /* carry: add_no_canonicalize_trashing_input */
    return ora_gl_canon(t2);
}

uint64_t ora_gl_mul(uint64_t a, uint64_t b) { return reduce128((u128)a * b); }

uint64_t ora_gl_pow(uint64_t a, uint64_t e) {
    uint64_t r = 1;
    while (e) {
        if (e & 1) r = ora_gl_mul(r, a);
        a = ora_gl_mul(a, a);
        e >>= 1;
    }
    return r;
}

uint64_t ora_gl_inv(uint64_t a) { return ora_gl_pow(a, P - 2); } /* Fermat; goldilocks_field.rs:108-147 */

/* types.rs:268-272: POWER_OF_TWO_GENERATOR^(2^(32 - log_n)); goldilocks_field.rs:87 */
uint64_t ora_gl_root_of_unity(unsigned log_n) {
    uint64_t w = 7277203076849721926ULL;
    for (unsigned i = log_n; i < 32; ++i) w = ora_gl_mul(w, w);
    return w;
}

/* extension/quadratic.rs:180-194 with W = 7 (goldilocks_extensions.rs:19) */
void ora_ext2_mul(const uint64_t a[2], const uint64_t b[2], uint64_t out[2]) {
    uint64_t c0 = ora_gl_add(ora_gl_mul(a[0], b[0]), ora_gl_mul(7, ora_gl_mul(a[1], b[1])));
    uint64_t c1 = ora_gl_add(ora_gl_mul(a[0], b[1]), ora_gl_mul(a[1], b[0]));
    out[0] = c0;
    out[1] = c1;
}

/* ------------------------------------------------------------------ util */
static size_t reverse_bits(size_t x, unsigned bits) { /* plonky2/src/util/mod.rs:33-41 */
    size_t r = 0;
    for (unsigned i = 0; i < bits; ++i) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

static unsigned log2_strict(size_t n) {
    unsigned l = 0;
    while (((size_t)1 << l) < n) ++l;
    return l;
}

/* util/src/lib.rs:185-234 semantics (a[i] <-> a[bitrev(i)]); elements of `w` words */
void ora_reverse_index_bits(uint64_t *a, size_t n, size_t w) {
    unsigned lg = log2_strict(n);
    uint64_t *tmp = (uint64_t *)malloc(w * sizeof(uint64_t));
    for (size_t i = 0; i < n; ++i) {
        size_t j = reverse_bits(i, lg);
        if (i < j) {
            memcpy(tmp, a + i * w, w * 8);
            memcpy(a + i * w, a + j * w, w * 8);
            memcpy(a + j * w, tmp, w * 8);
        }
    }
    free(tmp);
}

/* ------------------------------------------------------------------ fft */
/* fft.rs:165-202 fft_classic: bit-reverse, zero-tail replication (r), DIT layers with
 * per-layer twiddles root_table[lg_half_m][j] = w_m^j (fft.rs:14-33). */
void ora_fft(uint64_t *v, unsigned lg_n, unsigned r) {
    size_t n = (size_t)1 << lg_n;
    ora_reverse_index_bits(v, n, 1);
    if (r > 0) {
        size_t mask = ~(((size_t)1 << r) - 1);
        for (size_t i = 0; i < n; ++i) v[i] = v[i & mask];
    }
    for (unsigned lg_half_m = r; lg_half_m < lg_n; ++lg_half_m) {
        size_t half_m = (size_t)1 << lg_half_m, m = half_m << 1;
        uint64_t w_m = ora_gl_root_of_unity(lg_half_m + 1);
        uint64_t *tw = (uint64_t *)malloc(half_m * 8);
        tw[0] = 1;
        for (size_t j = 1; j < half_m; ++j) tw[j] = ora_gl_mul(tw[j - 1], w_m);
        for (size_t k = 0; k < n; k += m)
            for (size_t j = 0; j < half_m; ++j) {
                uint64_t t = ora_gl_mul(tw[j], v[k + half_m + j]);
                uint64_t u = v[k + j];
                v[k + j] = ora_gl_add(u, t);
                v[k + half_m + j] = ora_gl_sub(u, t);
            }
        free(tw);
    }
}

/* fft.rs:68-91 ifft_with_options: forward NTT, reverse all but the first, times n^-1 */
void ora_ifft(uint64_t *v, unsigned lg_n) {
    size_t n = (size_t)1 << lg_n;
    uint64_t n_inv = ora_gl_inv((uint64_t)n % P); /* types.rs:227-262 inverse_2exp */
    ora_fft(v, lg_n, 0);
    if (n == 1) {
        v[0] = ora_gl_mul(v[0], n_inv);
        return;
    }
    v[0] = ora_gl_mul(v[0], n_inv);
    v[n / 2] = ora_gl_mul(v[n / 2], n_inv);
    for (size_t i = 1; i < n / 2; ++i) {
        size_t j = n - i;
        uint64_t ci = ora_gl_mul(v[j], n_inv), cj = ora_gl_mul(v[i], n_inv);
        v[i] = ci;
        v[j] = cj;
    }
}

/* polynomial/mod.rs:280-293: scale coeff t by shift^t, then fft with zero_factor */
void ora_coset_fft(uint64_t *c, unsigned lg_n, uint64_t shift, unsigned zero_factor) {
    size_t n = (size_t)1 << lg_n;
    uint64_t pw = 1;
    for (size_t t = 0; t < n; ++t) {
        c[t] = ora_gl_mul(c[t], pw);
        pw = ora_gl_mul(pw, shift);
    }
    ora_fft(c, lg_n, zero_factor);
}

/* polynomial/mod.rs:63-73 */
void ora_coset_ifft(uint64_t *v, unsigned lg_n, uint64_t shift) {
    size_t n = (size_t)1 << lg_n;
    ora_ifft(v, lg_n);
    uint64_t si = ora_gl_inv(shift), pw = 1;
    for (size_t t = 0; t < n; ++t) {
        v[t] = ora_gl_mul(v[t], pw);
        pw = ora_gl_mul(pw, si);
    }
}

/* ------------------------------------------------------------------ poseidon */
static inline uint64_t sbox(uint64_t x) { /* poseidon.rs:690-696 */
    uint64_t x2 = ora_gl_mul(x, x), x4 = ora_gl_mul(x2, x2), x3 = ora_gl_mul(x, x2);
    return ora_gl_mul(x3, x4);
}

static void constant_layer(uint64_t s[12], unsigned round) { /* poseidon.rs:632-641 */
    for (int i = 0; i < 12; ++i) s[i] = ora_gl_add(s[i], P2_POSEIDON_ALL_ROUND_CONSTANTS[i + 12 * round]);
}

static void sbox_layer(uint64_t s[12]) { /* poseidon.rs:712-718 */
    for (int i = 0; i < 12; ++i) s[i] = sbox(s[i]);
}

/* poseidon.rs:180-199 mds_row_shf + :271-290 mds_layer (u128 accumulate, one reduction) */
static void mds_layer(uint64_t s[12]) {
    uint64_t lo[12], hi[12], out[12];
    for (int i = 0; i < 12; ++i) {
        lo[i] = s[i] & EPS;
        hi[i] = s[i] >> 32;
    }
    for (int r = 0; r < 12; ++r) {
        uint64_t al = 0, ah = 0; /* < 2^42 each */
        for (int i = 0; i < 12; ++i) {
            int j = i + r >= 12 ? i + r - 12 : i + r;
            al += lo[j] * P2_POSEIDON_MDS_CIRC[i];
            ah += hi[j] * P2_POSEIDON_MDS_CIRC[i];
        }
        al += lo[r] * P2_POSEIDON_MDS_DIAG[r];
        ah += hi[r] * P2_POSEIDON_MDS_DIAG[r];
        out[r] = reduce128((u128)al + ((u128)ah << 32));
    }
    memcpy(s, out, sizeof out);
}

static void full_rounds(uint64_t s[12], unsigned *round) { /* poseidon.rs:742-749 */
    for (int k = 0; k < 4; ++k) {
        constant_layer(s, *round);
        sbox_layer(s);
        mds_layer(s);
        ++*round;
    }
}

/* sum of up to 16 products x*y kept as two u128 halves (low / high 64-bit words summed separately), reduced once:
 * value = lo_sum + 2^64 * hi_sum, both sums < 2^68 */
typedef struct {
    u128 lo, hi;
} acc256;
static inline void acc_mul(acc256 *a, uint64_t x, uint64_t y) {
    u128 p = (u128)x * y;
    a->lo += (uint64_t)p;
    a->hi += (uint64_t)(p >> 64);
}
static inline uint64_t acc_reduce(const acc256 *a) {
    /* lo + 2^64*hi with lo, hi < 2^68: fold hi's top bits first (2^128 = -2^32 mod P is avoided by
     * reducing hi to 64 bits, then one reduce128 of the 128-bit value {lo64 + carry, hi64}) */
    uint64_t h = reduce128(a->hi);               /* hi mod P, < 2^64 */
    u128 t = (u128)h * EPS;                      /* 2^64 * h = h * EPS (mod P) */
    uint64_t l = reduce128(a->lo);
    return ora_gl_add(l, reduce128(t));
}

/* poseidon.rs:752-764 partial_rounds (fast): first-constant layer (:365-375), init matrix
 * (:415-441), then 22 x { sbox on s0; + scalar constant; sparse matrix (:516-542) } */
static void partial_rounds_fast(uint64_t s[12], unsigned *round) {
    for (int i = 0; i < 12; ++i) s[i] = ora_gl_add(s[i], P2_POSEIDON_FAST_PARTIAL_FIRST_ROUND_CONSTANT[i]);
    uint64_t t[12];
    t[0] = s[0];
    for (int c = 1; c < 12; ++c) {
        acc256 a = {0, 0};
        for (int r = 1; r < 12; ++r) acc_mul(&a, s[r], P2_POSEIDON_FAST_PARTIAL_ROUND_INITIAL_MATRIX[(r - 1) * 11 + (c - 1)]);
        t[c] = acc_reduce(&a);
    }
    memcpy(s, t, sizeof t);
    const uint64_t m00 = P2_POSEIDON_MDS_CIRC[0] + P2_POSEIDON_MDS_DIAG[0];
    for (int i = 0; i < 22; ++i) {
        s[0] = sbox(s[0]);
        s[0] = ora_gl_add(s[0], P2_POSEIDON_FAST_PARTIAL_ROUND_CONSTANTS[i]);
        acc256 a = {0, 0};
        acc_mul(&a, s[0], m00);
        for (int j = 1; j < 12; ++j) acc_mul(&a, s[j], P2_POSEIDON_FAST_PARTIAL_ROUND_W_HATS[i * 11 + j - 1]);
        for (int j = 1; j < 12; ++j) {  /* s_j + s_0 * v_j: one 128-bit multiply-add, one reduction */
            u128 p = (u128)s[0] * P2_POSEIDON_FAST_PARTIAL_ROUND_VS[i * 11 + j - 1];
            uint64_t plo = (uint64_t)p, phi = (uint64_t)(p >> 64);
            u128 q = (u128)plo + s[j];
            t[j] = reduce128(((u128)(phi + (uint64_t)(q >> 64)) << 64) | (uint64_t)q); /* phi < 2^64 - 1: no overflow */
        }
        t[0] = acc_reduce(&a);
        memcpy(s, t, sizeof t);
    }
    *round += 22;
}

static void partial_rounds_naive(uint64_t s[12], unsigned *round) { /* poseidon.rs:781-788 */
    for (int k = 0; k < 22; ++k) {
        constant_layer(s, *round);
        s[0] = sbox(s[0]);
        mds_layer(s);
        ++*round;
    }
}

void ora_poseidon(uint64_t s[12]) { /* poseidon.rs:767-777 */
    unsigned round = 0;
    full_rounds(s, &round);
    partial_rounds_fast(s, &round);
    full_rounds(s, &round);
    for (int i = 0; i < 12; ++i) s[i] = ora_gl_canon(s[i]);
}

void ora_poseidon_naive(uint64_t s[12]) { /* poseidon.rs:791-801 */
    unsigned round = 0;
    full_rounds(s, &round);
    partial_rounds_naive(s, &round);
    full_rounds(s, &round);
    for (int i = 0; i < 12; ++i) s[i] = ora_gl_canon(s[i]);
}

/* hashing.rs:118-145: zero state, overwrite-mode absorb of <=8-element chunks, no padding */
void ora_hash_no_pad(const uint64_t *in, size_t len, uint64_t out[4]) {
    uint64_t s[12] = {0};
    for (size_t off = 0; off < len; off += 8) {
        size_t c = len - off < 8 ? len - off : 8;
        for (size_t i = 0; i < c; ++i) s[i] = in[off + i];
        ora_poseidon(s);
    }
    /* len == 0: the reference squeezes the all-zero state without permuting */
    for (int i = 0; i < 4; ++i) out[i] = ora_gl_canon(s[i]);
}

/* plonk/config.rs:63-74: <= 4 elements are copied (canonical), not hashed */
void ora_hash_or_noop(const uint64_t *in, size_t len, uint64_t out[4]) {
    if (len * 8 <= 32) {
        for (size_t i = 0; i < 4; ++i) out[i] = i < len ? ora_gl_canon(in[i]) : 0;
    } else {
        ora_hash_no_pad(in, len, out);
    }
}

/* hashing.rs:97-114 compress */
void ora_two_to_one(const uint64_t l[4], const uint64_t r[4], uint64_t out[4]) {
    uint64_t s[12] = {l[0], l[1], l[2], l[3], r[0], r[1], r[2], r[3], 0, 0, 0, 0};
    ora_poseidon(s);
    for (int i = 0; i < 4; ++i) out[i] = s[i];
}

/* ------------------------------------------------------------------ merkle */
/* merkle_tree.rs:86-113 fill_subtree.  digests_buf has 2*(n_leaves-1) digests.  The leaf digests
 * (hash_or_noop of every leaf, the bulk of the work) are computed beforehand by a flat parallel loop,
 * which scales over many cores better than task recursion; the recursion below only combines. */
static void fill_subtree(uint64_t *digests_buf, size_t n_digests, const uint64_t *leaf_digests, size_t n_leaves,
                         uint64_t out[4]) {
    if (n_digests == 0) {
        memcpy(out, leaf_digests, 32);
        return;
    }
    size_t half = n_digests / 2;
    uint64_t *left_buf = digests_buf;                      /* [0, half-1) */
    uint64_t *left_digest = digests_buf + (half - 1) * 4;  /* split_last_mut */
    uint64_t *right_digest = digests_buf + half * 4;       /* split_first_mut */
    uint64_t *right_buf = digests_buf + (half + 1) * 4;
    uint64_t l[4], r[4];
    fill_subtree(left_buf, half - 1, leaf_digests, n_leaves / 2, l);
    fill_subtree(right_buf, half - 1, leaf_digests + (n_leaves / 2) * 4, n_leaves / 2, r);
    memcpy(left_digest, l, 32);
    memcpy(right_digest, r, 32);
    ora_two_to_one(l, r, out);
}

/* merkle_tree.rs:193-224 MerkleTree::new + :115-149 fill_digests_buf */
void ora_merkle_tree(const uint64_t *leaves, size_t n, size_t w, unsigned cap_height,
                     uint64_t *digests_out, uint64_t *cap_out) {
    size_t n_cap = (size_t)1 << cap_height;
    size_t n_digests = 2 * (n - n_cap);
    if (n_digests == 0) { /* all-cap tree, merkle_tree.rs:124-133 */
        for (size_t i = 0; i < n; ++i) ora_hash_or_noop(leaves + i * w, w, cap_out + 4 * i);
        return;
    }
    uint64_t *leaf_digests = (uint64_t *)malloc(n * 32);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) ora_hash_or_noop(leaves + i * w, w, leaf_digests + 4 * i);
    /* split every cap subtree further so that there are enough independent pieces for all cores:
     * pieces of `piece` leaves are filled in parallel, the few levels above them serially per subtree */
    size_t sub_digests = n_digests >> cap_height, sub_leaves = n >> cap_height;
#pragma omp parallel for schedule(dynamic)
    for (size_t s = 0; s < n_cap; ++s)
        fill_subtree(digests_out + s * sub_digests * 4, sub_digests, leaf_digests + s * sub_leaves * 4, sub_leaves,
                     cap_out + 4 * s);
    free(leaf_digests);
}

/* merkle_tree.rs:151-190 merkle_tree_prove */
void ora_merkle_prove(size_t leaf_index, size_t n, unsigned cap_height, const uint64_t *digests,
                      uint64_t *siblings_out) {
    unsigned num_layers = log2_strict(n) - cap_height;
    size_t digest_len = 2 * (n - ((size_t)1 << cap_height));
    size_t tree_index = leaf_index >> num_layers;
    size_t tree_len = digest_len >> cap_height;
    const uint64_t *tree = digests + tree_len * tree_index * 4;
    size_t pair_index = leaf_index & (((size_t)1 << num_layers) - 1);
    for (unsigned i = 0; i < num_layers; ++i) {
        size_t parity = pair_index & 1;
        pair_index >>= 1;
        size_t siblings_index = (pair_index << (i + 1)) + ((size_t)1 << i) - 1;
        size_t sibling_index = 2 * siblings_index + (1 - parity);
        memcpy(siblings_out + 4 * i, tree + 4 * sibling_index, 32);
    }
}

/* merkle_proofs.rs:55-108 verify_merkle_proof_to_cap */
int ora_merkle_verify(const uint64_t *leaf, size_t w, size_t leaf_index, const uint64_t *cap,
                      unsigned cap_height, const uint64_t *siblings, unsigned n_siblings) {
    (void)cap_height;
    uint64_t cur[4], nxt[4];
    size_t index = leaf_index;
    ora_hash_or_noop(leaf, w, cur);
    for (unsigned i = 0; i < n_siblings; ++i) {
        if (index & 1)
            ora_two_to_one(siblings + 4 * i, cur, nxt);
        else
            ora_two_to_one(cur, siblings + 4 * i, nxt);
        memcpy(cur, nxt, 32);
        index >>= 1;
    }
    for (int i = 0; i < 4; ++i)
        if (cur[i] != ora_gl_canon(cap[4 * index + i])) return 0;
    return 1;
}

/* ------------------------------------------------------------------ PolynomialBatch */
void ora_commit_timed(const uint64_t *cols, size_t W, unsigned lg_n, unsigned rate_bits,
                      unsigned cap_height, int is_values, uint64_t *coeffs_out,
                      uint64_t *leaves_out, uint64_t *digests_out, uint64_t *cap_out,
                      double seconds[4]) {
    size_t n = (size_t)1 << lg_n, N = n << rate_bits;
    double t0 = now_s();
    /* oracle.rs:65-69 "IFFT": par over columns */
    uint64_t *coeffs = coeffs_out ? coeffs_out : (uint64_t *)malloc(W * n * 8);
    memcpy(coeffs, cols, W * n * 8);
    if (is_values) {
#pragma omp parallel for schedule(dynamic)
        for (size_t c = 0; c < W; ++c) ora_ifft(coeffs + c * n, lg_n);
    }
    double t1 = now_s();
    /* oracle.rs:114-139 lde_values "FFT + blinding": lde = zero pad (polynomial/mod.rs:199-201),
     * coset_fft_with_options(coset_shift, Some(rate_bits)) */
    uint64_t *lde = (uint64_t *)malloc(W * N * 8);
#pragma omp parallel for schedule(dynamic)
    for (size_t c = 0; c < W; ++c) {
        uint64_t *col = lde + c * N;
        memcpy(col, coeffs + c * n, n * 8);
        memset(col + n, 0, (N - n) * 8);
        ora_coset_fft(col, lg_n + rate_bits, ORA_COSET_SHIFT, rate_bits);
    }
    double t2 = now_s();
    /* oracle.rs:97-98 transpose (plonky2/src/util/mod.rs:25-31, gather per output row) then
     * reverse_index_bits_in_place on the rows */
    unsigned lg_N = lg_n + rate_bits;
#pragma omp parallel for schedule(static)
    for (size_t L = 0; L < N; ++L) {
        size_t i = reverse_bits(L, lg_N);
        uint64_t *row = leaves_out + L * W;
        for (size_t c = 0; c < W; ++c) row[c] = lde[c * N + i];
    }
    double t3 = now_s();
    ora_merkle_tree(leaves_out, N, W, cap_height, digests_out, cap_out);
    double t4 = now_s();
    free(lde);
    if (!coeffs_out) free(coeffs);
    if (seconds) {
        seconds[0] += t1 - t0;
        seconds[1] += t2 - t1;
        seconds[2] += t3 - t2;
        seconds[3] += t4 - t3;
    }
}

void ora_commit(const uint64_t *cols, size_t W, unsigned lg_n, unsigned rate_bits,
                unsigned cap_height, int is_values, uint64_t *coeffs_out, uint64_t *leaves_out,
                uint64_t *digests_out, uint64_t *cap_out) {
    ora_commit_timed(cols, W, lg_n, rate_bits, cap_height, is_values, coeffs_out, leaves_out,
                     digests_out, cap_out, NULL);
}

/* ------------------------------------------------------------------ challenger */
void ora_challenger_init(ora_challenger *c) { memset(c, 0, sizeof *c); }

static void duplexing(ora_challenger *c) { /* challenger.rs:129-144 */
    for (uint32_t i = 0; i < c->n_in; ++i) c->state[i] = c->in[i];
    c->n_in = 0;
    ora_poseidon(c->state);
    memcpy(c->out, c->state, 8 * 8);
    c->n_out = 8;
}

void ora_challenger_observe(ora_challenger *c, const uint64_t *e, size_t n) { /* challenger.rs:39-48 */
    for (size_t i = 0; i < n; ++i) {
        c->n_out = 0;
        c->in[c->n_in++] = ora_gl_canon(e[i]);
        if (c->n_in == 8) duplexing(c);
    }
}

uint64_t ora_challenger_get(ora_challenger *c) { /* challenger.rs:82-92: pops from the back */
    if (c->n_in != 0 || c->n_out == 0) duplexing(c);
    return c->out[--c->n_out];
}

/* ------------------------------------------------------------------ FRI */
/* extension NTT with base-field twiddles == two interleaved base NTTs (extension/mod.rs:75-78) */
static void ext_coset_fft(uint64_t *v /*[n][2]*/, unsigned lg_n, uint64_t shift) {
    size_t n = (size_t)1 << lg_n;
    uint64_t *a = (uint64_t *)malloc(n * 8), *b = (uint64_t *)malloc(n * 8);
    for (size_t i = 0; i < n; ++i) {
        a[i] = v[2 * i];
        b[i] = v[2 * i + 1];
    }
#pragma omp parallel sections
    {
#pragma omp section
        ora_coset_fft(a, lg_n, shift, 0);
#pragma omp section
        ora_coset_fft(b, lg_n, shift, 0);
    }
    for (size_t i = 0; i < n; ++i) {
        v[2 * i] = a[i];
        v[2 * i + 1] = b[i];
    }
    free(a);
    free(b);
}

/* fri/prover.rs:84-150 fri_committed_trees */
void ora_fri_commit(const uint64_t *coeffs_in, unsigned lg_N, unsigned rate_bits,
                    unsigned cap_height, const unsigned *arity_bits, unsigned n_rounds,
                    ora_challenger *ch, uint64_t *leaves_out, uint64_t *digests_out,
                    uint64_t *caps_out, uint64_t *betas_out, uint64_t *final_out) {
    size_t m = (size_t)1 << lg_N;
    unsigned lg_m = lg_N;
    uint64_t *coeffs = (uint64_t *)malloc(m * 16), *values = (uint64_t *)malloc(m * 16);
    memcpy(coeffs, coeffs_in, m * 16);
    /* oracle.rs:215-220: lde_final_values = lde_final_poly.coset_fft(coset_shift) */
    memcpy(values, coeffs_in, m * 16);
    ext_coset_fft(values, lg_m, ORA_COSET_SHIFT);
    uint64_t shift = ORA_COSET_SHIFT;
    size_t n_cap = (size_t)1 << cap_height;
    for (unsigned rd = 0; rd < n_rounds; ++rd) {
        unsigned ab = arity_bits[rd];
        size_t arity = (size_t)1 << ab;
        ora_reverse_index_bits(values, m, 2);        /* prover.rs:98 */
        size_t n_leaves = m / arity, w = 2 * arity;  /* prover.rs:99-103 chunk + flatten */
        size_t n_dig = 2 * (n_leaves - n_cap);
        uint64_t *dig = digests_out ? digests_out : (uint64_t *)malloc((n_dig ? n_dig : 1) * 32);
        uint64_t capbuf[4 * 256];
        uint64_t *cap = caps_out ? caps_out : capbuf;
        ora_merkle_tree(values, n_leaves, w, cap_height, dig, cap); /* prover.rs:104 */
        if (leaves_out) {
            for (size_t i = 0; i < m * 2; ++i) leaves_out[i] = ora_gl_canon(values[i]);
            leaves_out += m * 2;
        }
        ora_challenger_observe(ch, cap, n_cap * 4); /* prover.rs:106 observe_cap */
        if (digests_out) digests_out += n_dig * 4; else free(dig);
        if (caps_out) caps_out += n_cap * 4;
        uint64_t beta[2];
        beta[0] = ora_challenger_get(ch); /* prover.rs:109 get_extension_challenge */
        beta[1] = ora_challenger_get(ch);
        if (betas_out) {
            betas_out[2 * rd] = beta[0];
            betas_out[2 * rd + 1] = beta[1];
        }
        /* prover.rs:111-117 + plonk_common.rs:120-132 reduce_with_powers (Horner from the back) */
        size_t m2 = m / arity;
#pragma omp parallel for schedule(static)
        for (size_t j = 0; j < m2; ++j) {
            uint64_t sum[2] = {0, 0}, t[2];
            for (size_t i = arity; i-- > 0;) {
                ora_ext2_mul(sum, beta, t);
                sum[0] = ora_gl_add(t[0], coeffs[2 * (arity * j + i)]);
                sum[1] = ora_gl_add(t[1], coeffs[2 * (arity * j + i) + 1]);
            }
            values[2 * j] = sum[0]; /* values is dead here; reuse as the new coeff buffer */
            values[2 * j + 1] = sum[1];
        }
        memcpy(coeffs, values, m2 * 16);
        m = m2;
        lg_m -= ab;
        shift = ora_gl_pow(shift, arity);            /* prover.rs:118 */
        ext_coset_fft(values, lg_m, shift);          /* prover.rs:119 (fresh NTT, no zero-tail) */
    }
    size_t n_final = m >> rate_bits;                 /* prover.rs:135-137 */
    for (size_t i = 0; i < 2 * n_final; ++i) coeffs[i] = ora_gl_canon(coeffs[i]);
    ora_challenger_observe(ch, coeffs, 2 * n_final); /* prover.rs:139 */
    if (final_out) memcpy(final_out, coeffs, n_final * 16);
    free(coeffs);
    free(values);
}

/* fri/prover.rs:153-202, deterministic (smallest witness) */
uint64_t ora_fri_pow(ora_challenger *ch, unsigned pow_bits) {
    uint64_t inter[12];
    memcpy(inter, ch->state, sizeof inter);
    for (uint32_t i = 0; i < ch->n_in; ++i) inter[i] = ch->in[i];
    uint32_t pos = ch->n_in;
    uint64_t found = 0;
    for (uint64_t cand = 0;; ++cand) {
        uint64_t s[12];
        memcpy(s, inter, sizeof s);
        s[pos] = cand;
        ora_poseidon(s);
        uint64_t resp = s[7]; /* squeeze().iter().last(): last of the RATE elements */
        unsigned lz = resp ? (unsigned)__builtin_clzll(resp) : 64;
        if (lz >= pow_bits) {
            found = cand;
            break;
        }
    }
    ora_challenger_observe(ch, &found, 1);
    (void)ora_challenger_get(ch);
    return found;
}

/* ------------------------------------------------------------------ prove_openings prelude */
void ora_reduce_polys_base(const uint64_t *const *polys, size_t n_polys, size_t n,
                           const uint64_t alpha[2], uint64_t *out) {
    memset(out, 0, n * 16);
    uint64_t pw[2] = {1, 0};
    for (size_t j = 0; j < n_polys; ++j) {
        for (size_t i = 0; i < n; ++i) { /* mul_extension: ext scalar times base coefficient */
            out[2 * i] = ora_gl_add(out[2 * i], ora_gl_mul(pw[0], polys[j][i]));
            out[2 * i + 1] = ora_gl_add(out[2 * i + 1], ora_gl_mul(pw[1], polys[j][i]));
        }
        uint64_t t[2];
        ora_ext2_mul(pw, alpha, t);
        pw[0] = t[0];
        pw[1] = t[1];
    }
}

void ora_divide_by_linear(const uint64_t *poly, size_t n, const uint64_t z[2], uint64_t *out) {
    uint64_t acc[2] = {0, 0}, t[2];
    /* bs[k] for k = n-1 .. 0; quotient coefficient j is bs at index j+1 */
    for (size_t k = n; k-- > 0;) {
        ora_ext2_mul(acc, z, t);
        acc[0] = ora_gl_add(t[0], poly[2 * k]);
        acc[1] = ora_gl_add(t[1], poly[2 * k + 1]);
        if (k >= 1) {
            out[2 * (k - 1)] = acc[0];
            out[2 * (k - 1) + 1] = acc[1];
        }
    }
    out[2 * (n - 1)] = 0;
    out[2 * (n - 1) + 1] = 0;
}

/* OpeningSet::new's eval_commitment (plonk/proof.rs:314-327): p.to_extension().eval(z) for every polynomial of a batch,
 * eval = coeffs.iter().rev().fold(ZERO, |acc, c| acc * x + c) (field/src/polynomial/mod.rs:155-160, :295-300).
 * polys: n_polys pointers to n base-field coefficients; out[n_polys][2], canonical. */
void ora_eval_polys_ext(const uint64_t *const *polys, size_t n_polys, size_t n, const uint64_t z[2], uint64_t *out) {
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t j = 0; j < n_polys; ++j) {
        uint64_t acc[2] = {0, 0}, t[2];
        for (size_t k = n; k-- > 0;) {
            ora_ext2_mul(acc, z, t);
            acc[0] = ora_gl_add(t[0], polys[j][k]);
            acc[1] = t[1];
        }
        out[2 * j] = ora_gl_canon(acc[0]);
        out[2 * j + 1] = ora_gl_canon(acc[1]);
    }
}

/* ---- the gate-independent part of compute_quotient_polys (plonk/prover.rs:609-815) with
 * eval_vanishing_poly_base_batch (plonk/vanishing_poly.rs:167-330): the permutation argument's vanishing terms on the quotient
 * coset, combined with the powers of alpha, divided by Z_H.  For every natural index i of the coset (size Nq = n << qbits,
 * qbits = log2_ceil(quotient_degree_factor); x = g * w_Nq^i, prover.rs:707):
 *   rows: get_lde_values(i, step) = leaves[reverse_bits(i * step, log_n + rate_bits)], step = 1 << (rate_bits - qbits)
 *         (oracle.rs:142-147, prover.rs:640); the "next" row i_next = (i + next_step) % Nq, next_step = 1 << qbits (:643, :708)
 *   terms, in the order of vanishing_poly.rs:326-330:
 *     for c in challenges:  L_0(x) * (Z_c(x) - 1),   L_0 = ZeroPolyOnCoset::eval_l_0 (field/src/zero_poly_coset.rs:58-61)
 *     for c in challenges:  check_partial_products(numerators, denominators, partials_c, Z_c(x), Z_c(g x), max_degree)
 *                           (util/partial_products.rs:52-79): prev_acc * prod(num chunk) - next_acc * prod(den chunk),
 *                           num_j = wire_j + beta_c * k_j * x + gamma_c, den_j = wire_j + beta_c * sigma_j(x) + gamma_c (:278-289)
 *     then the gate constraint terms -- circuit specific, OUT OF SCOPE here: `gate_sums[a][i]` (nullable = none) is the caller's
 *     reduce_with_powers of them alone, which enters as alpha_a^K * gate_sums (K = the number of permutation terms above)
 *   out[a][i] = reduce_with_powers_multi(terms, alphas)[a] * Z_H(x_i)^-1   (plonk_common.rs:99-115, prover.rs:797-803)
 * wires / cs / zs: the three commitments' leaf matrices, row-major [N][W_*], committed order; the sigma polynomials are columns
 * sigmas_first .. + num_routed of `cs`; `zs` holds Z_0..Z_{nc-1} then the partial products of challenge 0, 1, ... (prover.rs:224-229).
 * Returns -1 for an unsupported shape (quotient degree above the rate: the reference asserts, prover.rs:632-636). */
int ora_quotient_permutation(const uint64_t *wires, size_t W_w, const uint64_t *cs, size_t W_cs, size_t sigmas_first, const uint64_t *zs,
                             size_t W_z, unsigned log_n, unsigned rate_bits, const uint64_t *k_is, size_t num_routed, size_t qdf, unsigned nc,
                             const uint64_t *betas, const uint64_t *gammas, const uint64_t *alphas, const uint64_t *gate_sums, uint64_t *out) {
    unsigned qbits = 0;
    while (((size_t)1 << qbits) < qdf) ++qbits; /* log2_ceil */
    if (qbits > rate_bits || qdf < 2) return -1;
    const size_t n = (size_t)1 << log_n, Nq = n << qbits, step = (size_t)1 << (rate_bits - qbits), next_step = (size_t)1 << qbits;
    const unsigned log_N = log_n + rate_bits;
    const size_t num_chunks = (num_routed + qdf - 1) / qdf, num_prods = num_chunks - 1;
    if (W_z < nc * (1 + num_prods) || W_w < num_routed || W_cs < sigmas_first + num_routed) return -1;
    /* ZeroPolyOnCoset::new(n_log, qbits) (zero_poly_coset.rs:21-34) */
    const size_t rate = (size_t)1 << qbits;
    uint64_t zh[64], zh_inv[64];
    const uint64_t g_pow_n = ora_gl_pow(ORA_COSET_SHIFT, n), v = ora_gl_root_of_unity(qbits);
    for (size_t j = 0; j < rate; ++j) {
        zh[j] = ora_gl_sub(ora_gl_mul(g_pow_n, ora_gl_pow(v, j)), 1);
        zh_inv[j] = ora_gl_inv(zh[j]);
    }
    const uint64_t w = ora_gl_root_of_unity(log_n + qbits), n_f = (uint64_t)n % P;
    const size_t K = (size_t)nc + (size_t)nc * num_chunks;
    uint64_t *terms = (uint64_t *)malloc(sizeof(uint64_t) * (K ? K : 1));
    uint64_t xi = 1; /* points = two_adic_subgroup (prover.rs:645) */
    for (size_t i = 0; i < Nq; ++i) {
        const uint64_t x = ora_gl_mul(ORA_COSET_SHIFT, xi); /* shifted_x, prover.rs:707 */
        const size_t row = reverse_bits(i * step, log_N), row_next = reverse_bits(((i + next_step) % Nq) * step, log_N);
        const uint64_t *lw = wires + row * W_w, *lc = cs + row * W_cs, *lz = zs + row * W_z, *nz = zs + row_next * W_z;
        const uint64_t l0 = ora_gl_mul(zh[i % rate], ora_gl_inv(ora_gl_mul(n_f, ora_gl_sub(x, 1)))); /* eval_l_0 */
        size_t t = 0;
        for (unsigned c = 0; c < nc; ++c) terms[t++] = ora_gl_mul(l0, ora_gl_sub(lz[c], 1));
        for (unsigned c = 0; c < nc; ++c) {
            const uint64_t *partials = lz + nc + (size_t)c * num_prods;
            for (size_t ch = 0; ch < num_chunks; ++ch) {
                uint64_t pn = 1, pd = 1;
                for (size_t j = ch * qdf; j < (ch + 1) * qdf && j < num_routed; ++j) {
                    const uint64_t s_id = ora_gl_mul(k_is[j], x);
                    pn = ora_gl_mul(pn, ora_gl_add(ora_gl_add(lw[j], ora_gl_mul(betas[c], s_id)), gammas[c]));
                    pd = ora_gl_mul(pd, ora_gl_add(ora_gl_add(lw[j], ora_gl_mul(betas[c], lc[sigmas_first + j])), gammas[c]));
                }
                const uint64_t prev = ch == 0 ? lz[c] : partials[ch - 1], next = ch == num_chunks - 1 ? nz[c] : partials[ch];
                terms[t++] = ora_gl_sub(ora_gl_mul(prev, pn), ora_gl_mul(next, pd));
            }
        }
        for (unsigned a = 0; a < nc; ++a) {
            uint64_t cumul = gate_sums ? gate_sums[(size_t)a * Nq + i] : 0; /* the terms behind ours, already reduced */
            for (size_t k = K; k-- > 0;) cumul = ora_gl_add(terms[k], ora_gl_mul(cumul, alphas[a])); /* multiply_accumulate */
            out[(size_t)a * Nq + i] = ora_gl_canon(ora_gl_mul(cumul, zh_inv[i % rate]));
        }
        xi = ora_gl_mul(xi, w);
    }
    free(terms);
    return 0;
}
