/*
 * p2oracle -- CPU restatement of plonky2's LDE + Poseidon-commit + FRI-commit path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and
 * only as the checker / the timed CPU baseline.  The product path is the HIP
 * library behind include/p2hot.h and must fail loudly when that library is missing.
 *
 * Parity status: the reference is Rust (nightly) and cannot be compiled in the build
 * image (no cargo/rustc), so there is no oracle/_ref.  This restatement is pinned by
 *   - the reference's four Poseidon known-answer vectors
 *     (plonky2/src/hash/poseidon_goldilocks.rs:466-487),
 *   - the reference's bit-reversal table test (plonky2/src/util/mod.rs:61-126),
 *   - derivation of the fast-partial-round tables == reference tables
 *     (tools/gen_poseidon_constants.py),
 *   - the reference tests' own properties: NTT == naive evaluation, zero-tail option,
 *     coset-FFT == eval on the coset, iNTT o NTT == id, fast Poseidon == naive,
 *     every Merkle path verifies to the cap, FRI fold consistent with the verifier's
 *     interpolation (field/src/fft.rs:215-282, field/src/polynomial/mod.rs:477-516,
 *     plonky2/src/hash/merkle_tree.rs:253-311, plonky2/src/fri/verifier.rs:22-47).
 * NTT / LDE ordering / Merkle cap / FRI caps have no reference bytes to compare with:
 * "parity pinned by property + Poseidon KAT".
 *
 * All file:line citations are into /root/reference.
 */
#ifndef P2ORACLE_H
#define P2ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORA_P 0xFFFFFFFF00000001ULL
#define ORA_COSET_SHIFT 14293326489335486720ULL /* field/src/goldilocks_field.rs:80 (generator 7... see types.rs:441) */

/* ---- field (field/src/goldilocks_field.rs) ---- */
uint64_t ora_gl_canon(uint64_t x);
uint64_t ora_gl_add(uint64_t a, uint64_t b);
uint64_t ora_gl_sub(uint64_t a, uint64_t b);
uint64_t ora_gl_mul(uint64_t a, uint64_t b);
uint64_t ora_gl_pow(uint64_t a, uint64_t e);
uint64_t ora_gl_inv(uint64_t a);
uint64_t ora_gl_root_of_unity(unsigned log_n); /* field/src/types.rs:268-272 */
void ora_ext2_mul(const uint64_t a[2], const uint64_t b[2], uint64_t out[2]);

/* ---- util (util/src/lib.rs:53-62, :185-234) ---- */
void ora_reverse_index_bits(uint64_t *a, size_t n, size_t elem_words);

/* ---- fft (field/src/fft.rs) ---- */
/* fft_classic semantics: natural in / natural out, out[i] = sum a[t] w^(it); `r` = zero-tail option */
void ora_fft(uint64_t *vals, unsigned log_n, unsigned r);
void ora_ifft(uint64_t *vals, unsigned log_n);
/* PolynomialCoeffs::coset_fft_with_options (field/src/polynomial/mod.rs:280-293): in place */
void ora_coset_fft(uint64_t *coeffs, unsigned log_n, uint64_t shift, unsigned zero_factor);
/* PolynomialValues::coset_ifft (field/src/polynomial/mod.rs:63-73): in place */
void ora_coset_ifft(uint64_t *vals, unsigned log_n, uint64_t shift);

/* ---- poseidon (plonky2/src/hash/poseidon.rs, hashing.rs, plonk/config.rs) ---- */
void ora_poseidon(uint64_t s[12]);       /* poseidon.rs:767-777 (fast partial rounds) */
void ora_poseidon_naive(uint64_t s[12]); /* poseidon.rs:791-801 */
void ora_hash_no_pad(const uint64_t *in, size_t len, uint64_t out[4]);  /* hashing.rs:118-145 */
void ora_hash_or_noop(const uint64_t *in, size_t len, uint64_t out[4]); /* plonk/config.rs:63-74 */
void ora_two_to_one(const uint64_t l[4], const uint64_t r[4], uint64_t out[4]); /* hashing.rs:97-114 */

/* ---- merkle (plonky2/src/hash/merkle_tree.rs, merkle_proofs.rs) ---- */
/* leaves: n rows of w words, row-major.  digests_out: 2*(n - 2^cap_height) * 4 words in the
 * reference layout (merkle_tree.rs:50-57).  cap_out: 2^cap_height * 4 words. */
void ora_merkle_tree(const uint64_t *leaves, size_t n, size_t w, unsigned cap_height,
                     uint64_t *digests_out, uint64_t *cap_out);
/* merkle_tree_prove (merkle_tree.rs:151-190): writes log2(n)-cap_height siblings (4 words each) */
void ora_merkle_prove(size_t leaf_index, size_t n, unsigned cap_height, const uint64_t *digests,
                      uint64_t *siblings_out);
/* verify_merkle_proof_to_cap (merkle_proofs.rs:55-108): 1 = ok */
int ora_merkle_verify(const uint64_t *leaf, size_t w, size_t leaf_index, const uint64_t *cap,
                      unsigned cap_height, const uint64_t *siblings, unsigned n_siblings);

/* ---- PolynomialBatch (plonky2/src/fri/oracle.rs:57-139) ---- */
/* cols: W columns of n = 2^log_n words (column-major, each column contiguous, stride n).
 * is_values != 0: from_values (ifft first) else from_coeffs.
 * coeffs_out [W][n] (may be NULL), leaves_out [N][W] row-major, rows in bit-reversed order,
 * digests_out / cap_out as in ora_merkle_tree with N = n << rate_bits leaves. */
void ora_commit(const uint64_t *cols, size_t W, unsigned log_n, unsigned rate_bits,
                unsigned cap_height, int is_values, uint64_t *coeffs_out, uint64_t *leaves_out,
                uint64_t *digests_out, uint64_t *cap_out);
/* the same, split so that the CPU baseline can time the reference's TimingTree scopes
 * ("IFFT", "FFT + blinding", "transpose LDEs", "build Merkle tree"); seconds[4] accumulates. */
void ora_commit_timed(const uint64_t *cols, size_t W, unsigned log_n, unsigned rate_bits,
                      unsigned cap_height, int is_values, uint64_t *coeffs_out,
                      uint64_t *leaves_out, uint64_t *digests_out, uint64_t *cap_out,
                      double seconds[4]);

/* ---- Challenger (plonky2/src/iop/challenger.rs:16-153) ---- */
typedef struct {
    uint64_t state[12];
    uint64_t in[8];
    uint32_t n_in;
    uint64_t out[8];
    uint32_t n_out;
} ora_challenger;
void ora_challenger_init(ora_challenger *c);
void ora_challenger_observe(ora_challenger *c, const uint64_t *elems, size_t n);
uint64_t ora_challenger_get(ora_challenger *c);

/* ---- FRI commit phase (plonky2/src/fri/prover.rs:84-150) ---- */
/* coeffs: N = 2^log_N extension coefficients [N][2] (top N - N>>rate_bits are zero).
 * For round i (arity 2^arity_bits[i]) the tree has m_i/arity leaves of 2*arity words where
 * m_0 = N, m_{i+1} = m_i >> arity_bits[i].
 * leaves_out / digests_out / caps_out: concatenation over rounds (caller sizes them);
 * any may be NULL.  betas_out: [n_rounds][2].  final_out: [m_last >> rate_bits][2].
 * The challenger is advanced exactly like the reference (cap observed, beta drawn per round,
 * final coefficients observed). */
void ora_fri_commit(const uint64_t *coeffs, unsigned log_N, unsigned rate_bits,
                    unsigned cap_height, const unsigned *arity_bits, unsigned n_rounds,
                    ora_challenger *ch, uint64_t *leaves_out, uint64_t *digests_out,
                    uint64_t *caps_out, uint64_t *betas_out, uint64_t *final_out);
/* fri_proof_of_work (fri/prover.rs:153-202), made deterministic: the SMALLEST valid witness.
 * Advances the challenger like the reference (observe witness, draw response). */
uint64_t ora_fri_pow(ora_challenger *ch, unsigned pow_bits);

/* ---- "next" rows: prove_openings prelude ---- */
/* reduce_polys_base (util/reducing.rs:83-95): out[n][2] = sum_j alpha^j * polys[j] (base polys) */
void ora_reduce_polys_base(const uint64_t *const *polys, size_t n_polys, size_t n,
                           const uint64_t alpha[2], uint64_t *out);
/* divide_by_linear (field/src/polynomial/division.rs:79-92) on an extension polynomial of n coeffs:
 * quotient has n-1 coeffs; out[n][2] gets them plus a trailing zero ("pad back to power of two"). */
void ora_divide_by_linear(const uint64_t *poly, size_t n, const uint64_t z[2], uint64_t *out);
/* plonk/proof.rs:314-327 eval_commitment: every polynomial (n base coefficients) at the extension point z; out[n_polys][2] */
void ora_eval_polys_ext(const uint64_t *const *polys, size_t n_polys, size_t n, const uint64_t z[2], uint64_t *out);

/* ---- wires_permutation_partial_products_and_zs (plonk/prover.rs:392-449) for one (beta, gamma).
 * wires / sigmas: [num_routed][n] column-major (sigmas = the sigma polynomials' values on the subgroup);
 * out: [num_prods + 1][n], partial products then Z (the reference's transposed Vec order, :444-447). */
int ora_partial_products(const uint64_t *wires, const uint64_t *sigmas, const uint64_t *k_is, size_t num_routed,
                         unsigned log_n, size_t degree, uint64_t beta, uint64_t gamma, uint64_t *out);

/* ---- the gate-independent part of compute_quotient_polys (plonk/prover.rs:609-815) + the permutation terms of
 * eval_vanishing_poly_base_batch (plonk/vanishing_poly.rs:167-330); see the definition for the layout.  out [nc][n << qbits]. */
int ora_quotient_permutation(const uint64_t *wires, size_t W_w, const uint64_t *cs, size_t W_cs, size_t sigmas_first, const uint64_t *zs,
                             size_t W_z, unsigned log_n, unsigned rate_bits, const uint64_t *k_is, size_t num_routed, size_t qdf, unsigned nc,
                             const uint64_t *betas, const uint64_t *gammas, const uint64_t *alphas, const uint64_t *gate_sums, uint64_t *out);

int ora_num_threads(void);
void ora_set_num_threads(int n); /* e.g. the cgroup CPU quota of the job */

#ifdef __cplusplus
}
#endif
#endif
