/*
 * p2hot.h -- C ABI of libp2hot: the MI355X (gfx950) implementation of plonky2's
 * PolynomialBatch LDE + Poseidon-Merkle commit pipeline and FRI commit phase.
 *
 * This is the drop-in boundary a patched `plonky2` crate binds with `extern "C"` (see
 * INTEGRATION.md).  The reference has no plugin trait for this path; each entry point cites the
 * reference function whose body it replaces (paths relative to the plonky2 repository).
 *
 * Conventions
 *  - Field elements are uint64_t (GoldilocksField is #[repr(transparent)] u64,
 *    field/src/goldilocks_field.rs:23-25).  Inputs may be any representative < 2^64; every value
 *    written to an output buffer is canonical (< P = 2^64 - 2^32 + 1).
 *  - Extension elements (F^2) are two consecutive words [a0, a1] (field/src/extension/quadratic.rs:13).
 *  - Digests are 4 words (hash/hash_types.rs:25-27).
 *  - Functions return 0 (P2HOT_OK) or a P2HOT_E* code; p2hot_last_error() has the text.  No C++
 *    exception, abort or library-owned host memory crosses this boundary.
 *  - `*_dev` functions take DEVICE pointers and only enqueue work on the context's HIP stream;
 *    the others take HOST pointers, stage through device memory owned by the context and return
 *    after the results are in the caller's buffers.
 *  - A context is bound to one GPU and runs one call at a time; use one context per GPU (one process per GPU with
 *    p2hot_comm_*, or one process for all GPUs with p2hot_group_*, see the multi-GPU section).
 */
#ifndef P2HOT_H
#define P2HOT_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define P2HOT_OK 0
#define P2HOT_EINVAL 1       /* bad shape / argument (the reference would panic!, e.g. fft.rs:171, merkle_tree.rs:195) */
#define P2HOT_ENOMEM 2       /* device allocation failed */
#define P2HOT_EHIP 3         /* HIP runtime error */
#define P2HOT_EUNSUPPORTED 4 /* valid in the reference but not on this path */

#define P2HOT_P 0xFFFFFFFF00000001ULL
#define P2HOT_COSET_SHIFT 14293326489335486720ULL /* F::coset_shift(), field/src/goldilocks_field.rs:80 */

typedef struct p2hot_ctx p2hot_ctx;

/* ---------------------------------------------------------------- context */
/* hip_stream: the hipStream_t all work of this context is enqueued on; NULL = the legacy default
 * stream (which is also PyTorch's default stream).  The caller keeps the stream alive. */
int p2hot_ctx_create(int device, void *hip_stream, p2hot_ctx **out);
void p2hot_ctx_destroy(p2hot_ctx *ctx);
int p2hot_ctx_set_stream(p2hot_ctx *ctx, void *hip_stream);
int p2hot_ctx_sync(p2hot_ctx *ctx);
const char *p2hot_last_error(const p2hot_ctx *ctx);
const char *p2hot_version(void);
/* 1 when the library was built by the test-only kernel emulator (tests/emu), 0 for the HIP build */
int p2hot_is_emulated(void);

/* Live per-kernel timing: when enabled, the launches of each kernel family are bracketed with HIP
 * events on the context's stream.  p2hot_profile_json synchronises, drains them and returns
 * {"kernel": {"ms": total, "launches": count}, ...} (valid until the next call); reset != 0 clears
 * the totals.  This is the TimingTree analogue (plonky2/src/util/timing.rs) for the GPU stages. */
int p2hot_profile_enable(p2hot_ctx *ctx, int on);
/* tuning knob for the NTT pass kernels (0 = LDS radix-2 layers, 3 = register radix 8 on carry-free 24-bit limbs
 * [default; 4096-element tiles, other tiles as 8], 8 = register radix 8 on 64-bit words, 4 = radix 16);
 * results are identical, only the speed differs.  ABI note: until round 2 `3` meant the 64-bit-word radix-8 kernels, which are
 * `8` now; P2HOT_NTT_LIMB=0 in the environment is sticky (then `3` selects the word kernels as before). */
int p2hot_tune_ntt(p2hot_ctx *ctx, int radix_bits);
/* overlap the Poseidon leaf sponge of coset block b with the LDE of block b+1 on a second HIP stream (default 0: measured neutral on MI355X) */
int p2hot_tune_overlap(p2hot_ctx *ctx, int on);
/* leaf-hash / tree-level launches with at most max_perms permutations run the quad-cooperative kernels (4 lanes per
 * permutation, ~3x lower latency, 1.3x the work); default 2^15, 0 = never.  Results are identical. */
int p2hot_tune_quad(p2hot_ctx *ctx, size_t max_perms);
/* ... and launches with at most max_perms permutations the word-per-lane kernels (16 lanes per permutation, one state word
 * and one MDS row per lane through DPP row broadcasts: ~3x lower latency than the quad kernels, ~3.6x the work); default 2^13, 0 = never.
 * The Fiat-Shamir sponge always runs this mapping.  Results are identical. */
int p2hot_tune_row(p2hot_ctx *ctx, size_t max_perms);
const char *p2hot_profile_json(p2hot_ctx *ctx, int reset);

/* sizes: number of digests (4 words each) in MerkleTree::digests for n_leaves = 2^log_leaves
 * (hash/merkle_tree.rs:203: 2 * (n_leaves - 2^cap_height)) */
size_t p2hot_num_digests(unsigned log_leaves, unsigned cap_height);

/* ---------------------------------------------------------------- primitives (device pointers) */
/* fft_with_options(input, None, None) for `batch` polynomials, in place, natural order in and out:
 * out[i] = sum_t in[t] * w_n^(i*t)  (field/src/fft.rs:53-65).  poly_stride >= n elements. */
int p2hot_fft_dev(p2hot_ctx *ctx, uint64_t *d_data, size_t batch, size_t poly_stride, unsigned log_n);
/* ifft_with_options (field/src/fft.rs:68-91): values on H_n -> coefficients, in place. */
int p2hot_ifft_dev(p2hot_ctx *ctx, uint64_t *d_data, size_t batch, size_t poly_stride, unsigned log_n);
/* PolynomialValues::coset_ifft (field/src/polynomial/mod.rs:63-73): values on shift * H_n -> coefficients, in place
 * (the quotient-side transform of plonk/prover.rs:810-814) */
int p2hot_coset_ifft_dev(p2hot_ctx *ctx, uint64_t *d_data, size_t batch, size_t poly_stride, unsigned log_n,
                         uint64_t shift);
/* PolynomialBatch::lde_values (fri/oracle.rs:114-139) = lde(rate_bits) + coset_fft(shift) for W
 * polynomials, fused with the transpose + reverse_index_bits of oracle.rs:97-98:
 *   d_lde[c * lde_stride + (L - row_begin)] = p_c(shift * w_N^bitrev(L)),  L in [row_begin, row_begin + row_count)
 * where N = n << rate_bits.  Rows are produced in whole coset blocks of n rows: row_begin and
 * row_count must be multiples of n (row block b holds coset j = bitrev_rb(b)). */
int p2hot_coset_lde_dev(p2hot_ctx *ctx, const uint64_t *d_coeffs, size_t W, size_t coeff_stride, unsigned log_n,
                        unsigned rate_bits, uint64_t shift, size_t row_begin, size_t row_count, uint64_t *d_lde,
                        size_t lde_stride);
/* transpose (plonky2/src/util/mod.rs:25-31): column-major [W][rows] -> row-major [rows][W], canonical */
int p2hot_transpose_dev(p2hot_ctx *ctx, const uint64_t *d_colmajor, size_t col_stride, size_t W, size_t rows,
                        uint64_t *d_rowmajor);
/* reverse_index_bits (util/src/lib.rs:53-62) on `batch` arrays of 2^log_n words, out of place */
int p2hot_reverse_index_bits_dev(p2hot_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, size_t batch,
                                 size_t poly_stride, unsigned log_n);
/* Poseidon::poseidon on `count` states of 12 words, in place (hash/poseidon.rs:767-777) */
int p2hot_poseidon_permute_dev(p2hot_ctx *ctx, uint64_t *d_states, size_t count);
/* MerkleTree::new (hash/merkle_tree.rs:193-224) for the leaf range [leaf_begin, leaf_begin + leaf_count)
 * of a tree with 2^log_leaves leaves; the range must be a whole number of cap subtrees.
 * layout: 0 = leaves column-major (element (L, c) at d_leaves[c * leaf_stride + L - leaf_begin]),
 *         1 = leaves row-major    (d_leaves[(L - leaf_begin) * W + c]).
 * d_digests / d_cap point at the FULL tree's arrays (reference layout, merkle_tree.rs:50-57);
 * only the entries of the given range are written. */
int p2hot_merkle_dev(p2hot_ctx *ctx, const uint64_t *d_leaves, int layout, size_t leaf_stride, size_t W,
                     unsigned log_leaves, unsigned cap_height, size_t leaf_begin, size_t leaf_count,
                     uint64_t *d_digests, uint64_t *d_cap);
/* Field-arithmetic self test (goldilocks_field.rs:245-320, :402-415): for count operand pairs writes six arrays of
 * `count` words to d_out: a*b by the compiler-scheduled multiply, by the hand-scheduled single stream, by the 3-way
 * interleaved stream, a+b, a-b (all canonical), and a word that is 0 unless a hand-written instruction stream disagreed
 * with the compiler-scheduled arithmetic (bit 0 the other two mul3 lanes, 1 the low-register multiply, 2 the
 * power-of-two twiddle multiplies, 3 / 4 the MDS row recombinations fold1 / fold3, 5 / 6 the two-stream mul2 / fold2). */
int p2hot_field_selftest_dev(p2hot_ctx *ctx, const uint64_t *d_a, const uint64_t *d_b, size_t count, uint64_t *d_out);
/* rows of a column-major matrix of `rows` rows: d_out[q][c] = d_colmajor[c * col_stride + d_idx[q]]
 * (PolynomialBatch::get_lde_values / MerkleTree::get, fri/oracle.rs:142-147, merkle_tree.rs:227).
 * The indices are device-resident, so they cannot be validated before the launch: an index >= rows reads nothing,
 * its output row is zero, and the next p2hot_ctx_sync returns P2HOT_EINVAL (the reference panics on the slice index). */
int p2hot_gather_rows_dev(p2hot_ctx *ctx, const uint64_t *d_colmajor, size_t col_stride, size_t rows, size_t W,
                          const uint64_t *d_idx, size_t m, uint64_t *d_out);

/* ---------------------------------------------------------------- PolynomialBatch (device pointers) */
/* PolynomialBatch::from_values (is_values != 0, fri/oracle.rs:57-79) / from_coeffs (:82-112) with
 * blinding = false, restricted to the leaf rows [row_begin, row_begin + row_count) (whole coset
 * blocks and whole cap subtrees; pass 0, N for the full commitment).
 *   d_cols    [W][n] column-major input (values on H_n, or coefficients), stride col_stride
 *   d_coeffs  [W][n] out: the coefficient form (`polynomials`); may alias d_cols when
 *             is_values == 0; NULL only when is_values == 0
 *   d_lde     [W][row_count] out: column-major LDE, rows in committed (bit-reversed) order,
 *             stride lde_stride -- the device-resident form of MerkleTree::leaves
 *   d_leaves  [row_count][W] out, row-major copy (the reference's `leaves`), or NULL
 *   d_digests, d_cap: FULL-tree arrays (see p2hot_merkle_dev) */
int p2hot_commit_dev(p2hot_ctx *ctx, const uint64_t *d_cols, size_t col_stride, size_t W, unsigned log_n,
                     unsigned rate_bits, unsigned cap_height, int is_values, size_t row_begin, size_t row_count,
                     uint64_t *d_coeffs, size_t coeff_stride, uint64_t *d_lde, size_t lde_stride,
                     uint64_t *d_leaves, uint64_t *d_digests, uint64_t *d_cap);

/* ---------------------------------------------------------------- Challenger (device-resident) */
/* plonky2/src/iop/challenger.rs:16-153.  The sponge lives on the GPU so FRI rounds need no host
 * round trip; the host (or the Rust shim) moves its state in and out with load/store. */
typedef struct {
    uint64_t sponge_state[12];
    uint64_t input_buffer[8];
    uint64_t output_buffer[8];
    uint32_t input_len, output_len;
} p2hot_challenger_state;
typedef struct p2hot_challenger p2hot_challenger;
int p2hot_challenger_create(p2hot_ctx *ctx, p2hot_challenger **out); /* Challenger::new */
void p2hot_challenger_destroy(p2hot_challenger *ch);
int p2hot_challenger_load(p2hot_challenger *ch, const p2hot_challenger_state *host_state);
int p2hot_challenger_store(p2hot_challenger *ch, p2hot_challenger_state *host_state);
/* observe_elements (challenger.rs:50-54) then get_n_challenges (:93-95); host pointers; either count may be 0 */
int p2hot_challenger_step(p2hot_challenger *ch, const uint64_t *observe, size_t n_observe, uint64_t *challenges,
                          size_t n_challenges);

/* ---------------------------------------------------------------- FRI commit phase */
/* fri_committed_trees (fri/prover.rs:84-150) including the `lde_final_values` coset FFT of
 * prove_openings (fri/oracle.rs:215-220).  Host pointers.
 *   coeffs        [n][2]: the n = 2^log_n nonzero extension coefficients of final_poly (the
 *                 reference passes them zero-padded to N = n << rate_bits; the padding is implicit here)
 *   arity_bits    reduction_arity_bits (fri/reduction_strategies.rs:31-59), n_rounds entries
 *   challenger    advanced exactly like the reference: per round observe_cap then
 *                 get_extension_challenge; finally observe the final coefficients
 *   outputs (any may be NULL), concatenated over rounds r with m_r = N >> sum(arity_bits[0..r)):
 *     leaves_out   m_r * 2 words per round: tree r's leaves (m_r/arity rows of 2*arity words)
 *     digests_out  4 * p2hot_num_digests(log m_r - arity_bits[r], cap_height) words per round
 *     caps_out     4 << cap_height words per round
 *     betas_out    2 words per round
 *     final_out    [(m_last >> rate_bits)][2], m_last = N >> sum(arity_bits) */
int p2hot_fri_commit(p2hot_ctx *ctx, const uint64_t *coeffs, unsigned log_n, unsigned rate_bits,
                     unsigned cap_height, const unsigned *arity_bits, unsigned n_rounds,
                     unsigned max_num_query_steps, size_t final_poly_coeff_len,
                     p2hot_challenger *challenger, uint64_t *leaves_out, uint64_t *digests_out,
                     uint64_t *caps_out, uint64_t *betas_out, uint64_t *final_out);
/* max_num_query_steps / final_poly_coeff_len: the two Option<usize> arguments of fri_committed_trees
 * (fri/prover.rs:89-90, used by starky's multi-degree recursion: dummy cap observations + challenges up to
 * max_num_query_steps, :122-132, zero observations up to final_poly_coeff_len, :140-147); 0 = None. */
/* the same with the coefficients already on the device as two planes [2][n] (component 0, then component 1),
 * e.g. the output of p2hot_fri_final_poly_dev.  d_leaves_out is a DEVICE buffer (or NULL): the round trees' leaf
 * matrices stay on the GPU (same concatenated layout) and the query phase gathers the few rows it opens.
 * digests_out is a DEVICE buffer when digests_on_device != 0 (the round trees' digest arrays stay on the GPU next
 * to their leaves; Merkle paths then come from p2hot_merkle_paths_dev), else a host buffer; the other outputs are
 * host pointers as above. */
int p2hot_fri_commit_dev(p2hot_ctx *ctx, const uint64_t *d_coeffs_planar, unsigned log_n, unsigned rate_bits,
                         unsigned cap_height, const unsigned *arity_bits, unsigned n_rounds,
                         unsigned max_num_query_steps, size_t final_poly_coeff_len,
                         p2hot_challenger *challenger, uint64_t *d_leaves_out, uint64_t *digests_out,
                         int digests_on_device, uint64_t *caps_out, uint64_t *betas_out, uint64_t *final_out);
/* The final_poly construction of PolynomialBatch::prove_openings (fri/oracle.rs:186-213) on device-resident
 * coefficient polynomials: for every batch i with opening point z_i,
 *   F_i = ReducingFactor::reduce_polys_base(polys of the batch)      (util/reducing.rs:83-95)
 *   q_i = F_i.divide_by_linear(z_i), padded with a zero                 (field/src/polynomial/division.rs:79-92)
 *   final_poly = final_poly * alpha^(#polys of batch i) + q_i          (shift_poly, reducing.rs:103-106)
 * d_poly_table: DEVICE array of device pointers, each to n = 2^log_n base-field coefficients; batch i uses
 * entries [batch_offsets[i], batch_offsets[i+1]).  points [n_batches][2], alpha [2]: host.  d_final: planes [2][n]. */
int p2hot_fri_final_poly_dev(p2hot_ctx *ctx, const uint64_t *const *d_poly_table, const size_t *batch_offsets,
                             size_t n_batches, const uint64_t *points, const uint64_t alpha[2], unsigned log_n,
                             uint64_t *d_final);
/* OpeningSet::new (plonky2/src/plonk/proof.rs:314-327): evaluate n_polys device-resident coefficient
 * polynomials (d_poly_table: DEVICE array of device pointers, 2^log_n words each) at n_points extension points
 * (host [n_points][2]): d_out[p][j] = polys[j](points[p]) as [2] words, layout [n_points][n_polys][2]. */
int p2hot_eval_polys_dev(p2hot_ctx *ctx, const uint64_t *const *d_poly_table, size_t n_polys, unsigned log_n,
                         const uint64_t *points, size_t n_points, uint64_t *d_out);
/* all_wires_permutation_partial_products (plonky2/src/plonk/prover.rs:356-449): the permutation
 * argument's Z and partial-product polynomials (values on the trace subgroup) for num_challenges (beta, gamma) pairs.
 *   d_wires   DEVICE [num_routed][n] column-major (MatrixWitness.wire_values[col][row], iop/witness.rs:283-291)
 *   d_sigmas  DEVICE [num_routed][n] column-major: sigma_j(w_n^i) (the sigma polynomials' values; prover_data.sigmas
 *             holds their transpose)
 *   k_is      HOST [num_routed] coset shifts (common_data.k_is), betas / gammas HOST [num_challenges]
 *   degree    = common_data.quotient_degree_factor (chunk size); num_prods = ceil(num_routed / degree) - 1
 *   d_out     DEVICE [num_challenges * (num_prods + 1)][n], ordered as the prover batches them for the commit
 *             (prover.rs:224-229): Z of challenge 0 .. nc-1, then the num_prods partial products of challenge 0, 1, ...
 * Canonical outputs; feed d_out to p2hot_commit_dev(is_values = 1).  A zero denominator (the reference panics,
 * "Tried to invert zero") returns P2HOT_EINVAL. */
int p2hot_partial_products_dev(p2hot_ctx *ctx, const uint64_t *d_wires, size_t wires_stride, const uint64_t *d_sigmas,
                               size_t sigmas_stride, const uint64_t *k_is, unsigned num_routed, unsigned log_n,
                               unsigned degree, const uint64_t *betas, const uint64_t *gammas, unsigned num_challenges,
                               uint64_t *d_out, size_t out_stride);
/* merkle_tree_prove (hash/merkle_tree.rs:151-190) for m leaf indices from a device-resident digest array
 * (the query phase, fri/prover.rs:204-258): d_out [m][log_leaves - cap_height][4].  A leaf index
 * >= 2^log_leaves yields a zero path and P2HOT_EINVAL at the next p2hot_ctx_sync (see p2hot_gather_rows_dev). */
int p2hot_merkle_paths_dev(p2hot_ctx *ctx, const uint64_t *d_digests, unsigned log_leaves, unsigned cap_height,
                           const uint64_t *d_idx, size_t m, uint64_t *d_out);
/* fri_proof_of_work (fri/prover.rs:153-202), deterministic: returns the SMALLEST valid witness
 * (the reference's rayon find_any returns an arbitrary valid one), observes it and draws the
 * response like the reference does. */
int p2hot_fri_pow(p2hot_ctx *ctx, p2hot_challenger *challenger, unsigned pow_bits, uint64_t *witness_out);

/* ================================================================ prover session (HOST pointers)
 * What the patched plonky2 crate calls from the prover's main thread with ordinary Vec<F> buffers.  Everything between
 * the calls stays on the GPU behind opaque handles:
 *   p2hot_batch       a device-resident PolynomialBatch (fri/oracle.rs:30-37): polynomials (coefficients), the LDE
 *                     matrix = merkle_tree.leaves, merkle_tree.digests; optionally the input values
 *   p2hot_cols        a device-resident Vec<PolynomialValues> / Vec<PolynomialCoeffs> ([W][n]) that never visits the host
 *   p2hot_challenger  the Fiat-Shamir transcript (above)
 * A context runs ONE host-pointer call at a time; a second thread entering gets P2HOT_EBUSY (plonky2 calls these
 * from the main thread, outside its rayon closures).  Serialised by the library (busy guard): p2hot_commit*, p2hot_cols_upload,
 * p2hot_batch_coeffs / _rows / _paths / _digests / _subgroup_values, p2hot_eval_openings, p2hot_prove_openings, p2hot_partial_products,
 * p2hot_quotient_chunks, p2hot_quotient_polys, p2hot_ctx_trim.  p2hot_batch_free / p2hot_cols_free may be called from any thread at any time (a
 * Drop, a finaliser): the block cache has its own lock.  Everything else -- the *_dev building blocks, p2hot_fri_commit,
 * p2hot_fri_pow, p2hot_challenger_* -- enqueues on the context's stream without a guard: the CALLER serialises those with
 * each other and with the host-pointer calls of the same context (the Rust shim holds the context behind a Mutex). */
#define P2HOT_EBUSY 5        /* another host-pointer call is running on this context */
#define P2HOT_ECOMM 6        /* collective (RCCL / caller-supplied transport) failure in the multi-GPU mode */
#define P2HOT_KEEP_VALUES 1u /* from_values: keep the input values on the device (p2hot_batch_values) */
/* p2hot_commit / _salted / _cols: `coeffs_out` is a TABLE of W host pointers (cast from `uint64_t *const *`), polynomial c's
 * n coefficients go to table[c] -- the caller's `polynomials: Vec<PolynomialCoeffs<F>>` (fri/oracle.rs:32) is W separate
 * vectors, and with this flag each is filled in place instead of being split out of one flat block afterwards */
#define P2HOT_COEFFS_PER_COLUMN 2u
/* p2hot_commit / _salted with leaves_out AND handle_out: the call returns when the cap, the coefficients and the digests are back;
 * the row-major leaf matrix (9.1 GB at the C3 wires shape: 3 x the rest of the call over PCIe) keeps travelling into leaves_out on
 * the context's leaf-copy stream, in min(64, N / 1024) consecutive row blocks (p2hot_batch_leaves_block_rows), first block first.  A row may be READ only after
 * p2hot_batch_leaves_wait(handle, row_lo, row_hi) covered it; the buffer may be FREED only after p2hot_batch_free(handle) (which
 * waits for the copy) or a wait over all rows.  leaves_out should be pinned (p2hot_host_alloc): a pageable destination makes the
 * copy a synchronous one.  The next commitments, the partial products and the quotient run beside the copy. */
#define P2HOT_LEAVES_ASYNC 4u
/* leaves_out in NATURAL LDE order instead of the committed order: leaves_out[i] = merkle_tree.leaves[reverse_bits(i, log2 N)], i.e.
 * `transpose(lde_values)` without the `reverse_index_bits_in_place` of fri/oracle.rs:97-98.  get_lde_values(i, step) (oracle.rs:142-147)
 * is then row i * step of the buffer, so the quotient loop of plonk/prover.rs:712-722 walks the buffer FORWARD -- the order the
 * P2HOT_LEAVES_ASYNC blocks arrive in -- and MerkleTree::get(r) is row reverse_bits(r).  p2hot_batch_rows / _paths and the proofs keep
 * the committed indexing: only the host copy's row order changes. */
#define P2HOT_LEAVES_NATURAL 8u

typedef struct p2hot_batch p2hot_batch;
typedef struct p2hot_cols p2hot_cols;

/* from_values (is_values != 0, fri/oracle.rs:57-79) / from_coeffs (:82-112), blinding = false.
 * cols: W host pointers to n = 2^log_n words each (Vec<PolynomialValues<F>> / Vec<PolynomialCoeffs<F>>).
 * coeffs_out [W][n], leaves_out [N][W], digests_out, cap_out: caller-allocated or NULL (anything not asked for is not
 * copied back: the leaf matrix is 9 GB at the C3 shape, and the query phase needs only a few dozen rows and paths).
 * handle_out (optional): the device-resident batch for p2hot_batch_rows / _paths / _coeffs, p2hot_eval_openings and
 * p2hot_prove_openings; free with p2hot_batch_free.  flags: P2HOT_KEEP_VALUES, P2HOT_COEFFS_PER_COLUMN, P2HOT_LEAVES_ASYNC,
 * P2HOT_LEAVES_NATURAL.
 * W = 0 is P2HOT_EINVAL for every commit entry point (the reference panics on polynomials[0], fri/oracle.rs:90). */
int p2hot_commit(p2hot_ctx *ctx, const uint64_t *const *cols, size_t W, unsigned log_n, unsigned rate_bits,
                 unsigned cap_height, int is_values, unsigned flags, uint64_t *coeffs_out, uint64_t *leaves_out,
                 uint64_t *digests_out, uint64_t *cap_out, p2hot_batch **handle_out);
/* the same with blinding = true (fri/oracle.rs:123-137; standard_recursion_zk_config, circuit_data.rs:135-137): the
 * reference appends SALT_SIZE = 4 random LDE-value vectors of length N to the leaves.  The random numbers stay the
 * caller's (Rust draws them with F::rand_vec / OsRng and hands them over): salt_cols = n_salt host pointers to N words
 * each, in the order F::rand_vec produced them (natural index, like the LDE values before transpose +
 * reverse_index_bits, oracle.rs:97-98).  Leaves are W + n_salt wide: leaves_out [N][W + n_salt], p2hot_batch_rows and
 * the opened rows of p2hot_prove_openings carry the salts (MerkleTree::get does; get_lde_values strips them, :146);
 * p2hot_batch_width stays W (the polynomials), p2hot_batch_leaf_width is W + n_salt.  n_salt = 0 is p2hot_commit. */
int p2hot_commit_salted(p2hot_ctx *ctx, const uint64_t *const *cols, size_t W, unsigned log_n, unsigned rate_bits,
                        unsigned cap_height, int is_values, unsigned flags, const uint64_t *const *salt_cols, size_t n_salt,
                        uint64_t *coeffs_out, uint64_t *leaves_out, uint64_t *digests_out, uint64_t *cap_out,
                        p2hot_batch **handle_out);
/* the same on a device-resident column set (the output of p2hot_partial_products / p2hot_quotient_chunks, or an upload).
 * CONSUMES `cols` -- its block becomes the batch's coefficients or kept values, or is released -- on success and on failure,
 * except for the failures detected before the set is touched: P2HOT_EBUSY and EVERY P2HOT_EINVAL (a null set, a set of
 * another context, a borrowed view (p2hot_batch_values), an empty set, a bad rate / cap height / flag word: all arguments are
 * validated first).  After those two codes the caller still owns the handle. */
int p2hot_commit_cols(p2hot_ctx *ctx, p2hot_cols *cols, unsigned rate_bits, unsigned cap_height, int is_values,
                      unsigned flags, uint64_t *coeffs_out, uint64_t *leaves_out, uint64_t *digests_out, uint64_t *cap_out,
                      p2hot_batch **handle_out);
/* M commitments of ONE shape in one set of launches (recursion workloads: the 2^12-row proofs of bench_recursion's chain,
 * examples/bench_recursion.rs:317-345, are latency-bound one at a time).  M is a power of two, W * M <= 65535.
 * Results equal M separate p2hot_commit calls (fri/oracle.rs:57-112 per proof), bit for bit.
 * cols[m * W + e]: column e of proof m (n words).  coeffs_out [M][W][n], digests_out [M][p2hot_num_digests][4],
 * caps_out [M][2^cap_height][4], each optional.  handles_out [M] (optional): one p2hot_batch per proof (rows, paths,
 * coefficients, p2hot_eval_openings, p2hot_prove_openings); the proofs share their device blocks, which return to the
 * context's cache when the last of the M handles is freed. */
int p2hot_commit_many(p2hot_ctx *ctx, const uint64_t *const *cols, size_t M, size_t W, unsigned log_n, unsigned rate_bits,
                      unsigned cap_height, int is_values, uint64_t *coeffs_out, uint64_t *digests_out, uint64_t *caps_out,
                      p2hot_batch **handles_out);
/* the device-pointer form: d_cols [W][M][n] (column e of every proof, then column e + 1, ...; becomes the coefficients in
 * place), d_lde [W][M][N] in the same interleaving, d_digests [M][num_digests][4], d_cap [M][2^cap_height][4]. */
int p2hot_commit_many_dev(p2hot_ctx *ctx, uint64_t *d_cols, size_t M, size_t W, unsigned log_n, unsigned rate_bits,
                          unsigned cap_height, int is_values, uint64_t *d_lde, uint64_t *d_digests, uint64_t *d_cap);
/* a handle over device buffers the CALLER owns (the outputs of p2hot_commit_dev): nothing is copied, p2hot_batch_free
 * frees only the handle.  d_coeffs [W][n] stride n, d_lde [W][N] stride N, d_digests the full tree's array. */
int p2hot_batch_wrap_dev(p2hot_ctx *ctx, const uint64_t *d_coeffs, const uint64_t *d_lde, const uint64_t *d_digests, size_t W,
                         unsigned log_n, unsigned rate_bits, unsigned cap_height, p2hot_batch **out);
size_t p2hot_batch_width(const p2hot_batch *batch);      /* polynomials */
size_t p2hot_batch_leaf_width(const p2hot_batch *batch); /* words per leaf: polynomials + salt columns */
unsigned p2hot_batch_degree_log(const p2hot_batch *batch);
/* `polynomials[first .. first + count)` (fri/oracle.rs:32), canonical: out [count][n] */
int p2hot_batch_coeffs(p2hot_batch *batch, size_t first, size_t count, uint64_t *out);
/* MerkleTree::get for m leaf indices (merkle_tree.rs:227): out [m][p2hot_batch_leaf_width] */
int p2hot_batch_rows(p2hot_batch *batch, const uint64_t *row_idx, size_t m, uint64_t *out);
/* merkle_tree_prove (merkle_tree.rs:151-190) for m leaf indices from the batch's device-resident digests:
 * out [m][log2(N) - cap_height][4].  With it the caller may pass digests_out = NULL to p2hot_commit and never copy
 * the digest array (0.54 GB at the C3 shape) to the host. */
int p2hot_batch_paths(p2hot_batch *batch, const uint64_t *leaf_idx, size_t m, uint64_t *out);
/* merkle_tree.digests (reference layout, hash/merkle_tree.rs:50-57): out [p2hot_num_digests(log2 N, cap_height)][4] */
int p2hot_batch_digests(p2hot_batch *batch, uint64_t *out);
/* P2HOT_LEAVES_ASYNC: blocks until rows [row_lo, row_hi) of the caller's leaves_out (in ITS row order: natural with
 * P2HOT_LEAVES_NATURAL, committed otherwise) have landed.  A batch without a pending copy returns at once; row_hi beyond the
 * leaf count or row_lo > row_hi is EINVAL.  This is the fence behind MerkleTree::get (hash/merkle_tree.rs:227) in the Rust shim. */
int p2hot_batch_leaves_wait(p2hot_batch *batch, size_t row_lo, size_t row_hi);
/* rows per block of the pending copy: min(64, N / 1024) blocks (at least one), delivered in order -- once row r has landed, so has
 * every row below (r / block_rows + 1) * block_rows, which is what a reader walking forward raises its "landed" mark to instead of
 * asking once per row (integration/p2hot.rs DeviceTree::fence).  0 when the batch has no copy in flight.  Lock-free like the wait,
 * and like it never writes p2hot_last_error: both may run beside a locked call of the same context. */
size_t p2hot_batch_leaves_block_rows(const p2hot_batch *batch);
/* the kept input values (P2HOT_KEEP_VALUES) as a BORROWED column set: valid while the batch lives; p2hot_cols_free on
 * the view leaves the batch's memory alone */
int p2hot_batch_values(p2hot_batch *batch, p2hot_cols **out);
/* polynomials [first, first + count) of the batch as values on the subgroup H_n, as an OWNED column set (free with
 * p2hot_cols_free): a forward NTT (field/src/fft.rs:53-65) of a copy of their device-resident coefficients.  For the sigma range
 * of the constants_sigmas commitment this is `prover_data.sigmas` (plonk/prover.rs:413) column-major: the `sigmas` input of
 * p2hot_partial_products, with no host transpose and no upload.  count == 0 or a range beyond the batch is EINVAL. */
int p2hot_batch_subgroup_values(p2hot_batch *batch, size_t first, size_t count, p2hot_cols **out);
/* returns the batch's device blocks to its context's block cache; call it BEFORE p2hot_ctx_destroy of that context
 * (the host-pointer entry points keep their device blocks in a per-context cache: a fresh allocation of the 9 GB LDE
 * matrix costs up to a second; p2hot_ctx_trim gives the cached free blocks back to the driver) */
void p2hot_batch_free(p2hot_batch *batch);
/* frees what the context only keeps for speed: the cached free device and pinned blocks (its helpers' too) and the per-size
 * L_0 denominator tables of p2hot_quotient_polys (8 bytes per point of the quotient coset; rebuilt on demand) */
int p2hot_ctx_trim(p2hot_ctx *ctx);
/* PINNED host memory from a grow-only cache of the context, for output buffers that live as long as a commitment -- above all the
 * flat leaf matrix (`leaves_out`, 9 GB at the C3 shape) the Rust shim keeps behind MerkleTree::get (hash/merkle_tree.rs:227).
 * A device-to-host copy into pinned memory runs at the PCIe rate and touches no fresh page; a fresh pageable allocation of that
 * size pays a page fault per 4 KiB while the copy runs, and a fresh PIN of that size costs seconds -- hence the cache: a block
 * released with p2hot_host_free (any thread, any time: a Drop) is handed out again by the next p2hot_host_alloc of a similar
 * size; p2hot_ctx_trim and p2hot_ctx_destroy give the cached blocks back to the OS.  Free every block before destroying the context. */
int p2hot_host_alloc(p2hot_ctx *ctx, size_t bytes, void **out);
void p2hot_host_free(p2hot_ctx *ctx, void *p);

/* device-resident column sets */
int p2hot_cols_upload(p2hot_ctx *ctx, const uint64_t *const *cols, size_t W, unsigned log_n, p2hot_cols **out);
int p2hot_cols_download(p2hot_cols *cols, size_t first, size_t count, uint64_t *out /* [count][n] */);
size_t p2hot_cols_width(const p2hot_cols *cols);
unsigned p2hot_cols_degree_log(const p2hot_cols *cols);
void p2hot_cols_free(p2hot_cols *cols);

/* OpeningSet::new (plonky2/src/plonk/proof.rs:314-327; starky/src/proof.rs StarkOpeningSet): every polynomial of every
 * listed batch at each extension point.  points [n_points][2].  out: for batch b in order, [n_points][W_b][2], concatenated. */
int p2hot_eval_openings(p2hot_ctx *ctx, const p2hot_batch *const *batches, size_t n_batches, const uint64_t *points,
                        size_t n_points, uint64_t *out);

/* FriBatchInfo (fri/structure.rs): an opening point and the polynomials opened there as (oracle_index, polynomial_index) */
typedef struct {
    uint64_t point[2];
    const uint32_t *oracle_index;
    const uint32_t *poly_index;
    size_t n_polys;
} p2hot_fri_batch_info;
/* FriParams / FriConfig (fri/mod.rs:19-60, :62-104) + the two Option<usize> of prove_openings (0 = None) */
typedef struct {
    unsigned rate_bits, cap_height, proof_of_work_bits, num_query_rounds;
    const unsigned *reduction_arity_bits;
    unsigned n_reduction_rounds;
    int hiding;                     /* FriParams::hiding: carried for the caller's transcript (fri/mod.rs:148); the prover's FRI path does not depend on it */
    unsigned max_num_query_steps;   /* fri/prover.rs:90 */
    size_t final_poly_coeff_len;    /* fri/prover.rs:89 */
} p2hot_fri_params;
/* FriProof (fri/proof.rs:95-110) as flat caller-allocated buffers; Q = num_query_rounds, R = rounds, m_0 = N,
 * m_{r+1} = m_r >> arity_bits[r]:
 *   commit_phase_merkle_caps [R][2^cap_height][4]
 *   final_poly               [m_R >> rate_bits][2]
 *   pow_witness
 *   query_indices            [Q] x_index of every query round (optional, may be NULL; the verifier re-derives them)
 *   query_round_proofs[q].initial_trees_proof.evals_proofs[o] = (leaf, path):
 *     initial_leaves         [Q][sum_o leaf_width_o]   the opened row (salts included) of every oracle, oracles in order
 *     initial_paths          [Q][n_oracles][log2(N) - cap_height][4]
 *   query_round_proofs[q].steps[r] = (evals, merkle_proof):
 *     step_evals             [Q][sum_r 2 * 2^arity_bits[r]]       evals of round r = 2^arity_bits[r] extension elements
 *     step_paths             [Q][sum_r (log2(m_r) - arity_bits[r] - cap_height) * 4] */
typedef struct {
    uint64_t *commit_phase_merkle_caps;
    uint64_t *final_poly;
    uint64_t pow_witness;
    uint64_t *query_indices;
    uint64_t *initial_leaves;
    uint64_t *initial_paths;
    uint64_t *step_evals;
    uint64_t *step_paths;
} p2hot_fri_proof;
typedef struct {
    size_t caps_words, final_poly_words, initial_leaves_words, initial_paths_words, step_evals_words, step_paths_words;
} p2hot_fri_proof_layout;
int p2hot_fri_proof_sizes(const p2hot_batch *const *oracles, size_t n_oracles, const p2hot_fri_params *params,
                          p2hot_fri_proof_layout *out);
/* PolynomialBatch::prove_openings (fri/oracle.rs:176-237) + fri_proof (fri/prover.rs:24-82): alpha, final_poly =
 * sum_i alpha^(k_i) (F_i - F_i(z_i)) / (X - z_i), its LDE, the commit phase (fri_committed_trees, :84-150), the
 * proof-of-work grind (:153-202, smallest witness) and the query rounds (:204-258), with the transcript advanced exactly
 * like the reference.  All oracles must share degree, rate and cap height. */
int p2hot_prove_openings(p2hot_ctx *ctx, const p2hot_fri_batch_info *batches, size_t n_batches,
                         const p2hot_batch *const *oracles, size_t n_oracles, p2hot_challenger *challenger,
                         const p2hot_fri_params *params, p2hot_fri_proof *proof);
/* M independent opening proofs side by side (recursion workloads: many 2^12-row proofs, each a latency-bound chain of small
 * launches).  Proof j opens oracles[j * n_oracles .. + n_oracles) (handles of THIS context, e.g. from p2hot_commit_many) at
 * batches[j][0 .. n_batches[j]) with transcript challengers[j] and fills proofs[j]; every proof is computed exactly as by
 * p2hot_prove_openings (same buffers, same transcript afterwards).  Inside, up to 4 sibling contexts of the same GPU --
 * own streams, driven by their own host threads -- run the proofs concurrently. */
int p2hot_prove_openings_many(p2hot_ctx *ctx, size_t M, const p2hot_fri_batch_info *const *batches, const size_t *n_batches,
                              const p2hot_batch *const *oracles, size_t n_oracles, p2hot_challenger *const *challengers,
                              const p2hot_fri_params *fp, p2hot_fri_proof *proofs);

/* all_wires_permutation_partial_products (plonk/prover.rs:356-449) on device-resident columns: wires = columns
 * [wires_first_col, +num_routed) of `wires` (e.g. p2hot_batch_values of the wires commitment), sigmas likewise (the
 * constants_sigmas commitment's values: sigma_j(w_n^i)).  k_is HOST [num_routed], betas / gammas HOST [num_challenges],
 * degree = quotient_degree_factor.  Output rows ordered as the prover commits them (prover.rs:224-229): Z of challenge
 * 0 .. nc-1, then the partial products of challenge 0, 1, ...: out_host [nc * (num_prods + 1)][n] and / or out_cols
 * (feed it to p2hot_commit_cols(is_values = 1)); either may be NULL. */
int p2hot_partial_products(p2hot_ctx *ctx, const p2hot_cols *wires, size_t wires_first_col, const p2hot_cols *sigmas,
                           size_t sigmas_first_col, const uint64_t *k_is, unsigned num_routed, unsigned degree,
                           const uint64_t *betas, const uint64_t *gammas, unsigned num_challenges, uint64_t *out_host,
                           p2hot_cols **out_cols);
/* The gate-independent tail of compute_quotient_polys (plonk/prover.rs:274-289, :810-815): quotient_values[c] are the
 * values of challenge c's quotient on the coset g*H of size n << ceil(log2(quotient_degree_factor)) (natural order);
 * coset_ifft, trim to quotient_degree_factor * n coefficients (P2HOT_EINVAL "Quotient has failed ..." if the tail is
 * not zero, where the reference panics) and split into chunks of n: chunks_out = [nc * quotient_degree_factor][n]
 * coefficients, ready for p2hot_commit_cols(is_values = 0). */
int p2hot_quotient_chunks(p2hot_ctx *ctx, const uint64_t *const *quotient_values, unsigned num_challenges,
                          unsigned degree_bits, unsigned quotient_degree_factor, p2hot_cols **chunks_out);
/* compute_quotient_polys (plonk/prover.rs:609-815) WITHOUT its gate evaluation: the permutation argument's share of
 * eval_vanishing_poly_base_batch (plonk/vanishing_poly.rs:167-330) evaluated on the quotient coset from the device-resident LDE
 * matrices of the three commitments -- for every x = g * w^i of the coset of size n << log2_ceil(quotient_degree_factor):
 *   terms  L_0(x) (Z_c(x) - 1)  for every challenge c, then check_partial_products (util/partial_products.rs:52-79) per challenge:
 *          prev_acc * prod(wire_j + beta_c k_j x + gamma_c) - next_acc * prod(wire_j + beta_c sigma_j(x) + gamma_c) per chunk of
 *          quotient_degree_factor routed wires, accumulators Z_c(x), the partial products, Z_c(g x);
 *   value_a(x) = (reduce_with_powers(terms, alpha_a) + alpha_a^K * gate_sums[a][i]) / Z_H(x),  K = the number of terms above
 * -- the gate constraint terms (circuit specific, out of scope) are the CALLER's: gate_sums[a] = their own reduce_with_powers
 * by alpha_a on the same coset (n << qbits words, natural order), or gate_sums = NULL for none --, followed by the tail of
 * p2hot_quotient_chunks (coset_ifft, the divisibility check, chunks of n coefficients).
 *   wires, constants_sigmas, zs_partial_products: commitments of THIS context with the same degree and rate; the routed wires are
 *     columns 0 .. num_routed of `wires`, the sigma polynomials columns sigmas_first_col .. of `constants_sigmas`
 *     (common_data.sigmas_range()), `zs_partial_products` holds Z_0 .. Z_{nc-1} then the partial products of challenge 0, 1, ...
 *     (prover.rs:224-229: the order p2hot_partial_products emits)
 *   k_is HOST [num_routed]; betas / gammas / alphas HOST [num_challenges], 1 <= num_challenges <= 4
 *   values_out (optional) HOST [num_challenges][n << qbits]: the quotient VALUES on the coset (prover.rs:805: quotient_values)
 *   chunks_out (optional): [num_challenges * quotient_degree_factor][n] coefficients for p2hot_commit_cols(is_values = 0)
 * P2HOT_EINVAL "Quotient has failed ..." when the result is not a polynomial of the expected degree (the reference panics). */
int p2hot_quotient_polys(p2hot_ctx *ctx, const p2hot_batch *wires, const p2hot_batch *constants_sigmas, size_t sigmas_first_col,
                         const p2hot_batch *zs_partial_products, const uint64_t *k_is, unsigned num_routed,
                         unsigned quotient_degree_factor, const uint64_t *betas, const uint64_t *gammas, const uint64_t *alphas,
                         unsigned num_challenges, const uint64_t *const *gate_sums, uint64_t *values_out, p2hot_cols **chunks_out);

/* ================================================================ multi-GPU: the coset-sharded commit
 * The rate-1/B LDE is B independent coset transforms and coset j is the contiguous row block bitrev(j) of the
 * committed order, so rank r of G owns rows [r*N/G, (r+1)*N/G) = whole cosets = whole cap subtrees: it runs the LDE,
 * the leaf sponge and the Merkle levels of its rows with no data-path exchange.  Exchanges: the coefficients after the
 * column-sharded iNTT (W*n*8 bytes, pipelined in column chunks beside the transforms), the 2^cap_height cap entries,
 * and -- on request -- the digest slices.  G <= 2^cap_height, G a power of two.  G > 2^rate_bits (starky's rate-1/2 traces on 4 or
 * 8 GPUs): the cosets are split into sub-cosets of H_n -- rank r folds every polynomial mod x^n' - c_r (n' = N/G rows, c_r = the
 * n'-th power of its block's coset shift) and runs the same LDE at (log n', log(N/n')); same exchange, same results.
 * Two ways to run it:
 *   one process per GPU   p2hot_comm_create_rccl (RCCL over xGMI; the launcher distributes the unique id) or
 *                         p2hot_comm_create_callback (the host application's own all-gather), then p2hot_commit_sharded_dev
 *   one process, N GPUs   p2hot_group_create + p2hot_group_commit: host pointers in, like p2hot_commit -- a patched
 *                         plonky2 gets every GPU of the node from its main thread
 * RCCL is bound at run time (dlopen), libp2hot.so has no link-time dependency on it. */
#define P2HOT_UNIQUE_ID_BYTES 128 /* NCCL_UNIQUE_ID_BYTES, rccl.h:40 */
typedef struct p2hot_comm p2hot_comm;
typedef struct p2hot_group p2hot_group;
typedef struct p2hot_sharded_batch p2hot_sharded_batch;
/* caller-supplied transport: rank r's slice is `bytes` bytes at d_base + offsets[r] (valid on rank r); on return every
 * slice must be complete on this rank.  Called on the host with the context's stream idle; return 0 on success. */
typedef int (*p2hot_allgather_fn)(void *user, void *d_base, const size_t *offsets, int world, size_t bytes, void *hip_stream);
int p2hot_comm_unique_id(uint8_t out[P2HOT_UNIQUE_ID_BYTES]);               /* ncclGetUniqueId on one rank */
int p2hot_comm_create_rccl(p2hot_ctx *ctx, int rank, int world, const uint8_t id[P2HOT_UNIQUE_ID_BYTES], p2hot_comm **out);
int p2hot_comm_create_callback(p2hot_ctx *ctx, int rank, int world, p2hot_allgather_fn fn, void *user, p2hot_comm **out);
void p2hot_comm_destroy(p2hot_comm *comm);
int p2hot_comm_rank(const p2hot_comm *comm);
int p2hot_comm_world(const p2hot_comm *comm);
/* preflight (collective: every rank calls it): a `bytes`-sized pattern slice per rank is all-gathered the way the commit
 * exchanges coefficients and caps, and every rank checks every slice; P2HOT_ECOMM with a named cause on failure */
int p2hot_comm_selftest(p2hot_comm *comm, size_t bytes);
/* How equal, contiguous slices (whole-rank coefficient slices, cap, digest slices; the pipelined column chunks through a
 * chunk-major staging block) travel on an RCCL transport: 0 = one grouped ncclBroadcast per slice, 1 = ncclAllGather.  On an
 * RCCL communicator p2hot_comm_selftest checks and times both and keeps the faster (rank 0's verdict, broadcast to all);
 * p2hot_group_create does the same over its ranks; P2HOT_EXCHANGE=broadcast|allgather pins one.  -1 for a null handle. */
int p2hot_comm_exchange_mode(const p2hot_comm *comm);
/* Which RCCL this process is bound to: the library file (dladdr of ncclGetUniqueId) into path_out (NUL-terminated, truncated to
 * path_cap) and ncclGetVersion's code (e.g. 22203) into version_out; either may be NULL.  The library prefers the copy the
 * process already loaded (PyTorch's, under torch.distributed.run) and falls back to /opt/rocm/lib/librccl.so.1 by path (a patched
 * plonky2, no torch): two deployment modes, two RCCL builds -- a run reports which one it used.  P2HOT_ECOMM if none binds. */
int p2hot_rccl_info(char *path_out, size_t path_cap, int *version_out);
/* columns [first, first + count) of W are the ones rank `rank` of `world` transforms in the iNTT stage */
int p2hot_shard_columns(size_t W, int world, int rank, size_t *first, size_t *count);
/* from_values / from_coeffs (fri/oracle.rs:57-112) of this rank's share, DEVICE pointers, asynchronous:
 *   d_cols_local  [count][n] this rank's columns (p2hot_shard_columns), stride col_stride
 *   d_coeffs_all  [world * ceil(W/world)][n] out: row c = polynomial c for c < W (`polynomials`, complete on every rank)
 *   d_lde         [W][N/world] out: this rank's rows of the LDE matrix, column-major, stride lde_stride
 *   d_leaves      [N/world][W] out or NULL
 *   d_digests, d_cap: FULL-tree arrays; this rank's digest slice [rank * nd/world, +nd/world) and the whole cap are
 *                 valid on return, the other digest slices only with gather_digests != 0 (a Merkle path below the
 *                 cap never leaves the cap subtree of its leaf: the owner of a row can always serve its path)
 *   pipeline_chunks: column chunks of the coefficient exchange (0 / 1 = one blocking exchange; 8 is a good default) */
int p2hot_commit_sharded_dev(p2hot_ctx *ctx, p2hot_comm *comm, const uint64_t *d_cols_local, size_t col_stride, size_t W,
                             unsigned log_n, unsigned rate_bits, unsigned cap_height, int is_values, int gather_digests,
                             unsigned pipeline_chunks, uint64_t *d_coeffs_all, uint64_t *d_lde, size_t lde_stride,
                             uint64_t *d_leaves, uint64_t *d_digests, uint64_t *d_cap);
/* one process, n_gpus devices (devices == NULL: 0 .. n_gpus-1), one context + stream per device.  Distinct devices use
 * RCCL (ncclCommInitAll); a repeated device id (one-GPU test boxes) or P2HOT_GROUP_PEER_COPY=1 uses peer copies. */
int p2hot_group_create(int n_gpus, const int *devices, p2hot_group **out);
void p2hot_group_destroy(p2hot_group *group);
int p2hot_group_size(const p2hot_group *group);
p2hot_ctx *p2hot_group_ctx(p2hot_group *group, int i);
int p2hot_group_uses_rccl(const p2hot_group *group);
int p2hot_group_exchange_mode(const p2hot_group *group); /* see p2hot_comm_exchange_mode */
const char *p2hot_group_last_error(const p2hot_group *group);
/* from_values / from_coeffs over every GPU of the group, HOST pointers (the multi-GPU p2hot_commit): each GPU is sent
 * only the columns it transforms; coeffs_out / leaves_out / digests_out / cap_out (any may be NULL) are assembled from
 * the owning ranks.  handle_out: the sharded batch for p2hot_sharded_batch_open. */
int p2hot_group_commit(p2hot_group *group, const uint64_t *const *cols, size_t W, unsigned log_n, unsigned rate_bits,
                       unsigned cap_height, int is_values, int shard_mode, unsigned pipeline_chunks, uint64_t *coeffs_out,
                       uint64_t *leaves_out, uint64_t *digests_out, uint64_t *cap_out, p2hot_sharded_batch **handle_out);
/* shard_mode: P2HOT_SHARD_COSETS (default design, above) or P2HOT_SHARD_COLUMNS -- the fallback of SURVEY 8e's last row:
 * each rank runs the iNTT and the WHOLE LDE of its ceil(W/G) columns, then the LDE matrix is re-partitioned to row blocks
 * by an all-to-all of strided peer copies (the leaf sponge chains across the columns of a row) before hashing.  It moves
 * W*N*8 bytes instead of W*n*8 (2^rate_bits times more); kept as a second, independently derived partition (the coset mode
 * serves G > 2^rate_bits by sub-cosets).  Same results bit for bit. */
#define P2HOT_SHARD_COSETS 0
#define P2HOT_SHARD_COLUMNS 1
/* MerkleTree::get + merkle_tree_prove (merkle_tree.rs:227, :151-190) for m leaves, each answered by the rank that owns
 * the row: rows_out [m][W], paths_out [m][log2(N) - cap_height][4]; either may be NULL */
int p2hot_sharded_batch_open(p2hot_sharded_batch *batch, const uint64_t *leaf_idx, size_t m, uint64_t *rows_out, uint64_t *paths_out);
void p2hot_sharded_batch_free(p2hot_sharded_batch *batch);
/* OpeningSet::new, p2hot_fri_proof_sizes and PolynomialBatch::prove_openings + fri_proof (see the single-GPU entry points
 * above) over sharded oracles of one group, coset mode only (every rank then holds all coefficients): rank 0 runs what
 * needs the polynomials -- the evaluations, final_poly, the FRI commit phase (tiny after round 0), the grind --
 * and the rows and Merkle paths of the initial trees come from the ranks that own them.  `challenger` belongs to
 * p2hot_group_ctx(group, 0).  Same proof, bit for bit, as the single-GPU path on the same polynomials. */
int p2hot_group_eval_openings(p2hot_group *group, const p2hot_sharded_batch *const *oracles, size_t n_oracles, const uint64_t *points,
                              size_t n_points, uint64_t *out);
int p2hot_group_fri_proof_sizes(const p2hot_sharded_batch *const *oracles, size_t n_oracles, const p2hot_fri_params *params,
                                p2hot_fri_proof_layout *out);
int p2hot_group_prove_openings(p2hot_group *group, const p2hot_fri_batch_info *batches, size_t n_batches,
                               const p2hot_sharded_batch *const *oracles, size_t n_oracles, p2hot_challenger *challenger,
                               const p2hot_fri_params *params, p2hot_fri_proof *proof);

#ifdef __cplusplus
}
#endif
#endif
