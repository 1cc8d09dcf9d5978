//! `crate::p2hot` -- the plonky2 side of libp2hot (include/p2hot.h): the MI355X implementation of the
//! PolynomialBatch LDE + Poseidon-Merkle commit pipeline, the FRI commit phase and `prove_openings`.
//!
//! This file is added to the crate as `plonky2/src/p2hot.rs` by `integration/plonky2_p2hot.patch`
//! (feature `p2hot`).  With the feature on and `F = GoldilocksField`, `C::Hasher = PoseidonHash`, `D = 2`
//! (blinded commitments included: the salts are drawn here with `F::rand_vec` and handed to the library), the bodies of
//!   `PolynomialBatch::from_values` / `from_coeffs`   (fri/oracle.rs:57-112)
//!   `PolynomialBatch::prove_openings`                 (fri/oracle.rs:176-237)
//!   `fri_committed_trees`                             (fri/prover.rs:84-150)
//!   `MerkleTree::get` / `::prove`                     (hash/merkle_tree.rs:227, :231-237)
//!   `all_wires_permutation_partial_products`          (plonk/prover.rs:356-390)
//!   `compute_quotient_polys` (its permutation terms, the division by Z_H and the coset_ifft; plonk/prover.rs:609-815)
//!   `OpeningSet::new`'s `eval_commitment`             (plonk/proof.rs:323-328)
//! call into the library; every other instantiation takes the unchanged CPU path.  Signatures, struct fields
//! and results are unchanged, so `CircuitBuilder::build`, `prove`, starky's `prove`, serialization and the
//! verifier are untouched.
//!
//! Environment:
//!   P2HOT_LIB_DIR   directory of libp2hot.so (build.rs adds it to the link search path)
//!   P2HOT_DEVICE    GPU index of the single-GPU context (default 0)
//!   P2HOT_LEAVES    "host" (default): the leaf matrix of every commitment is copied back ONCE, as one flat row-major
//!                   buffer owned by the tree's `DeviceTree`; `merkle_tree.leaves` stays EMPTY and `MerkleTree::get` (which
//!                   `get_lde_values`, the CPU quotient evaluation and the serializer go through) returns
//!                   `&flat[i * w..(i + 1) * w]` -- no per-row allocation, no second copy (the round-3 shim rebuilt
//!                   `Vec<Vec<F>>` serially: 8.4 M allocations + 9 GB per wires commitment at 2^20 gates);
//!                   "vec": additionally materialises `merkle_tree.leaves: Vec<Vec<F>>` (in parallel, `par_chunks_exact`)
//!                   for code that indexes the field directly; "device": nothing comes back, rows and Merkle paths are
//!                   fetched from the GPU on demand (`MerkleTree::get` / `::prove`) -- 9 GB less PCIe traffic per wires commit.
//!
//!   P2HOT_DISABLE   "1": every call takes the unchanged CPU body (the same switch as `set_enabled(false)`)
//!
//! Bit-exact harness: `#[cfg(test)] mod tests` at the end of this file runs every replaced body twice in one process --
//! CPU (`set_enabled(false)`) and GPU -- on the same inputs and `assert_eq!`s polynomials, trees, FRI caps, final_poly
//! and `proof.to_bytes()`:
//!     P2HOT_LIB_DIR=/path/to/plonky2_amd cargo test --release --features p2hot p2hot:: -- --test-threads=1
//! `examples/p2hot_dump_goldens.rs` (added by the same patch) writes the CPU prover's commitments of the synthetic
//! inputs in the schema of tests/golden/commit_caps.json, so the oracle of the GPU repository can be pinned against
//! the real reference on any box with cargo (no GPU needed for that step).
//!
//! The build image of the GPU repository has no Rust toolchain: this file is checked against include/p2hot.h symbol by
//! symbol (tests/test_integration_files.py) but has not been compiled there.
#![allow(non_camel_case_types, clippy::missing_safety_doc, clippy::too_many_arguments)]

use core::any::type_name;
use core::ffi::{c_char, c_int, c_uint, c_void};
use core::sync::atomic::{AtomicBool, AtomicUsize, Ordering};
use std::collections::HashMap;
use std::sync::{Mutex, OnceLock};

use plonky2_maybe_rayon::*;

use crate::field::extension::{Extendable, FieldExtension};
use crate::field::goldilocks_field::GoldilocksField;
use crate::field::polynomial::{PolynomialCoeffs, PolynomialValues};
use crate::field::types::{Field, PrimeField64};
use crate::fri::oracle::{PolynomialBatch, SALT_SIZE};
use crate::fri::proof::{FriInitialTreeProof, FriProof, FriQueryRound, FriQueryStep};
use crate::fri::structure::FriInstanceInfo;
use crate::fri::FriParams;
use crate::hash::hash_types::RichField;
use crate::hash::hashing::PlonkyPermutation;
use crate::hash::merkle_proofs::MerkleProof;
use crate::hash::merkle_tree::{MerkleCap, MerkleTree};
use crate::hash::poseidon::PoseidonHash;
use crate::iop::challenger::Challenger;
use crate::iop::witness::MatrixWitness;
use crate::plonk::circuit_data::{CommonCircuitData, ProverOnlyCircuitData};
use crate::plonk::config::{GenericConfig, Hasher};
use crate::plonk::plonk_common::reduce_with_powers_multi;
use crate::plonk::vanishing_poly::evaluate_gate_constraints_base_batch;
use crate::plonk::vars::EvaluationVarsBaseBatch;
use crate::util::strided_view::PackedStridedView;
use crate::util::{log2_ceil, log2_strict, reverse_bits};

// ------------------------------------------------------------------------------------------------
// Raw bindings: one declaration per symbol of include/p2hot.h, same order.
// ------------------------------------------------------------------------------------------------
#[repr(C)]
pub struct P2hotCtx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct P2hotChallenger {
    _private: [u8; 0],
}
#[repr(C)]
pub struct P2hotBatch {
    _private: [u8; 0],
}
#[repr(C)]
pub struct P2hotCols {
    _private: [u8; 0],
}
#[repr(C)]
pub struct P2hotComm {
    _private: [u8; 0],
}
#[repr(C)]
pub struct P2hotGroup {
    _private: [u8; 0],
}
#[repr(C)]
pub struct P2hotShardedBatch {
    _private: [u8; 0],
}

/// p2hot_challenger_state: the Fiat-Shamir sponge (iop/challenger.rs:16-20) as plain words
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct P2hotChallengerState {
    pub sponge_state: [u64; 12],
    pub input_buffer: [u64; 8],
    pub output_buffer: [u64; 8],
    pub input_len: u32,
    pub output_len: u32,
}

/// p2hot_fri_batch_info: FriBatchInfo (fri/structure.rs)
#[repr(C)]
pub struct P2hotFriBatchInfo {
    pub point: [u64; 2],
    pub oracle_index: *const u32,
    pub poly_index: *const u32,
    pub n_polys: usize,
}

/// p2hot_fri_params: FriParams / FriConfig (fri/mod.rs:31-46, :103-118) + the two Option<usize> of prove_openings
#[repr(C)]
pub struct P2hotFriParams {
    pub rate_bits: c_uint,
    pub cap_height: c_uint,
    pub proof_of_work_bits: c_uint,
    pub num_query_rounds: c_uint,
    pub reduction_arity_bits: *const c_uint,
    pub n_reduction_rounds: c_uint,
    pub hiding: c_int,
    pub max_num_query_steps: c_uint,
    pub final_poly_coeff_len: usize,
}

/// p2hot_fri_proof: FriProof (fri/proof.rs:95-110) as flat caller-allocated buffers (layout: include/p2hot.h)
#[repr(C)]
pub struct P2hotFriProof {
    pub commit_phase_merkle_caps: *mut u64,
    pub final_poly: *mut u64,
    pub pow_witness: u64,
    pub query_indices: *mut u64,
    pub initial_leaves: *mut u64,
    pub initial_paths: *mut u64,
    pub step_evals: *mut u64,
    pub step_paths: *mut u64,
}

#[repr(C)]
#[derive(Default)]
pub struct P2hotFriProofLayout {
    pub caps_words: usize,
    pub final_poly_words: usize,
    pub initial_leaves_words: usize,
    pub initial_paths_words: usize,
    pub step_evals_words: usize,
    pub step_paths_words: usize,
}

/// p2hot_allgather_fn
pub type P2hotAllgatherFn = Option<
    unsafe extern "C" fn(user: *mut c_void, d_base: *mut c_void, offsets: *const usize, world: c_int, bytes: usize, hip_stream: *mut c_void) -> c_int,
>;

pub const P2HOT_OK: c_int = 0;
pub const P2HOT_KEEP_VALUES: c_uint = 1;
pub const P2HOT_COEFFS_PER_COLUMN: c_uint = 2;
pub const P2HOT_LEAVES_ASYNC: c_uint = 4;
pub const P2HOT_LEAVES_NATURAL: c_uint = 8;

#[link(name = "p2hot")]
extern "C" {
    // ---- context
    pub fn p2hot_ctx_create(device: c_int, hip_stream: *mut c_void, out: *mut *mut P2hotCtx) -> c_int;
    pub fn p2hot_ctx_destroy(ctx: *mut P2hotCtx);
    pub fn p2hot_ctx_set_stream(ctx: *mut P2hotCtx, hip_stream: *mut c_void) -> c_int;
    pub fn p2hot_ctx_sync(ctx: *mut P2hotCtx) -> c_int;
    pub fn p2hot_last_error(ctx: *const P2hotCtx) -> *const c_char;
    pub fn p2hot_version() -> *const c_char;
    pub fn p2hot_is_emulated() -> c_int;
    pub fn p2hot_profile_enable(ctx: *mut P2hotCtx, on: c_int) -> c_int;
    pub fn p2hot_tune_ntt(ctx: *mut P2hotCtx, radix_bits: c_int) -> c_int;
    pub fn p2hot_tune_overlap(ctx: *mut P2hotCtx, on: c_int) -> c_int;
    pub fn p2hot_tune_quad(ctx: *mut P2hotCtx, max_perms: usize) -> c_int;
    pub fn p2hot_tune_row(ctx: *mut P2hotCtx, max_perms: usize) -> c_int;
    pub fn p2hot_profile_json(ctx: *mut P2hotCtx, reset: c_int) -> *const c_char;
    pub fn p2hot_num_digests(log_leaves: c_uint, cap_height: c_uint) -> usize;
    // ---- primitives (device pointers)
    pub fn p2hot_fft_dev(ctx: *mut P2hotCtx, d_data: *mut u64, batch: usize, poly_stride: usize, log_n: c_uint) -> c_int;
    pub fn p2hot_ifft_dev(ctx: *mut P2hotCtx, d_data: *mut u64, batch: usize, poly_stride: usize, log_n: c_uint) -> c_int;
    pub fn p2hot_coset_ifft_dev(ctx: *mut P2hotCtx, d_data: *mut u64, batch: usize, poly_stride: usize, log_n: c_uint, shift: u64) -> c_int;
    pub fn p2hot_coset_lde_dev(
        ctx: *mut P2hotCtx, d_coeffs: *const u64, W: usize, coeff_stride: usize, log_n: c_uint, rate_bits: c_uint, shift: u64,
        row_begin: usize, row_count: usize, d_lde: *mut u64, lde_stride: usize,
    ) -> c_int;
    pub fn p2hot_transpose_dev(ctx: *mut P2hotCtx, d_colmajor: *const u64, col_stride: usize, W: usize, rows: usize, d_rowmajor: *mut u64) -> c_int;
    pub fn p2hot_reverse_index_bits_dev(ctx: *mut P2hotCtx, d_in: *const u64, d_out: *mut u64, batch: usize, poly_stride: usize, log_n: c_uint) -> c_int;
    pub fn p2hot_poseidon_permute_dev(ctx: *mut P2hotCtx, d_states: *mut u64, count: usize) -> c_int;
    pub fn p2hot_merkle_dev(
        ctx: *mut P2hotCtx, d_leaves: *const u64, layout: c_int, leaf_stride: usize, W: usize, log_leaves: c_uint, cap_height: c_uint,
        leaf_begin: usize, leaf_count: usize, d_digests: *mut u64, d_cap: *mut u64,
    ) -> c_int;
    pub fn p2hot_field_selftest_dev(ctx: *mut P2hotCtx, d_a: *const u64, d_b: *const u64, count: usize, d_out: *mut u64) -> c_int;
    pub fn p2hot_gather_rows_dev(
        ctx: *mut P2hotCtx, d_colmajor: *const u64, col_stride: usize, rows: usize, W: usize, d_idx: *const u64, m: usize, d_out: *mut u64,
    ) -> c_int;
    pub fn p2hot_commit_dev(
        ctx: *mut P2hotCtx, d_cols: *const u64, col_stride: usize, W: usize, log_n: c_uint, rate_bits: c_uint, cap_height: c_uint,
        is_values: c_int, row_begin: usize, row_count: usize, d_coeffs: *mut u64, coeff_stride: usize, d_lde: *mut u64, lde_stride: usize,
        d_leaves: *mut u64, d_digests: *mut u64, d_cap: *mut u64,
    ) -> c_int;
    // ---- Challenger
    pub fn p2hot_challenger_create(ctx: *mut P2hotCtx, out: *mut *mut P2hotChallenger) -> c_int;
    pub fn p2hot_challenger_destroy(ch: *mut P2hotChallenger);
    pub fn p2hot_challenger_load(ch: *mut P2hotChallenger, host_state: *const P2hotChallengerState) -> c_int;
    pub fn p2hot_challenger_store(ch: *mut P2hotChallenger, host_state: *mut P2hotChallengerState) -> c_int;
    pub fn p2hot_challenger_step(ch: *mut P2hotChallenger, observe: *const u64, n_observe: usize, challenges: *mut u64, n_challenges: usize) -> c_int;
    // ---- FRI commit phase and its building blocks
    pub fn p2hot_fri_commit(
        ctx: *mut P2hotCtx, coeffs: *const u64, log_n: c_uint, rate_bits: c_uint, cap_height: c_uint, arity_bits: *const c_uint,
        n_rounds: c_uint, max_num_query_steps: c_uint, final_poly_coeff_len: usize, challenger: *mut P2hotChallenger, leaves_out: *mut u64,
        digests_out: *mut u64, caps_out: *mut u64, betas_out: *mut u64, final_out: *mut u64,
    ) -> c_int;
    pub fn p2hot_fri_commit_dev(
        ctx: *mut P2hotCtx, d_coeffs_planar: *const u64, log_n: c_uint, rate_bits: c_uint, cap_height: c_uint, arity_bits: *const c_uint,
        n_rounds: c_uint, max_num_query_steps: c_uint, final_poly_coeff_len: usize, challenger: *mut P2hotChallenger, d_leaves_out: *mut u64,
        digests_out: *mut u64, digests_on_device: c_int, caps_out: *mut u64, betas_out: *mut u64, final_out: *mut u64,
    ) -> c_int;
    pub fn p2hot_fri_final_poly_dev(
        ctx: *mut P2hotCtx, d_poly_table: *const *const u64, batch_offsets: *const usize, n_batches: usize, points: *const u64,
        alpha: *const u64, log_n: c_uint, d_final: *mut u64,
    ) -> c_int;
    pub fn p2hot_eval_polys_dev(
        ctx: *mut P2hotCtx, d_poly_table: *const *const u64, n_polys: usize, log_n: c_uint, points: *const u64, n_points: usize, d_out: *mut u64,
    ) -> c_int;
    pub fn p2hot_partial_products_dev(
        ctx: *mut P2hotCtx, d_wires: *const u64, wires_stride: usize, d_sigmas: *const u64, sigmas_stride: usize, k_is: *const u64,
        num_routed: c_uint, log_n: c_uint, degree: c_uint, betas: *const u64, gammas: *const u64, num_challenges: c_uint, d_out: *mut u64,
        out_stride: usize,
    ) -> c_int;
    pub fn p2hot_merkle_paths_dev(
        ctx: *mut P2hotCtx, d_digests: *const u64, log_leaves: c_uint, cap_height: c_uint, d_idx: *const u64, m: usize, d_out: *mut u64,
    ) -> c_int;
    pub fn p2hot_fri_pow(ctx: *mut P2hotCtx, challenger: *mut P2hotChallenger, pow_bits: c_uint, witness_out: *mut u64) -> c_int;
    // ---- prover session (host pointers)
    pub fn p2hot_commit(
        ctx: *mut P2hotCtx, cols: *const *const u64, W: usize, log_n: c_uint, rate_bits: c_uint, cap_height: c_uint, is_values: c_int,
        flags: c_uint, coeffs_out: *mut u64, leaves_out: *mut u64, digests_out: *mut u64, cap_out: *mut u64, handle_out: *mut *mut P2hotBatch,
    ) -> c_int;
    pub fn p2hot_commit_salted(
        ctx: *mut P2hotCtx, cols: *const *const u64, W: usize, log_n: c_uint, rate_bits: c_uint, cap_height: c_uint, is_values: c_int,
        flags: c_uint, salt_cols: *const *const u64, n_salt: usize, coeffs_out: *mut u64, leaves_out: *mut u64, digests_out: *mut u64,
        cap_out: *mut u64, handle_out: *mut *mut P2hotBatch,
    ) -> c_int;
    pub fn p2hot_commit_cols(
        ctx: *mut P2hotCtx, cols: *mut P2hotCols, rate_bits: c_uint, cap_height: c_uint, is_values: c_int, flags: c_uint, coeffs_out: *mut u64,
        leaves_out: *mut u64, digests_out: *mut u64, cap_out: *mut u64, handle_out: *mut *mut P2hotBatch,
    ) -> c_int;
    pub fn p2hot_commit_many(
        ctx: *mut P2hotCtx, cols: *const *const u64, M: usize, W: usize, log_n: c_uint, rate_bits: c_uint, cap_height: c_uint, is_values: c_int,
        coeffs_out: *mut u64, digests_out: *mut u64, caps_out: *mut u64, handles_out: *mut *mut P2hotBatch,
    ) -> c_int;
    pub fn p2hot_commit_many_dev(
        ctx: *mut P2hotCtx, d_cols: *mut u64, M: usize, W: usize, log_n: c_uint, rate_bits: c_uint, cap_height: c_uint, is_values: c_int,
        d_lde: *mut u64, d_digests: *mut u64, d_cap: *mut u64,
    ) -> c_int;
    pub fn p2hot_batch_wrap_dev(
        ctx: *mut P2hotCtx, d_coeffs: *const u64, d_lde: *const u64, d_digests: *const u64, W: usize, log_n: c_uint, rate_bits: c_uint,
        cap_height: c_uint, out: *mut *mut P2hotBatch,
    ) -> c_int;
    pub fn p2hot_batch_width(batch: *const P2hotBatch) -> usize;
    pub fn p2hot_batch_leaf_width(batch: *const P2hotBatch) -> usize;
    pub fn p2hot_batch_degree_log(batch: *const P2hotBatch) -> c_uint;
    pub fn p2hot_batch_coeffs(batch: *mut P2hotBatch, first: usize, count: usize, out: *mut u64) -> c_int;
    pub fn p2hot_batch_rows(batch: *mut P2hotBatch, row_idx: *const u64, m: usize, out: *mut u64) -> c_int;
    pub fn p2hot_batch_paths(batch: *mut P2hotBatch, leaf_idx: *const u64, m: usize, out: *mut u64) -> c_int;
    pub fn p2hot_batch_digests(batch: *mut P2hotBatch, out: *mut u64) -> c_int;
    pub fn p2hot_batch_leaves_wait(batch: *mut P2hotBatch, row_lo: usize, row_hi: usize) -> c_int;
    pub fn p2hot_batch_leaves_block_rows(batch: *const P2hotBatch) -> usize;
    pub fn p2hot_batch_values(batch: *mut P2hotBatch, out: *mut *mut P2hotCols) -> c_int;
    pub fn p2hot_batch_subgroup_values(batch: *mut P2hotBatch, first: usize, count: usize, out: *mut *mut P2hotCols) -> c_int;
    pub fn p2hot_batch_free(batch: *mut P2hotBatch);
    pub fn p2hot_ctx_trim(ctx: *mut P2hotCtx) -> c_int;
    pub fn p2hot_host_alloc(ctx: *mut P2hotCtx, bytes: usize, out: *mut *mut c_void) -> c_int;
    pub fn p2hot_host_free(ctx: *mut P2hotCtx, p: *mut c_void);
    pub fn p2hot_cols_upload(ctx: *mut P2hotCtx, cols: *const *const u64, W: usize, log_n: c_uint, out: *mut *mut P2hotCols) -> c_int;
    pub fn p2hot_cols_download(cols: *mut P2hotCols, first: usize, count: usize, out: *mut u64) -> c_int;
    pub fn p2hot_cols_width(cols: *const P2hotCols) -> usize;
    pub fn p2hot_cols_degree_log(cols: *const P2hotCols) -> c_uint;
    pub fn p2hot_cols_free(cols: *mut P2hotCols);
    pub fn p2hot_eval_openings(
        ctx: *mut P2hotCtx, batches: *const *const P2hotBatch, n_batches: usize, points: *const u64, n_points: usize, out: *mut u64,
    ) -> c_int;
    pub fn p2hot_fri_proof_sizes(
        oracles: *const *const P2hotBatch, n_oracles: usize, params: *const P2hotFriParams, out: *mut P2hotFriProofLayout,
    ) -> c_int;
    pub fn p2hot_prove_openings(
        ctx: *mut P2hotCtx, batches: *const P2hotFriBatchInfo, n_batches: usize, oracles: *const *const P2hotBatch, n_oracles: usize,
        challenger: *mut P2hotChallenger, params: *const P2hotFriParams, proof: *mut P2hotFriProof,
    ) -> c_int;
    pub fn p2hot_prove_openings_many(
        ctx: *mut P2hotCtx, M: usize, batches: *const *const P2hotFriBatchInfo, n_batches: *const usize, oracles: *const *const P2hotBatch,
        n_oracles: usize, challengers: *const *mut P2hotChallenger, fp: *const P2hotFriParams, proofs: *mut P2hotFriProof,
    ) -> c_int;
    pub fn p2hot_partial_products(
        ctx: *mut P2hotCtx, wires: *const P2hotCols, wires_first_col: usize, sigmas: *const P2hotCols, sigmas_first_col: usize,
        k_is: *const u64, num_routed: c_uint, degree: c_uint, betas: *const u64, gammas: *const u64, num_challenges: c_uint,
        out_host: *mut u64, out_cols: *mut *mut P2hotCols,
    ) -> c_int;
    pub fn p2hot_quotient_chunks(
        ctx: *mut P2hotCtx, quotient_values: *const *const u64, num_challenges: c_uint, degree_bits: c_uint, quotient_degree_factor: c_uint,
        chunks_out: *mut *mut P2hotCols,
    ) -> c_int;
    pub fn p2hot_quotient_polys(
        ctx: *mut P2hotCtx, wires: *const P2hotBatch, constants_sigmas: *const P2hotBatch, sigmas_first_col: usize, zs_partial_products: *const P2hotBatch,
        k_is: *const u64, num_routed: c_uint, quotient_degree_factor: c_uint, betas: *const u64, gammas: *const u64, alphas: *const u64,
        num_challenges: c_uint, gate_sums: *const *const u64, values_out: *mut u64, chunks_out: *mut *mut P2hotCols,
    ) -> c_int;
    // ---- multi-GPU
    pub fn p2hot_comm_unique_id(out: *mut u8) -> c_int;
    pub fn p2hot_comm_create_rccl(ctx: *mut P2hotCtx, rank: c_int, world: c_int, id: *const u8, out: *mut *mut P2hotComm) -> c_int;
    pub fn p2hot_comm_create_callback(
        ctx: *mut P2hotCtx, rank: c_int, world: c_int, r#fn: P2hotAllgatherFn, user: *mut c_void, out: *mut *mut P2hotComm,
    ) -> c_int;
    pub fn p2hot_comm_destroy(comm: *mut P2hotComm);
    pub fn p2hot_comm_rank(comm: *const P2hotComm) -> c_int;
    pub fn p2hot_comm_world(comm: *const P2hotComm) -> c_int;
    pub fn p2hot_comm_selftest(comm: *mut P2hotComm, bytes: usize) -> c_int;
    pub fn p2hot_comm_exchange_mode(comm: *const P2hotComm) -> c_int;
    pub fn p2hot_rccl_info(path_out: *mut c_char, path_cap: usize, version_out: *mut c_int) -> c_int;
    pub fn p2hot_shard_columns(W: usize, world: c_int, rank: c_int, first: *mut usize, count: *mut usize) -> c_int;
    pub fn p2hot_commit_sharded_dev(
        ctx: *mut P2hotCtx, comm: *mut P2hotComm, d_cols_local: *const u64, col_stride: usize, W: usize, log_n: c_uint, rate_bits: c_uint,
        cap_height: c_uint, is_values: c_int, gather_digests: c_int, pipeline_chunks: c_uint, d_coeffs_all: *mut u64, d_lde: *mut u64,
        lde_stride: usize, d_leaves: *mut u64, d_digests: *mut u64, d_cap: *mut u64,
    ) -> c_int;
    pub fn p2hot_group_create(n_gpus: c_int, devices: *const c_int, out: *mut *mut P2hotGroup) -> c_int;
    pub fn p2hot_group_destroy(group: *mut P2hotGroup);
    pub fn p2hot_group_size(group: *const P2hotGroup) -> c_int;
    pub fn p2hot_group_ctx(group: *mut P2hotGroup, i: c_int) -> *mut P2hotCtx;
    pub fn p2hot_group_uses_rccl(group: *const P2hotGroup) -> c_int;
    pub fn p2hot_group_exchange_mode(group: *const P2hotGroup) -> c_int;
    pub fn p2hot_group_last_error(group: *const P2hotGroup) -> *const c_char;
    pub fn p2hot_group_commit(
        group: *mut P2hotGroup, cols: *const *const u64, W: usize, log_n: c_uint, rate_bits: c_uint, cap_height: c_uint, is_values: c_int,
        shard_mode: c_int, pipeline_chunks: c_uint, coeffs_out: *mut u64, leaves_out: *mut u64, digests_out: *mut u64, cap_out: *mut u64,
        handle_out: *mut *mut P2hotShardedBatch,
    ) -> c_int;
    pub fn p2hot_sharded_batch_open(batch: *mut P2hotShardedBatch, leaf_idx: *const u64, m: usize, rows_out: *mut u64, paths_out: *mut u64) -> c_int;
    pub fn p2hot_sharded_batch_free(batch: *mut P2hotShardedBatch);
    pub fn p2hot_group_eval_openings(
        group: *mut P2hotGroup, oracles: *const *const P2hotShardedBatch, n_oracles: usize, points: *const u64, n_points: usize, out: *mut u64,
    ) -> c_int;
    pub fn p2hot_group_fri_proof_sizes(
        oracles: *const *const P2hotShardedBatch, n_oracles: usize, params: *const P2hotFriParams, out: *mut P2hotFriProofLayout,
    ) -> c_int;
    pub fn p2hot_group_prove_openings(
        group: *mut P2hotGroup, batches: *const P2hotFriBatchInfo, n_batches: usize, oracles: *const *const P2hotShardedBatch, n_oracles: usize,
        challenger: *mut P2hotChallenger, params: *const P2hotFriParams, proof: *mut P2hotFriProof,
    ) -> c_int;
}

// ------------------------------------------------------------------------------------------------
// Layout facts the casts below rely on
// ------------------------------------------------------------------------------------------------
// GoldilocksField is #[repr(transparent)] over u64 (field/src/goldilocks_field.rs:23-25): &[F] <-> *const u64.
static_assertions::assert_eq_size!(GoldilocksField, u64);
static_assertions::assert_eq_align!(GoldilocksField, u64);
// HashOut<F> { elements: [F; 4] } (hash/hash_types.rs:25-27) is a single-field struct: 32 bytes, 4 words.
static_assertions::assert_eq_size!(crate::hash::hash_types::HashOut<GoldilocksField>, [u64; 4]);
static_assertions::assert_eq_align!(crate::hash::hash_types::HashOut<GoldilocksField>, u64);
// the degree-2 extension is [F; 2] (field/src/extension/quadratic.rs:13): [a0, a1] per element at the ABI
static_assertions::assert_eq_size!(<GoldilocksField as Extendable<2>>::Extension, [u64; 2]);

// ------------------------------------------------------------------------------------------------
// The process-wide context
// ------------------------------------------------------------------------------------------------
struct CtxPtr(*mut P2hotCtx);
// The library serialises host-pointer calls per context (P2HOT_EBUSY) and plonky2 calls them from the prover's
// main thread, outside its rayon closures; the Mutex makes that explicit on this side.
unsafe impl Send for CtxPtr {}

static CTX: OnceLock<Mutex<CtxPtr>> = OnceLock::new();

fn with_ctx<R>(f: impl FnOnce(*mut P2hotCtx) -> R) -> R {
    let m = CTX.get_or_init(|| {
        let device = std::env::var("P2HOT_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
        let mut ctx: *mut P2hotCtx = core::ptr::null_mut();
        let rc = unsafe { p2hot_ctx_create(device, core::ptr::null_mut(), &mut ctx) };
        if rc != P2HOT_OK {
            let msg = error_text(ctx);
            unsafe { p2hot_ctx_destroy(ctx) };
            panic!("p2hot_ctx_create(device {device}) failed: {msg}"); // no silent CPU fallback once the feature is on
        }
        Mutex::new(CtxPtr(ctx))
    });
    // The reference's own failure panics surface inside these closures ("Quotient has failed ...", "Tried to invert zero", shape
    // violations).  A panic must not unwind THROUGH the guard: that would poison the process-wide mutex, and the `Drop` impls of the
    // batches that unwind next (DeviceTree, FlatLeaves) lock it again -- a panic inside a panic, i.e. an abort where the CPU prover
    // gives `#[should_panic]` tests and `catch_unwind` callers an ordinary panic.  So the payload is caught, the guard released, the
    // panic resumed; and a mutex poisoned some other way is taken over (the library keeps its own state consistent: every failure is
    // a return code).  Nothing inside a closure re-enters `with_ctx` (ColsGuard / DeviceChallenger free without the lock).
    let guard = m.lock().unwrap_or_else(|e| e.into_inner());
    let ctx = guard.0;
    let r = std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| f(ctx)));
    drop(guard);
    match r {
        Ok(v) => v,
        Err(payload) => std::panic::resume_unwind(payload),
    }
}

fn error_text(ctx: *const P2hotCtx) -> String {
    unsafe { std::ffi::CStr::from_ptr(p2hot_last_error(ctx)) }.to_string_lossy().into_owned()
}

/// The reference panics on shape violations (fft.rs:171, oracle.rs:128, merkle_tree.rs:195); so does the shim.
fn check(ctx: *const P2hotCtx, rc: c_int, what: &str) {
    if rc != P2HOT_OK {
        panic!("libp2hot {what} failed ({rc}): {}", error_text(ctx));
    }
}

fn leaves_on_device() -> bool {
    matches!(std::env::var("P2HOT_LEAVES").as_deref(), Ok("device"))
}

/// P2HOT_LEAVES=vec: also fill `merkle_tree.leaves` (for code that indexes the field instead of calling `get`)
fn leaves_as_vecs() -> bool {
    matches!(std::env::var("P2HOT_LEAVES").as_deref(), Ok("vec"))
}

static ENABLED: AtomicBool = AtomicBool::new(true);

/// `false` sends every call down the unchanged CPU bodies (the bit-exact harness proves the same circuit both ways in
/// one process); `P2HOT_DISABLE=1` does the same from the environment.
pub fn set_enabled(on: bool) {
    ENABLED.store(on, Ordering::SeqCst);
}

pub fn enabled() -> bool {
    static ENV_OFF: OnceLock<bool> = OnceLock::new();
    ENABLED.load(Ordering::SeqCst) && !*ENV_OFF.get_or_init(|| matches!(std::env::var("P2HOT_DISABLE").as_deref(), Ok("1")))
}

/// Does the GPU path apply to this instantiation?  (type / size checks instead of a plugin trait;
/// `type_name` rather than `TypeId` because the config's associated types carry no `'static` bound.)
/// `blinding` no longer matters: a blinded commitment takes the same path with its salt columns (`commit`).
pub fn applies<F: RichField + Extendable<D>, C: GenericConfig<D, F = F>, const D: usize>(_blinding: bool) -> bool {
    enabled()
        && D == 2
        && type_name::<F>() == type_name::<GoldilocksField>()
        && type_name::<C::Hasher>() == type_name::<PoseidonHash>()
        && core::mem::size_of::<F>() == 8
        && core::mem::size_of::<<C::Hasher as Hasher<F>>::Hash>() == 32
}

#[inline]
fn words<F: Field>(s: &[F]) -> *const u64 {
    debug_assert_eq!(core::mem::size_of::<F>(), 8);
    s.as_ptr() as *const u64
}

/// A Vec<T> of `len` elements filled through its raw u64 view (T is F or H::Hash: plain words, see the assertions above)
fn vec_from_words<T>(len: usize, fill: impl FnOnce(*mut u64)) -> Vec<T> {
    let mut v: Vec<T> = Vec::with_capacity(len);
    fill(v.as_mut_ptr() as *mut u64);
    unsafe { v.set_len(len) }; // SAFETY: `fill` initialised len * size_of::<T>() bytes of plain integers
    v
}

/// The output-buffer idiom of MerkleTree::new (merkle_tree.rs:203-219): reserve, let the library fill the spare
/// capacity through the raw pointer, then `set_len`.  `ptr` is null for a buffer that is not wanted.
struct Out<T> {
    v: Vec<T>,
    len: usize,
}
impl<T> Out<T> {
    fn new(len: usize, wanted: bool) -> Self {
        Out { v: Vec::with_capacity(if wanted { len } else { 0 }), len: if wanted { len } else { 0 } }
    }
    fn ptr(&mut self) -> *mut u64 {
        if self.len == 0 {
            core::ptr::null_mut()
        } else {
            self.v.as_mut_ptr() as *mut u64
        }
    }
    /// SAFETY: the library call that received `ptr()` returned P2HOT_OK, i.e. it wrote all `len` elements
    unsafe fn finish(mut self) -> Vec<T> {
        self.v.set_len(self.len);
        self.v
    }
}

// ------------------------------------------------------------------------------------------------
// Device-resident tree: what MerkleTree::get / ::prove read when `leaves` / `digests` stayed on the GPU
// ------------------------------------------------------------------------------------------------
/// The leaf matrix on the host, row-major [num_leaves][width]: the ONE buffer the library's copy filled.  Pinned memory from the
/// context's block cache when it can be had (`p2hot_host_alloc`: the 9 GB copy then runs at the PCIe rate and touches no fresh
/// page, and the block is reused by the next commitment of that size), an ordinary Vec otherwise; empty = the matrix stayed on the GPU.
enum FlatLeaves<F> {
    Pinned { ptr: *mut F, len: usize },
    Heap(Vec<F>),
}
impl<F> FlatLeaves<F> {
    /// `len` elements for the library to fill; `ptr()` is what goes into `leaves_out`
    fn for_output(len: usize) -> Self {
        if len == 0 {
            return FlatLeaves::Heap(Vec::new());
        }
        let mut p: *mut c_void = core::ptr::null_mut();
        let rc = with_ctx(|ctx| unsafe { p2hot_host_alloc(ctx, len * core::mem::size_of::<F>(), &mut p) });
        if rc == P2HOT_OK && !p.is_null() {
            FlatLeaves::Pinned { ptr: p as *mut F, len }
        } else {
            FlatLeaves::Heap(Vec::with_capacity(len)) // filled through the spare capacity, `filled` sets the length
        }
    }
    fn ptr(&mut self) -> *mut u64 {
        match self {
            FlatLeaves::Pinned { ptr, .. } => *ptr as *mut u64,
            FlatLeaves::Heap(v) if v.capacity() > 0 => v.as_mut_ptr() as *mut u64,
            FlatLeaves::Heap(_) => core::ptr::null_mut(),
        }
    }
    /// SAFETY: the library call that received `ptr()` returned P2HOT_OK, i.e. it wrote all `len` elements
    unsafe fn filled(mut self, len: usize) -> Self {
        if let FlatLeaves::Heap(v) = &mut self {
            v.set_len(if v.capacity() > 0 { len } else { 0 });
        }
        self
    }
    fn as_slice(&self) -> &[F] {
        match self {
            // SAFETY: `ptr` is a live pinned block of at least `len` initialised elements, owned by this value until Drop
            FlatLeaves::Pinned { ptr, len } => unsafe { core::slice::from_raw_parts(*ptr, *len) },
            FlatLeaves::Heap(v) => v.as_slice(),
        }
    }
}
impl<F> Drop for FlatLeaves<F> {
    fn drop(&mut self) {
        if let FlatLeaves::Pinned { ptr, .. } = self {
            with_ctx(|ctx| unsafe { p2hot_host_free(ctx, *ptr as *mut c_void) }); // back to the context's cache, not to the OS
        }
    }
}

pub struct DeviceTree<F: RichField> {
    batch: *mut P2hotBatch,
    width: usize,
    num_leaves: usize,
    num_layers: usize,
    /// P2HOT_LEAVES=host (default): the whole leaf matrix, row-major [num_leaves][width], exactly the buffer the library
    /// filled (`leaves_out`) -- ONE allocation; empty when the matrix stayed on the GPU.  Rows in NATURAL LDE order
    /// (P2HOT_LEAVES_NATURAL): buffer row i is `leaves[reverse_bits(i)]`, so `get_lde_values(i, step)` reads row i * step and
    /// the quotient loop (plonk/prover.rs:712-722) walks the buffer forward
    flat: FlatLeaves<F>,
    /// P2HOT_LEAVES_ASYNC: the buffer rows [0, landed) are known to have arrived (the copy delivers row blocks in order, first
    /// block first); `num_leaves` once everything has.  A reader of a row at or beyond it asks `p2hot_batch_leaves_wait` first
    landed: AtomicUsize,
    /// rows fetched so far in the device mode: `get` hands out `&[F]`, so fetched rows are kept (never moved) for the tree's lifetime
    rows: Mutex<HashMap<usize, Box<[F]>>>,
}
unsafe impl<F: RichField> Send for DeviceTree<F> {}
unsafe impl<F: RichField> Sync for DeviceTree<F> {}

impl<F: RichField> DeviceTree<F> {
    pub fn raw(&self) -> *const P2hotBatch {
        self.batch
    }

    /// `leaves.len()` of the reference tree (merkle_tree.rs:234 passes it to merkle_tree_prove)
    pub fn num_leaves(&self) -> usize {
        self.num_leaves
    }

    /// words per leaf (salt columns included)
    pub fn leaf_width(&self) -> usize {
        self.width
    }

    /// MerkleTree::get (merkle_tree.rs:227)
    pub fn row(&self, i: usize) -> &[F] {
        let flat = self.flat.as_slice();
        if !flat.is_empty() {
            assert!(i < self.num_leaves, "index out of bounds: the len is {} but the index is {}", self.num_leaves, i); // as `&self.leaves[i]`
            let r = reverse_bits(i, self.num_leaves.trailing_zeros() as usize); // committed index -> buffer (natural) row
            self.fence(r);
            return &flat[r * self.width..(r + 1) * self.width];
        }
        let mut cache = self.rows.lock().unwrap();
        if !cache.contains_key(&i) {
            let idx = [i as u64];
            let row: Vec<F> = vec_from_words(self.width, |p| {
                with_ctx(|c| check(c, unsafe { p2hot_batch_rows(self.batch, idx.as_ptr(), 1, p) }, "p2hot_batch_rows"));
            });
            cache.insert(i, row.into_boxed_slice());
        }
        let r: &[F] = &cache[&i];
        // SAFETY: the boxed slice is never removed or moved while `self` lives; only the map's buckets move.
        unsafe { core::slice::from_raw_parts(r.as_ptr(), r.len()) }
    }

    /// The fence of the asynchronous leaf copy: returns once buffer row `r` has landed.  The copy delivers the buffer front to
    /// back, so one atomic load serves every row below the mark; beyond it the library waits on the row block's event
    /// (`p2hot_batch_leaves_wait` takes no context lock: rayon workers of the quotient loop call it side by side).
    fn fence(&self, r: usize) {
        if r < self.landed.load(Ordering::Acquire) {
            return;
        }
        let rc = unsafe { p2hot_batch_leaves_wait(self.batch, r, r + 1) };
        assert!(rc == P2HOT_OK, "libp2hot p2hot_batch_leaves_wait failed ({rc})");
        // the copy delivers whole row blocks in order: with row r, everything up to the end of r's block has landed, so a
        // reader walking forward pays one FFI wait per block (64 per matrix), not one per row
        let block = unsafe { p2hot_batch_leaves_block_rows(self.batch) };
        let mark = if block == 0 { self.num_leaves } else { core::cmp::min((r / block + 1) * block, self.num_leaves) };
        self.landed.fetch_max(mark, Ordering::AcqRel);
    }

    /// waits for the whole leaf matrix (serializers, `leaves_as_vecs`, equality: readers that walk every row)
    pub fn fence_all(&self) {
        if self.num_leaves > 0 && !self.flat.as_slice().is_empty() {
            self.fence(self.num_leaves - 1);
        }
    }

    /// the reference's `leaves: Vec<Vec<F>>` (merkle_tree.rs:47) rebuilt from the flat buffer, rows in parallel
    /// (P2HOT_LEAVES=vec, and the bit-exact harness, which compares the field itself)
    pub fn leaves_as_vecs(&self) -> Vec<Vec<F>> {
        if self.flat.as_slice().is_empty() {
            return (0..self.num_leaves).map(|i| self.row(i).to_vec()).collect();
        }
        self.fence_all();
        (0..self.num_leaves).into_par_iter().map(|i| self.row(i).to_vec()).collect() // committed order out of the natural-order buffer
    }

    /// merkle_tree_prove (merkle_tree.rs:151-190)
    pub fn path<H: Hasher<F>>(&self, leaf_index: usize) -> Vec<H::Hash> {
        let idx = [leaf_index as u64];
        vec_from_words(self.num_layers, |p| {
            with_ctx(|c| check(c, unsafe { p2hot_batch_paths(self.batch, idx.as_ptr(), 1, p) }, "p2hot_batch_paths"));
        })
    }
}

impl<F: RichField> Drop for DeviceTree<F> {
    fn drop(&mut self) {
        // returns the LDE matrix, the digests and the coefficients to the context's block cache
        with_ctx(|_| unsafe { p2hot_batch_free(self.batch) });
    }
}

// ------------------------------------------------------------------------------------------------
// PolynomialBatch::from_values / from_coeffs (fri/oracle.rs:57-112)
// ------------------------------------------------------------------------------------------------
/// `cols`: the W input vectors (values on H_n, or coefficients), each of length n.
pub fn commit<F: RichField + Extendable<D>, C: GenericConfig<D, F = F>, const D: usize>(
    cols: &[&[F]],
    rate_bits: usize,
    cap_height: usize,
    is_values: bool,
    blinding: bool,
) -> PolynomialBatch<F, C, D> {
    // "If blinding, salt with two random elements to each leaf vector" (oracle.rs:122-123, :133-137): the random numbers
    // are drawn HERE, by the reference's own generator, in the reference's order; the library only places and hashes them
    let salts: Vec<Vec<F>> = if blinding {
        let big_n = cols[0].len() << rate_bits;
        (0..SALT_SIZE).map(|_| F::rand_vec(big_n)).collect()
    } else {
        Vec::new()
    };
    commit_with_salts::<F, C, D>(cols, rate_bits, cap_height, is_values, &salts)
}

/// `commit` with the salt vectors given (each of length N = n << rate_bits, in LDE-value order: oracle.rs:133-137 appends
/// them to the LDE vectors BEFORE transpose + reverse_index_bits); empty = an unblinded commitment.
pub(crate) fn commit_with_salts<F: RichField + Extendable<D>, C: GenericConfig<D, F = F>, const D: usize>(
    cols: &[&[F]],
    rate_bits: usize,
    cap_height: usize,
    is_values: bool,
    salts: &[Vec<F>],
) -> PolynomialBatch<F, C, D> {
    let w = cols.len();
    let lw = w + salts.len(); // words per leaf
    let n = cols[0].len(); // polynomials[0].len(), oracle.rs:90
    let log_n = log2_strict(n);
    assert!(cols.iter().all(|c| c.len() == n));
    let big_n = n << rate_bits;
    let num_digests = 2 * (big_n - (1usize << cap_height));
    let ptrs: Vec<*const u64> = cols.iter().map(|c| words(c)).collect();
    assert!(salts.iter().all(|v| v.len() == big_n));
    let salt_ptrs: Vec<*const u64> = salts.iter().map(|v| words(v)).collect();
    let on_device = leaves_on_device();
    let mut handle: *mut P2hotBatch = core::ptr::null_mut();
    // `polynomials` always comes back (W * n words): the quotient evaluation and the openings read it on the host.  One Vec per
    // polynomial, filled IN PLACE by the library (P2HOT_COEFFS_PER_COLUMN): no flat block to split afterwards
    let mut coeff_vecs: Vec<Out<F>> = (0..w).map(|_| Out::<F>::new(n, true)).collect();
    let coeff_ptrs: Vec<*mut u64> = coeff_vecs.iter_mut().map(|o| o.ptr()).collect();
    let mut cap = Out::<<C::Hasher as Hasher<F>>::Hash>::new(1 << cap_height, true);
    let mut digests = Out::<<C::Hasher as Hasher<F>>::Hash>::new(num_digests, !on_device);
    let mut flat_leaves = FlatLeaves::<F>::for_output(if on_device { 0 } else { big_n * lw });
    // the leaf matrix comes back in natural LDE order, and -- into a pinned block -- ASYNCHRONOUSLY: the call returns with the cap,
    // the coefficients and the digests (about a third of the time at the wires shape), the 9 GB matrix keeps landing block by block
    // behind `DeviceTree::fence` while the partial products, the Zs commitment and the first quotient batches run.  Into a heap
    // Vec the copy would be synchronous anyway (pageable memory), and P2HOT_LEAVES_SYNC=1 asks for that on purpose
    let async_leaves = !on_device && matches!(flat_leaves, FlatLeaves::Pinned { .. }) && std::env::var_os("P2HOT_LEAVES_SYNC").is_none();
    let flags = P2HOT_COEFFS_PER_COLUMN | if on_device { 0 } else { P2HOT_LEAVES_NATURAL } | if async_leaves { P2HOT_LEAVES_ASYNC } else { 0 };
    with_ctx(|ctx| {
        let rc = unsafe {
            p2hot_commit_salted(
                ctx, ptrs.as_ptr(), w, log_n as c_uint, rate_bits as c_uint, cap_height as c_uint, is_values as c_int, flags,
                salt_ptrs.as_ptr(), salt_ptrs.len(), coeff_ptrs.as_ptr() as *mut u64, flat_leaves.ptr(), digests.ptr(), cap.ptr(), &mut handle,
            )
        };
        check(ctx, rc, "p2hot_commit_salted");
    });
    let (cap, digests, flat_leaves) = unsafe { (cap.finish(), digests.finish(), flat_leaves.filled(if on_device { 0 } else { big_n * lw })) };
    // `polynomials` (oracle.rs:32): the W vectors the library filled (SAFETY: the call returned P2HOT_OK, every one holds n words)
    let polynomials = coeff_vecs.into_iter().map(|o| PolynomialCoeffs::new(unsafe { o.finish() })).collect();
    // merkle_tree.leaves (oracle.rs:97-98) is NOT rebuilt: the flat buffer the library filled moves into the DeviceTree and
    // MerkleTree::get serves `&flat[i * lw..]`; the field itself stays empty unless P2HOT_LEAVES=vec asks for it
    let device = std::sync::Arc::new(DeviceTree {
        batch: handle,
        width: lw, // MerkleTree::get returns the whole leaf; get_lde_values strips the salt (oracle.rs:146)
        num_leaves: big_n,
        num_layers: log_n + rate_bits - cap_height,
        flat: flat_leaves,
        landed: AtomicUsize::new(if async_leaves { 0 } else { big_n }),
        rows: Mutex::new(HashMap::new()),
    });
    let leaves: Vec<Vec<F>> = if leaves_as_vecs() { device.leaves_as_vecs() } else { Vec::new() };
    PolynomialBatch {
        polynomials,
        merkle_tree: MerkleTree { leaves, digests, cap: MerkleCap(cap), device: Some(device) },
        degree_log: log_n,
        rate_bits,
        blinding: !salts.is_empty(),
    }
}

// ------------------------------------------------------------------------------------------------
// Challenger <-> p2hot_challenger (iop/challenger.rs:16-153)
// ------------------------------------------------------------------------------------------------
struct DeviceChallenger(*mut P2hotChallenger);
impl Drop for DeviceChallenger {
    fn drop(&mut self) {
        unsafe { p2hot_challenger_destroy(self.0) };
    }
}

fn challenger_to_device<F: RichField, H: Hasher<F>>(ctx: *mut P2hotCtx, ch: &mut Challenger<F, H>) -> DeviceChallenger {
    let (state, input, output) = ch.p2hot_parts();
    let mut st = P2hotChallengerState::default();
    for (d, s) in st.sponge_state.iter_mut().zip(state.as_ref()) {
        *d = s.to_canonical_u64();
    }
    for (d, s) in st.input_buffer.iter_mut().zip(input.iter()) {
        *d = s.to_canonical_u64();
    }
    for (d, s) in st.output_buffer.iter_mut().zip(output.iter()) {
        *d = s.to_canonical_u64();
    }
    st.input_len = input.len() as u32;
    st.output_len = output.len() as u32;
    let mut h: *mut P2hotChallenger = core::ptr::null_mut();
    check(ctx, unsafe { p2hot_challenger_create(ctx, &mut h) }, "p2hot_challenger_create");
    check(ctx, unsafe { p2hot_challenger_load(h, &st) }, "p2hot_challenger_load");
    DeviceChallenger(h)
}

fn challenger_from_device<F: RichField, H: Hasher<F>>(ctx: *mut P2hotCtx, dev: &DeviceChallenger, ch: &mut Challenger<F, H>) {
    let mut st = P2hotChallengerState::default();
    check(ctx, unsafe { p2hot_challenger_store(dev.0, &mut st) }, "p2hot_challenger_store");
    let (state, input, output) = ch.p2hot_parts();
    state.set_from_iter(st.sponge_state.iter().map(|&x| F::from_canonical_u64(x)), 0);
    input.clear();
    input.extend(st.input_buffer[..st.input_len as usize].iter().map(|&x| F::from_canonical_u64(x)));
    output.clear();
    output.extend(st.output_buffer[..st.output_len as usize].iter().map(|&x| F::from_canonical_u64(x)));
}

fn fri_params_c(fri_params: &FriParams, arity: &[c_uint], final_poly_coeff_len: Option<usize>, max_num_query_steps: Option<usize>) -> P2hotFriParams {
    P2hotFriParams {
        rate_bits: fri_params.config.rate_bits as c_uint,
        cap_height: fri_params.config.cap_height as c_uint,
        proof_of_work_bits: fri_params.config.proof_of_work_bits as c_uint,
        num_query_rounds: fri_params.config.num_query_rounds as c_uint,
        reduction_arity_bits: arity.as_ptr(),
        n_reduction_rounds: arity.len() as c_uint,
        hiding: fri_params.hiding as c_int,
        max_num_query_steps: max_num_query_steps.unwrap_or(0) as c_uint, // 0 = None: a Some(0) pads nothing either
        final_poly_coeff_len: final_poly_coeff_len.unwrap_or(0),
    }
}

// ------------------------------------------------------------------------------------------------
// fri_committed_trees (fri/prover.rs:84-150): used by starky and by plonky2 when an oracle has no device handle
// ------------------------------------------------------------------------------------------------
pub fn fri_committed_trees<F: RichField + Extendable<D>, C: GenericConfig<D, F = F>, const D: usize>(
    coeffs: &PolynomialCoeffs<F::Extension>,
    challenger: &mut Challenger<F, C::Hasher>,
    fri_params: &FriParams,
    final_poly_coeff_len: Option<usize>,
    max_num_query_steps: Option<usize>,
) -> (Vec<MerkleTree<F, C::Hasher>>, PolynomialCoeffs<F::Extension>) {
    let big_n = coeffs.len();
    let rate_bits = fri_params.config.rate_bits;
    let cap_height = fri_params.config.cap_height;
    let n = big_n >> rate_bits; // "Only the first 1/rate coefficients are non-zero" (prover.rs:27)
    let arity: Vec<c_uint> = fri_params.reduction_arity_bits.iter().map(|&a| a as c_uint).collect();
    // output sizes per round (include/p2hot.h, p2hot_fri_commit)
    let mut m = big_n;
    let mut shapes = Vec::new(); // (leaves, row width, digests)
    for &ab in &fri_params.reduction_arity_bits {
        let n_leaves = m >> ab;
        shapes.push((n_leaves, 2usize << ab, 2 * (n_leaves - (1usize << cap_height))));
        m >>= ab;
    }
    let n_final = m >> rate_bits;
    let leaf_words: usize = shapes.iter().map(|s| s.0 * s.1).sum();
    let digest_count: usize = shapes.iter().map(|s| s.2).sum();
    let cap_len = 1usize << cap_height;
    let mut leaves_flat = Out::<F>::new(leaf_words, true);
    let mut digests_flat = Out::<<C::Hasher as Hasher<F>>::Hash>::new(digest_count, true);
    let mut caps_flat = Out::<<C::Hasher as Hasher<F>>::Hash>::new(cap_len * shapes.len(), true);
    let mut final_coeffs = Out::<F::Extension>::new(n_final, true);
    with_ctx(|ctx| {
        let dev = challenger_to_device(ctx, challenger);
        let rc = unsafe {
            p2hot_fri_commit(
                ctx, coeffs.coeffs.as_ptr() as *const u64, log2_strict(n.max(1)) as c_uint, rate_bits as c_uint, cap_height as c_uint,
                arity.as_ptr(), arity.len() as c_uint, max_num_query_steps.unwrap_or(0) as c_uint, final_poly_coeff_len.unwrap_or(0), dev.0,
                leaves_flat.ptr(), digests_flat.ptr(), caps_flat.ptr(), core::ptr::null_mut(), final_coeffs.ptr(),
            )
        };
        check(ctx, rc, "p2hot_fri_commit");
        challenger_from_device(ctx, &dev, challenger);
    });
    let (leaves_flat, digests_flat, caps_flat, final_coeffs) =
        unsafe { (leaves_flat.finish(), digests_flat.finish(), caps_flat.finish(), final_coeffs.finish()) };
    let mut trees = Vec::with_capacity(shapes.len());
    let (mut lo, mut dg) = (0usize, 0usize);
    for (i, &(n_leaves, width, nd)) in shapes.iter().enumerate() {
        let leaves = leaves_flat[lo..lo + n_leaves * width].par_chunks_exact(width).map(|r| r.to_vec()).collect();
        trees.push(MerkleTree {
            leaves,
            digests: digests_flat[dg..dg + nd].to_vec(),
            cap: MerkleCap(caps_flat[i * cap_len..(i + 1) * cap_len].to_vec()),
            device: None,
        });
        lo += n_leaves * width;
        dg += nd;
    }
    (trees, PolynomialCoeffs::new(final_coeffs))
}

// ------------------------------------------------------------------------------------------------
// PolynomialBatch::prove_openings + fri_proof (fri/oracle.rs:176-237, fri/prover.rs:24-82, :204-258)
// ------------------------------------------------------------------------------------------------
/// One library call when every oracle carries a device handle; `None` sends the caller down the CPU path.
pub fn prove_openings<F: RichField + Extendable<D>, C: GenericConfig<D, F = F>, const D: usize>(
    instance: &FriInstanceInfo<F, D>,
    oracles: &[&PolynomialBatch<F, C, D>],
    challenger: &mut Challenger<F, C::Hasher>,
    fri_params: &FriParams,
    final_poly_coeff_len: Option<usize>,
    max_num_query_steps: Option<usize>,
) -> Option<FriProof<F, C::Hasher, D>> {
    // FriParams::hiding does not reach the prover's FRI path (it is observed into the transcript by the caller and tells the
    // VERIFIER to strip the salts, fri/verifier.rs:149-151): blinded oracles simply have wider leaves
    if !applies::<F, C, D>(false) {
        return None;
    }
    let handles: Option<Vec<*const P2hotBatch>> = oracles.iter().map(|o| o.merkle_tree.device.as_ref().map(|d| d.raw())).collect();
    let handles = handles?;
    let arity: Vec<c_uint> = fri_params.reduction_arity_bits.iter().map(|&a| a as c_uint).collect();
    let params = fri_params_c(fri_params, &arity, final_poly_coeff_len, max_num_query_steps);
    // FriInstanceInfo.batches -> flat index arrays
    let idx: Vec<(Vec<u32>, Vec<u32>)> = instance
        .batches
        .iter()
        .map(|b| (b.polynomials.iter().map(|p| p.oracle_index as u32).collect(), b.polynomials.iter().map(|p| p.polynomial_index as u32).collect()))
        .collect();
    let infos: Vec<P2hotFriBatchInfo> = instance
        .batches
        .iter()
        .zip(&idx)
        .map(|(b, (oi, pi))| {
            let pt = b.point.to_basefield_array();
            P2hotFriBatchInfo { point: [pt[0].to_canonical_u64(), pt[1].to_canonical_u64()], oracle_index: oi.as_ptr(), poly_index: pi.as_ptr(), n_polys: oi.len() }
        })
        .collect();
    let mut lay = P2hotFriProofLayout::default();
    assert_eq!(unsafe { p2hot_fri_proof_sizes(handles.as_ptr(), handles.len(), &params, &mut lay) }, P2HOT_OK);
    let q = fri_params.config.num_query_rounds;
    let cap_len = 1usize << fri_params.config.cap_height;
    let mut caps = vec![0u64; lay.caps_words];
    let mut final_poly = vec![0u64; lay.final_poly_words];
    let mut initial_leaves = vec![0u64; lay.initial_leaves_words];
    let mut initial_paths = vec![0u64; lay.initial_paths_words];
    let mut step_evals = vec![0u64; lay.step_evals_words];
    let mut step_paths = vec![0u64; lay.step_paths_words];
    let mut proof = P2hotFriProof {
        commit_phase_merkle_caps: caps.as_mut_ptr(),
        final_poly: final_poly.as_mut_ptr(),
        pow_witness: 0,
        query_indices: core::ptr::null_mut(),
        initial_leaves: initial_leaves.as_mut_ptr(),
        initial_paths: initial_paths.as_mut_ptr(),
        step_evals: step_evals.as_mut_ptr(),
        step_paths: step_paths.as_mut_ptr(),
    };
    with_ctx(|ctx| {
        let dev = challenger_to_device(ctx, challenger);
        let rc = unsafe { p2hot_prove_openings(ctx, infos.as_ptr(), infos.len(), handles.as_ptr(), handles.len(), dev.0, &params, &mut proof) };
        check(ctx, rc, "p2hot_prove_openings");
        challenger_from_device(ctx, &dev, challenger);
    });
    // flat buffers -> FriProof (layout: include/p2hot.h, p2hot_fri_proof)
    let f = |x: u64| F::from_canonical_u64(x);
    let hash = |w: &[u64]| -> <C::Hasher as Hasher<F>>::Hash { vec_from_words::<<C::Hasher as Hasher<F>>::Hash>(1, |p| unsafe { core::ptr::copy_nonoverlapping(w.as_ptr(), p, 4) })[0] };
    let ext = |w: &[u64]| F::Extension::from_basefield_array({
        let mut a = [F::ZERO; D];
        a[0] = f(w[0]);
        a[1] = f(w[1]);
        a
    });
    let degree_bits = oracles[0].degree_log;
    let layers0 = degree_bits + fri_params.config.rate_bits - fri_params.config.cap_height;
    // the opened leaf of a blinded oracle carries its SALT_SIZE salt words behind the polynomial values (fri/proof.rs:45-52)
    let widths: Vec<usize> = oracles.iter().map(|o| o.polynomials.len() + if o.blinding { SALT_SIZE } else { 0 }).collect();
    let wsum: usize = widths.iter().sum();
    let mut ev_w = Vec::new();
    let mut pa_w = Vec::new();
    let mut lm = degree_bits + fri_params.config.rate_bits;
    for &ab in &fri_params.reduction_arity_bits {
        ev_w.push(2usize << ab);
        pa_w.push(lm - ab - fri_params.config.cap_height);
        lm -= ab;
    }
    let (ev_sum, pa_sum): (usize, usize) = (ev_w.iter().sum(), pa_w.iter().sum());
    let query_round_proofs = (0..q)
        .map(|qi| {
            let mut wo = 0usize;
            let evals_proofs = widths
                .iter()
                .enumerate()
                .map(|(oi, &w)| {
                    let row = &initial_leaves[qi * wsum + wo..qi * wsum + wo + w];
                    wo += w;
                    let base = (qi * oracles.len() + oi) * layers0 * 4;
                    let siblings = (0..layers0).map(|l| hash(&initial_paths[base + 4 * l..base + 4 * l + 4])).collect();
                    (row.iter().map(|&x| f(x)).collect::<Vec<F>>(), MerkleProof { siblings })
                })
                .collect();
            let (mut eo, mut po) = (0usize, 0usize);
            let steps = ev_w
                .iter()
                .zip(&pa_w)
                .map(|(&ew, &pw)| {
                    let e = &step_evals[qi * ev_sum + eo..qi * ev_sum + eo + ew];
                    let p = &step_paths[qi * pa_sum * 4 + po..qi * pa_sum * 4 + po + 4 * pw];
                    eo += ew;
                    po += 4 * pw;
                    FriQueryStep { evals: e.chunks_exact(2).map(ext).collect(), merkle_proof: MerkleProof { siblings: p.chunks_exact(4).map(hash).collect() } }
                })
                .collect();
            FriQueryRound { initial_trees_proof: FriInitialTreeProof { evals_proofs }, steps }
        })
        .collect();
    Some(FriProof {
        commit_phase_merkle_caps: caps.chunks_exact(4 * cap_len).map(|c| MerkleCap(c.chunks_exact(4).map(hash).collect())).collect(),
        query_round_proofs,
        final_poly: PolynomialCoeffs::new(final_poly.chunks_exact(2).map(ext).collect()),
        pow_witness: f(proof.pow_witness),
    })
}

// ------------------------------------------------------------------------------------------------
// all_wires_permutation_partial_products (plonk/prover.rs:356-449) on the GPU
// ------------------------------------------------------------------------------------------------
struct ColsGuard(*mut P2hotCols);
impl Drop for ColsGuard {
    fn drop(&mut self) {
        if !self.0.is_null() {
            unsafe { p2hot_cols_free(self.0) };
        }
    }
}

/// The permutation argument's Z and partial-product polynomials for every (beta, gamma): the routed wire columns go up as they
/// lie in the witness (`wire_values[col][row]`, iop/witness.rs:283-291), the sigma values come from the constants_sigmas
/// commitment's device-resident coefficients (`p2hot_batch_subgroup_values`: no host transpose of `prover_data.sigmas`), one
/// `p2hot_partial_products` call (lane = row, one inversion per row, the sequential row walk as a prefix-product scan).
/// Returns the reference's shape -- per challenge the `num_partial_products` polynomials followed by Z (prover.rs:437-447) --
/// or `None` for the CPU body.
pub fn all_wires_permutation_partial_products<F: RichField + Extendable<D>, C: GenericConfig<D, F = F>, const D: usize>(
    witness: &MatrixWitness<F>,
    betas: &[F],
    gammas: &[F],
    prover_data: &ProverOnlyCircuitData<F, C, D>,
    common_data: &CommonCircuitData<F, D>,
) -> Option<Vec<Vec<PolynomialValues<F>>>> {
    if !applies::<F, C, D>(false) {
        return None;
    }
    let h_cs = prover_data.constants_sigmas_commitment.merkle_tree.device.as_ref()?.raw();
    let nc = common_data.config.num_challenges;
    let num_routed = common_data.config.num_routed_wires;
    let degree = common_data.quotient_degree_factor;
    let num_prods = common_data.num_partial_products;
    let n = 1usize << common_data.degree_bits();
    let wire_ptrs: Vec<*const u64> = witness.wire_values[..num_routed].iter().map(|c| words(c)).collect();
    assert!(witness.wire_values[..num_routed].iter().all(|c| c.len() == n));
    let k_is: Vec<u64> = common_data.k_is[..num_routed].iter().map(|k| k.to_canonical_u64()).collect();
    let b: Vec<u64> = betas.iter().map(|x| x.to_canonical_u64()).collect();
    let g: Vec<u64> = gammas.iter().map(|x| x.to_canonical_u64()).collect();
    let flat: Vec<F> = vec_from_words(nc * (num_prods + 1) * n, |out| {
        with_ctx(|ctx| {
            let (mut wires, mut sigmas) = (ColsGuard(core::ptr::null_mut()), ColsGuard(core::ptr::null_mut()));
            check(ctx, unsafe { p2hot_cols_upload(ctx, wire_ptrs.as_ptr(), num_routed, log2_strict(n) as c_uint, &mut wires.0) }, "p2hot_cols_upload");
            check(
                ctx,
                unsafe { p2hot_batch_subgroup_values(h_cs as *mut P2hotBatch, common_data.sigmas_range().start, num_routed, &mut sigmas.0) },
                "p2hot_batch_subgroup_values",
            );
            let rc = unsafe {
                p2hot_partial_products(
                    ctx, wires.0, 0, sigmas.0, 0, k_is.as_ptr(), num_routed as c_uint, degree as c_uint, b.as_ptr(), g.as_ptr(), nc as c_uint, out,
                    core::ptr::null_mut(),
                )
            };
            check(ctx, rc, "p2hot_partial_products"); // a zero denominator: the reference panics with "Tried to invert zero"
        })
    });
    // the library's order is the batch order (Z_0 .. Z_{nc-1}, then the partial products of challenge 0, 1, ...: prover.rs:224-229);
    // the function's own result has, per challenge, the partial products and THEN Z (the caller pops it, prover.rs:226-229)
    let col = |r: usize| PolynomialValues::new(flat[r * n..(r + 1) * n].to_vec());
    Some((0..nc).map(|c| (0..num_prods).map(|p| col(nc + c * num_prods + p)).chain(core::iter::once(col(c))).collect()).collect())
}

// ------------------------------------------------------------------------------------------------
// OpeningSet::new (plonk/proof.rs:314-351): `eval_commitment` on the GPU
// ------------------------------------------------------------------------------------------------
/// `c.polynomials.par_iter().map(|p| p.to_extension().eval(z))` (proof.rs:323-328) as ONE `p2hot_eval_openings` call on the
/// coefficients the commitment already holds on the device: W Horner evaluations of degree n in F^2.  `None` (no device handle,
/// another field / hasher, the path switched off) sends the caller down the CPU closure.
pub fn eval_commitment<F: RichField + Extendable<D>, C: GenericConfig<D, F = F>, const D: usize>(
    z: F::Extension,
    c: &PolynomialBatch<F, C, D>,
) -> Option<Vec<F::Extension>> {
    if !applies::<F, C, D>(false) {
        return None;
    }
    let handle = c.merkle_tree.device.as_ref()?.raw();
    let w = c.polynomials.len();
    let zb = z.to_basefield_array();
    let point = [zb[0].to_canonical_u64(), zb[1].to_canonical_u64()];
    let handles = [handle];
    // out: [n_points = 1][W][2] words = W extension elements ([a0, a1], field/src/extension/quadratic.rs:13)
    Some(vec_from_words::<F::Extension>(w, |out| {
        with_ctx(|ctx| check(ctx, unsafe { p2hot_eval_openings(ctx, handles.as_ptr(), 1, point.as_ptr(), 1, out) }, "p2hot_eval_openings"));
    }))
}

// ------------------------------------------------------------------------------------------------
// compute_quotient_polys (plonk/prover.rs:609-815): the permutation argument's share on the GPU
// ------------------------------------------------------------------------------------------------
/// The vanishing polynomial's terms are `[L_0 (Z - 1) ..] ++ [partial product checks ..] ++ [gate constraint terms ..]`
/// (plonk/vanishing_poly.rs:326-330) reduced with the powers of each alpha.  The gate terms are circuit specific: they are
/// evaluated HERE by the reference's own `evaluate_gate_constraints_base_batch` on the reference's own point batches
/// (prover.rs:684-779, BATCH_SIZE = 32) and reduced on their own; everything in front of them -- 80 routed wires x 2 challenges at
/// 2^23 points for a 2^20-gate circuit -- and the division by Z_H, the coset_ifft and the trim run in ONE `p2hot_quotient_polys`
/// call on the LDE matrices the three commitments already hold on the device.  Returns the reference's result (one polynomial of
/// `quotient_degree_factor * n` coefficients per challenge); `None` sends the caller down the CPU body (lookups, a commitment
/// without a device handle, more than 4 challenges).
pub fn compute_quotient_polys<F: RichField + Extendable<D>, C: GenericConfig<D, F = F>, const D: usize>(
    common_data: &CommonCircuitData<F, D>,
    prover_data: &ProverOnlyCircuitData<F, C, D>,
    public_inputs_hash: &<<C as GenericConfig<D>>::InnerHasher as Hasher<F>>::Hash,
    wires_commitment: &PolynomialBatch<F, C, D>,
    zs_partial_products_commitment: &PolynomialBatch<F, C, D>,
    betas: &[F],
    gammas: &[F],
    alphas: &[F],
) -> Option<Vec<PolynomialCoeffs<F>>> {
    if !applies::<F, C, D>(false) || common_data.num_lookup_polys != 0 {
        return None;
    }
    let cs = &prover_data.constants_sigmas_commitment;
    let h_wires = wires_commitment.merkle_tree.device.as_ref()?.raw();
    let h_cs = cs.merkle_tree.device.as_ref()?.raw();
    let h_zs = zs_partial_products_commitment.merkle_tree.device.as_ref()?.raw();
    let nc = common_data.config.num_challenges;
    let qdf = common_data.quotient_degree_factor;
    let quotient_degree_bits = log2_ceil(qdf);
    let rate_bits = common_data.config.fri_config.rate_bits;
    if nc > 4 || quotient_degree_bits > rate_bits {
        return None; // (the CPU body asserts on the second, prover.rs:632-636)
    }
    assert!(betas.len() == nc && gammas.len() == nc && alphas.len() == nc);
    let n = 1usize << common_data.degree_bits();
    let step = 1usize << (rate_bits - quotient_degree_bits); // prover.rs:640
    let lde_size = n << quotient_degree_bits;
    let num_constants = common_data.constants_range().len();
    let num_wires = common_data.config.num_wires;
    // gate_sums[a][i] = reduce_with_powers(gate constraint terms at point i, alpha_a)
    const BATCH_SIZE: usize = 32; // prover.rs:607
    let indices: Vec<usize> = (0..lde_size).collect();
    let per_point: Vec<F> = indices // point-major: [lde_size][nc]
        .par_chunks(BATCH_SIZE)
        .flat_map(|batch| {
            let m = batch.len();
            let mut local_constants_batch = vec![F::ZERO; m * num_constants];
            let mut local_wires_batch = vec![F::ZERO; m * num_wires];
            for (j, &i) in batch.iter().enumerate() {
                let local_constants_sigmas = cs.get_lde_values(i, step);
                let local_wires = wires_commitment.get_lde_values(i, step);
                for (k, &v) in local_constants_sigmas[common_data.constants_range()].iter().enumerate() {
                    local_constants_batch[k * m + j] = v; // the transposed layout of prover.rs:748-761
                }
                for (k, &v) in local_wires.iter().enumerate() {
                    local_wires_batch[k * m + j] = v;
                }
            }
            let vars_batch = EvaluationVarsBaseBatch::new(m, &local_constants_batch, &local_wires_batch, public_inputs_hash);
            let constraint_terms_batch = evaluate_gate_constraints_base_batch::<F, D>(common_data, vars_batch);
            let mut sums = vec![F::ZERO; m * nc];
            if !constraint_terms_batch.is_empty() {
                for j in 0..m {
                    let r = reduce_with_powers_multi(PackedStridedView::<F>::new(&constraint_terms_batch, m, j), alphas);
                    sums[j * nc..(j + 1) * nc].copy_from_slice(&r);
                }
            }
            sums
        })
        .collect();
    let gate_sums: Vec<Vec<u64>> = (0..nc).map(|a| per_point.iter().skip(a).step_by(nc).map(|v| v.to_canonical_u64()).collect()).collect();
    let gate_ptrs: Vec<*const u64> = gate_sums.iter().map(|v| v.as_ptr()).collect();
    let num_routed = common_data.config.num_routed_wires;
    let k_is: Vec<u64> = common_data.k_is[..num_routed].iter().map(|k| k.to_canonical_u64()).collect();
    let to_u64 = |v: &[F]| -> Vec<u64> { v.iter().map(|x| x.to_canonical_u64()).collect() };
    let (b, g, a) = (to_u64(betas), to_u64(gammas), to_u64(alphas));
    let coeffs: Vec<F> = vec_from_words(nc * qdf * n, |out| {
        with_ctx(|ctx| {
            let mut chunks = ColsGuard(core::ptr::null_mut()); // freed on every path, the panicking ones included
            let rc = unsafe {
                p2hot_quotient_polys(
                    ctx, h_wires, h_cs, common_data.sigmas_range().start, h_zs, k_is.as_ptr(), num_routed as c_uint, qdf as c_uint,
                    b.as_ptr(), g.as_ptr(), a.as_ptr(), nc as c_uint, gate_ptrs.as_ptr(), core::ptr::null_mut(), &mut chunks.0,
                )
            };
            check(ctx, rc, "p2hot_quotient_polys"); // "Quotient has failed ..." panics here as trim_to_len does on the CPU path
            check(ctx, unsafe { p2hot_cols_download(chunks.0, 0, nc * qdf, out) }, "p2hot_cols_download");
        })
    });
    // challenge c's polynomial = its quotient_degree_factor chunks of n coefficients, back to back (prover.rs:810-814 + :279-287)
    Some(coeffs.par_chunks_exact((qdf * n).max(1)).map(|c| PolynomialCoeffs::new(c.to_vec())).collect())
}

/// PolynomialValues -> column slices for `commit`
pub fn value_slices<F: Field>(values: &[PolynomialValues<F>]) -> Vec<&[F]> {
    values.iter().map(|v| v.values.as_slice()).collect()
}
pub fn coeff_slices<F: Field>(polys: &[PolynomialCoeffs<F>]) -> Vec<&[F]> {
    polys.iter().map(|p| p.coeffs.as_slice()).collect()
}

// MerkleTree's value is (leaves, digests, cap) (hash/merkle_tree.rs:45-61); the device copy is a cache of the same tree, not part
// of its value.
impl<F: RichField> core::fmt::Debug for DeviceTree<F> {
    fn fmt(&self, f: &mut core::fmt::Formatter<'_>) -> core::fmt::Result {
        write!(f, "DeviceTree({:p}, {} leaves of {}, {} layers, {} host words)", self.batch, self.num_leaves, self.width, self.num_layers, self.flat.as_slice().len())
    }
}
impl<F: RichField> PartialEq for DeviceTree<F> {
    /// the device copy is a cache of the tree it hangs off: it never decides an equality on its own.  The VALUE comparison of two
    /// trees is `MerkleTree::eq`, written by hand under the feature (hash/merkle_tree.rs): rows through `get`, digests through
    /// `digests_or_device`, cap -- so a GPU-built tree (`leaves` empty, the matrix behind `device`) equals its serialise /
    /// deserialise round trip (`leaves` populated, `device: None`), as `examples/square_root.rs:152` asserts of CircuitData
    fn eq(&self, _other: &Self) -> bool {
        true
    }
}
impl<F: RichField> Eq for DeviceTree<F> {}

impl<F: RichField> DeviceTree<F> {
    /// `MerkleTree::digests` (reference layout) fetched from the device: what the field holds in the default mode, and what it
    /// would hold in P2HOT_LEAVES=device, where it is left empty
    pub fn digests<H: Hasher<F>>(&self) -> Vec<H::Hash> {
        let cap_len = self.num_leaves >> self.num_layers;
        vec_from_words(2 * (self.num_leaves - cap_len), |p| {
            with_ctx(|c| check(c, unsafe { p2hot_batch_digests(self.batch, p) }, "p2hot_batch_digests"));
        })
    }
}

// ------------------------------------------------------------------------------------------------
// Bit-exact harness (SURVEY 8c "Parity status" / "Bit-exact vs CPU prover protocol"): every replaced body, CPU vs GPU, in one
// process on the same inputs.  Needs an MI355X and libp2hot.so:
//     P2HOT_LIB_DIR=/path/to/plonky2_amd cargo test --release --features p2hot p2hot:: -- --test-threads=1
// (`--test-threads=1`: `set_enabled` is process-wide.)  The 2^16 / 2^20 circuits are `#[ignore]`d: add `--ignored`.
// ------------------------------------------------------------------------------------------------
#[cfg(test)]
mod tests {
    use anyhow::Result;

    use super::*;
    use crate::field::types::Sample;
    use crate::fri::reduction_strategies::FriReductionStrategy;
    use crate::fri::FriConfig;
    use crate::gates::noop::NoopGate;
    use crate::iop::witness::{PartialWitness, WitnessWrite};
    use crate::plonk::circuit_builder::CircuitBuilder;
    use crate::plonk::circuit_data::{CircuitConfig, CircuitData, CommonCircuitData, VerifierOnlyCircuitData};
    use crate::plonk::config::PoseidonGoldilocksConfig;
    use crate::plonk::proof::ProofWithPublicInputs;
    use crate::plonk::prover::prove;
    use crate::util::serialization::Write;
    use crate::util::timing::TimingTree;
    use crate::util::{reverse_index_bits_in_place, transpose};

    const D: usize = 2;
    type C = PoseidonGoldilocksConfig;
    type F = <C as GenericConfig<D>>::F;
    type H = <C as GenericConfig<D>>::Hasher;
    type Batch = PolynomialBatch<F, C, D>;

    /// the unchanged reference bodies
    fn on_cpu<R>(f: impl FnOnce() -> R) -> R {
        set_enabled(false);
        let r = f();
        set_enabled(true);
        r
    }

    /// `gpu` is compared twice: as the shim built it (default mode: `leaves` empty, the flat buffer behind `get`) -- rows through
    /// `MerkleTree::get`, the serializer's bytes -- and then field by field after `leaves` is materialised from the flat buffer
    fn assert_same_batch(cpu: &Batch, gpu: &mut Batch) {
        assert_eq!(cpu.polynomials, gpu.polynomials, "polynomials (oracle.rs:32)");
        assert_eq!(cpu.merkle_tree.cap, gpu.merkle_tree.cap, "merkle_tree.cap");
        assert_eq!(cpu.merkle_tree.digests, gpu.merkle_tree.digests, "merkle_tree.digests (reference layout, merkle_tree.rs:50-57)");
        assert_eq!((cpu.degree_log, cpu.rate_bits, cpu.blinding), (gpu.degree_log, gpu.rate_bits, gpu.blinding));
        let big_n = cpu.merkle_tree.leaves.len();
        if !leaves_as_vecs() {
            assert!(gpu.merkle_tree.leaves.is_empty(), "default mode: the leaf matrix is ONE flat buffer behind MerkleTree::get");
        }
        assert_eq!(gpu.merkle_tree.device.as_ref().unwrap().num_leaves(), big_n);
        for i in [0usize, 1, big_n / 2, big_n - 1] {
            assert_eq!(cpu.merkle_tree.get(i), gpu.merkle_tree.get(i), "MerkleTree::get({i}) through the flat buffer");
            assert_eq!(cpu.merkle_tree.prove(i), gpu.merkle_tree.prove(i), "MerkleTree::prove({i}) with `leaves` empty");
            assert_eq!(cpu.get_lde_values(i, 1), gpu.get_lde_values(i, 1));
        }
        // 8f-4 wire formats: the reference serializer over the shim-built structs (util/serialization/mod.rs:1417-1431, :1744-1763);
        // under the feature it reads the rows through `get`, so the flat buffer serialises to the same bytes
        let (mut a, mut b) = (Vec::<u8>::new(), Vec::<u8>::new());
        a.write_polynomial_batch(cpu).unwrap();
        b.write_polynomial_batch(gpu).unwrap();
        assert_eq!(a, b, "write_polynomial_batch bytes");
        let (mut a, mut b) = (Vec::<u8>::new(), Vec::<u8>::new());
        a.write_merkle_tree(&cpu.merkle_tree).unwrap();
        b.write_merkle_tree(&gpu.merkle_tree).unwrap();
        assert_eq!(a, b, "write_merkle_tree bytes");
        // the field itself, rebuilt in parallel from the flat buffer (what P2HOT_LEAVES=vec does at commit time)
        if gpu.merkle_tree.leaves.is_empty() {
            gpu.merkle_tree.leaves = gpu.merkle_tree.device.as_ref().unwrap().leaves_as_vecs();
        }
        assert_eq!(cpu.merkle_tree.leaves, gpu.merkle_tree.leaves, "merkle_tree.leaves (LDE values, transposed, bit-reversed)");
        assert_eq!(cpu, &*gpu); // PartialEq (oracle.rs:29) over all of the above
    }

    /// a GPU-built batch (default mode: `leaves` empty, the matrix behind `device`) equals its own serialise / deserialise round
    /// trip (`leaves` populated, `device: None`) and the CPU-built batch, in both argument orders: `MerkleTree::eq` compares rows
    /// through `get` (what `examples/square_root.rs:152` `assert_eq!(data, data_from_bytes)` relies on for CircuitData)
    #[test]
    fn gpu_built_batch_equals_its_round_trip() {
        use crate::util::serialization::Buffer;
        use crate::util::serialization::Read;
        let (w, log_n, rate_bits, cap_height) = (7usize, 6usize, 3usize, 2usize);
        let values: Vec<PolynomialValues<F>> = (0..w).map(|_| PolynomialValues::new(F::rand_vec(1 << log_n))).collect();
        let cpu = on_cpu(|| Batch::from_values(values.clone(), rate_bits, false, cap_height, &mut TimingTree::default(), None));
        let gpu = Batch::from_values(values.clone(), rate_bits, false, cap_height, &mut TimingTree::default(), None);
        assert!(gpu.merkle_tree.device.is_some() && (leaves_as_vecs() || gpu.merkle_tree.leaves.is_empty()));
        let mut bytes = Vec::<u8>::new();
        bytes.write_polynomial_batch(&gpu).unwrap();
        let back: Batch = Buffer::new(&bytes).read_polynomial_batch().unwrap();
        assert!(back.merkle_tree.device.is_none() && !back.merkle_tree.leaves.is_empty());
        assert_eq!(gpu, back);
        assert_eq!(back, gpu);
        assert_eq!(cpu, gpu);
        assert_eq!(gpu, cpu);
        // and the comparison still tells two different GPU-built trees apart
        let mut other = values;
        other[0].values[1] += F::ONE;
        let gpu2 = Batch::from_values(other, rate_bits, false, cap_height, &mut TimingTree::default(), None);
        assert_ne!(gpu, gpu2);
    }

    /// the reference's failure panics stay panics under the feature (no abort through a poisoned context mutex): a zero
    /// denominator in the partial products panics with the reference's text, and the context works afterwards
    #[test]
    fn a_failure_panic_does_not_poison_the_context() {
        let r = std::panic::catch_unwind(|| with_ctx(|_| panic!("Tried to invert zero")));
        assert!(r.is_err());
        let values: Vec<PolynomialValues<F>> = (0..3).map(|_| PolynomialValues::new(F::rand_vec(8))).collect();
        let gpu = Batch::from_values(values, 1, false, 0, &mut TimingTree::default(), None); // locks the same mutex
        assert!(gpu.merkle_tree.device.is_some());
    }

    /// from_values and from_coeffs at the widths of a proof's four commitments (constants_sigmas ~84, wires 135, Zs 20, quotient 16)
    #[test]
    fn commit_matches_the_cpu_prover() {
        assert!(leaves_on_device() == false, "run the harness with P2HOT_LEAVES unset (or \"vec\"): it compares the whole leaf matrix");
        for &(w, log_n, rate_bits, cap_height) in &[(3usize, 5usize, 3usize, 4usize), (135, 12, 3, 4), (84, 12, 3, 4), (20, 12, 3, 4), (16, 13, 3, 4), (2, 10, 1, 0), (1, 0, 3, 0)] {
            let values: Vec<PolynomialValues<F>> = (0..w).map(|_| PolynomialValues::new(F::rand_vec(1 << log_n))).collect();
            let cpu = on_cpu(|| Batch::from_values(values.clone(), rate_bits, false, cap_height, &mut TimingTree::default(), None));
            let mut gpu = Batch::from_values(values.clone(), rate_bits, false, cap_height, &mut TimingTree::default(), None);
            assert!(gpu.merkle_tree.device.is_some(), "the p2hot body did not run");
            assert_same_batch(&cpu, &mut gpu);
            let polys: Vec<PolynomialCoeffs<F>> = values.into_iter().map(|v| PolynomialCoeffs::new(v.values)).collect();
            let cpu = on_cpu(|| Batch::from_coeffs(polys.clone(), rate_bits, false, cap_height, &mut TimingTree::default(), None));
            let mut gpu = Batch::from_coeffs(polys, rate_bits, false, cap_height, &mut TimingTree::default(), None);
            assert_same_batch(&cpu, &mut gpu);
            // MerkleTree::get / ::prove through the device handle
            let dev = gpu.merkle_tree.device.as_ref().unwrap();
            for i in [0usize, 1, (1 << (log_n + rate_bits)) - 1] {
                assert_eq!(dev.row(i), cpu.merkle_tree.get(i));
                assert_eq!(dev.path::<H>(i), cpu.merkle_tree.prove(i).siblings);
            }
        }
    }

    /// OpeningSet::new's eval_commitment (plonk/proof.rs:323-328) on the device vs the CPU closure
    #[test]
    fn eval_commitment_matches_the_cpu_horner() {
        let values: Vec<PolynomialValues<F>> = (0..7).map(|_| PolynomialValues::new(F::rand_vec(1 << 9))).collect();
        let batch = Batch::from_values(values, 3, false, 4, &mut TimingTree::default(), None);
        let z = <F as Extendable<D>>::Extension::rand();
        let gpu = eval_commitment::<F, C, D>(z, &batch).expect("the p2hot body did not run");
        let cpu: Vec<_> = batch.polynomials.iter().map(|p| p.to_extension::<D>().eval(z)).collect();
        assert_eq!(cpu, gpu);
        assert!(on_cpu(|| eval_commitment::<F, C, D>(z, &batch)).is_none());
    }

    /// blinding = true with the SAME salts on both sides: the CPU tree is assembled exactly as from_coeffs does it
    /// (lde_values + salts -> transpose -> reverse_index_bits -> MerkleTree::new, oracle.rs:91-103, :114-139)
    #[test]
    fn salted_commit_matches_the_cpu_prover() {
        for &(w, log_n, rate_bits, cap_height) in &[(5usize, 6usize, 3usize, 4usize), (135, 10, 3, 4), (2, 4, 1, 1)] {
            let polys: Vec<PolynomialCoeffs<F>> = (0..w).map(|_| PolynomialCoeffs::new(F::rand_vec(1 << log_n))).collect();
            let salts: Vec<Vec<F>> = (0..SALT_SIZE).map(|_| F::rand_vec(1 << (log_n + rate_bits))).collect();
            let mut lde_values = on_cpu(|| Batch::lde_values(&polys, rate_bits, false, None));
            lde_values.extend(salts.iter().cloned());
            let mut leaves = transpose(&lde_values);
            reverse_index_bits_in_place(&mut leaves);
            let cpu_tree = MerkleTree::<F, H>::new(leaves, cap_height);
            let gpu = commit_with_salts::<F, C, D>(&coeff_slices(&polys), rate_bits, cap_height, false, &salts);
            assert!(gpu.blinding);
            assert_eq!(gpu.polynomials, polys);
            assert_eq!(cpu_tree.cap, gpu.merkle_tree.cap);
            assert_eq!(cpu_tree.digests, gpu.merkle_tree.digests);
            assert_eq!(cpu_tree.leaves, gpu.merkle_tree.device.as_ref().unwrap().leaves_as_vecs());
            // get_lde_values strips the salt (oracle.rs:142-147)
            assert_eq!(gpu.get_lde_values(3, 1).len(), w);
        }
    }

    /// fri_committed_trees (fri/prover.rs:84-150), the starky entry: trees, final_poly and the transcript afterwards
    #[test]
    fn fri_commit_phase_matches_the_cpu_prover() {
        for &(log_n, rate_bits, cap_height, ref arity) in &[(8usize, 3usize, 4usize, vec![4usize]), (10, 1, 2, vec![1, 2, 1]), (6, 3, 0, vec![])] {
            let n = 1usize << log_n;
            let mut coeffs = <F as Extendable<D>>::Extension::rand_vec(n);
            coeffs.resize(n << rate_bits, <F as Extendable<D>>::Extension::ZERO); // "only the first 1/rate coefficients are non-zero"
            let coeffs = PolynomialCoeffs::new(coeffs);
            let fri_params = FriParams {
                config: FriConfig { rate_bits, cap_height, proof_of_work_bits: 4, reduction_strategy: FriReductionStrategy::Fixed(arity.clone()), num_query_rounds: 5 },
                hiding: false,
                degree_bits: log_n,
                reduction_arity_bits: arity.clone(),
            };
            let seed = F::rand_vec(7);
            let run = || {
                let mut challenger = Challenger::<F, H>::new();
                challenger.observe_elements(&seed);
                let out = crate::fri::prover::p2hot_fri_committed_trees_for_tests::<F, C, D>(&coeffs, &mut challenger, &fri_params);
                (out, challenger.get_n_challenges(4))
            };
            let ((cpu_trees, cpu_final), cpu_next) = on_cpu(run);
            let ((gpu_trees, gpu_final), gpu_next) = run();
            assert_eq!(cpu_trees, gpu_trees, "round trees (leaves, digests, caps)");
            assert_eq!(cpu_final, gpu_final, "final_poly");
            assert_eq!(cpu_next, gpu_next, "the transcript after the commit phase");
        }
    }

    fn dummy_circuit(config: &CircuitConfig, log2_size: usize) -> CircuitData<F, C, D> {
        // examples/bench_recursion.rs:78-105 (dummy_proof)
        let num_dummy_gates = match log2_size {
            0 | 1 => 0,
            2 => 1,
            n => (1 << (n - 1)) + 1,
        };
        let mut builder = CircuitBuilder::<F, D>::new(config.clone());
        for _ in 0..num_dummy_gates {
            builder.add_gate(NoopGate, vec![]);
        }
        builder.build::<C>()
    }

    type ProofTuple = (ProofWithPublicInputs<F, C, D>, VerifierOnlyCircuitData<C, D>, CommonCircuitData<F, D>);

    /// builds the circuit and proves it on both paths; everything the prover commits to and the proof bytes must agree
    /// (the CPU grind takes the smallest witness under the `p2hot` feature: fri/prover.rs, `find_first`)
    fn prove_both_ways(build: &dyn Fn() -> (CircuitData<F, C, D>, PartialWitness<F>)) -> Result<ProofTuple> {
        let (cpu_data, cpu_pw) = on_cpu(build);
        let (mut gpu_data, gpu_pw) = build();
        // CircuitBuilder::build commits constants + sigmas (circuit_builder.rs:1182-1191)
        assert_eq!(cpu_data.verifier_only.constants_sigmas_cap, gpu_data.verifier_only.constants_sigmas_cap);
        assert_eq!(cpu_data.verifier_only.circuit_digest, gpu_data.verifier_only.circuit_digest);
        assert_same_batch(&cpu_data.prover_only.constants_sigmas_commitment, &mut gpu_data.prover_only.constants_sigmas_commitment);
        let cpu_proof = on_cpu(|| prove::<F, C, D>(&cpu_data.prover_only, &cpu_data.common, cpu_pw, &mut TimingTree::default()))?;
        let gpu_proof = prove::<F, C, D>(&gpu_data.prover_only, &gpu_data.common, gpu_pw, &mut TimingTree::default())?;
        let (c, g) = (&cpu_proof.proof, &gpu_proof.proof);
        assert_eq!(c.wires_cap, g.wires_cap, "wires_cap");
        assert_eq!(c.plonk_zs_partial_products_cap, g.plonk_zs_partial_products_cap, "plonk_zs_partial_products_cap");
        assert_eq!(c.quotient_polys_cap, g.quotient_polys_cap, "quotient_polys_cap");
        assert_eq!(c.openings, g.openings, "OpeningSet");
        assert_eq!(c.opening_proof.commit_phase_merkle_caps, g.opening_proof.commit_phase_merkle_caps, "FRI commit_phase_merkle_caps");
        assert_eq!(c.opening_proof.final_poly, g.opening_proof.final_poly, "FRI final_poly");
        assert_eq!(c.opening_proof.pow_witness, g.opening_proof.pow_witness, "pow_witness (smallest on both sides)");
        assert_eq!(c.opening_proof.query_round_proofs, g.opening_proof.query_round_proofs, "FRI query rounds");
        assert_eq!(cpu_proof.to_bytes(), gpu_proof.to_bytes(), "proof.to_bytes()");
        gpu_data.verify(gpu_proof.clone())?;
        cpu_data.verify(gpu_proof.clone())?;
        Ok((gpu_proof, gpu_data.verifier_only, gpu_data.common))
    }

    fn dummy_both_ways(log2_size: usize) -> Result<ProofTuple> {
        let config = CircuitConfig::standard_recursion_config();
        prove_both_ways(&|| (dummy_circuit(&config, log2_size), PartialWitness::new()))
    }

    /// one recursion layer over `inner` (examples/bench_recursion.rs:200-245), proved both ways
    fn recursive_both_ways(inner: &ProofTuple, min_degree_bits: Option<usize>) -> Result<ProofTuple> {
        let config = CircuitConfig::standard_recursion_config();
        let (inner_proof, inner_vd, inner_cd) = inner;
        prove_both_ways(&|| {
            let mut builder = CircuitBuilder::<F, D>::new(config.clone());
            let pt = builder.add_virtual_proof_with_pis(inner_cd);
            let inner_data = builder.add_virtual_verifier_data(inner_cd.config.fri_config.cap_height);
            builder.verify_proof::<C>(&pt, &inner_data, inner_cd);
            if let Some(min_degree_bits) = min_degree_bits {
                let min_gates = (1 << (min_degree_bits - 1)) + 1;
                for _ in builder.num_gates()..min_gates {
                    builder.add_gate(NoopGate, vec![]);
                }
            }
            let data = builder.build::<C>();
            let mut pw = PartialWitness::new();
            pw.set_proof_with_pis_target(&pt, inner_proof).unwrap();
            pw.set_verifier_data_target(&inner_data, inner_vd).unwrap();
            (data, pw)
        })
    }

    #[test]
    fn prove_2_12_matches_the_cpu_prover() -> Result<()> {
        dummy_both_ways(12).map(|_| ())
    }

    /// the 3-proof chain of bench_recursion (:317-345) at its recursion-threshold size: inner -> middle -> outer
    #[test]
    fn recursion_chain_matches_the_cpu_prover() -> Result<()> {
        let inner = dummy_both_ways(12)?;
        let middle = recursive_both_ways(&inner, None)?;
        recursive_both_ways(&middle, None).map(|_| ())
    }

    #[test]
    #[ignore = "BASELINE C2: a 2^16-gate circuit (about a minute on the CPU side)"]
    fn prove_2_16_matches_the_cpu_prover() -> Result<()> {
        dummy_both_ways(16).map(|_| ())
    }

    #[test]
    #[ignore = "BASELINE C3: bench_recursion --size 20 (2^20 gates; the CPU side takes minutes and ~40 GB)"]
    fn recursion_chain_2_20_matches_the_cpu_prover() -> Result<()> {
        let inner = dummy_both_ways(20)?;
        let middle = recursive_both_ways(&inner, None)?;
        recursive_both_ways(&middle, None).map(|_| ())
    }

    /// zero-knowledge configuration: the salts (and the blinding gates' witness values) are fresh randomness on each run,
    /// so the check is the reference's own acceptance test -- the GPU-built proof verifies -- plus the shapes
    #[test]
    fn zk_config_proof_verifies() -> Result<()> {
        let config = CircuitConfig::standard_recursion_zk_config();
        let data = dummy_circuit(&config, 12);
        assert!(!data.prover_only.constants_sigmas_commitment.blinding); // PlonkOracle::CONSTANTS_SIGMAS
        let proof = prove::<F, C, D>(&data.prover_only, &data.common, PartialWitness::new(), &mut TimingTree::default())?;
        let q = &proof.proof.opening_proof.query_round_proofs[0].initial_trees_proof.evals_proofs;
        assert_eq!(q[1].0.len(), config.num_wires + SALT_SIZE, "the opened wires leaf carries its salt");
        data.verify(proof)
    }
}
