"""ctypes binding of libp2hot (include/p2hot.h).

The product library is ``plonky2_amd/libp2hot.so`` -- built by ``__graft_entry__.build()`` with
``hipcc --offload-arch=gfx950`` -- and there is NO CPU fallback: if it is missing, or no GPU is
visible, importing the engine raises.  ``load(path)`` exists so the test-suite can bind the
test-only kernel-emulator build (tests/emu) through the very same signatures.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
PRODUCT_SO = os.path.join(_HERE, "libp2hot.so")

OK, EINVAL, ENOMEM, EHIP, EUNSUPPORTED, EBUSY, ECOMM = 0, 1, 2, 3, 4, 5, 6
_ERR_NAMES = {EINVAL: "EINVAL", ENOMEM: "ENOMEM", EHIP: "EHIP", EUNSUPPORTED: "EUNSUPPORTED", EBUSY: "EBUSY",
              ECOMM: "ECOMM"}
KEEP_VALUES = 1
COEFFS_PER_COLUMN = 2
LEAVES_ASYNC = 4
LEAVES_NATURAL = 8

vp = C.c_void_p
sz = C.c_size_t
u = C.c_uint
u64 = C.c_uint64
i = C.c_int


class ChallengerState(C.Structure):
    _fields_ = [("sponge_state", u64 * 12), ("input_buffer", u64 * 8), ("output_buffer", u64 * 8),
                ("input_len", C.c_uint32), ("output_len", C.c_uint32)]


class FriBatchInfo(C.Structure):  # p2hot_fri_batch_info
    _fields_ = [("point", u64 * 2), ("oracle_index", C.POINTER(C.c_uint32)), ("poly_index", C.POINTER(C.c_uint32)),
                ("n_polys", sz)]


class FriParams(C.Structure):  # p2hot_fri_params
    _fields_ = [("rate_bits", u), ("cap_height", u), ("proof_of_work_bits", u), ("num_query_rounds", u),
                ("reduction_arity_bits", C.POINTER(u)), ("n_reduction_rounds", u), ("hiding", i),
                ("max_num_query_steps", u), ("final_poly_coeff_len", sz)]


class FriProof(C.Structure):  # p2hot_fri_proof
    _fields_ = [("commit_phase_merkle_caps", vp), ("final_poly", vp), ("pow_witness", u64), ("query_indices", vp),
                ("initial_leaves", vp), ("initial_paths", vp), ("step_evals", vp), ("step_paths", vp)]


class FriProofLayout(C.Structure):  # p2hot_fri_proof_layout
    _fields_ = [("caps_words", sz), ("final_poly_words", sz), ("initial_leaves_words", sz), ("initial_paths_words", sz),
                ("step_evals_words", sz), ("step_paths_words", sz)]


# p2hot_allgather_fn: (user, d_base, offsets, world, bytes, hip_stream) -> int
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, vp, vp, C.POINTER(sz), C.c_int, sz, vp)

# name -> (restype, argtypes); every symbol include/p2hot.h declares
SIGNATURES = {
    "p2hot_ctx_create": (i, [i, vp, C.POINTER(vp)]),
    "p2hot_ctx_destroy": (None, [vp]),
    "p2hot_ctx_set_stream": (i, [vp, vp]),
    "p2hot_ctx_sync": (i, [vp]),
    "p2hot_last_error": (C.c_char_p, [vp]),
    "p2hot_version": (C.c_char_p, []),
    "p2hot_is_emulated": (i, []),
    "p2hot_profile_enable": (i, [vp, i]),
    "p2hot_tune_ntt": (i, [vp, i]),
    "p2hot_tune_overlap": (i, [vp, i]),
    "p2hot_tune_quad": (i, [vp, sz]),
    "p2hot_tune_row": (i, [vp, sz]),
    "p2hot_profile_json": (C.c_char_p, [vp, i]),
    "p2hot_num_digests": (sz, [u, u]),
    "p2hot_fft_dev": (i, [vp, vp, sz, sz, u]),
    "p2hot_ifft_dev": (i, [vp, vp, sz, sz, u]),
    "p2hot_coset_ifft_dev": (i, [vp, vp, sz, sz, u, u64]),
    "p2hot_coset_lde_dev": (i, [vp, vp, sz, sz, u, u, u64, sz, sz, vp, sz]),
    "p2hot_transpose_dev": (i, [vp, vp, sz, sz, sz, vp]),
    "p2hot_reverse_index_bits_dev": (i, [vp, vp, vp, sz, sz, u]),
    "p2hot_poseidon_permute_dev": (i, [vp, vp, sz]),
    "p2hot_merkle_dev": (i, [vp, vp, i, sz, sz, u, u, sz, sz, vp, vp]),
    "p2hot_field_selftest_dev": (i, [vp, vp, vp, sz, vp]),
    "p2hot_gather_rows_dev": (i, [vp, vp, sz, sz, sz, vp, sz, vp]),
    "p2hot_commit_dev": (i, [vp, vp, sz, sz, u, u, u, i, sz, sz, vp, sz, vp, sz, vp, vp, vp]),
    "p2hot_challenger_create": (i, [vp, C.POINTER(vp)]),
    "p2hot_challenger_destroy": (None, [vp]),
    "p2hot_challenger_load": (i, [vp, C.POINTER(ChallengerState)]),
    "p2hot_challenger_store": (i, [vp, C.POINTER(ChallengerState)]),
    "p2hot_challenger_step": (i, [vp, vp, sz, vp, sz]),
    "p2hot_fri_commit": (i, [vp, vp, u, u, u, C.POINTER(u), u, u, sz, vp, vp, vp, vp, vp, vp]),
    "p2hot_fri_commit_dev": (i, [vp, vp, u, u, u, C.POINTER(u), u, u, sz, vp, vp, vp, i, vp, vp, vp]),
    "p2hot_fri_final_poly_dev": (i, [vp, vp, C.POINTER(sz), sz, vp, vp, u, vp]),
    "p2hot_eval_polys_dev": (i, [vp, vp, sz, u, vp, sz, vp]),
    "p2hot_merkle_paths_dev": (i, [vp, vp, u, u, vp, sz, vp]),
    "p2hot_partial_products_dev": (i, [vp, vp, sz, vp, sz, vp, u, u, u, vp, vp, u, vp, sz]),
    "p2hot_fri_pow": (i, [vp, vp, u, C.POINTER(u64)]),
    "p2hot_commit": (i, [vp, C.POINTER(vp), sz, u, u, u, i, u, vp, vp, vp, vp, C.POINTER(vp)]),
    "p2hot_commit_salted": (i, [vp, C.POINTER(vp), sz, u, u, u, i, u, C.POINTER(vp), sz, vp, vp, vp, vp, C.POINTER(vp)]),
    "p2hot_commit_cols": (i, [vp, vp, u, u, i, u, vp, vp, vp, vp, C.POINTER(vp)]),
    "p2hot_commit_many": (i, [vp, C.POINTER(vp), sz, sz, u, u, u, i, vp, vp, vp, C.POINTER(vp)]),
    "p2hot_commit_many_dev": (i, [vp, vp, sz, sz, u, u, u, i, vp, vp, vp]),
    "p2hot_batch_wrap_dev": (i, [vp, vp, vp, vp, sz, u, u, u, C.POINTER(vp)]),
    "p2hot_batch_width": (sz, [vp]),
    "p2hot_batch_leaf_width": (sz, [vp]),
    "p2hot_batch_degree_log": (u, [vp]),
    "p2hot_batch_coeffs": (i, [vp, sz, sz, vp]),
    "p2hot_batch_rows": (i, [vp, vp, sz, vp]),
    "p2hot_batch_paths": (i, [vp, vp, sz, vp]),
    "p2hot_batch_digests": (i, [vp, vp]),
    "p2hot_batch_leaves_wait": (i, [vp, sz, sz]),
    "p2hot_batch_leaves_block_rows": (sz, [vp]),
    "p2hot_batch_values": (i, [vp, C.POINTER(vp)]),
    "p2hot_batch_subgroup_values": (i, [vp, sz, sz, C.POINTER(vp)]),
    "p2hot_batch_free": (None, [vp]),
    "p2hot_ctx_trim": (i, [vp]),
    "p2hot_cols_upload": (i, [vp, C.POINTER(vp), sz, u, C.POINTER(vp)]),
    "p2hot_cols_download": (i, [vp, sz, sz, vp]),
    "p2hot_cols_width": (sz, [vp]),
    "p2hot_cols_degree_log": (u, [vp]),
    "p2hot_cols_free": (None, [vp]),
    "p2hot_eval_openings": (i, [vp, C.POINTER(vp), sz, vp, sz, vp]),
    "p2hot_host_alloc": (i, [vp, sz, C.POINTER(vp)]),
    "p2hot_host_free": (None, [vp, vp]),
    "p2hot_fri_proof_sizes": (i, [C.POINTER(vp), sz, C.POINTER(FriParams), C.POINTER(FriProofLayout)]),
    "p2hot_prove_openings": (i, [vp, C.POINTER(FriBatchInfo), sz, C.POINTER(vp), sz, vp, C.POINTER(FriParams),
                                 C.POINTER(FriProof)]),
    "p2hot_prove_openings_many": (i, [vp, sz, C.POINTER(C.POINTER(FriBatchInfo)), C.POINTER(sz), C.POINTER(vp), sz, C.POINTER(vp),
                                      C.POINTER(FriParams), C.POINTER(FriProof)]),
    "p2hot_partial_products": (i, [vp, vp, sz, vp, sz, vp, u, u, vp, vp, u, vp, C.POINTER(vp)]),
    "p2hot_quotient_chunks": (i, [vp, C.POINTER(vp), u, u, u, C.POINTER(vp)]),
    "p2hot_quotient_polys": (i, [vp, vp, vp, sz, vp, vp, u, u, vp, vp, vp, u, C.POINTER(vp), vp, C.POINTER(vp)]),
    "p2hot_comm_unique_id": (i, [vp]),
    "p2hot_comm_create_rccl": (i, [vp, i, i, vp, C.POINTER(vp)]),
    "p2hot_comm_create_callback": (i, [vp, i, i, ALLGATHER_FN, vp, C.POINTER(vp)]),
    "p2hot_comm_destroy": (None, [vp]),
    "p2hot_comm_rank": (i, [vp]),
    "p2hot_comm_world": (i, [vp]),
    "p2hot_comm_selftest": (i, [vp, sz]),
    "p2hot_comm_exchange_mode": (i, [vp]),
    "p2hot_rccl_info": (i, [C.c_char_p, sz, C.POINTER(i)]),
    "p2hot_shard_columns": (i, [sz, i, i, C.POINTER(sz), C.POINTER(sz)]),
    "p2hot_commit_sharded_dev": (i, [vp, vp, vp, sz, sz, u, u, u, i, i, u, vp, vp, sz, vp, vp, vp]),
    "p2hot_group_create": (i, [i, C.POINTER(i), C.POINTER(vp)]),
    "p2hot_group_destroy": (None, [vp]),
    "p2hot_group_size": (i, [vp]),
    "p2hot_group_ctx": (vp, [vp, i]),
    "p2hot_group_uses_rccl": (i, [vp]),
    "p2hot_group_exchange_mode": (i, [vp]),
    "p2hot_group_last_error": (C.c_char_p, [vp]),
    "p2hot_group_commit": (i, [vp, C.POINTER(vp), sz, u, u, u, i, i, u, vp, vp, vp, vp, C.POINTER(vp)]),
    "p2hot_sharded_batch_open": (i, [vp, vp, sz, vp, vp]),
    "p2hot_sharded_batch_free": (None, [vp]),
    "p2hot_group_eval_openings": (i, [vp, C.POINTER(vp), sz, vp, sz, vp]),
    "p2hot_group_fri_proof_sizes": (i, [C.POINTER(vp), sz, C.POINTER(FriParams), C.POINTER(FriProofLayout)]),
    "p2hot_group_prove_openings": (i, [vp, C.POINTER(FriBatchInfo), sz, C.POINTER(vp), sz, vp, C.POINTER(FriParams),
                                       C.POINTER(FriProof)]),
}


class P2HotError(RuntimeError):
    def __init__(self, code, text):
        self.code = code
        super().__init__("libp2hot %s: %s" % (_ERR_NAMES.get(code, code), text))


def load(path):
    """Bind a libp2hot build; raises if any declared symbol is missing."""
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return lib


_product = None


def product():
    """The HIP build.  Fails loudly when it has not been built -- there is no fallback."""
    global _product
    if _product is None:
        if not os.path.exists(PRODUCT_SO):
            raise RuntimeError(
                "plonky2_amd/libp2hot.so is missing: build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
                "plonky2_amd has no CPU fallback.")
        # PyTorch-ROCm ships its own libamdhip64; it must be the HIP runtime this process uses, so it is
        # loaded first and libp2hot's libamdhip64 dependency resolves to the already-loaded copy.
        import torch  # noqa: F401
        _product = load(PRODUCT_SO)
        if _product.p2hot_is_emulated():
            raise RuntimeError("plonky2_amd/libp2hot.so is an emulator build; refusing to use it as the product")
    return _product
