"""FRI commit phase -- mirror of plonky2/src/fri/prover.rs:84-202 (fri_committed_trees,
fri_proof_of_work) on the GPU.  The query phase (prover.rs:204-258) is SURVEY 8(f) "next".
"""
import ctypes as C

import numpy as np

from ..engine import default_engine
from ..hash.merkle_tree import MerkleTree


def fri_committed_trees(coeffs, challenger, rate_bits, cap_height, reduction_arity_bits, engine=None,
                        final_poly_coeff_len=None, max_num_query_steps=None):
    """fri_committed_trees (prover.rs:84-150).

    coeffs: [n][2] uint64 -- the nonzero extension coefficients of final_poly (the reference passes
    them zero-padded to N = n << rate_bits together with their coset-FFT `values`; both are implicit
    here: the LDE runs on the GPU).  challenger: plonky2_amd.iop.challenger.Challenger, advanced
    like the reference.  Returns (trees, final_coeffs, betas): trees are MerkleTree objects whose
    leaves are rows of 2*arity words.
    """
    eng = engine or challenger.engine or default_engine()
    coeffs = np.ascontiguousarray(np.asarray(coeffs, dtype=np.uint64))
    if coeffs.ndim != 2 or coeffs.shape[1] != 2:
        raise ValueError("coeffs must be [n][2]")
    n = coeffs.shape[0]
    log_n = n.bit_length() - 1
    if n != 1 << log_n:
        raise ValueError("coefficient count must be a power of two")
    return _commit(coeffs, None, log_n, challenger, rate_bits, cap_height, reduction_arity_bits, eng,
                   max_num_query_steps or 0, final_poly_coeff_len or 0)   # prover.rs:89-90: None -> 0


def _commit(coeffs, planes, log_n, challenger, rate_bits, cap_height, reduction_arity_bits, engine,
            max_num_query_steps=0, final_poly_coeff_len=0):
    eng = engine or challenger.engine or default_engine()
    n = 1 << log_n
    arity = [int(a) for a in reduction_arity_bits]
    N = n << rate_bits
    ncap = 1 << cap_height
    m, sizes = N, []
    for ab in arity:
        nl = m >> ab
        sizes.append((m, nl, max(0, 2 * (nl - ncap))))
        m >>= ab
    n_final = max(m >> rate_bits, 0)
    n_leaf_words = max(1, 2 * sum(s[0] for s in sizes))
    # host-coefficient entry: leaves come back to the host; device-plane entry: they stay on the GPU
    leaves = np.zeros(n_leaf_words, dtype=np.uint64) if planes is None else eng.mem.empty(n_leaf_words)
    n_dig_words = max(1, 4 * sum(s[2] for s in sizes))
    # device-plane entry: the digest arrays stay on the GPU as well (paths come from p2hot_merkle_paths_dev)
    digests = np.zeros(n_dig_words, dtype=np.uint64) if planes is None else eng.mem.empty(n_dig_words)
    caps = np.zeros(max(1, 4 * ncap * len(sizes)), dtype=np.uint64)
    betas = np.zeros((max(1, len(sizes)), 2), dtype=np.uint64)
    final = np.zeros((max(1, n_final), 2), dtype=np.uint64)
    ab = (C.c_uint * max(1, len(arity)))(*arity)
    if planes is None:
        eng.check(eng.lib.p2hot_fri_commit(eng.ctx, coeffs.ctypes.data, log_n, rate_bits, cap_height, ab, len(arity),
                                           max_num_query_steps, final_poly_coeff_len, challenger._h, leaves.ctypes.data,
                                           digests.ctypes.data, caps.ctypes.data, betas.ctypes.data, final.ctypes.data))
    else:
        eng.check(eng.lib.p2hot_fri_commit_dev(eng.ctx, eng.ptr(planes), log_n, rate_bits, cap_height, ab, len(arity),
                                               max_num_query_steps, final_poly_coeff_len, challenger._h, eng.ptr(leaves),
                                               eng.ptr(digests), 1, caps.ctypes.data, betas.ctypes.data, final.ctypes.data))
    trees, lo, do = [], 0, 0
    for i, (mi, nl, nd) in enumerate(sizes):
        lv = leaves[lo:lo + 2 * mi].reshape(nl, -1)
        if planes is None:
            tree = MerkleTree(leaves=lv, digests=digests[do:do + 4 * nd].reshape(nd, 4),
                              cap=caps[4 * ncap * i:4 * ncap * (i + 1)].reshape(ncap, 4), cap_height=cap_height)
        else:  # rows are fetched from the device on demand (MerkleTree::get of the few queried leaves)
            tree = MerkleTree(leaves=None, digests=digests[do:do + 4 * nd].reshape(nd, 4) if nd else np.zeros((0, 4), np.uint64),
                              cap=caps[4 * ncap * i:4 * ncap * (i + 1)].reshape(ncap, 4), cap_height=cap_height, n_leaves=nl,
                              leaf_getter=(lambda idx, lv=lv: eng.host(lv[np.asarray(idx, dtype=np.int64)])),
                              engine=eng if nd else None)
        trees.append(tree)
        lo += 2 * mi
        do += 4 * nd
    return trees, final[:n_final], betas[:len(sizes)]


def fri_committed_trees_device(planes, log_n, challenger, rate_bits, cap_height, reduction_arity_bits, engine=None,
                               final_poly_coeff_len=None, max_num_query_steps=None):
    """fri_committed_trees on coefficients that are already on the device as planes [2][n]
    (e.g. plonky2_amd.fri.oracle.final_poly_device)."""
    return _commit(None, planes, log_n, challenger, rate_bits, cap_height, reduction_arity_bits, engine,
                   max_num_query_steps or 0, final_poly_coeff_len or 0)


def fri_proof_of_work(challenger, proof_of_work_bits, engine=None):
    """fri_proof_of_work (prover.rs:153-202); deterministic smallest witness."""
    eng = engine or challenger.engine or default_engine()
    w = C.c_uint64()
    eng.check(eng.lib.p2hot_fri_pow(eng.ctx, challenger._h, proof_of_work_bits, C.byref(w)))
    return int(w.value)
