"""PolynomialBatch -- mirror of plonky2/src/fri/oracle.rs:30-147 with the pipeline on the GPU.

from_values / from_coeffs keep the reference's signature (values, rate_bits, blinding, cap_height,
timing, fft_root_table); `timing` and `fft_root_table` are accepted and ignored (the GPU path has
its own twiddle tables), `blinding=True` raises (salts come from OsRng in the reference,
oracle.rs:133-137).  The LDE matrix stays on the GPU in column-major form; `merkle_tree.leaves`
and `get_lde_values` fetch rows on demand.
"""
import numpy as np

from ..engine import default_engine
from ..hash.merkle_tree import MerkleTree


class PolynomialBatch:
    def __init__(self, engine, coeffs, lde, digests, cap, degree_log, rate_bits, cap_height, blinding=False):
        self.engine = engine
        self._coeffs = coeffs      # device [W][n]
        self.lde = lde             # device [W][N], rows in committed (bit-reversed) order
        self.degree_log = degree_log
        self.rate_bits = rate_bits
        self.blinding = blinding
        W = coeffs.shape[0]
        N = 1 << (degree_log + rate_bits)
        self.merkle_tree = MerkleTree(None, digests, engine.host(cap), cap_height, n_leaves=N, engine=engine,
                                      leaf_getter=lambda idx: engine.host(engine.gather_rows(lde, idx)) if W else
                                      np.zeros((len(idx), 0), dtype=np.uint64))

    @property
    def polynomials(self):
        """coefficient form, host [W][n] (oracle.rs:32), canonical representatives"""
        a = self.engine.host(self._coeffs)
        P = np.uint64(0xFFFFFFFF00000001)
        return np.where(a >= P, a - P, a)

    @classmethod
    def from_values(cls, values, rate_bits, blinding, cap_height, timing=None, fft_root_table=None, engine=None):
        """oracle.rs:57-79.  values: [W][n] (host ndarray or device buffer), values on H_n."""
        return cls._build(values, rate_bits, blinding, cap_height, True, engine)

    @classmethod
    def from_coeffs(cls, polynomials, rate_bits, blinding, cap_height, timing=None, fft_root_table=None, engine=None):
        """oracle.rs:82-112.  polynomials: [W][n] coefficients."""
        return cls._build(polynomials, rate_bits, blinding, cap_height, False, engine)

    @classmethod
    def _build(cls, cols, rate_bits, blinding, cap_height, is_values, engine):
        eng = engine or default_engine()
        if blinding:
            raise NotImplementedError("blinding=True draws salts from OsRng in the reference (oracle.rs:133-137); "
                                      "use the CPU prover for zero-knowledge configs")
        d_cols = eng.dev(cols)
        if d_cols.ndim != 2:
            raise ValueError("expected [W][n]")
        W, n = d_cols.shape
        log_n = int(n).bit_length() - 1
        if n != 1 << log_n:
            raise ValueError("polynomial length must be a power of two")  # log2_strict, oracle.rs:88
        r = eng.commit(d_cols, log_n, rate_bits, cap_height, is_values)
        return cls(eng, r["coeffs"], r["lde"], r["digests"], r["cap"], log_n, rate_bits, cap_height, blinding)

    def get_lde_values(self, index, step=1):
        """oracle.rs:142-147: row reverse_bits(index*step, degree_log + rate_bits) of the leaf matrix"""
        bits = self.degree_log + self.rate_bits
        i = index * step
        rev = int(format(i, "0%db" % bits)[::-1], 2) if bits else 0
        return self.merkle_tree.get(rev)


# ------------------------------------------------------------------ prove_openings (oracle.rs:176-237)
class FriBatchInfo:
    """fri/structure.rs FriBatchInfo: an opening point and the (oracle_index, polynomial_index) pairs opened there"""

    def __init__(self, point, polynomials):
        self.point = [int(point[0]), int(point[1])]
        self.polynomials = [(int(o), int(p)) for o, p in polynomials]


def eval_openings(oracles, points, engine=None):
    """OpeningSet::new (plonk/proof.rs:314-327): every polynomial of every oracle at each extension point.
    Returns a list (per oracle) of arrays [n_points][W][2]."""
    eng = engine or oracles[0].engine
    pts = np.ascontiguousarray(np.asarray(points, dtype=np.uint64).reshape(-1, 2))
    out = []
    for o in oracles:
        co = o._coeffs
        W, n = co.shape
        table = eng.dev(np.asarray([eng.mem.ptr(co) + 8 * j * n for j in range(W)] or [0], dtype=np.uint64))
        res = eng.mem.zeros(len(pts), max(W, 1), 2)
        eng.check(eng.lib.p2hot_eval_polys_dev(eng.ctx, eng.ptr(table), W, o.degree_log, pts.ctypes.data, len(pts),
                                               eng.ptr(res)))
        out.append(eng.host(res)[:, :W])
    return out


def final_poly_device(batches, oracles, alpha, engine=None):
    """The final_poly of prove_openings (oracle.rs:186-213) on the GPU: per batch reduce_polys_base, divide_by_linear,
    shift_poly + accumulate.  Returns device planes [2][n] (component 0 plane, component 1 plane)."""
    import ctypes as C
    eng = engine or oracles[0].engine
    n = 1 << oracles[0].degree_log
    ptrs, offsets = [], [0]
    for b in batches:
        for (oi, pi) in b.polynomials:
            co = oracles[oi]._coeffs
            if co.shape[1] != n:
                raise ValueError("all oracles must have the same degree")
            ptrs.append(eng.mem.ptr(co) + 8 * pi * co.shape[1])
        offsets.append(len(ptrs))
    table = eng.dev(np.asarray(ptrs if ptrs else [0], dtype=np.uint64))
    points = np.ascontiguousarray(np.asarray([b.point for b in batches] or [[0, 0]], dtype=np.uint64))
    al = np.asarray(alpha, dtype=np.uint64)
    out = eng.mem.zeros(2, n)
    offs = (C.c_size_t * len(offsets))(*offsets)
    eng.check(eng.lib.p2hot_fri_final_poly_dev(eng.ctx, eng.ptr(table), offs, len(batches), points.ctypes.data,
                                               al.ctypes.data, oracles[0].degree_log, eng.ptr(out)))
    return out


def prove_openings(batches, oracles, challenger, rate_bits, cap_height, reduction_arity_bits, proof_of_work_bits,
                   num_query_rounds, engine=None, timing=None):
    """PolynomialBatch::prove_openings + fri_proof (oracle.rs:176-237, fri/prover.rs:24-82) with everything
    but the transcript bookkeeping on the GPU.  Returns a dict shaped like FriProof:
      commit_phase_merkle_caps, query_round_proofs [{initial_trees_proof: [(leaf, siblings)...], steps: [(evals, siblings)...]}],
      final_poly [[c0, c1]...], pow_witness.
    `timing` (a dict, the reference's TimingTree argument): filled with synchronised wall milliseconds per stage."""
    import time
    from .prover import fri_committed_trees_device, fri_proof_of_work
    eng = engine or oracles[0].engine
    t_last = [time.perf_counter()]

    def lap(label):
        if timing is not None:
            eng.sync()
            now = time.perf_counter()
            timing[label] = timing.get(label, 0.0) + (now - t_last[0]) * 1e3
            t_last[0] = now

    alpha = challenger.get_extension_challenge()                      # oracle.rs:186
    planes = final_poly_device(batches, oracles, alpha, eng)
    lap("reduce + divide_by_linear (final_poly)")
    log_n = oracles[0].degree_log
    trees, final, _betas = fri_committed_trees_device(planes, log_n, challenger, rate_bits, cap_height,
                                                      reduction_arity_bits, eng)   # prover.rs:40-51
    lap("final FFT + fold codewords in the commitment phase")
    pow_witness = fri_proof_of_work(challenger, proof_of_work_bits, eng)           # prover.rs:53-58
    lap("find proof-of-work witness")
    lde_size = 1 << (log_n + rate_bits)
    xs = [rand % lde_size for rand in challenger.get_n_challenges(num_query_rounds)]   # prover.rs:215-220
    # initial trees: one batched row gather and one batched path gather per oracle, on the device (prover.rs:238-241)
    idx = np.asarray(xs, dtype=np.uint64)
    init_rows = [o.merkle_tree._getter(idx) if o.merkle_tree._leaves is None else o.merkle_tree.leaves[idx.astype(np.int64)]
                 for o in oracles]
    init_paths = [o.merkle_tree.prove_many(idx) for o in oracles]
    # FRI trees: one batched row fetch + one batched path walk per round (prover.rs:242-253 for all queries at once)
    step_rows, step_paths, idx_r = [], [], idx.copy()
    for i, tree in enumerate(trees):
        idx_r = idx_r >> np.uint64(reduction_arity_bits[i])
        rows = tree._getter(idx_r) if tree._leaves is None else tree.leaves[idx_r.astype(np.int64)]
        step_rows.append(np.asarray(rows).reshape(len(xs), -1, 2))
        step_paths.append(tree.prove_many(idx_r))
    queries = []
    for q in range(len(xs)):
        initial = [(init_rows[oi][q], init_paths[oi][q]) for oi in range(len(oracles))]
        steps = [(step_rows[i][q], step_paths[i][q]) for i in range(len(trees))]
        queries.append({"initial_trees_proof": initial, "steps": steps})
    lap("produce batch opening proof: query rounds")
    return {"commit_phase_merkle_caps": [t.cap.entries for t in trees], "query_round_proofs": queries,
            "final_poly": final, "pow_witness": pow_witness}
