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
        self.merkle_tree = MerkleTree(None, engine.host(digests), engine.host(cap), cap_height, n_leaves=N,
                                      leaf_getter=lambda idx: engine.host(engine.gather_rows(lde, idx)) if W else
                                      np.zeros((len(idx), 0), dtype=np.uint64))

    @property
    def polynomials(self):
        """coefficient form, host [W][n] (oracle.rs:32)"""
        return self.engine.host(self._coeffs)

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
