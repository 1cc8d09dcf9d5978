"""PolynomialBatch -- mirror of plonky2/src/fri/oracle.rs:30-237 over libp2hot.

from_values / from_coeffs keep the reference's signature (values, rate_bits, blinding, cap_height,
timing, fft_root_table); `timing` and `fft_root_table` are accepted and ignored (the GPU path has
its own twiddle tables), `blinding=True` takes the caller's salt vectors (they come from OsRng in the reference,
oracle.rs:133-137).  Every batch is a `p2hot_batch` handle of the library: host arrays go through
the host-pointer entry point (p2hot_commit -- what the Rust shim calls), device buffers through
p2hot_commit_dev + p2hot_batch_wrap_dev.  The LDE matrix stays on the GPU in column-major form;
`merkle_tree.leaves` and `get_lde_values` fetch rows on demand.

OpeningSet evaluation and prove_openings are ONE library call each (p2hot_eval_openings,
p2hot_prove_openings): this module only marshals arguments and reshapes the flat result buffers.
"""
import ctypes as C

import numpy as np

from .. import _lib
from ..engine import default_engine
from ..hash.merkle_tree import MerkleTree

P = 0xFFFFFFFF00000001


class _BatchHandle:
    """Owns one p2hot_batch* and the lazy fetches on it.  PolynomialBatch and its MerkleTree both point HERE, not at
    each other: no reference cycle, so a dropped batch returns its 9 GB LDE block immediately (refcount, not the
    cycle collector) -- a cycle cost a fresh hipMalloc of the LDE matrix per commit (310 ms instead of 67)."""

    def __init__(self, engine, handle, W, degree_log, rate_bits, cap_height, keepalive=(), salt=0):
        self.engine, self.h, self.W = engine, handle, W + salt  # leaf width
        self.degree_log, self.rate_bits, self.cap_height = degree_log, rate_bits, cap_height
        self.keepalive = keepalive  # device buffers a wrapped (borrowing) handle points into

    def __del__(self):
        try:
            if self.h and getattr(self.engine, "_ctx", None):
                self.engine.lib.p2hot_batch_free(self.h)
        except Exception:
            pass
        self.h = None

    # MerkleTree::get / ::prove / .digests on the device-resident tree
    def rows(self, idx):
        idx = np.ascontiguousarray(np.asarray(idx, dtype=np.uint64).reshape(-1))
        out = np.zeros((len(idx), self.W), dtype=np.uint64)
        if len(idx) and self.W:
            self.engine.check(self.engine.lib.p2hot_batch_rows(self.h, idx.ctypes.data, len(idx), out.ctypes.data))
        return out

    def paths(self, idx):
        idx = np.ascontiguousarray(np.asarray(idx, dtype=np.uint64).reshape(-1))
        layers = self.degree_log + self.rate_bits - self.cap_height
        out = np.zeros((len(idx), layers, 4), dtype=np.uint64)
        if len(idx) and layers:
            self.engine.check(self.engine.lib.p2hot_batch_paths(self.h, idx.ctypes.data, len(idx), out.ctypes.data))
        return out

    def digests(self):
        nd = self.engine.num_digests(self.degree_log + self.rate_bits, self.cap_height)
        out = np.zeros((nd, 4), dtype=np.uint64)
        if nd:
            self.engine.check(self.engine.lib.p2hot_batch_digests(self.h, out.ctypes.data))
        return out


class PolynomialBatch:
    def __init__(self, engine, handle, W, degree_log, rate_bits, cap_height, cap, digests=None, coeffs=None, lde=None,
                 blinding=False, salt=0):
        self.engine = engine
        self._owner = _BatchHandle(engine, handle, W, degree_log, rate_bits, cap_height, keepalive=(coeffs, lde, digests), salt=salt)
        self.salt_size = salt
        self._W = W
        self._coeffs = coeffs      # device [W][n] when the batch was built from device buffers, else None
        self.lde = lde             # device [W][N] (same condition)
        self.degree_log = degree_log
        self.rate_bits = rate_bits
        self.cap_height = cap_height
        self.blinding = blinding
        N = 1 << (degree_log + rate_bits)
        own = self._owner
        self.merkle_tree = MerkleTree(None, digests, cap, cap_height, n_leaves=N,
                                      engine=engine if digests is not None and engine.mem.is_buffer(digests) else None,
                                      leaf_getter=own.rows, path_getter=own.paths if digests is None else None,
                                      digests_getter=own.digests)

    @property
    def _h(self):
        return self._owner.h

    @property
    def polynomials(self):
        """coefficient form, host [W][n] (oracle.rs:32), canonical representatives"""
        out = np.zeros((self._W, 1 << self.degree_log), dtype=np.uint64)
        if self._W:
            self.engine.check(self.engine.lib.p2hot_batch_coeffs(self._h, 0, self._W, out.ctypes.data))
        return out

    @classmethod
    def from_values(cls, values, rate_bits, blinding, cap_height, timing=None, fft_root_table=None, engine=None,
                    keep_values=False, salts=None):
        """oracle.rs:57-79.  values: [W][n] (host ndarray or device buffer), values on H_n.
        keep_values: keep the values on the device for plonk.prover (P2HOT_KEEP_VALUES).
        blinding=True: `salts` = the SALT_SIZE random vectors [4][N] the reference draws with F::rand_vec (oracle.rs:133-137);
        the caller owns the randomness."""
        return cls._build(values, rate_bits, blinding, cap_height, True, engine, keep_values, salts)

    @classmethod
    def from_coeffs(cls, polynomials, rate_bits, blinding, cap_height, timing=None, fft_root_table=None, engine=None, salts=None):
        """oracle.rs:82-112.  polynomials: [W][n] coefficients."""
        return cls._build(polynomials, rate_bits, blinding, cap_height, False, engine, False, salts)

    @classmethod
    def _build(cls, cols, rate_bits, blinding, cap_height, is_values, engine, keep_values, salts=None):
        eng = engine or default_engine()
        if blinding:
            if salts is None:
                raise ValueError("blinding=True needs the caller's salt vectors (the reference draws them from OsRng, "
                                 "oracle.rs:133-137): pass salts=[SALT_SIZE][N]")
            return cls._build_salted(cols, rate_bits, cap_height, is_values, eng, keep_values, salts)
        if isinstance(cols, DeviceColumns):
            return cls._from_device_columns(cols, rate_bits, cap_height, is_values, eng, keep_values)
        if not eng.mem.is_buffer(cols):
            cols = np.asarray(cols, dtype=np.uint64)
        if cols.ndim != 2:
            raise ValueError("expected [W][n]")
        W, n = cols.shape
        log_n = int(n).bit_length() - 1
        if n != 1 << log_n:
            raise ValueError("polynomial length must be a power of two")  # log2_strict, oracle.rs:88
        h = C.c_void_p()
        if eng.mem.is_buffer(cols):
            # device buffers: the *_dev flow, wrapped into a handle that borrows them
            r = eng.commit(cols, log_n, rate_bits, cap_height, is_values)
            eng.check(eng.lib.p2hot_batch_wrap_dev(eng.ctx, eng.ptr(r["coeffs"]), eng.ptr(r["lde"]), eng.ptr(r["digests"]), W,
                                                   log_n, rate_bits, cap_height, C.byref(h)))
            return cls(eng, h, W, log_n, rate_bits, cap_height, eng.host(r["cap"]), digests=r["digests"], coeffs=r["coeffs"],
                       lde=r["lde"])
        # host arrays: exactly what the Rust shim does -- W column pointers in, the cap out, the rest stays on the GPU
        cols = np.ascontiguousarray(cols, dtype=np.uint64)
        ptrs = (C.c_void_p * max(W, 1))(*[cols[c].ctypes.data for c in range(W)])
        cap = np.zeros((1 << cap_height, 4), dtype=np.uint64)
        eng.check(eng.lib.p2hot_commit(eng.ctx, ptrs, W, log_n, rate_bits, cap_height, 1 if is_values else 0,
                                       _lib.KEEP_VALUES if (keep_values and is_values) else 0, None, None, None,
                                       cap.ctypes.data, C.byref(h)))
        return cls(eng, h, W, log_n, rate_bits, cap_height, cap)

    @classmethod
    def _build_salted(cls, cols, rate_bits, cap_height, is_values, eng, keep_values, salts):
        """p2hot_commit_salted: host columns + host salt vectors (what the Rust shim passes for a zk config)"""
        cols = np.ascontiguousarray(np.asarray(cols, dtype=np.uint64))
        salts = np.ascontiguousarray(np.asarray(salts, dtype=np.uint64))
        if cols.ndim != 2 or salts.ndim != 2:
            raise ValueError("expected [W][n] columns and [SALT_SIZE][N] salts")
        W, n = cols.shape
        log_n = int(n).bit_length() - 1
        if n != 1 << log_n:
            raise ValueError("polynomial length must be a power of two")
        S = salts.shape[0]
        if salts.shape[1] != n << rate_bits:
            raise ValueError("salt vectors have length N = n << rate_bits (oracle.rs:136)")
        ptrs = (C.c_void_p * max(W, 1))(*[cols[c].ctypes.data for c in range(W)])
        sptrs = (C.c_void_p * max(S, 1))(*[salts[j].ctypes.data for j in range(S)])
        cap = np.zeros((1 << cap_height, 4), dtype=np.uint64)
        h = C.c_void_p()
        eng.check(eng.lib.p2hot_commit_salted(eng.ctx, ptrs, W, log_n, rate_bits, cap_height, 1 if is_values else 0,
                                              _lib.KEEP_VALUES if (keep_values and is_values) else 0, sptrs, S, None, None, None,
                                              cap.ctypes.data, C.byref(h)))
        return cls(eng, h, W, log_n, rate_bits, cap_height, cap, blinding=True, salt=S)

    @classmethod
    def _from_device_columns(cls, dc, rate_bits, cap_height, is_values, eng, keep_values):
        W, log_n = dc.width, dc.degree_log
        cap = np.zeros((1 << cap_height, 4), dtype=np.uint64)
        h = C.c_void_p()
        handle, dc._h = dc._h, None  # consumed by the library ...
        rc = eng.lib.p2hot_commit_cols(eng.ctx, handle, rate_bits, cap_height, 1 if is_values else 0,
                                       _lib.KEEP_VALUES if (keep_values and is_values) else 0, None, None, None,
                                       cap.ctypes.data, C.byref(h))
        if rc in (_lib.EBUSY, _lib.EINVAL):
            dc._h = handle  # ... except when the call never touched it (include/p2hot.h: both codes mean "not consumed")
        eng.check(rc)
        return cls(eng, h, W, log_n, rate_bits, cap_height, cap)

    def values(self):
        """the kept input values (keep_values=True) as a borrowed DeviceColumns view"""
        h = C.c_void_p()
        self.engine.check(self.engine.lib.p2hot_batch_values(self._h, C.byref(h)))
        return DeviceColumns(self.engine, h, owner=self)

    def subgroup_values(self, first, count):
        """values on H_n of polynomials [first, first+count) of the batch, from the device-resident coefficients
        (p2hot_batch_subgroup_values) -- how plonk.prover reads the sigma values off the constants_sigmas commitment
        instead of ProverOnlyCircuitData.sigmas (circuit_data.rs:455-456) -- as owned DeviceColumns"""
        h = C.c_void_p()
        self.engine.check(self.engine.lib.p2hot_batch_subgroup_values(self._h, first, count, C.byref(h)))
        return DeviceColumns(self.engine, h)

    def get_lde_values(self, index, step=1):
        """oracle.rs:142-147: row reverse_bits(index*step, degree_log + rate_bits) of the leaf matrix"""
        bits = self.degree_log + self.rate_bits
        i = index * step
        rev = int(format(i, "0%db" % bits)[::-1], 2) if bits else 0
        row = self.merkle_tree.get(rev)
        return row[..., :row.shape[-1] - self.salt_size] if self.salt_size else row  # oracle.rs:146


class DeviceColumns:
    """p2hot_cols: a device-resident Vec<PolynomialValues> / Vec<PolynomialCoeffs>"""

    def __init__(self, engine, handle, owner=None):
        self.engine, self._h, self._owner = engine, handle, owner

    @classmethod
    def upload(cls, cols, engine=None):
        eng = engine or default_engine()
        cols = np.ascontiguousarray(np.asarray(cols, dtype=np.uint64))
        W, n = cols.shape
        log_n = int(n).bit_length() - 1
        if n != 1 << log_n:
            raise ValueError("column length must be a power of two")
        ptrs = (C.c_void_p * max(W, 1))(*[cols[c].ctypes.data for c in range(W)])
        h = C.c_void_p()
        eng.check(eng.lib.p2hot_cols_upload(eng.ctx, ptrs, W, log_n, C.byref(h)))
        return cls(eng, h)

    @property
    def width(self):
        return self.engine.lib.p2hot_cols_width(self._h)

    @property
    def degree_log(self):
        return self.engine.lib.p2hot_cols_degree_log(self._h)

    def host(self):
        out = np.zeros((self.width, 1 << self.degree_log), dtype=np.uint64)
        if self.width:
            self.engine.check(self.engine.lib.p2hot_cols_download(self._h, 0, self.width, out.ctypes.data))
        return out

    def __del__(self):
        try:
            if self._h and getattr(self.engine, "_ctx", None):
                self.engine.lib.p2hot_cols_free(self._h)
        except Exception:
            pass
        self._h = None


# ------------------------------------------------------------------ prove_openings (oracle.rs:176-237)
class FriBatchInfo:
    """fri/structure.rs FriBatchInfo: an opening point and the (oracle_index, polynomial_index) pairs opened there"""

    def __init__(self, point, polynomials):
        self.point = [int(point[0]), int(point[1])]
        self.polynomials = [(int(o), int(p)) for o, p in polynomials]


def _handles(oracles):
    return (C.c_void_p * max(len(oracles), 1))(*[o._h for o in oracles])


def eval_openings(oracles, points, engine=None):
    """OpeningSet::new (plonk/proof.rs:314-327): every polynomial of every oracle at each extension point.
    Returns a list (per oracle) of arrays [n_points][W][2]."""
    eng = engine or oracles[0].engine
    pts = np.ascontiguousarray(np.asarray(points, dtype=np.uint64).reshape(-1, 2))
    total = sum(o._W for o in oracles)
    flat = np.zeros(max(1, 2 * len(pts) * total), dtype=np.uint64)
    eng.check(eng.lib.p2hot_eval_openings(eng.ctx, _handles(oracles), len(oracles), pts.ctypes.data, len(pts), flat.ctypes.data))
    out, off = [], 0
    for o in oracles:
        cnt = 2 * len(pts) * o._W
        out.append(flat[off:off + cnt].reshape(len(pts), o._W, 2))
        off += cnt
    return out


def final_poly_device(batches, oracles, alpha, engine=None):
    """The final_poly of prove_openings (oracle.rs:186-213) on the GPU (p2hot_fri_final_poly_dev, the building block
    p2hot_prove_openings uses): device planes [2][n].  Only for oracles built from device buffers; kept for the
    stage-level parity test."""
    eng = engine or oracles[0].engine
    n = 1 << oracles[0].degree_log
    ptrs, offsets = [], [0]
    for b in batches:
        for (oi, pi) in b.polynomials:
            co = oracles[oi]._coeffs
            if co is None:
                raise ValueError("final_poly_device needs oracles committed from device buffers")
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
                   num_query_rounds, engine=None, timing=None, final_poly_coeff_len=None, max_num_query_steps=None):
    """PolynomialBatch::prove_openings + fri_proof (oracle.rs:176-237, fri/prover.rs:24-82): one p2hot_prove_openings
    call.  Returns a dict shaped like FriProof:
      commit_phase_merkle_caps, query_round_proofs [{initial_trees_proof: [(leaf, siblings)...], steps: [(evals, siblings)...]}],
      final_poly [[c0, c1]...], pow_witness.
    `timing` (a dict): total synchronised wall milliseconds of the call under "prove_openings"."""
    import time
    eng = engine or oracles[0].engine
    arity = [int(a) for a in reduction_arity_bits]
    R, Q = len(arity), int(num_query_rounds)
    ab = (C.c_uint * max(R, 1))(*arity)
    fp = _lib.FriParams(rate_bits, cap_height, proof_of_work_bits, Q, ab, R, 0, max_num_query_steps or 0,
                        final_poly_coeff_len or 0)
    keep = []  # index arrays referenced by the batch structs
    infos = (_lib.FriBatchInfo * max(len(batches), 1))()
    for k, b in enumerate(batches):
        oi = (C.c_uint32 * max(len(b.polynomials), 1))(*[o for o, _ in b.polynomials])
        pi = (C.c_uint32 * max(len(b.polynomials), 1))(*[p for _, p in b.polynomials])
        keep += [oi, pi]
        infos[k].point[0], infos[k].point[1] = b.point
        infos[k].oracle_index, infos[k].poly_index, infos[k].n_polys = oi, pi, len(b.polynomials)
    handles = _handles(oracles)
    lay = _lib.FriProofLayout()
    rc = eng.lib.p2hot_fri_proof_sizes(handles, len(oracles), C.byref(fp), C.byref(lay))
    if rc != _lib.OK:
        raise _lib.P2HotError(rc, "inconsistent FRI parameters")
    bufs = {k: np.zeros(max(1, getattr(lay, k + "_words")), dtype=np.uint64)
            for k in ("caps", "final_poly", "initial_leaves", "initial_paths", "step_evals", "step_paths")}
    qidx = np.zeros(max(1, Q), dtype=np.uint64)
    proof = _lib.FriProof(bufs["caps"].ctypes.data, bufs["final_poly"].ctypes.data, 0, qidx.ctypes.data,
                          bufs["initial_leaves"].ctypes.data, bufs["initial_paths"].ctypes.data,
                          bufs["step_evals"].ctypes.data, bufs["step_paths"].ctypes.data)
    t0 = time.perf_counter()
    eng.check(eng.lib.p2hot_prove_openings(eng.ctx, infos, len(batches), handles, len(oracles), challenger._h, C.byref(fp),
                                           C.byref(proof)))
    if timing is not None:
        timing["prove_openings"] = timing.get("prove_openings", 0.0) + (time.perf_counter() - t0) * 1e3
    widths = [o._W + getattr(o, "salt_size", 0) for o in oracles]  # evals_proofs carry whole leaves, salts included
    bufs["final_poly"] = bufs["final_poly"][:lay.final_poly_words]
    return shape_fri_proof(bufs, int(proof.pow_witness), qidx, widths, oracles[0].degree_log + rate_bits, cap_height, arity, Q)


def shape_fri_proof(bufs, pow_witness, qidx, widths, log_N, cap_height, arity, Q):
    """the flat proof buffers of p2hot_prove_openings / p2hot_group_prove_openings (layout: include/p2hot.h, p2hot_fri_proof) as the
    FriProof-shaped dict documented at prove_openings.  widths: leaf words per oracle; log_N: log2 of the LDE size"""
    R, ncap = len(arity), 1 << cap_height
    wsum = sum(widths)
    il = bufs["initial_leaves"][:Q * wsum].reshape(Q, wsum) if wsum else np.zeros((Q, 0), dtype=np.uint64)
    layers0 = log_N - cap_height
    ip = bufs["initial_paths"][:Q * len(widths) * layers0 * 4].reshape(Q, len(widths), layers0, 4)
    ev_w, pa_w, lm = [], [], log_N
    for a in arity:
        ev_w.append(2 << a)
        pa_w.append(lm - a - cap_height)
        lm -= a
    se = bufs["step_evals"][:Q * sum(ev_w)].reshape(Q, sum(ev_w)) if R else None
    sp = bufs["step_paths"][:Q * 4 * sum(pa_w)].reshape(Q, 4 * sum(pa_w)) if R else None
    queries = []
    for q in range(Q):
        initial, wo = [], 0
        for oi, w in enumerate(widths):
            initial.append((il[q, wo:wo + w], ip[q, oi]))
            wo += w
        steps, eo, po = [], 0, 0
        for r in range(R):
            steps.append((se[q, eo:eo + ev_w[r]].reshape(-1, 2), sp[q, po:po + 4 * pa_w[r]].reshape(pa_w[r], 4)))
            eo += ev_w[r]
            po += 4 * pa_w[r]
        queries.append({"initial_trees_proof": initial, "steps": steps})
    # final_poly: (N / 2^rate) / prod(2^arity) extension coefficients -- whatever the buffer holds
    return {"commit_phase_merkle_caps": [bufs["caps"][4 * ncap * r:4 * ncap * (r + 1)].reshape(ncap, 4) for r in range(R)],
            "query_round_proofs": queries,
            "final_poly": np.asarray(bufs["final_poly"]).reshape(-1, 2),
            "pow_witness": int(pow_witness), "query_indices": [int(x) for x in qidx[:Q]]}
