"""Host-side mirror of the gate-independent pieces of plonky2/src/plonk/prover.rs that sit next to the commit path
(SURVEY 8f-3): the permutation argument's partial products / Z polynomials, computed on the GPU from device-resident
wire and sigma columns and handed straight to PolynomialBatch.from_values."""
import ctypes as C

import numpy as np

from ..engine import default_engine


def num_partial_products(n, max_degree):
    """plonky2/src/util/partial_products.rs:40-47"""
    return -(-n // max_degree) - 1


def all_wires_permutation_partial_products(wires, sigmas, k_is, quotient_degree_factor, betas, gammas, engine=None):
    """all_wires_permutation_partial_products (prover.rs:356-390) followed by the batch ordering of prover.rs:224-229.

    wires, sigmas: [num_routed][n] column-major -- MatrixWitness.wire_values[col][row] and the sigma polynomials' values
    on the subgroup -- as host ndarrays / DeviceColumns (the host-pointer entry point p2hot_partial_products, what the
    Rust shim calls; returns DeviceColumns) or as device buffers (p2hot_partial_products_dev; returns a device buffer).
    Result [nc * (num_prods + 1)][n]: the Z of every challenge first, then the partial products of challenge 0, 1, ...
    (`zs_partial_products`)."""
    from ..fri.oracle import DeviceColumns
    eng = engine or default_engine()
    if not (eng.mem.is_buffer(wires) and eng.mem.is_buffer(sigmas)):
        for x in (wires, sigmas):
            if not isinstance(x, DeviceColumns) and (np.ndim(x) != 2 or np.shape(x)[1] & (np.shape(x)[1] - 1)):
                raise ValueError("wires and sigmas must be [num_routed][n] with n a power of two")
        dw = wires if isinstance(wires, DeviceColumns) else DeviceColumns.upload(eng.host(wires), eng)
        ds = sigmas if isinstance(sigmas, DeviceColumns) else DeviceColumns.upload(eng.host(sigmas), eng)
        k = np.ascontiguousarray(np.asarray(k_is, dtype=np.uint64))
        if dw.degree_log != ds.degree_log or dw.width < len(k) or ds.width < len(k):
            raise ValueError("wires and sigmas must both be [num_routed][n]")
        if not quotient_degree_factor < len(k):  # prover.rs:215-218
            raise ValueError("quotient_degree_factor must be smaller than the number of routed wires")
        b = np.ascontiguousarray(np.asarray(betas, dtype=np.uint64))
        g = np.ascontiguousarray(np.asarray(gammas, dtype=np.uint64))
        if b.shape != g.shape or b.ndim != 1:
            raise ValueError("betas and gammas must be equally long vectors")
        h = C.c_void_p()
        eng.check(eng.lib.p2hot_partial_products(eng.ctx, dw._h, 0, ds._h, 0, k.ctypes.data_as(C.c_void_p), len(k),
                                                 quotient_degree_factor, b.ctypes.data_as(C.c_void_p),
                                                 g.ctypes.data_as(C.c_void_p), len(b), None, C.byref(h)))
        return DeviceColumns(eng, h)
    d_w, d_s = eng.dev(wires), eng.dev(sigmas)
    if d_w.ndim != 2 or d_s.shape != d_w.shape:
        raise ValueError("wires and sigmas must both be [num_routed][n]")
    r, n = d_w.shape
    log_n = int(n).bit_length() - 1
    if n != 1 << log_n:
        raise ValueError("row count must be a power of two")
    k = np.ascontiguousarray(np.asarray(k_is, dtype=np.uint64))
    if k.shape != (r,):
        raise ValueError("k_is must have one shift per routed wire")
    b = np.ascontiguousarray(np.asarray(betas, dtype=np.uint64))
    g = np.ascontiguousarray(np.asarray(gammas, dtype=np.uint64))
    if b.shape != g.shape or b.ndim != 1:
        raise ValueError("betas and gammas must be equally long vectors")
    nc = b.shape[0]
    # prover.rs:215-218: "When the number of routed wires is smaller that the degree, we should change the logic"
    if not quotient_degree_factor < r:
        raise ValueError("quotient_degree_factor must be smaller than the number of routed wires")
    num_prods = num_partial_products(r, quotient_degree_factor)
    out = eng.mem.empty(nc * (num_prods + 1), n)
    eng.check(eng.lib.p2hot_partial_products_dev(eng.ctx, eng.ptr(d_w), n, eng.ptr(d_s), n, k.ctypes.data_as(C.c_void_p), r,
                                                 log_n, quotient_degree_factor, b.ctypes.data_as(C.c_void_p),
                                                 g.ctypes.data_as(C.c_void_p), nc, eng.ptr(out), n))
    return out


def partial_products_and_zs_commitment(wires, sigmas, k_is, quotient_degree_factor, betas, gammas, rate_bits, cap_height,
                                       engine=None):
    """prover.rs:219-247 without lookups: compute partial products + Zs and commit to them (from_values, no blinding)."""
    from ..fri.oracle import PolynomialBatch
    eng = engine or default_engine()
    zs_pp = all_wires_permutation_partial_products(wires, sigmas, k_is, quotient_degree_factor, betas, gammas, eng)
    return PolynomialBatch.from_values(zs_pp, rate_bits, False, cap_height, engine=eng)


def quotient_poly_chunks(quotient_values, degree_bits, quotient_degree_factor, engine=None):
    """The gate-independent tail of the quotient computation (prover.rs:274-289, :810-815): the quotient polynomials
    arrive as values on the coset g*H of size n << ceil(log2(quotient_degree_factor)) (one row per challenge, natural
    order), are interpolated with coset_ifft, trimmed to quotient_degree_factor * n coefficients (the reference panics
    with "Quotient has failed, the vanishing polynomial is not divisible by Z_H" if the tail is not zero) and split
    into quotient_degree_factor chunks of n coefficients.  Returns DeviceColumns
    [num_challenges * quotient_degree_factor][n] ready for PolynomialBatch.from_coeffs."""
    from ..fri.oracle import DeviceColumns
    eng = engine or default_engine()
    # one p2hot_quotient_chunks call (the Rust shim's path: compute_quotient_polys leaves host Vecs)
    q = np.ascontiguousarray(eng.host(quotient_values))
    if q.ndim != 2:
        raise ValueError("expected [num_challenges][n << quotient_degree_bits]")
    qb = max(0, (quotient_degree_factor - 1).bit_length())
    if q.shape[1] != (1 << degree_bits) << qb:
        raise ValueError("quotient values must live on the coset of size n << ceil(log2(quotient_degree_factor))")
    ptrs = (C.c_void_p * max(q.shape[0], 1))(*[q[c].ctypes.data for c in range(q.shape[0])])
    h = C.c_void_p()
    rc = eng.lib.p2hot_quotient_chunks(eng.ctx, ptrs, q.shape[0], degree_bits, quotient_degree_factor, C.byref(h))
    if rc == 1 and b"Quotient has failed" in eng.lib.p2hot_last_error(eng._ctx):
        raise ValueError(eng.lib.p2hot_last_error(eng._ctx).decode())  # the reference panics (polynomial/mod.rs:164-178)
    eng.check(rc)
    return DeviceColumns(eng, h)


def compute_quotient_polys(wires_commitment, constants_sigmas_commitment, sigmas_first_col, zs_partial_products_commitment, k_is,
                           quotient_degree_factor, betas, gammas, alphas, gate_sums=None, want_values=False, engine=None):
    """compute_quotient_polys (prover.rs:609-815) without its gate evaluation -- one p2hot_quotient_polys call: the permutation
    argument's vanishing terms (vanishing_poly.rs:167-330) on the quotient coset from the three commitments' device-resident LDE
    matrices, plus `gate_sums` (the caller's reduce_with_powers of the gate constraint terms, [num_challenges][n << qbits], or
    None), over Z_H, then coset_ifft / trim / chunks.  Returns DeviceColumns [num_challenges * quotient_degree_factor][n] for
    PolynomialBatch.from_coeffs (and the quotient values [num_challenges][n << qbits] when want_values)."""
    from ..fri.oracle import DeviceColumns
    eng = engine or wires_commitment.engine
    k = np.ascontiguousarray(np.asarray(k_is, dtype=np.uint64))
    b, g, a = (np.ascontiguousarray(np.asarray(v, dtype=np.uint64)) for v in (betas, gammas, alphas))
    if not (b.shape == g.shape == a.shape and b.ndim == 1):
        raise ValueError("betas, gammas and alphas must be equally long vectors")
    nc = len(b)
    qb = max(0, (quotient_degree_factor - 1).bit_length())
    m = (1 << wires_commitment.degree_log) << qb
    gs = None
    if gate_sums is not None:
        gs = np.ascontiguousarray(np.asarray(gate_sums, dtype=np.uint64))
        if gs.shape != (nc, m):
            raise ValueError("gate_sums must be [num_challenges][n << ceil(log2(quotient_degree_factor))]")
    gptrs = (C.c_void_p * nc)(*[gs[c].ctypes.data for c in range(nc)]) if gs is not None else None
    vals = np.zeros((nc, m), dtype=np.uint64) if want_values else None
    h = C.c_void_p()
    rc = eng.lib.p2hot_quotient_polys(eng.ctx, wires_commitment._h, constants_sigmas_commitment._h, sigmas_first_col,
                                      zs_partial_products_commitment._h, k.ctypes.data_as(C.c_void_p), len(k), quotient_degree_factor,
                                      b.ctypes.data_as(C.c_void_p), g.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p), nc, gptrs,
                                      vals.ctypes.data_as(C.c_void_p) if want_values else None, C.byref(h))
    if rc == 1 and b"Quotient has failed" in eng.lib.p2hot_last_error(eng._ctx):
        raise ValueError(eng.lib.p2hot_last_error(eng._ctx).decode())  # the reference panics (polynomial/mod.rs:164-178)
    eng.check(rc)
    cols = DeviceColumns(eng, h)
    return (cols, vals) if want_values else cols
