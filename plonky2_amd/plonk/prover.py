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

    wires, sigmas: [num_routed][n] column-major (host ndarray or device buffer) -- MatrixWitness.wire_values[col][row]
    and the sigma polynomials' values on the subgroup.  Returns a device buffer [nc * (num_prods + 1)][n]: the Z of every
    challenge first, then the partial products of challenge 0, 1, ... (`zs_partial_products`)."""
    eng = engine or default_engine()
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
    into quotient_degree_factor chunks of n coefficients.  Returns a device buffer
    [num_challenges * quotient_degree_factor][n] ready for PolynomialBatch.from_coeffs."""
    from ..engine import COSET_SHIFT
    eng = engine or default_engine()
    d = eng.dev(quotient_values)
    if d.ndim != 2:
        raise ValueError("expected [num_challenges][n << quotient_degree_bits]")
    nc, m = d.shape
    n = 1 << degree_bits
    qbits = max(0, (quotient_degree_factor - 1).bit_length())
    if m != n << qbits:
        raise ValueError("quotient values must live on the coset of size n << ceil(log2(quotient_degree_factor))")
    work = eng.mem.empty(nc, m)
    work[:] = d                                                             # coset_ifft works in place
    eng.check(eng.lib.p2hot_coset_ifft_dev(eng.ctx, eng.ptr(work), nc, m, degree_bits + qbits, COSET_SHIFT))
    keep = n * quotient_degree_factor
    if keep < m and eng.host(work[:, keep:]).any():                         # trim_to_len (polynomial/mod.rs:164-178)
        raise ValueError("Quotient has failed, the vanishing polynomial is not divisible by Z_H")
    out = eng.mem.empty(nc * quotient_degree_factor, n)
    for ch in range(nc):                                                    # PolynomialCoeffs::chunks (mod.rs:136-142)
        out[ch * quotient_degree_factor:(ch + 1) * quotient_degree_factor] = \
            work[ch, :keep].reshape(quotient_degree_factor, n)
    return out
