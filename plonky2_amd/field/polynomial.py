"""coset LDE -- mirror of PolynomialCoeffs::lde + coset_fft_with_options
(field/src/polynomial/mod.rs:199-201, :280-293) for a batch of polynomials."""
import numpy as np

from ..engine import COSET_SHIFT, default_engine


def lde_coset_fft(coeffs, rate_bits, shift=COSET_SHIFT, bit_reversed=False, engine=None):
    """[W][n] coefficients -> [W][n << rate_bits] values p(shift * w_N^i).

    bit_reversed=False returns natural order (coset_fft_with_options); True returns the committed
    order (row L = natural index bitrev(L)), which is what the kernels produce natively."""
    eng = engine or default_engine()
    a = np.ascontiguousarray(np.asarray(coeffs, dtype=np.uint64))
    W, n = a.shape
    log_n = n.bit_length() - 1
    if n != 1 << log_n:
        raise ValueError("length must be a power of two")
    lde = eng.coset_lde(eng.dev(a), log_n, rate_bits, shift)
    if not bit_reversed:
        lde = eng.reverse_index_bits(lde, log_n + rate_bits)
    return eng.host(lde)


def coset_ifft(values, shift=COSET_SHIFT, engine=None):
    """PolynomialValues::coset_ifft (field/src/polynomial/mod.rs:63-73) for a batch [W][n]"""
    eng = engine or default_engine()
    a = np.ascontiguousarray(np.asarray(values, dtype=np.uint64))
    one = a.ndim == 1
    a = a.reshape(1, -1) if one else a
    W, n = a.shape
    log_n = n.bit_length() - 1
    if n != 1 << log_n:
        raise ValueError("length must be a power of two")
    d = eng.dev(a.copy())
    eng.check(eng.lib.p2hot_coset_ifft_dev(eng.ctx, eng.ptr(d), W, n, log_n, shift))
    out = eng.host(d)
    return out[0] if one else out
