"""fft / ifft -- mirror of field/src/fft.rs:53-91 (natural order in and out), batched on the GPU."""
import numpy as np

from ..engine import default_engine


def _run(vals, inverse, engine):
    eng = engine or default_engine()
    a = np.ascontiguousarray(np.asarray(vals, dtype=np.uint64))
    one = a.ndim == 1
    a = a.reshape(1, -1) if one else a
    n = a.shape[1]
    log_n = n.bit_length() - 1
    if n != 1 << log_n:
        raise ValueError("length must be a power of two")
    d = eng.dev(a.copy())
    (eng.ifft if inverse else eng.fft)(d, log_n)
    out = eng.host(d)
    return out[0] if one else out


def fft(coeffs, engine=None):
    """fft_with_options(coeffs, None, None): out[i] = sum_t c[t] w^(i t)"""
    return _run(coeffs, False, engine)


def ifft(values, engine=None):
    """ifft_with_options(values, None, None)"""
    return _run(values, True, engine)
