"""transpose / reverse_index_bits -- mirror of plonky2/src/util/mod.rs:25-41 and util/src/lib.rs:53-62."""
import numpy as np

from ..engine import default_engine


def transpose(matrix, engine=None):
    eng = engine or default_engine()
    return eng.host(eng.transpose(eng.dev(np.ascontiguousarray(np.asarray(matrix, dtype=np.uint64)))))


def reverse_index_bits(arr, engine=None):
    eng = engine or default_engine()
    a = np.ascontiguousarray(np.asarray(arr, dtype=np.uint64)).reshape(1, -1)
    n = a.shape[1]
    log_n = n.bit_length() - 1
    if n != 1 << log_n:
        raise ValueError("length must be a power of two")
    return eng.host(eng.reverse_index_bits(eng.dev(a), log_n))[0]


def reverse_bits(n, num_bits):
    return int(format(n, "0%db" % num_bits)[::-1], 2) if num_bits else 0
