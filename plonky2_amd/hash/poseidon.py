"""Poseidon width-12 entry points -- mirror of plonky2/src/hash/poseidon.rs (Poseidon::poseidon
:767-777, PoseidonHash::hash_no_pad :880, two_to_one :884) and plonk/config.rs:63-74 hash_or_noop,
evaluated by the same device code that hashes Merkle leaves (batched; one lane per permutation).
"""
import numpy as np

from ..engine import default_engine

SPONGE_WIDTH, SPONGE_RATE = 12, 8


def poseidon(states, engine=None):
    """[count][12] -> [count][12] (canonical)"""
    eng = engine or default_engine()
    s = np.ascontiguousarray(np.asarray(states, dtype=np.uint64)).reshape(-1, 12)
    return eng.host(eng.poseidon_permute(eng.dev(s)))


def hash_or_noop_batch(rows, engine=None):
    """hash_or_noop of every row of [n][w] (n a power of two) -> [n][4]"""
    eng = engine or default_engine()
    rows = np.ascontiguousarray(np.asarray(rows, dtype=np.uint64))
    n, w = rows.shape
    log_n = n.bit_length() - 1
    if n != 1 << log_n:
        raise ValueError("row count must be a power of two")
    _, cap = eng.merkle(eng.dev(rows), 1, w, log_n, log_n)  # all-cap tree: cap[i] = hash_or_noop(row i)
    return eng.host(cap)


def hash_no_pad(x, engine=None):
    x = np.asarray(x, dtype=np.uint64).reshape(1, -1)
    if x.shape[1] <= 4:  # force the sponge: pad-free hash of a short input = one permutation
        s = np.zeros((1, 12), dtype=np.uint64)
        s[0, :x.shape[1]] = x[0]
        return poseidon(s, engine)[0, :4] if x.shape[1] else np.zeros(4, dtype=np.uint64)
    return hash_or_noop_batch(x, engine)[0]


def two_to_one(left, right, engine=None):
    s = np.zeros((1, 12), dtype=np.uint64)
    s[0, 0:4] = np.asarray(left, dtype=np.uint64)
    s[0, 4:8] = np.asarray(right, dtype=np.uint64)
    return poseidon(s, engine)[0, :4]
