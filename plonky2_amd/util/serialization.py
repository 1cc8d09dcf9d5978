"""Wire format of MerkleTree / PolynomialBatch -- mirror of plonky2/src/util/serialization/mod.rs
(write_merkle_tree :1417-1431, write_polynomial_batch :1744-1763 and the matching readers): every
usize and field element is a little-endian u64 (`write_usize` :1237, `write_field` :1254-1260 canonical),
a hash is 4 such words (hash/hash_types.rs:87-92), `blinding` is one byte.  Lets a GPU-built commitment
(e.g. the build-time constants_sigmas_commitment) be cached in the reference's own format."""
import numpy as np

from ..hash.merkle_tree import MerkleTree


def _u64(x):
    return np.asarray([x], dtype="<u8").tobytes()


def write_merkle_tree(tree):
    leaves = np.asarray(tree.leaves, dtype=np.uint64)
    n, w = leaves.shape
    rows = np.empty((n, w + 1), dtype="<u8")
    rows[:, 0] = w                                   # write_usize(tree.leaves[i].len())
    rows[:, 1:] = leaves                             # write_field_vec
    digests = np.asarray(tree.digests, dtype="<u8").reshape(-1, 4)
    return b"".join([_u64(n), rows.tobytes(), _u64(digests.shape[0]), digests.tobytes(), _u64(tree.cap.height()),
                     np.asarray(tree.cap.entries, dtype="<u8").tobytes()])


def read_merkle_tree(buf, off=0):
    def u64():
        nonlocal off
        v = int(np.frombuffer(buf, dtype="<u8", count=1, offset=off)[0])
        off += 8
        return v
    n = u64()
    if n:
        w = int(np.frombuffer(buf, dtype="<u8", count=1, offset=off)[0])
        rows = np.frombuffer(buf, dtype="<u8", count=n * (w + 1), offset=off).reshape(n, w + 1)
        if not (rows[:, 0] == w).all():
            raise ValueError("ragged leaves are not produced by PolynomialBatch")
        leaves = rows[:, 1:].astype(np.uint64)
        off += 8 * n * (w + 1)
    else:
        leaves = np.zeros((0, 0), dtype=np.uint64)
    nd = u64()
    digests = np.frombuffer(buf, dtype="<u8", count=4 * nd, offset=off).reshape(nd, 4).astype(np.uint64)
    off += 32 * nd
    h = u64()
    cap = np.frombuffer(buf, dtype="<u8", count=4 << h, offset=off).reshape(1 << h, 4).astype(np.uint64)
    off += 32 << h
    return MerkleTree(leaves, digests, cap, h), off


def write_polynomial_batch(batch):
    polys = np.asarray(batch.polynomials, dtype=np.uint64)
    W, n = polys.shape
    rows = np.empty((W, n + 1), dtype="<u8")
    rows[:, 0] = n
    rows[:, 1:] = polys
    return b"".join([_u64(W), rows.tobytes(), write_merkle_tree(batch.merkle_tree), _u64(batch.degree_log),
                     _u64(batch.rate_bits), bytes([1 if batch.blinding else 0])])


def read_polynomial_batch(buf):
    """-> dict(polynomials [W][n], merkle_tree, degree_log, rate_bits, blinding)"""
    off = 0
    W = int(np.frombuffer(buf, dtype="<u8", count=1, offset=off)[0])
    off += 8
    if W:
        n = int(np.frombuffer(buf, dtype="<u8", count=1, offset=off)[0])
        rows = np.frombuffer(buf, dtype="<u8", count=W * (n + 1), offset=off).reshape(W, n + 1)
        polys = rows[:, 1:].astype(np.uint64)
        off += 8 * W * (n + 1)
    else:
        polys = np.zeros((0, 0), dtype=np.uint64)
    tree, off = read_merkle_tree(buf, off)
    degree_log, rate_bits = (int(x) for x in np.frombuffer(buf, dtype="<u8", count=2, offset=off))
    off += 16
    blinding = bool(buf[off])
    return {"polynomials": polys, "merkle_tree": tree, "degree_log": degree_log, "rate_bits": rate_bits, "blinding": blinding}
