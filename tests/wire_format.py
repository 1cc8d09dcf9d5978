"""TEST INFRASTRUCTURE (round 3: moved out of the product package).  In the drop-in the reference's OWN serializer runs on the
structs the Rust shim builds (integration/p2hot.rs; the bit-exact harness compares write_polynomial_batch / write_merkle_tree
bytes of CPU-built and GPU-built batches), so the product needs no wire-format code.  This Python restatement stays as a
checker: it pins the byte layout the tests expect of device-produced proofs.

Wire format of MerkleTree / PolynomialBatch -- mirror of plonky2/src/util/serialization/mod.rs
(write_merkle_tree :1417-1431, write_polynomial_batch :1744-1763 and the matching readers): every
usize and field element is a little-endian u64 (`write_usize` :1237, `write_field` :1254-1260 canonical),
a hash is 4 such words (hash/hash_types.rs:87-92), `blinding` is one byte.  Lets a GPU-built commitment
(e.g. the build-time constants_sigmas_commitment) be cached in the reference's own format."""
import numpy as np

from plonky2_amd.hash.merkle_tree import MerkleTree


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


def iter_polynomial_batch_bytes(polys, row_chunks, digests, cap, cap_height, degree_log, rate_bits, blinding):
    """write_polynomial_batch's bytes piece by piece, for batches whose leaf matrix is not held whole (C3 quotient chunks: 1.1 GB
    of leaves): `row_chunks` yields [m][W] blocks of leaf rows in committed order, everything else is given whole.  Layout as
    above (serialization/mod.rs:1744-1763 around :1417-1431); b"".join(...) == write_polynomial_batch(batch)."""
    polys = np.asarray(polys, dtype=np.uint64)
    W, n = polys.shape
    yield _u64(W)
    for c in range(W):
        yield _u64(n)
        yield polys[c].astype("<u8").tobytes()
    n_rows = (np.asarray(digests).size // 4 + 2 * (1 << cap_height)) // 2     # digests.len() = 2 * (leaves - 2^cap_height)
    yield _u64(n_rows)
    seen = 0
    for block in row_chunks:
        block = np.asarray(block, dtype=np.uint64)
        rows = np.empty((block.shape[0], block.shape[1] + 1), dtype="<u8")
        rows[:, 0] = block.shape[1]
        rows[:, 1:] = block
        seen += block.shape[0]
        yield rows.tobytes()
    if seen != n_rows:
        raise ValueError("row_chunks produced %d leaf rows, the digest array belongs to %d" % (seen, n_rows))
    d = np.asarray(digests, dtype="<u8").reshape(-1, 4)
    yield _u64(d.shape[0])
    yield d.tobytes()
    yield _u64(cap_height)
    yield np.asarray(cap, dtype="<u8").tobytes()
    yield _u64(degree_log)
    yield _u64(rate_bits)
    yield bytes([1 if blinding else 0])


def polynomial_batch_sha256(polys, row_chunks, digests, cap, cap_height, degree_log, rate_bits, blinding):
    import hashlib
    h = hashlib.sha256()
    for piece in iter_polynomial_batch_bytes(polys, row_chunks, digests, cap, cap_height, degree_log, rate_bits, blinding):
        h.update(piece)
    return h.hexdigest()


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


# ------------------------------------------------------------------ FriProof (serialization/mod.rs:1470-1611)
def _merkle_proof_bytes(siblings):
    s = np.asarray(siblings, dtype="<u8").reshape(-1, 4)
    if s.shape[0] > 255:
        raise ValueError("Merkle proof length must fit in u8.")        # write_merkle_proof :1476-1480
    return bytes([s.shape[0]]) + s.tobytes()


def write_fri_proof(proof):
    """write_fri_proof (:1595-1611) of the dict plonky2_amd.fri.oracle.prove_openings returns: the commit-phase caps
    (write_merkle_cap: bare hashes), the query rounds (write_fri_query_rounds :1558-1573: per round the initial-tree
    (leaf, Merkle proof) pairs then the (evals, Merkle proof) steps -- vectors WITHOUT length prefixes, Merkle proofs
    with a u8 length), the final polynomial's extension coefficients and the PoW witness; every word a canonical
    little-endian u64 (write_field :1254-1260)."""
    out = []
    for cap in proof["commit_phase_merkle_caps"]:
        out.append(np.asarray(cap, dtype="<u8").tobytes())
    for qr in proof["query_round_proofs"]:
        for leaf, siblings in qr["initial_trees_proof"]:
            out.append(np.asarray(leaf, dtype="<u8").tobytes())
            out.append(_merkle_proof_bytes(siblings))
        for evals, siblings in qr["steps"]:
            out.append(np.asarray(evals, dtype="<u8").tobytes())
            out.append(_merkle_proof_bytes(siblings))
    out.append(np.asarray(proof["final_poly"], dtype="<u8").tobytes())
    out.append(_u64(int(proof["pow_witness"])))
    return b"".join(out)


def read_fri_proof(buf, oracle_widths, degree_bits, rate_bits, cap_height, reduction_arity_bits, num_query_rounds):
    """read_fri_proof (serialization/mod.rs:626-660): the shape comes from the circuit's common data, as in the reference"""
    off = 0

    def words(count, shape):
        nonlocal off
        v = np.frombuffer(buf, dtype="<u8", count=count, offset=off).astype(np.uint64).reshape(shape)
        off += 8 * count
        return v

    def merkle_proof():
        nonlocal off
        length = buf[off]
        off += 1
        return words(4 * length, (length, 4))

    ncap = 1 << cap_height
    caps = [words(4 * ncap, (ncap, 4)) for _ in reduction_arity_bits]
    queries = []
    for _ in range(num_query_rounds):
        initial = []
        for w in oracle_widths:
            leaf = words(w, (w,))
            initial.append((leaf, merkle_proof()))
        steps = []
        for ab in reduction_arity_bits:
            evals = words(2 << ab, (1 << ab, 2))
            steps.append((evals, merkle_proof()))
        queries.append({"initial_trees_proof": initial, "steps": steps})
    n_final = (1 << degree_bits) >> sum(reduction_arity_bits)
    final = words(2 * n_final, (n_final, 2))
    pow_witness = int(words(1, (1,))[0])
    if off != len(buf):
        raise ValueError("trailing bytes after the FRI proof")
    return {"commit_phase_merkle_caps": caps, "query_round_proofs": queries, "final_poly": final, "pow_witness": pow_witness}
