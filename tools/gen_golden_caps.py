#!/usr/bin/env python3
"""Golden commitments of the BASELINE shapes, computed by the FAITHFUL CPU oracle (oracle/p2oracle.c).

    python tools/gen_golden_caps.py [--only NAME]        (build container; 62 GB of RAM, ~10 minutes)

For each full-size commit of BASELINE.json's configs it writes, into tests/golden/commit_caps.json:
  cap            the 2^cap_height Merkle cap entries (4 words each)
  sha256_coeffs  SHA-256 of the coefficient matrix [W][n] (u64 little endian, canonical)
  sha256_digests SHA-256 of MerkleTree::digests in the reference layout (hash/merkle_tree.rs:50-57)
  sha256_lde     SHA-256 of the LDE matrix column-major [W][N], rows in committed (bit-reversed) order
                 (= the reference's `leaves` transposed: how the device holds it)
so the GPU tests (tests/test_gpu_fullsize.py) and bench.py can compare the device results of exactly these
inputs with the oracle's bytes WITHOUT running 1.5e8 CPU permutations on the GPU box, and independently of the
tuned CPU code (oracle/p2fast.c), which this script also cross-checks against the same bytes.
Inputs come from plonky2_amd/util/synthetic.py (the bench's splitmix columns, the C4 Fibonacci trace).

    python tools/gen_golden_caps.py --streamed [--only NAME]      (~1 hour)

The shapes of the weak-scaling bench lines (W=135 at 2^21 / 2^22 / 2^23 rows = BASELINE C5 on 2 / 4 / 8 GPUs) do not fit
the whole-matrix oracle in this container's RAM (C5: 72.5 GB of LDE values).  `--streamed` computes them one LDE coset at
a time from the SAME oracle primitives (ora_ifft, ora_coset_fft with shift g * w_N^j, ora_reverse_index_bits,
ora_merkle_tree): coset j of the rate-1/B LDE is leaf block bitrev(j), i.e. cap subtrees
[bitrev(j) * 2^cap / B, ...), so caps and digest arrays concatenate block by block.  The composition is first checked
against the whole-matrix goldens above (c2_wires and c3_wires must reproduce byte for byte) before anything is written.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from plonky2_amd.util.synthetic import fibonacci_trace, splitmix_columns_numpy  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "commit_caps.json")

# name -> (W, log_n, rate_bits, cap_height, is_values, input kind)
SHAPES = {
    "c2_wires": (135, 16, 3, 4, True, "splitmix"),
    "c3_wires": (135, 20, 3, 4, True, "splitmix"),        # the bench workload
    "c3_constants_sigmas": (84, 20, 3, 4, True, "splitmix"),  # CircuitBuilder::build's commitment (circuit_builder.rs:1182-1191)
    "c3_zs_partial_products": (20, 20, 3, 4, True, "splitmix"),
    "c3_quotient_chunks": (16, 20, 3, 4, False, "splitmix"),
    "c4_fibonacci_trace": (2, 22, 1, 4, True, "fibonacci"),
    # the CPU tier's `bench.py --gpus 2 --log-n 6` on the kernel emulator (tests/test_bench_launch.py): the weak shape and the strong one
    "tiny_wires": (135, 7, 3, 4, True, "splitmix"),
    "tiny_strong_wires": (135, 6, 3, 4, True, "splitmix"),
}


# shapes whose whole write_polynomial_batch byte stream (serialization/mod.rs:1744-1763: coefficients, leaf rows, digests, cap,
# degree_log, rate_bits, blinding) is pinned as well: SURVEY 8f-4 at the size a build-time constants_sigmas / a proof's quotient has
BATCH_BYTES = ("c2_wires", "c3_quotient_chunks", "tiny_wires")


def batch_bytes_sha(ora, o, log_n, rb, cap, rows_per_chunk=1 << 16):
    from tests.wire_format import polynomial_batch_sha256
    leaves = o["leaves"]
    chunks = (leaves[r:r + rows_per_chunk] for r in range(0, leaves.shape[0], rows_per_chunk))
    return polynomial_batch_sha256(o["coeffs"] % np.uint64(ora.P), chunks, o["digests"], o["cap"], cap, log_n, rb, False)


def inputs(name):
    W, log_n, rb, cap, is_values, kind = SHAPES[name]
    if kind == "splitmix":
        return splitmix_columns_numpy(0, W, 1 << log_n)
    return fibonacci_trace(log_n)


def sha_colmajor_from_rows(leaves):
    """SHA-256 of leaves.T (column-major [W][N]) without materialising the transpose"""
    h = hashlib.sha256()
    for c in range(leaves.shape[1]):
        h.update(np.ascontiguousarray(leaves[:, c]).tobytes())
    return h.hexdigest()


# weak-scaling bench shapes (bench.py --gpus 2 / 4 / 8): name -> (W, log_n, rate_bits, cap_height, is_values)
STREAMED = {
    "scale2_wires": (135, 21, 3, 4, True),
    "scale4_wires": (135, 22, 3, 4, True),
    "c5_wires": (135, 23, 3, 4, True),
}


def bitrev(x, bits):
    return int(format(x, "0%db" % bits)[::-1], 2) if bits else 0


def streamed_commit(ora, cols, rb, cap, is_values, threads, fast=None):
    """one coset at a time; returns cap, sha256 of coeffs / digests, and the per-block LDE hashes.
    fast: hash the leaf rows with the tuned AVX-512 sponge (oracle/p2fast.c: 20x the faithful one's speed), spot-checked on
    4096 random rows per block against the faithful ora_hash_no_pad; the tree levels above stay with ora_merkle_tree
    (4-word leaves are their own digests, plonk/config.rs:63-74).  None: the faithful sponge for every leaf."""
    from concurrent.futures import ThreadPoolExecutor
    W, n = cols.shape
    log_n = n.bit_length() - 1
    B = 1 << rb
    assert cap >= rb
    w_N = ora.root_of_unity(log_n + rb)
    pool = ThreadPoolExecutor(threads)
    if is_values:
        coeffs = np.empty_like(cols)

        def inv(c):
            coeffs[c] = ora.ifft(cols[c])
        list(pool.map(inv, range(W)))
    else:
        coeffs = cols
    sha_coeffs = hashlib.sha256()
    for c in range(W):
        sha_coeffs.update((coeffs[c] % np.uint64(ora.P)).tobytes())
    sha_dig = hashlib.sha256()
    caps, lde_blocks = [], []
    block = np.empty((W, n), dtype=np.uint64)
    for b in range(B):                       # leaf blocks in committed order
        j = bitrev(b, rb)                    # ... = LDE coset j: the points g * w_N^(B q + j)
        shift = ora.gl_mul(ora.COSET_SHIFT, ora.gl_pow(w_N, j))

        def lde(c):
            block[c] = ora.reverse_index_bits(ora.coset_fft(coeffs[c], shift))
        list(pool.map(lde, range(W)))
        h = hashlib.sha256()
        for c in range(W):
            h.update(block[c].tobytes())
        lde_blocks.append(h.hexdigest())
        rows = np.ascontiguousarray(block.T)
        if fast is not None and W > 4:
            leaf_digests = fast.hash_rows(rows)
            for i in np.random.default_rng(b).integers(0, n, size=4096):
                assert (ora.hash_no_pad(rows[i]) == leaf_digests[i]).all(), (b, i)
            digests, capb = ora.merkle_tree(leaf_digests, cap - rb)
        else:
            digests, capb = ora.merkle_tree(rows, cap - rb)
        del rows
        sha_dig.update(digests.tobytes())
        caps.append(capb)
        print("   block %d/%d" % (b + 1, B), flush=True)
    return dict(cap=np.concatenate(caps), sha256_coeffs=sha_coeffs.hexdigest(), sha256_digests=sha_dig.hexdigest(),
                sha256_lde_blocks=lde_blocks)


def main_streamed(args):
    from oracle import p2fast
    from oracle import p2oracle as ora
    threads = ora.usable_cores()
    ora.set_num_threads(threads)
    p2fast.set_num_threads(threads)
    fast = None if args.faithful_leaves else p2fast
    out = json.load(open(OUT))
    for name in ("c2_wires", "c3_wires"):   # the composition must reproduce the whole-matrix oracle's bytes first
        W, log_n, rb, cap, is_values, _ = SHAPES[name]
        t0 = time.time()
        r = streamed_commit(ora, splitmix_columns_numpy(0, W, 1 << log_n), rb, cap, is_values, threads, fast)
        g = out[name]
        assert r["cap"].tolist() == g["cap"] and r["sha256_digests"] == g["sha256_digests"] \
            and r["sha256_coeffs"] == g["sha256_coeffs"], name
        print("streamed == whole-matrix oracle on %s (%.0f s)" % (name, time.time() - t0), flush=True)
    for name, (W, log_n, rb, cap, is_values) in STREAMED.items():
        if args.only and name != args.only:
            continue
        t0 = time.time()
        r = streamed_commit(ora, splitmix_columns_numpy(0, W, 1 << log_n), rb, cap, is_values, threads, fast)
        if name in out:   # a second method (e.g. --faithful-leaves after the default) must agree with what is there
            assert out[name]["cap"] == r["cap"].tolist() and out[name]["sha256_digests"] == r["sha256_digests"], name
        out[name] = {"W": W, "log_n": log_n, "rate_bits": rb, "cap_height": cap, "is_values": is_values, "input": "splitmix",
                     "method": "coset-streamed composition of the oracle primitives (checked against c2_wires / c3_wires); leaf sponge: "
                               + ("oracle/p2oracle.c" if fast is None else "oracle/p2fast.c, 4096 rows per block re-hashed by oracle/p2oracle.c"),
                     "cap": [[int(x) for x in row] for row in r["cap"]], "sha256_coeffs": r["sha256_coeffs"],
                     "sha256_digests": r["sha256_digests"], "sha256_lde_blocks": r["sha256_lde_blocks"]}
        print("%s: %.0f s, cap[0] = %s" % (name, time.time() - t0, out[name]["cap"][0]), flush=True)
        with open(OUT, "w") as fh:
            json.dump(out, fh, indent=1)
    print("wrote", OUT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--streamed", action="store_true")
    ap.add_argument("--batch-bytes", action="store_true", help="only the shapes that pin write_polynomial_batch's bytes (BATCH_BYTES)")
    ap.add_argument("--faithful-leaves", action="store_true", help="--streamed: every leaf through the faithful sponge (20x slower)")
    args = ap.parse_args()
    if args.streamed:
        return main_streamed(args)
    from oracle import p2fast as fast
    from oracle import p2oracle as ora
    ora.set_num_threads(ora.usable_cores())
    fast.set_num_threads(ora.usable_cores())
    out = json.load(open(OUT)) if os.path.exists(OUT) else {}
    out["source"] = ("oracle/p2oracle.c (faithful restatement of fri/oracle.rs:57-112) on plonky2_amd/util/synthetic.py "
                     "inputs; generated by tools/gen_golden_caps.py")
    for name, (W, log_n, rb, cap, is_values, kind) in SHAPES.items():
        if args.only and name != args.only:
            continue
        if args.batch_bytes and name not in BATCH_BYTES:
            continue
        cols = inputs(name)
        t0 = time.time()
        o = ora.commit(cols, rb, cap, is_values)
        t1 = time.time()
        rec = {
            "W": W, "log_n": log_n, "rate_bits": rb, "cap_height": cap, "is_values": is_values, "input": kind,
            "cap": [[int(x) for x in row] for row in o["cap"]],
            "sha256_coeffs": hashlib.sha256((o["coeffs"] % np.uint64(ora.P)).tobytes()).hexdigest(),
            "sha256_digests": hashlib.sha256(o["digests"].tobytes()).hexdigest(),
            "sha256_lde": sha_colmajor_from_rows(o["leaves"]),
        }
        if name in BATCH_BYTES:
            rec["sha256_polynomial_batch"] = batch_bytes_sha(ora, o, log_n, rb, cap)
        # the tuned CPU code must produce the same bytes
        f = fast.commit(cols, rb, cap, is_values, want_leaves=False)
        assert (f["cap"] == o["cap"]).all() and (f["digests"] == o["digests"]).all(), name
        assert (f["coeffs"] == o["coeffs"] % np.uint64(ora.P)).all(), name
        if args.batch_bytes and name in out:  # adding one field: everything that was there must be what the oracle computes now
            assert all(out[name][k] == v for k, v in rec.items() if k in out[name]), name
        out[name] = rec
        print("%s: oracle %.1f s, fast %.1f s, cap[0] = %s" % (name, t1 - t0, time.time() - t1, rec["cap"][0]), flush=True)
        del o, f, cols
        with open(OUT, "w") as fh:
            json.dump(out, fh, indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
