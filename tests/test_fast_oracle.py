"""The tuned CPU implementation (oracle/p2fast.c: bench.py's tuned "port" cpu_baseline and the fast checker of the
full-size GPU tests) is pinned bit-for-bit to the faithful restatement (oracle/p2oracle.c) and to the reference's
own Poseidon vectors.  CPU tier."""
import json
import os

import numpy as np
import pytest

from tests.conftest import P, ROOT, rand_field


@pytest.fixture(scope="module")
def fast(ora):
    from oracle import p2fast
    p2fast.set_num_threads(ora.usable_cores())
    return p2fast


def test_fast_poseidon_reference_kats(fast, kats):
    """plonky2/src/hash/poseidon_goldilocks.rs:455-490"""
    for k in kats["poseidon12"]:
        got = fast.poseidon(np.array([k["input"]], dtype=np.uint64))[0]
        assert [int(x) for x in got] == k["output"]


def test_fast_poseidon_vs_oracle_random_and_boundary(fast, ora):
    rng = np.random.default_rng(5)
    st = rng.integers(0, 2**64 - 1, size=(600, 12), dtype=np.uint64)       # any representative < 2^64
    edge = np.array([0, 1, P - 1, P, P + 1, 2**64 - 1, 2**32 - 1, 2**32, 2**63, 0xFFFFFFFF00000000], dtype=np.uint64)
    for i in range(60):
        st[i] = edge[rng.integers(0, len(edge), size=12)]
    ref = np.stack([ora.poseidon(s) for s in st])
    assert (fast.poseidon(st) == ref).all()


def test_fast_hash_rows_every_width(fast, ora):
    """hash_or_noop (plonk/config.rs:63-74) incl. the no-hash widths <= 4 and every last-chunk length"""
    rng = np.random.default_rng(6)
    for w in list(range(0, 20)) + [135]:
        rows = rand_field(rng, 5, w, noncanonical=True) if w else np.zeros((5, 0), dtype=np.uint64)
        exp = np.stack([ora.hash_or_noop(r) for r in rows])
        assert (fast.hash_rows(rows) == exp).all(), w


@pytest.mark.parametrize("W,log_n,rb,cap,is_values", [
    (135, 9, 3, 4, True), (20, 8, 3, 4, True), (16, 7, 3, 4, False), (2, 9, 1, 4, True),   # the BASELINE shapes, small
    (5, 4, 2, 0, True), (3, 3, 3, 6, True), (1, 0, 3, 2, True), (9, 5, 0, 2, False), (4, 6, 3, 9, True), (7, 1, 1, 1, False),
])
def test_fast_commit_equals_faithful_oracle(fast, ora, W, log_n, rb, cap, is_values):
    rng = np.random.default_rng(W * 100 + log_n)
    cols = rand_field(rng, W, 1 << log_n, noncanonical=True)
    o = ora.commit(cols, rb, cap, is_values)
    tm = {}
    f = fast.commit(cols, rb, cap, is_values, want_leaves=True, timed=tm)
    assert (f["coeffs"] == o["coeffs"] % np.uint64(P)).all()
    assert (f["leaves"] == o["leaves"]).all()
    assert (f["digests"] == o["digests"]).all()
    assert (f["cap"] == o["cap"]).all()
    assert set(tm) == {"IFFT", "FFT + blinding", "transpose LDEs", "build Merkle tree"}   # fri/oracle.rs:65-103


def test_golden_caps_file_matches_the_oracle_on_the_small_config(ora):
    """tests/golden/commit_caps.json was produced by the faithful oracle (tools/gen_golden_caps.py); re-derive its C2
    entry here (15 s on 8 cores is too long for every shape, C2 pins the generator and the input recipe)"""
    import hashlib
    from plonky2_amd.util.synthetic import splitmix_columns_numpy
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "commit_caps.json")))
    for name in ("c2_wires", "c3_wires", "c3_constants_sigmas", "c3_zs_partial_products", "c3_quotient_chunks", "c4_fibonacci_trace"):
        assert name in g and len(g[name]["cap"]) == 1 << g[name]["cap_height"]
    r = g["c2_wires"]
    cols = splitmix_columns_numpy(0, r["W"], 1 << 12)   # the recipe at a smaller row count: cheap regression of the generator
    assert cols.shape == (135, 4096) and int(cols.max()) < P
    from oracle import p2fast
    f = p2fast.commit(splitmix_columns_numpy(0, r["W"], 1 << r["log_n"]), r["rate_bits"], r["cap_height"], True)
    assert [[int(x) for x in row] for row in f["cap"]] == r["cap"]
    assert hashlib.sha256(f["digests"].tobytes()).hexdigest() == r["sha256_digests"]
    assert hashlib.sha256(f["coeffs"].tobytes()).hexdigest() == r["sha256_coeffs"]


def test_fast_scalar_and_avx512_paths_agree(ora):
    """oracle/p2fast.c picks its AVX-512 leaf sponge (eight rows per instruction stream) at run time; P2FAST_SCALAR=1
    forces the scalar path.  Both must produce the faithful oracle's bytes -- run the scalar one in a fresh process (the
    choice is cached per process)."""
    import subprocess
    import sys
    code = ("import numpy as np, sys; sys.path.insert(0, %r)\n"
            "from oracle import p2fast as f, p2oracle as o\n"
            "rng = np.random.default_rng(5)\n"
            "c = rng.integers(0, 2**64 - 1, size=(19, 64), dtype=np.uint64)\n"
            "a, b = f.commit(c, 3, 2, True, want_leaves=True), o.commit(c, 3, 2, True)\n"
            "assert (a['digests'] == b['digests']).all() and (a['cap'] == b['cap']).all() and (a['leaves'] == b['leaves']).all()\n"
            "st = rng.integers(0, 2**64 - 1, size=(21, 12), dtype=np.uint64)\n"
            "assert (f.poseidon(st) == np.stack([o.poseidon(s) for s in st])).all()\n"
            "print('ok')\n" % ROOT)
    for env in ({"P2FAST_SCALAR": "1"}, {}):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env={**os.environ, **env})
        assert r.returncode == 0 and "ok" in r.stdout, r.stderr
