"""Parity of the HIP path (through the C ABI) against the CPU oracle and the reference's KATs.

Every test is parametrised over `eng`: "emu" runs the same kernel sources under the test-only
emulator (CPU tier, small sizes), "gpu" is the real thing on the MI355X (-m gpu).  Bit-exact: the
path is integer arithmetic mod P.
"""
import numpy as np
import pytest

from tests.conftest import P, rand_field
from tests.emu_backend import EMU_TUNE_QUAD, EMU_TUNE_ROW


def is_gpu(eng):
    return not eng.lib.p2hot_is_emulated()


# ---------------------------------------------------------------- field arithmetic
def test_field_ops_edge_grid(eng):
    """add / sub / mul (compiler-scheduled, hand-scheduled single stream, 3-way interleaved stream) on the
    cartesian grid of edge values the reference uses (field/src/prime_field_testing.rs:8-17: 0, 1, small, 2^32 +- 1,
    P - 1, and non-canonical representatives up to 2^64 - 1) plus random pairs, against Python big integers"""
    rng = np.random.default_rng(1)
    E32 = 2**32
    edge = [0, 1, 2, 3, E32 - 2, E32 - 1, E32, E32 + 1, 2**63 - 1, 2**63, 2**63 + 1, P - E32, P - 2, P - 1, P, P + 1,
            P + E32 - 2, 2**64 - E32, 2**64 - 2, 2**64 - 1, 0xFFFFFFFF_00000000, 0x00000000_FFFFFFFF,
            0xFFFFFFFE_FFFFFFFF, 0x80000000_80000000]
    a = [x for x in edge for _ in edge] + [int(v) for v in rng.integers(0, 2**63, 4000, dtype=np.uint64) * 2 + 1]
    b = [y for _ in edge for y in edge] + [int(v) for v in rng.integers(0, 2**63, 4000, dtype=np.uint64) * 2 + rng.integers(0, 2, 4000, dtype=np.uint64)]
    n = len(a)
    da, db = eng.dev(np.array(a, dtype=np.uint64)), eng.dev(np.array(b, dtype=np.uint64))
    out = eng.mem.zeros(6, n)
    eng.check(eng.lib.p2hot_field_selftest_dev(eng.ctx, eng.ptr(da), eng.ptr(db), n, eng.ptr(out)))
    o = eng.host(out)
    mul = np.array([x * y % P for x, y in zip(a, b)], dtype=np.uint64)
    assert (o[0] == mul).all() and (o[1] == mul).all() and (o[2] == mul).all() and (o[5] == 0).all()
    assert (o[3] == np.array([(x + y) % P for x, y in zip(a, b)], dtype=np.uint64)).all()
    assert (o[4] == np.array([(x - y) % P for x, y in zip(a, b)], dtype=np.uint64)).all()


# ---------------------------------------------------------------- Poseidon
def test_poseidon_reference_kats(eng, kats):
    """plonky2/src/hash/poseidon_goldilocks.rs:455-490"""
    from plonky2_amd.hash.poseidon import poseidon
    inp = np.array([k["input"] for k in kats["poseidon12"]], dtype=np.uint64)
    exp = np.array([k["output"] for k in kats["poseidon12"]], dtype=np.uint64)
    assert (poseidon(inp, eng) == exp).all()


def test_poseidon_random_vs_oracle(eng, ora):
    from plonky2_amd.hash.poseidon import poseidon
    rng = np.random.default_rng(11)
    count = 4096 if is_gpu(eng) else 200
    s = rand_field(rng, count, 12, noncanonical=True)
    s[0] = P - 1
    s[1] = np.uint64(2**64 - 1)  # non-canonical representatives are accepted like in the reference
    got = poseidon(s, eng)
    exp = np.stack([ora.poseidon(x) for x in s])
    assert (got == exp).all()
    assert (got < np.uint64(P)).all()


def test_poseidon_edge_states_vs_oracle(eng, ora):
    """states built from the boundary words of the representation (0, 1, 2^32 -+ 1, P -+ 1, P, 2^64 - 1) in every
    position pattern: the carry / borrow folds of the hand-written multiply and row-fold streams see their extremes"""
    from plonky2_amd.hash.poseidon import poseidon
    edge = [0, 1, 2**32 - 1, 2**32, 2**32 + 1, P - 1, P, P + 1, 2**63, 2**64 - 2**32, 2**64 - 1]
    rng = np.random.default_rng(3)
    states = []
    for v in edge:
        states.append([v] * 12)                                   # the same word everywhere
        states.append([v if i % 2 else edge[(i + 3) % len(edge)] for i in range(12)])
        states.append([v if i == 0 else 0 for i in range(12)])    # only the word the partial rounds raise
    for _ in range(64 + 31):                                       # random mixtures of edge words
        states.append([edge[int(k)] for k in rng.integers(0, len(edge), 12)])
    s = np.array(states, dtype=np.uint64)
    got = poseidon(s.copy(), eng)                                  # one permutation per lane (poseidon.hpp)
    exp = np.stack([ora.poseidon(x) for x in s])
    assert (got == exp).all()
    # the quad-cooperative kernel (poseidon4.hpp) through the leaf hash of a small batch: 8-word rows = one permutation
    from plonky2_amd.hash.poseidon import hash_or_noop_batch
    rows = s[:64, :8].copy()
    assert (hash_or_noop_batch(rows, eng) == np.stack([ora.hash_or_noop(r) for r in rows])).all()


def test_hash_no_pad_and_two_to_one(eng, ora):
    from plonky2_amd.hash.poseidon import hash_no_pad, hash_or_noop_batch, two_to_one
    rng = np.random.default_rng(12)
    # every length of the last chunk (the lane kernel skips output words the next chunk overwrites); 2 and 32 rows
    # put the launch on the quad-cooperative and on the one-permutation-per-lane kernel respectively
    for w in (1, 3, 4, 5, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 23, 135):
        for nrows in (2, 32):
            rows = rand_field(rng, nrows, w, noncanonical=True)
            got = hash_or_noop_batch(rows, eng)
            exp = np.stack([ora.hash_or_noop(r) for r in rows])
            assert (got == exp).all(), (w, nrows)
        assert (hash_no_pad(rows[0], eng) == ora.hash_no_pad(rows[0])).all(), w
    l, r = rand_field(rng, 4), rand_field(rng, 4)
    assert (two_to_one(l, r, eng) == ora.two_to_one(l, r)).all()


# ---------------------------------------------------------------- bit reversal / transpose
def test_reverse_index_bits_reference_table(eng, kats):
    """plonky2/src/util/mod.rs:61-126"""
    from plonky2_amd.util import reverse_index_bits
    got = reverse_index_bits(np.arange(256, dtype=np.uint64), eng)
    assert got.tolist() == kats["reverse_index_bits_256"]


def test_transpose(eng):
    from plonky2_amd.util import transpose
    rng = np.random.default_rng(13)
    for (w, rows) in ((1, 1), (5, 64), (33, 100), (135, 130)):
        m = rand_field(rng, w, rows)
        assert (transpose(m, eng) == m.T).all()


# ---------------------------------------------------------------- NTT
@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 7, 10, 12, 13, 15, 16, 18, 19, 21])
def test_fft_ifft_vs_oracle(eng, ora, log_n):
    """field/src/fft.rs:215-249 semantics: natural order in and out; ifft(fft(x)) == x.  The sizes walk the strided-pass
    shapes: 2^16 (4 bits), 2^18 (6: radix-8 last round, stored from LDS), 2^19 (7), 2^21 (9), beside 2^20 / 2^22 / 2^23 of the
    full-size tests"""
    from plonky2_amd.field.fft import fft, ifft
    rng = np.random.default_rng(100 + log_n)
    a = rand_field(rng, 3 if log_n <= 15 else 1, 1 << log_n, noncanonical=True)
    f = fft(a, eng)
    Pm = np.uint64(P)  # equality in the reference is canonical-value equality (goldilocks_field.rs:33-37)
    assert (f == np.stack([ora.fft(x.copy()) for x in a]) % Pm).all()
    assert (ifft(a, eng) == np.stack([ora.ifft(x.copy()) for x in a]) % Pm).all()
    assert (ifft(f, eng) == a % np.uint64(P)).all()


@pytest.mark.parametrize("log_n", [3, 6, 12, 14])
def test_fft_boundary_words_vs_oracle(eng, ora, log_n):
    """transforms of vectors made of the representation's boundary words (0, 1, 2^32 -+ 1, P -+ 1, P, 2^64 - 1):
    the butterflies' add / sub double folds and the hand-written power-of-two twiddle multiplies at their extremes"""
    from plonky2_amd.field.fft import fft, ifft
    edge = np.array([0, 1, 2**32 - 1, 2**32, 2**32 + 1, P - 1, P, P + 1, 2**63, 2**64 - 2**32, 2**64 - 1], dtype=np.uint64)
    rng = np.random.default_rng(log_n)
    n = 1 << log_n
    rows = [np.full(n, v, dtype=np.uint64) for v in (P - 1, 2**64 - 1, 2**32 - 1)]
    rows += [edge[rng.integers(0, len(edge), n)] for _ in range(3)]
    a = np.stack(rows)
    f = fft(a.copy(), eng)
    for k in range(len(a)):
        assert (f[k] == ora.fft(a[k].copy()) % np.uint64(P)).all(), k
    back = ifft(f.copy(), eng)
    assert (back == a % np.uint64(P)).all()


def test_fft_three_pass_sizes(eng, ora):
    """log_n = 22 is the largest two-pass size (a 10-bit strided pass: 1024 x 4 tiles, the workgroups sharing a 128-byte
    line placed on one XCD); log_n >= 23 takes three passes"""
    from plonky2_amd.field.fft import ifft
    from plonky2_amd.field.fft import fft
    rng = np.random.default_rng(5)
    # pass plans (9,12), (10,12) and (6,5,12); 2^23 is the per-rank transform of config C5.  The CPU tier runs 2^22 only
    for log_n in ((21, 22, 23) if is_gpu(eng) else (22,)):
        a = rand_field(rng, 1, 1 << log_n)
        f = fft(a, eng)
        assert (f[0] == ora.fft(a[0].copy())).all(), log_n
        assert (ifft(f, eng) == a).all(), log_n


def test_coset_ifft_vs_oracle(eng, ora):
    """polynomial/mod.rs:477-516: coset_ifft inverts coset_fft; bit-exact vs the oracle"""
    from plonky2_amd.field.polynomial import coset_ifft, lde_coset_fft
    rng = np.random.default_rng(31)
    for log_n in (0, 1, 5, 9, 13):
        v = rand_field(rng, 2, 1 << log_n, noncanonical=True)
        got = coset_ifft(v, engine=eng)
        assert (got == np.stack([ora.coset_ifft(x.copy()) for x in v]) % np.uint64(P)).all(), log_n
        assert (lde_coset_fft(got, 0, engine=eng) == v % np.uint64(P)).all(), log_n
    s = 123456789
    v = rand_field(rng, 1, 64)
    assert (coset_ifft(v, shift=s, engine=eng)[0] == ora.coset_ifft(v[0].copy(), shift=s)).all()


def test_ntt_kernel_variants_agree(eng, ora):
    """p2hot_tune_ntt: LDS radix-2 layers (0), register radix 8 on 64-bit words (8), on 24-bit limbs (3, default) and
    radix 16 (4) are the same function"""
    from plonky2_amd.field.fft import fft, ifft
    rng = np.random.default_rng(21)
    a = rand_field(rng, 2, 1 << 13)
    exp = np.stack([ora.fft(x.copy()) for x in a])
    try:
        for mode in (0, 4, 8, 3):
            eng.check(eng.lib.p2hot_tune_ntt(eng.ctx, mode))
            assert (fft(a, eng) == exp).all(), mode
            assert (ifft(exp, eng) == a).all(), mode
    finally:
        eng.check(eng.lib.p2hot_tune_ntt(eng.ctx, 3))


def test_coset_lde_vs_oracle_and_naive(eng, ora):
    """polynomial/mod.rs:477-516: coset FFT == evaluation on {shift * w^i}"""
    from plonky2_amd.field.polynomial import lde_coset_fft
    from tests import pyref
    rng = np.random.default_rng(14)
    for (log_n, rb) in ((3, 3), (5, 1), (8, 3), (12, 2), (13, 1)):
        n = 1 << log_n
        co = rand_field(rng, 2, n)
        got = lde_coset_fft(co, rb, engine=eng)
        for c in range(2):
            pad = np.zeros(n << rb, dtype=np.uint64)
            pad[:n] = co[c]
            assert (got[c] == ora.coset_fft(pad, zero_factor=rb)).all(), (log_n, rb)
        if log_n == 3:
            rows = pyref.naive_coset_lde_rows([int(x) for x in co[0]], rb)
            br = lde_coset_fft(co, rb, bit_reversed=True, engine=eng)[0]
            assert [int(x) for x in br] == rows
    # a different shift (FRI rounds use shift^arity)
    co = rand_field(rng, 1, 64)
    s = pow(ora.COSET_SHIFT, 16, P)
    pad = np.zeros(128, dtype=np.uint64)
    pad[:64] = co[0]
    assert (lde_coset_fft(co, 1, shift=s, engine=eng)[0] == ora.coset_fft(pad, shift=s, zero_factor=1)).all()


# ---------------------------------------------------------------- Merkle
@pytest.mark.parametrize("n,w,cap", [(1, 7, 0), (2, 3, 0), (2, 3, 1), (16, 4, 2), (64, 5, 0), (64, 135, 4), (256, 32, 8),
                                     (1024, 20, 4)])
def test_merkle_tree_vs_oracle(eng, ora, n, w, cap):
    """merkle_tree.rs:253-311: same digests (reference layout), same cap, every path verifies"""
    from plonky2_amd.hash.merkle_tree import MerkleTree
    rng = np.random.default_rng(n * 1000 + w)
    leaves = rand_field(rng, n, w, noncanonical=True)
    tree = MerkleTree.new(leaves, cap, engine=eng)
    digests, capv = ora.merkle_tree(leaves, cap)
    assert (tree.cap.entries == capv).all()
    assert (np.asarray(tree.digests).reshape(-1, 4) == digests).all()
    for i in sorted({0, n - 1, n // 2, (n * 3) // 7}):
        proof = tree.prove(i)
        assert (proof == ora.merkle_prove(i, n, cap, digests)).all()
        assert ora.merkle_verify(leaves[i], i, capv, proof)


# ---------------------------------------------------------------- PolynomialBatch
COMMIT_CASES = [  # (W, log_n, rate_bits, cap_height, is_values)
    (3, 4, 3, 4, True), (9, 5, 3, 2, True), (135, 6, 3, 4, True), (2, 7, 1, 4, False), (16, 10, 3, 4, False),
    (20, 9, 3, 4, True), (5, 13, 1, 0, True), (1, 3, 3, 6, True), (4, 2, 1, 3, True),
    (1, 0, 0, 0, True), (2, 0, 3, 1, False),
]


@pytest.mark.parametrize("W,log_n,rb,cap,is_values", COMMIT_CASES)
def test_polynomial_batch_vs_oracle(eng, ora, W, log_n, rb, cap, is_values):
    """from_values / from_coeffs (fri/oracle.rs:57-112): polynomials, leaves, digests and cap bit-exact"""
    from plonky2_amd.fri.oracle import PolynomialBatch
    rng = np.random.default_rng(W * 131 + log_n * 7 + rb)
    cols = rand_field(rng, W, 1 << log_n, noncanonical=True)
    ctor = PolynomialBatch.from_values if is_values else PolynomialBatch.from_coeffs
    b = ctor(cols, rb, False, cap, engine=eng)
    o = ora.commit(cols, rb, cap, is_values)
    assert (b.polynomials == o["coeffs"] % np.uint64(P)).all()  # canonical-value equality (goldilocks_field.rs:33-37)
    assert (b.merkle_tree.cap.entries == o["cap"]).all()
    assert (np.asarray(b.merkle_tree.digests).reshape(-1, 4) == o["digests"]).all()
    assert (b.merkle_tree.leaves.reshape(o["leaves"].shape) == o["leaves"]).all()
    if W:
        N = 1 << (log_n + rb)
        for idx in {0, N - 1, N // 3}:
            from plonky2_amd.util import reverse_bits
            assert (b.get_lde_values(idx, 1) == o["leaves"][reverse_bits(idx, log_n + rb)]).all()
        # the row-major `leaves` output of p2hot_commit_dev is the same matrix
        r = eng.commit(eng.dev(cols), log_n, rb, cap, is_values, want_leaves=True)
        assert (eng.host(r["leaves"]) == o["leaves"]).all()


def test_polynomial_batch_without_polynomials_is_refused(eng):
    """W = 0: the reference panics on polynomials[0].len() (fri/oracle.rs:90); the mirror raises"""
    from plonky2_amd import _lib
    from plonky2_amd.fri.oracle import PolynomialBatch
    with pytest.raises(_lib.P2HotError, match="no polynomials"):
        PolynomialBatch.from_values(np.zeros((0, 8), dtype=np.uint64), 1, False, 2, engine=eng)
    with pytest.raises(_lib.P2HotError, match="no polynomials"):
        eng.commit(eng.dev(np.zeros((0, 8), dtype=np.uint64)), 3, 1, 2, True)


def test_commit_row_ranges_assemble_to_the_full_tree(eng, ora):
    """the multi-GPU unit: whole coset blocks computed independently give the same tree"""
    rng = np.random.default_rng(77)
    W, log_n, rb, cap = 6, 6, 3, 4
    cols = rand_field(rng, W, 1 << log_n)
    o = ora.commit(cols, rb, cap, True)
    N, n = 1 << (log_n + rb), 1 << log_n
    nd = eng.num_digests(log_n + rb, cap)
    digests, capbuf = eng.mem.zeros(nd, 4), eng.mem.zeros(1 << cap, 4)
    d_cols = eng.dev(cols)
    for part in range(4):
        r = eng.commit(d_cols, log_n, rb, cap, True, row_begin=part * 2 * n, row_count=2 * n, want_leaves=True,
                       digests=digests, cap=capbuf)
        assert (eng.host(r["leaves"]) == o["leaves"][part * 2 * n:(part + 1) * 2 * n]).all()
    assert (eng.host(digests) == o["digests"]).all()
    assert (eng.host(capbuf) == o["cap"]).all()


@pytest.mark.parametrize("shape", [(5, 4, 3, 2, True), (9, 6, 1, 3, False), (1, 3, 2, 0, True), (135, 5, 3, 4, True)])
def test_salted_commit_vs_oracle(eng, ora, shape):
    """blinding = true (fri/oracle.rs:114-139, standard_recursion_zk_config): SALT_SIZE = 4 caller-supplied random
    vectors become leaf columns W..W+3 (through transpose + reverse_index_bits like the LDE values), are hashed by the
    leaf sponge and come back from MerkleTree::get; get_lde_values strips them (:146)"""
    import ctypes as C
    from plonky2_amd.fri.oracle import PolynomialBatch
    W, log_n, rb, cap, is_values = shape
    rng = np.random.default_rng(W * 31 + log_n)
    n, N, S = 1 << log_n, 1 << (log_n + rb), 4
    cols = [rand_field(rng, n) for _ in range(W)]
    salts = rand_field(rng, S, N, noncanonical=True)
    o = ora.commit_salted(np.stack(cols), salts, rb, cap, is_values)
    ptrs = (C.c_void_p * W)(*[c.ctypes.data for c in cols])
    sptrs = (C.c_void_p * S)(*[salts[j].ctypes.data for j in range(S)])
    coeffs = np.zeros((W, n), dtype=np.uint64)
    leaves = np.zeros((N, W + S), dtype=np.uint64)
    digests = np.zeros((eng.num_digests(log_n + rb, cap), 4), dtype=np.uint64)
    capv = np.zeros((1 << cap, 4), dtype=np.uint64)
    handle = C.c_void_p()
    eng.check(eng.lib.p2hot_commit_salted(eng.ctx, ptrs, W, log_n, rb, cap, 1 if is_values else 0, 0, sptrs, S, coeffs.ctypes.data,
                                          leaves.ctypes.data, digests.ctypes.data, capv.ctypes.data, C.byref(handle)))
    assert (coeffs == o["coeffs"]).all() and (leaves == o["leaves"]).all()
    assert (digests == o["digests"]).all() and (capv == o["cap"]).all()
    assert eng.lib.p2hot_batch_width(handle) == W and eng.lib.p2hot_batch_leaf_width(handle) == W + S
    idx = np.array([0, N // 3, N - 1], dtype=np.uint64)
    rows = np.zeros((3, W + S), dtype=np.uint64)
    eng.check(eng.lib.p2hot_batch_rows(handle, idx.ctypes.data, 3, rows.ctypes.data))
    assert (rows == o["leaves"][idx.astype(np.int64)]).all()
    eng.lib.p2hot_batch_free(handle)
    # the salts change the tree, not the polynomials; without salts the unblinded commitment comes back
    plain = ora.commit(np.stack(cols), rb, cap, is_values)
    assert (plain["coeffs"] == o["coeffs"]).all() and (plain["cap"] != o["cap"]).any()
    # the mirror of the reference signature: from_values / from_coeffs(…, blinding = true) with the caller's salts
    build = PolynomialBatch.from_values if is_values else PolynomialBatch.from_coeffs
    b = build(np.stack(cols), rb, True, cap, engine=eng, salts=salts)
    assert b.blinding and (np.asarray(b.merkle_tree.cap.entries) == o["cap"]).all()
    L = 5 % N
    rev = int(format(L, "0%db" % (log_n + rb))[::-1], 2) if log_n + rb else 0
    assert (np.asarray(b.get_lde_values(L)).reshape(-1) == o["leaves"][rev][:W]).all()
    assert (np.asarray(b.merkle_tree.get(rev)).reshape(-1) == o["leaves"][rev]).all()


@pytest.mark.parametrize("shape", [(4, 5, 6, 3, 2, True), (2, 135, 5, 3, 4, True), (8, 3, 4, 1, 0, False), (1, 7, 5, 2, 1, True)])
def test_commit_many_equals_separate_commits(eng, ora, shape):
    """p2hot_commit_many: M same-shape commitments sharing every launch (columns interleaved [W][M][n], one forest of
    M * 2^cap subtrees) -- coefficients, digests and caps of each proof equal its own from_values / from_coeffs
    (fri/oracle.rs:57-112), and each proof's handle serves rows, paths and openings like a separate commitment's"""
    import ctypes as C
    from plonky2_amd.fri.oracle import FriBatchInfo, PolynomialBatch, prove_openings
    from plonky2_amd.iop.challenger import Challenger
    M, W, log_n, rb, cap, is_values = shape
    rng = np.random.default_rng(M * 1000 + W)
    n, N = 1 << log_n, 1 << (log_n + rb)
    cols = [[rand_field(rng, n, noncanonical=True) for _ in range(W)] for _ in range(M)]
    ptrs = (C.c_void_p * (M * W))(*[cols[m][e].ctypes.data for m in range(M) for e in range(W)])
    nd = eng.num_digests(log_n + rb, cap)
    coeffs = np.zeros((M, W, n), dtype=np.uint64)
    digests = np.zeros((M, max(nd, 1), 4), dtype=np.uint64)
    caps = np.zeros((M, 1 << cap, 4), dtype=np.uint64)
    handles = (C.c_void_p * M)()
    eng.check(eng.lib.p2hot_commit_many(eng.ctx, ptrs, M, W, log_n, rb, cap, 1 if is_values else 0, coeffs.ctypes.data,
                                        digests.ctypes.data, caps.ctypes.data, handles))
    idx = np.array([0, N // 2 + 1, N - 1], dtype=np.uint64)
    for m in range(M):
        o = ora.commit(np.stack(cols[m]), rb, cap, is_values)
        o["coeffs"] = o["coeffs"] % np.uint64(P)  # from_coeffs hands the caller's representatives through; outputs here are canonical
        assert (coeffs[m] == o["coeffs"]).all(), m
        assert (caps[m] == o["cap"]).all(), m
        if nd:
            assert (digests[m][:nd] == o["digests"]).all(), m
        rows = np.zeros((3, W), dtype=np.uint64)
        eng.check(eng.lib.p2hot_batch_rows(handles[m], idx.ctypes.data, 3, rows.ctypes.data))
        assert (rows == o["leaves"][idx.astype(np.int64)]).all(), m
        layers = log_n + rb - cap
        if layers:
            paths = np.zeros((3, layers, 4), dtype=np.uint64)
            eng.check(eng.lib.p2hot_batch_paths(handles[m], idx.ctypes.data, 3, paths.ctypes.data))
            for q, i in enumerate(idx):
                assert ora.merkle_verify(rows[q], int(i), o["cap"], paths[q])
        back = np.zeros((W, n), dtype=np.uint64)
        eng.check(eng.lib.p2hot_batch_coeffs(handles[m], 0, W, back.ctypes.data))
        assert (back == o["coeffs"]).all()
    # an opening proof over a member of the batch equals the one over a separate commitment of the same proof
    if rb >= 1 and log_n >= 5 and cap <= log_n:
        m = M - 1
        sep = PolynomialBatch.from_values(np.stack(cols[m]), rb, False, cap, engine=eng) if is_values else \
            PolynomialBatch.from_coeffs(np.stack(cols[m]), rb, False, cap, engine=eng)
        mem = PolynomialBatch(eng, C.c_void_p(handles[m]), W, log_n, rb, cap, caps[m])
        handles[m] = None  # owned by `mem` now
        pfs = []
        for b in (sep, mem):
            ch = Challenger(eng)
            ch.observe_cap(caps[m])
            zeta = ch.get_extension_challenge()
            pfs.append(prove_openings([FriBatchInfo(zeta, [(0, pi) for pi in range(W)])], [b], ch, rb, cap, [2, 1], 3, 4, engine=eng))
        assert repr(pfs[0]) == repr(pfs[1])
        del mem
    for m in range(M):
        if handles[m]:
            eng.lib.p2hot_batch_free(handles[m])
    assert eng.lib.p2hot_ctx_trim(eng.ctx) == 0
    # shape errors
    assert eng.lib.p2hot_commit_many(eng.ctx, ptrs, 3, W, log_n, rb, cap, 1, None, None, caps.ctypes.data, None) != 0 or M < 3
    # the device-pointer form on the interleaved layout [W][M][n] -> LDE [W][M][N], digests [M][nd][4], caps [M][2^cap][4]
    inter = np.stack([np.stack([cols[m][e] for m in range(M)]) for e in range(W)])          # [W][M][n]
    d_cols = eng.dev(inter.reshape(W * M, n))
    d_lde, d_dig, d_cap = eng.mem.empty(W * M, N), eng.mem.empty(M * max(nd, 1), 4), eng.mem.empty(M << cap, 4)
    eng.check(eng.lib.p2hot_commit_many_dev(eng.ctx, eng.ptr(d_cols), M, W, log_n, rb, cap, 1 if is_values else 0, eng.ptr(d_lde),
                                            eng.ptr(d_dig), eng.ptr(d_cap)))
    assert (eng.host(d_cap).reshape(M, 1 << cap, 4) == caps).all()
    lde = eng.host(d_lde).reshape(W, M, N)
    o0 = ora.commit(np.stack(cols[0]), rb, cap, is_values)
    assert (lde[:, 0, :].T == o0["leaves"]).all()
    assert (eng.host(d_cols).reshape(W, M, n)[:, 0, :] % np.uint64(P) == o0["coeffs"] % np.uint64(P)).all()


def test_host_pointer_commit_abi(eng, ora):
    """p2hot_commit / p2hot_batch_rows: the entry points the Rust shim binds (W separate host vectors)"""
    import ctypes as C
    rng = np.random.default_rng(78)
    W, log_n, rb, cap = 7, 5, 3, 4
    n, N = 1 << log_n, 1 << (log_n + rb)
    cols = [rand_field(rng, n) for _ in range(W)]
    ptrs = (C.c_void_p * W)(*[c.ctypes.data for c in cols])
    coeffs = np.zeros((W, n), dtype=np.uint64)
    leaves = np.zeros((N, W), dtype=np.uint64)
    nd = eng.num_digests(log_n + rb, cap)
    digests = np.zeros((nd, 4), dtype=np.uint64)
    capv = np.zeros((1 << cap, 4), dtype=np.uint64)
    handle = C.c_void_p()
    eng.check(eng.lib.p2hot_commit(eng.ctx, ptrs, W, log_n, rb, cap, 1, 0, coeffs.ctypes.data, leaves.ctypes.data,
                                   digests.ctypes.data, capv.ctypes.data, C.byref(handle)))
    o = ora.commit(np.stack(cols), rb, cap, True)
    assert (coeffs == o["coeffs"]).all() and (leaves == o["leaves"]).all()
    assert (digests == o["digests"]).all() and (capv == o["cap"]).all()
    idx = np.array([0, 5, N - 1], dtype=np.uint64)
    rows = np.zeros((3, W), dtype=np.uint64)
    eng.check(eng.lib.p2hot_batch_rows(handle, idx.ctypes.data, 3, rows.ctypes.data))
    assert (rows == o["leaves"][idx.astype(np.int64)]).all()
    # Merkle paths from the handle's device-resident digests (merkle_tree_prove, merkle_tree.rs:151-190)
    layers = log_n + rb - cap
    paths = np.zeros((3, layers, 4), dtype=np.uint64)
    eng.check(eng.lib.p2hot_batch_paths(handle, idx.ctypes.data, 3, paths.ctypes.data))
    for q, i in enumerate(idx):
        assert (paths[q] == ora.merkle_prove(int(i), N, cap, o["digests"])).all()
        assert ora.merkle_verify(rows[q], int(i), o["cap"], paths[q])
    bad = np.array([N], dtype=np.uint64)
    assert eng.lib.p2hot_batch_paths(handle, bad.ctypes.data, 1, paths.ctypes.data) != 0
    eng.lib.p2hot_batch_free(handle)
    # digests_out = NULL: the digest array never leaves the device, paths still come from the handle
    handle = C.c_void_p()
    eng.check(eng.lib.p2hot_commit(eng.ctx, ptrs, W, log_n, rb, cap, 1, 0, None, None, None, capv.ctypes.data, C.byref(handle)))
    eng.check(eng.lib.p2hot_batch_paths(handle, idx.ctypes.data, 3, paths.ctypes.data))
    assert (paths[1] == ora.merkle_prove(int(idx[1]), N, cap, o["digests"])).all() and (capv == o["cap"]).all()
    eng.lib.p2hot_batch_free(handle)
    # P2HOT_COEFFS_PER_COLUMN: coeffs_out is a table of W destinations (the Rust side's Vec<PolynomialCoeffs>): each polynomial
    # lands in its own vector, nothing to split afterwards
    from plonky2_amd import _lib
    dst = [np.zeros(n, dtype=np.uint64) for _ in range(W)]
    table = (C.c_void_p * W)(*[d.ctypes.data for d in dst])
    eng.check(eng.lib.p2hot_commit(eng.ctx, ptrs, W, log_n, rb, cap, 1, _lib.COEFFS_PER_COLUMN, C.cast(table, C.c_void_p), None, None,
                                   capv.ctypes.data, None))
    assert (np.stack(dst) == o["coeffs"]).all() and (capv == o["cap"]).all()
    h = C.c_void_p()
    eng.check(eng.lib.p2hot_cols_upload(eng.ctx, ptrs, W, log_n, C.byref(h)))
    dst = [np.zeros(n, dtype=np.uint64) for _ in range(W)]
    table = (C.c_void_p * W)(*[d.ctypes.data for d in dst])
    eng.check(eng.lib.p2hot_commit_cols(eng.ctx, h, rb, cap, 1, _lib.COEFFS_PER_COLUMN, C.cast(table, C.c_void_p), None, None, capv.ctypes.data, None))
    assert (np.stack(dst) == o["coeffs"]).all() and (capv == o["cap"]).all()
    table[W - 1] = None  # a null destination is refused before anything runs
    assert eng.lib.p2hot_commit(eng.ctx, ptrs, W, log_n, rb, cap, 1, _lib.COEFFS_PER_COLUMN, C.cast(table, C.c_void_p), None, None,
                                capv.ctypes.data, None) == _lib.EINVAL


@pytest.mark.parametrize("W,log_n,rb,cap,S,block", [(37, 5, 3, 2, 0, 16), (37, 5, 3, 2, 4, 16), (135, 4, 1, 0, 0, 16),
                                                     (24, 3, 2, 5, 0, 8), (20, 4, 0, 1, 4, 3), (9, 3, 1, 2, 0, 5)])
def test_host_commit_in_blocks_hashes_as_columns_arrive(eng, ora, monkeypatch, W, log_n, rb, cap, S, block):
    """p2hot_commit with more than one column block (what every commit of >= 2^22 values does): the leaf sponge absorbs the
    complete 8-column chunks after each block's LDE, its state parked between launches, the salts' chunks last
    (hashing.rs:118-145 overwrite-mode sponge, oracle.rs:123-137 salts) -- same tree as the one-launch sponge and the oracle;
    block widths that are not multiples of the sponge rate, a last chunk shorter than the rate, all-cap trees"""
    import ctypes as C
    from plonky2_amd.engine import Engine
    monkeypatch.setenv("P2HOT_HOST_BLOCK_COLS", str(block))
    e2 = Engine(0, lib=eng.lib, memory=eng.mem)  # a context that splits every commit into blocks of `block` columns
    try:
        rng = np.random.default_rng(W * 7 + S)
        n, N = 1 << log_n, 1 << (log_n + rb)
        cols = [rand_field(rng, n, noncanonical=True) for _ in range(W)]
        salts = rand_field(rng, max(S, 1), N, noncanonical=True)
        o = ora.commit_salted(np.stack(cols), salts[:S], rb, cap, True) if S else ora.commit(np.stack(cols), rb, cap, True)
        ptrs = (C.c_void_p * W)(*[c.ctypes.data for c in cols])
        sptrs = (C.c_void_p * max(S, 1))(*[salts[j].ctypes.data for j in range(max(S, 1))])
        for chunked in ("1", "0", "grouped", "leaves"):
            # "leaves": the row-major leaf matrix is asked for as well -- the transforms then run first and ONE sponge launch runs
            # beside the matrix's copy (the copy is the long pole); the other variants leave the leaves behind the handle
            monkeypatch.setenv("P2HOT_HOST_CHUNKED_HASH", "0" if chunked == "0" else "1")
            # "grouped": the tail (last chunk + tree levels) per group of cap subtrees, each group's digest slice copied back
            # behind its own levels (the default from 2^18 leaves per group)
            monkeypatch.setenv("P2HOT_HOST_TAIL_MIN_LEAVES", "1" if chunked == "grouped" else str(1 << 40))
            e3 = Engine(0, lib=eng.lib, memory=eng.mem)
            coeffs = np.zeros((W, n), dtype=np.uint64)
            leaves = np.zeros((N, W + S), dtype=np.uint64)
            digests = np.zeros((max(eng.num_digests(log_n + rb, cap), 1), 4), dtype=np.uint64)
            capv = np.zeros((1 << cap, 4), dtype=np.uint64)
            h = C.c_void_p()
            e3.profile(True)
            want_leaves = chunked in ("0", "leaves")
            e3.check(e3.lib.p2hot_commit_salted(e3.ctx, ptrs, W, log_n, rb, cap, 1, 0, sptrs if S else None, S, coeffs.ctypes.data,
                                                leaves.ctypes.data if want_leaves else None, digests.ctypes.data, capv.ctypes.data,
                                                C.byref(h)))
            if not want_leaves:  # the rows from the handle instead
                idx = np.arange(N, dtype=np.uint64)
                e3.check(e3.lib.p2hot_batch_rows(h, idx.ctypes.data, N, leaves.ctypes.data))
            launches = e3.profile_results()["hash_leaves"]["launches"]
            e3.profile(False)
            # chunked: one sponge launch per block that completes at least one new 8-column chunk (+ the salts' launch)
            nd_ = eng.num_digests(log_n + rb, cap)
            groups = 1
            while chunked == "grouped" and nd_ and groups < 8 and groups * 2 <= (1 << cap):
                groups *= 2
            done, want = 0, 0
            for c_end in list(range(block, W, block)) + [W] + ([W + S] if S and groups == 1 else []):
                end = c_end if (c_end >= W + S and groups == 1) else min(c_end, W + S - 1) // 8 * 8
                want, done = want + (end > done), max(done, end)
            want += groups if groups > 1 else 0
            assert launches == (1 if want_leaves else want), (chunked, launches, want)
            assert (coeffs == o["coeffs"] % np.uint64(P)).all() and (leaves == o["leaves"]).all(), chunked
            assert (capv == o["cap"]).all(), chunked
            nd = eng.num_digests(log_n + rb, cap)
            assert nd == 0 or (digests[:nd] == o["digests"]).all(), chunked
            e3.lib.p2hot_batch_free(h)
            e3.close()
    finally:
        e2.close()


def test_host_commit_in_blocks_random_shapes(eng, ora, monkeypatch):
    """the chunked sponge, the grouped tail and the leaves-first order over random widths (around the sponge rate and its
    multiples), block widths, salts, cap heights incl. all-cap trees: every output equals the oracle's"""
    import ctypes as C
    from plonky2_amd.engine import Engine
    rng = np.random.default_rng(20260925)
    monkeypatch.setenv("P2HOT_HOST_TAIL_MIN_LEAVES", "1")
    for trial in range(12 if not is_gpu(eng) else 30):
        W = int(rng.choice([9, 10, 15, 16, 17, 23, 24, 25, 31, 40, 41]))
        block = int(rng.choice([1, 3, 7, 8, 9, 16]))
        S = int(rng.choice([0, 0, 4]))
        log_n, rb = int(rng.integers(2, 6)), int(rng.integers(0, 4))
        cap = int(rng.integers(0, log_n + rb + 1))
        want_leaves, want_dig = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        monkeypatch.setenv("P2HOT_HOST_BLOCK_COLS", str(block))
        e2 = Engine(0, lib=eng.lib, memory=eng.mem)
        try:
            n, N = 1 << log_n, 1 << (log_n + rb)
            cols = [rand_field(rng, n, noncanonical=True) for _ in range(W)]
            salts = rand_field(rng, max(S, 1), N, noncanonical=True)
            o = ora.commit_salted(np.stack(cols), salts[:S], rb, cap, True) if S else ora.commit(np.stack(cols), rb, cap, True)
            ptrs = (C.c_void_p * W)(*[c.ctypes.data for c in cols])
            sptrs = (C.c_void_p * max(S, 1))(*[salts[j].ctypes.data for j in range(max(S, 1))])
            coeffs = np.zeros((W, n), dtype=np.uint64)
            leaves = np.zeros((N, W + S), dtype=np.uint64)
            nd = eng.num_digests(log_n + rb, cap)
            digests = np.zeros((max(nd, 1), 4), dtype=np.uint64)
            capv = np.zeros((1 << cap, 4), dtype=np.uint64)
            h = C.c_void_p()
            e2.check(e2.lib.p2hot_commit_salted(e2.ctx, ptrs, W, log_n, rb, cap, 1, 0, sptrs if S else None, S, coeffs.ctypes.data,
                                                leaves.ctypes.data if want_leaves else None, digests.ctypes.data if want_dig else None,
                                                capv.ctypes.data, C.byref(h)))
            tag = (trial, W, block, S, log_n, rb, cap, want_leaves, want_dig)
            assert (coeffs == o["coeffs"] % np.uint64(P)).all() and (capv == o["cap"]).all(), tag
            if want_leaves:
                assert (leaves == o["leaves"]).all(), tag
            if want_dig and nd:
                assert (digests[:nd] == o["digests"]).all(), tag
            d2 = np.zeros((max(nd, 1), 4), dtype=np.uint64)
            e2.check(e2.lib.p2hot_batch_digests(h, d2.ctypes.data))
            assert nd == 0 or (d2[:nd] == o["digests"]).all(), tag
            e2.lib.p2hot_batch_free(h)
        finally:
            e2.close()


# ---------------------------------------------------------------- Challenger / FRI
def test_challenger_vs_oracle(eng, ora):
    from plonky2_amd.iop.challenger import Challenger
    rng = np.random.default_rng(15)
    c, oc = Challenger(eng), ora.Challenger()
    for kind, k in [("o", 3), ("g", 2), ("o", 8), ("g", 1), ("o", 5), ("o", 11), ("g", 9), ("g", 3), ("o", 64), ("g", 2)]:
        if kind == "o":
            x = rand_field(rng, k, noncanonical=True)
            c.observe_elements(x)
            oc.observe_elements(x)
        else:
            assert c.get_n_challenges(k) == oc.get_n_challenges(k)
    # state round trip (what the Rust shim does with the Challenger's fields)
    st = c.state()
    c2 = Challenger(eng)
    c2.load_state(st)
    assert c2.get_n_challenges(5) == oc.get_n_challenges(5)


FRI_CASES = [(8, 3, 4, [4]), (6, 3, 2, [1, 2]), (9, 1, 4, [4]), (12, 3, 4, [4, 4]), (4, 3, 0, []), (10, 1, 4, [3, 2])]


@pytest.mark.parametrize("log_n,rb,cap,arity", FRI_CASES)
def test_fri_commit_phase_vs_oracle(eng, ora, log_n, rb, cap, arity):
    """fri_committed_trees (fri/prover.rs:84-150) + fri_proof_of_work (:153-202)"""
    from plonky2_amd.fri.prover import fri_committed_trees, fri_proof_of_work
    from plonky2_amd.iop.challenger import Challenger
    rng = np.random.default_rng(log_n * 10 + rb)
    n = 1 << log_n
    co = rand_field(rng, n, 2, noncanonical=True)
    pad = np.zeros((n << rb, 2), dtype=np.uint64)
    pad[:n] = co
    c, oc = Challenger(eng), ora.Challenger()
    pre = rand_field(rng, 5)
    c.observe_elements(pre)
    oc.observe_elements(pre)
    trees, final, betas = fri_committed_trees(co, c, rb, cap, arity, engine=eng)
    o = ora.fri_commit(pad, rb, cap, arity, oc)
    assert (final == o["final"]).all()
    if arity:
        assert (betas == o["betas"]).all()
    for i, t in enumerate(trees):
        assert (t.leaves == o["leaves"][i]).all(), i
        assert (np.asarray(t.digests).reshape(-1, 4) == o["digests"][i]).all(), i
        assert (t.cap.entries == o["caps"][i]).all(), i
    bits = 10 if is_gpu(eng) else 6
    assert fri_proof_of_work(c, bits, engine=eng) == ora.fri_pow(oc, bits)
    assert c.get_n_challenges(3) == oc.get_n_challenges(3)


def test_fri_bad_schedule_is_rejected(eng):
    from plonky2_amd import _lib
    from plonky2_amd.fri.prover import fri_committed_trees
    from plonky2_amd.iop.challenger import Challenger
    co = np.zeros((16, 2), dtype=np.uint64)
    with pytest.raises(_lib.P2HotError):
        fri_committed_trees(co, Challenger(eng), 1, 4, [4, 4], engine=eng)  # second round folds past the degree


def test_commit_random_shapes(eng, ora):
    """seeded sweep over ragged shapes (odd widths, rate_bits 0..4, cap heights up to the all-cap tree, tiny and
    multi-pass sizes): every output of from_values / from_coeffs equals the oracle's"""
    from plonky2_amd.fri.oracle import PolynomialBatch
    rng = np.random.default_rng(2024)
    cases = 40 if is_gpu(eng) else 14
    for _ in range(cases):
        log_n = int(rng.integers(0, 15 if is_gpu(eng) else 11))
        rb = int(rng.integers(0, 5))
        W = int(rng.integers(1, 41))
        cap = int(rng.integers(0, min(log_n + rb, 6) + 1))
        is_values = bool(rng.integers(0, 2))
        cols = rand_field(rng, W, 1 << log_n, noncanonical=True)
        ctor = PolynomialBatch.from_values if is_values else PolynomialBatch.from_coeffs
        b = ctor(cols, rb, False, cap, engine=eng)
        o = ora.commit(cols, rb, cap, is_values)
        tag = (W, log_n, rb, cap, is_values)
        assert (b.polynomials == o["coeffs"] % np.uint64(P)).all(), tag
        assert (b.merkle_tree.cap.entries == o["cap"]).all(), tag
        assert (np.asarray(b.merkle_tree.digests).reshape(-1, 4) == o["digests"]).all(), tag
        assert (b.merkle_tree.leaves.reshape(o["leaves"].shape) == o["leaves"]).all(), tag


def test_fri_transcript_padding_options(eng, ora):
    """final_poly_coeff_len / max_num_query_steps (fri/prover.rs:122-147): the challenger observes zero caps, draws
    dummy challenges and observes zero coefficients exactly like the reference's loop"""
    from plonky2_amd.fri.prover import fri_committed_trees
    from plonky2_amd.iop.challenger import Challenger
    rng = np.random.default_rng(77)
    log_n, rb, cap, arity = 8, 1, 2, [2, 2]
    n = 1 << log_n
    co = rand_field(rng, n, 2)
    pad = np.zeros((n << rb, 2), dtype=np.uint64)
    pad[:n] = co
    c = Challenger(eng)
    trees, final, betas = fri_committed_trees(co, c, rb, cap, arity, engine=eng, final_poly_coeff_len=24, max_num_query_steps=4)
    # reference transcript, step by step, on the oracle challenger
    o = ora.fri_commit(pad, rb, cap, arity, ora.Challenger())
    oc = ora.Challenger()
    for i in range(len(arity)):
        oc.observe_cap(o["caps"][i])
        assert oc.get_extension_challenge() == [int(x) for x in betas[i]]
    for _ in range(len(arity), 4):
        oc.observe_elements(np.zeros(4 << cap, dtype=np.uint64))
        oc.get_extension_challenge()
    oc.observe_elements(o["final"].reshape(-1))
    for _ in range(len(o["final"]), 24):
        oc.observe_elements(np.zeros(2, dtype=np.uint64))
    assert (final == o["final"]).all()
    assert c.get_n_challenges(4) == oc.get_n_challenges(4)
    # the options are consumed: the next commit is unpadded
    c2, oc2 = Challenger(eng), ora.Challenger()
    fri_committed_trees(co, c2, rb, cap, arity, engine=eng)
    ora.fri_commit(pad, rb, cap, arity, oc2)
    assert c2.get_n_challenges(2) == oc2.get_n_challenges(2)


def test_quad_and_lane_poseidon_kernels_agree(eng, ora):
    """p2hot_tune_quad / p2hot_tune_row: the word-per-lane kernels (16 lanes per permutation, DPP row broadcasts), the
    quad-cooperative kernels (4 lanes, DPP quad rotations) and the one-permutation-per-lane kernels produce the same
    trees; all equal the oracle"""
    from plonky2_amd.hash.merkle_tree import MerkleTree
    rng = np.random.default_rng(91)
    gpu = is_gpu(eng)
    cases = [(64, 135, 2), (32, 9, 0), (16, 3, 1), (128, 20, 4)] if not gpu else [(4096, 135, 4), (1024, 20, 0), (256, 3, 2), (8192, 16, 4)]
    try:
        for (n, w, cap) in cases:
            leaves = rand_field(rng, n, w, noncanonical=True)
            digests, capv = ora.merkle_tree(leaves, cap)
            for (quad, row) in ((0, 0), (1 << 20, 0), (0, 1 << 20) if (gpu or n * w <= 64 * 9) else (0, 8)):
                eng.check(eng.lib.p2hot_tune_quad(eng.ctx, quad))
                eng.check(eng.lib.p2hot_tune_row(eng.ctx, row))
                t = MerkleTree.new(leaves, cap, engine=eng)
                assert (t.cap.entries == capv).all(), (n, w, cap, quad, row)
                assert (np.asarray(t.digests).reshape(-1, 4) == digests).all(), (n, w, cap, quad, row)
    finally:
        eng.check(eng.lib.p2hot_tune_quad(eng.ctx, EMU_TUNE_QUAD if not gpu else 1 << 15))
        eng.check(eng.lib.p2hot_tune_row(eng.ctx, EMU_TUNE_ROW if not gpu else 1 << 13))


def test_row_poseidon_kats_and_edges(eng, ora, kats):
    """the word-per-lane permutation (what the challenger and the smallest launches run) on the reference's four KATs
    (poseidon_goldilocks.rs:455-490), boundary-word states and random states, vs the oracle; batch sizes that are not a
    multiple of the four states a wave holds"""
    rng = np.random.default_rng(1234)
    gpu = is_gpu(eng)
    edge = np.array([0, 1, P - 1, P, P + 1, 2**64 - 1, 2**32 - 1, 2**32, 2**63, 0xFFFFFFFF00000000], dtype=np.uint64)
    states = [np.array(k["input"], dtype=np.uint64) for k in kats["poseidon12"]]
    states += [edge[rng.integers(0, len(edge), size=12)] for _ in range(5 if not gpu else 200)]
    states += [rng.integers(0, 2**64 - 1, size=12, dtype=np.uint64) for _ in range(4 if not gpu else 300)]
    st = np.stack(states)
    try:
        eng.check(eng.lib.p2hot_tune_row(eng.ctx, 1 << 20))
        for m in (1, 3, len(st)):
            got = eng.host(eng.poseidon_permute(eng.dev(st[:m].copy())))
            assert (got == np.stack([ora.poseidon(s) for s in st[:m]])).all(), m
        for k, g in zip(kats["poseidon12"], got[:4]):
            assert [int(x) for x in g] == k["output"]
    finally:
        eng.check(eng.lib.p2hot_tune_row(eng.ctx, EMU_TUNE_ROW if not gpu else 1 << 13))


def test_c2_wires_golden_through_the_emulated_kernels(eng):
    """BASELINE config C2 (W = 135, 2^16 rows, rate 1/8, cap 4) at full size in the CPU tier: the kernel sources, compiled for
    the emulator, reproduce the faithful oracle's golden cap and digest-array hash (tests/golden/commit_caps.json).  The
    GPU tier checks the same golden, and the larger ones, in tests/test_gpu_fullsize.py."""
    import hashlib
    import json
    import os
    from plonky2_amd.util.synthetic import splitmix_columns_numpy
    from tests.conftest import ROOT
    if is_gpu(eng):
        pytest.skip("covered by tests/test_gpu_fullsize.py on the GPU")
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "commit_caps.json")))["c2_wires"]
    r = eng.commit(eng.dev(splitmix_columns_numpy(0, g["W"], 1 << g["log_n"])), g["log_n"], g["rate_bits"], g["cap_height"], True)
    assert eng.host(r["cap"]).tolist() == g["cap"]
    assert hashlib.sha256(eng.host(r["digests"]).tobytes()).hexdigest() == g["sha256_digests"]
    co = eng.host(r["coeffs"])
    assert hashlib.sha256(np.where(co >= np.uint64(P), co - np.uint64(P), co).tobytes()).hexdigest() == g["sha256_coeffs"]


def test_host_pointer_commit_random_shapes(eng, ora):
    """p2hot_commit (host pointers, what the Rust shim calls) over random shapes incl. single-row polynomials, rate 1,
    all-cap trees, widths around the sponge rate and the no-hash widths, with and without kept values / leaves /
    digests: every requested output equals the oracle's, and the handle serves rows, paths, digests and coefficients"""
    import ctypes as C
    from plonky2_amd import _lib
    rng = np.random.default_rng(20260924)
    for trial in range(24):
        W = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 33]))
        log_n = int(rng.integers(0, 7))
        rb = int(rng.integers(0, 4))
        cap = int(rng.integers(0, log_n + rb + 1))
        is_values = bool(rng.integers(0, 2))
        keep = bool(rng.integers(0, 2))
        n, N = 1 << log_n, 1 << (log_n + rb)
        cols = rand_field(rng, W, n, noncanonical=True)
        o = ora.commit(cols, rb, cap, is_values)
        ptrs = (C.c_void_p * W)(*[cols[c].ctypes.data for c in range(W)])
        want_leaves, want_dig = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        coeffs = np.zeros((W, n), dtype=np.uint64)
        leaves = np.zeros((N, W), dtype=np.uint64)
        nd = eng.num_digests(log_n + rb, cap)
        digests = np.zeros((max(nd, 1), 4), dtype=np.uint64)
        capv = np.zeros((1 << cap, 4), dtype=np.uint64)
        h = C.c_void_p()
        eng.check(eng.lib.p2hot_commit(eng.ctx, ptrs, W, log_n, rb, cap, 1 if is_values else 0, _lib.KEEP_VALUES if keep else 0,
                                       coeffs.ctypes.data, leaves.ctypes.data if want_leaves else None,
                                       digests.ctypes.data if want_dig else None, capv.ctypes.data, C.byref(h)))
        tag = (trial, W, log_n, rb, cap, is_values)
        assert (coeffs == o["coeffs"] % np.uint64(P)).all(), tag
        assert (capv == o["cap"]).all(), tag
        if want_leaves:
            assert (leaves == o["leaves"]).all(), tag
        if want_dig and nd:
            assert (digests[:nd] == o["digests"]).all(), tag
        idx = rng.integers(0, N, size=5).astype(np.uint64)
        rows = np.zeros((5, W), dtype=np.uint64)
        eng.check(eng.lib.p2hot_batch_rows(h, idx.ctypes.data, 5, rows.ctypes.data))
        assert (rows == o["leaves"][idx.astype(np.int64)]).all(), tag
        layers = log_n + rb - cap
        paths = np.zeros((5, max(layers, 1), 4), dtype=np.uint64)
        eng.check(eng.lib.p2hot_batch_paths(h, idx.ctypes.data, 5, paths.ctypes.data))
        for q, x in enumerate(idx):
            if layers:
                assert (paths[q][:layers] == ora.merkle_prove(int(x), N, cap, o["digests"])).all(), tag
        d2 = np.zeros((max(nd, 1), 4), dtype=np.uint64)
        eng.check(eng.lib.p2hot_batch_digests(h, d2.ctypes.data))
        assert (d2[:nd] == o["digests"]).all(), tag
        c2 = np.zeros((W, n), dtype=np.uint64)
        eng.check(eng.lib.p2hot_batch_coeffs(h, 0, W, c2.ctypes.data))
        assert (c2 == coeffs).all(), tag
        v = C.c_void_p()
        rc = eng.lib.p2hot_batch_values(h, C.byref(v))
        if keep and is_values:
            assert rc == _lib.OK
            back = np.zeros((W, n), dtype=np.uint64)
            eng.check(eng.lib.p2hot_cols_download(v, 0, W, back.ctypes.data))
            assert (back == cols).all(), tag          # the values as uploaded (any representative)
            eng.lib.p2hot_cols_free(v)
        else:
            assert rc == _lib.EINVAL
        eng.lib.p2hot_batch_free(h)


@pytest.mark.parametrize("W,log_n,rb,cap", [(2, 16, 1, 2), (1, 17, 2, 3), (3, 18, 0, 1)])
def test_from_values_reads_the_bit_reversed_inverse_transform(eng, ora, W, log_n, rb, cap):
    """from_values at sizes whose coset LDE is two limb passes (2^16 rows and up): the bit reversal between the inverse transform and
    the LDE (fri/oracle.rs:65-69 then :91-98) has no kernel of its own -- the LDE's first pass reads the bit-reversed array and
    writes `polynomials` in natural order (nttl.hpp load_inputs_bitrev, tile rows 2^4 / 2^5 / 2^6 here; 2^8 at the bench's 2^20) --
    and coefficients, LDE matrix, digests and cap are the oracle's."""
    rng = np.random.default_rng(log_n * 13 + W)
    cols = rand_field(rng, W, 1 << log_n, noncanonical=True)
    o = ora.commit(cols, rb, cap, True)
    eng.profile(True)
    eng.profile_results(reset=True)
    r = eng.commit(eng.dev(cols), log_n, rb, cap, True)
    eng.sync()
    prof = eng.profile_results(reset=True)
    eng.profile(False)
    assert "bitrev_permute" not in prof, "the stand-alone bit reversal ran"
    assert (eng.host(r["coeffs"]) == o["coeffs"]).all(), "natural-order canonical coefficients"
    assert (eng.host(r["lde"]).T == o["leaves"]).all() and (eng.host(r["digests"]) == o["digests"]).all() and (eng.host(r["cap"]) == o["cap"]).all()
