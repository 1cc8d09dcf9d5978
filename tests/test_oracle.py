"""Pins the CPU oracle (oracle/p2oracle.c) before it is trusted as the checker:
reference known-answer vectors + the reference tests' own properties + a pure-Python twin.
CPU only.  Citations are into /root/reference.
"""
import random

import numpy as np
import pytest

from tests import pyref
from tests.pyref import P, G


def rnd_elems(rng, n):
    return [rng.randrange(P) for _ in range(n)]


# ---------------------------------------------------------------- golden vectors
def test_poseidon_kats(ora, kats):
    # plonky2/src/hash/poseidon_goldilocks.rs:455-490 (check_test_vectors poseidon.rs:926-942)
    for kv in kats["poseidon12"]:
        assert list(map(int, ora.poseidon(kv["input"]))) == kv["output"]
        assert list(map(int, ora.poseidon(kv["input"], naive=True))) == kv["output"]
        assert pyref.poseidon_naive(kv["input"]) == kv["output"]


def test_reverse_index_bits_kat(ora, kats):
    # plonky2/src/util/mod.rs:61-85
    assert list(map(int, ora.reverse_index_bits(list(range(256))))) == kats["reverse_index_bits_256"]
    assert list(map(int, ora.reverse_index_bits([10, 20, 30, 40]))) == [10, 30, 20, 40]
    assert list(map(int, ora.reverse_index_bits([10]))) == [10]
    assert list(map(int, ora.reverse_index_bits([10, 20]))) == [10, 20]


def test_constants(ora):
    # goldilocks_field.rs:80, :87, :198; SURVEY section 9 [probed]
    assert ora.root_of_unity(32) == pyref.W32
    assert pow(pyref.W32, 1 << 31, P) == P - 1
    assert pow(G, (P - 1) >> 32, P) == pyref.W32
    assert ora.gl_pow(G, (P - 1) // 2) == P - 1  # generator is a non-residue


# ---------------------------------------------------------------- field
def test_field_edge_grid(ora):
    # field/src/prime_field_testing.rs:8-17 style edge-value grid, incl. non-canonical inputs
    edge = [0, 1, 2, 0xFFFFFFFF, 0x100000000, 0x100000001, P - 2, P - 1, P, P + 1, 2 ** 64 - 1,
            2 ** 63, 0xFFFFFFFE00000001, 0xFFFFFFFF00000000]
    for a in edge:
        for b in edge:
            assert ora.gl_mul(a, b) == (a * b) % P
            assert ora.lib.ora_gl_add(a, b) == (a + b) % P
            assert ora.lib.ora_gl_sub(a, b) == (a - b) % P
    rng = random.Random(1)
    for _ in range(2000):
        a, b = rng.randrange(2 ** 64), rng.randrange(2 ** 64)
        assert ora.gl_mul(a, b) == (a * b) % P
    for a in [1, 2, 7, P - 1, 123456789]:
        assert ora.gl_mul(a, ora.gl_inv(a)) == 1


# ---------------------------------------------------------------- NTT properties
@pytest.mark.parametrize("lg", [0, 1, 2, 3, 5, 8])
def test_fft_matches_naive_eval(ora, lg):
    # field/src/fft.rs:215-249 fft_and_ifft: fft == evaluate_naive; zero-tail option r in 0..4
    rng = random.Random(lg)
    n = 1 << lg
    for deg_frac in (1, 2, 4):
        coeffs = rnd_elems(rng, max(1, n // deg_frac)) + [0] * (n - max(1, n // deg_frac))
        want = pyref.naive_ntt(coeffs)
        assert list(map(int, ora.fft(coeffs))) == want
        zeros_lg = {1: 0, 2: 1, 4: 2}[deg_frac]
        for r in range(0, min(zeros_lg, lg) + 1):
            assert list(map(int, ora.fft(coeffs, r=r))) == want
        assert list(map(int, ora.ifft(want))) == coeffs


@pytest.mark.parametrize("lg", [1, 4, 7])
def test_coset_fft_and_ifft(ora, lg):
    # field/src/polynomial/mod.rs:477-516 test_coset_fft / test_coset_ifft
    rng = random.Random(100 + lg)
    n = 1 << lg
    coeffs = rnd_elems(rng, n)
    shift = rng.randrange(1, P)
    w = pyref.root_of_unity(lg)
    want = [pyref.eval_poly(coeffs, shift * pow(w, i, P) % P) for i in range(n)]
    got = list(map(int, ora.coset_fft(coeffs, shift)))
    assert got == want
    assert list(map(int, ora.coset_ifft(got, shift))) == coeffs


def test_ifft_large_roundtrip(ora):
    rng = np.random.default_rng(5)
    v = rng.integers(0, P, size=1 << 14, dtype=np.uint64)
    c = ora.ifft(v)
    assert np.array_equal(ora.fft(c), v)


# ---------------------------------------------------------------- hashing
def test_sponge_vs_python_twin(ora):
    # hashing.rs:118-145: overwrite mode, no padding, stale tail kept for a short last chunk
    rng = random.Random(7)
    for ln in [0, 1, 4, 5, 7, 8, 9, 15, 16, 17, 20, 32, 135]:
        xs = rnd_elems(rng, ln)
        assert list(map(int, ora.hash_no_pad(xs))) == pyref.hash_no_pad(xs)
        assert list(map(int, ora.hash_or_noop(xs))) == pyref.hash_or_noop(xs)
    l, r = rnd_elems(rng, 4), rnd_elems(rng, 4)
    assert list(map(int, ora.two_to_one(l, r))) == pyref.two_to_one(l, r)
    # hash_or_noop canonicalises (plonk/config.rs:63-74)
    assert list(map(int, ora.hash_or_noop([P + 3, 2 ** 64 - 1]))) == [3, (2 ** 64 - 1) % P, 0, 0]


def test_fast_equals_naive_poseidon(ora):
    # poseidon.rs:944-957 check_consistency
    rng = random.Random(9)
    for _ in range(50):
        s = [rng.randrange(2 ** 64) for _ in range(12)]  # non-canonical inputs allowed
        assert np.array_equal(ora.poseidon(s), ora.poseidon(s, naive=True))
        assert list(map(int, ora.poseidon(s))) == pyref.poseidon_naive(s)


# ---------------------------------------------------------------- Merkle
@pytest.mark.parametrize("log_n,w,cap_height", [(3, 5, 0), (4, 9, 1), (4, 2, 2), (5, 32, 4), (3, 7, 3), (0, 6, 0)])
def test_merkle_layout_and_paths(ora, log_n, w, cap_height):
    # merkle_tree.rs:50-57 layout (closed form) and :253-311 prove/verify every leaf
    rng = random.Random(log_n * 100 + w)
    n = 1 << log_n
    leaves = [rnd_elems(rng, w) for _ in range(n)]
    digests, cap = ora.merkle_tree(leaves, cap_height)
    levels = pyref.merkle_levels(leaves, cap_height)
    assert [list(map(int, c)) for c in cap] == levels[-1]
    sub_leaves = n >> cap_height
    sub_digests = digests.shape[0] >> cap_height
    for lvl, row in enumerate(levels[:-1]):
        per_sub = sub_leaves >> lvl
        for j, d in enumerate(row):
            s, jj = divmod(j, per_sub)
            idx = s * sub_digests + pyref.digest_index(lvl, jj)
            assert list(map(int, digests[idx])) == d
    for i in range(n):
        sib = ora.merkle_prove(i, n, cap_height, digests) if digests.size else np.zeros((0, 4), dtype=np.uint64)
        assert ora.merkle_verify(leaves[i], i, cap, sib)
        if w > 4:
            bad = list(leaves[i])
            bad[0] = (bad[0] + 1) % P
            assert not ora.merkle_verify(bad, i, cap, sib)


# ---------------------------------------------------------------- PolynomialBatch
@pytest.mark.parametrize("W,log_n,rate_bits,cap_height,is_values", [
    (3, 3, 3, 1, True), (5, 4, 1, 2, False), (2, 5, 1, 4, True), (9, 2, 3, 0, True), (1, 0, 3, 0, False)])
def test_commit_semantics(ora, W, log_n, rate_bits, cap_height, is_values):
    # oracle.rs:57-112: leaves[L][c] = p_c(g * w_N^bitrev(L)); coeffs = ifft(values)
    rng = random.Random(W * 1000 + log_n)
    n = 1 << log_n
    cols = [rnd_elems(rng, n) for _ in range(W)]
    out = ora.commit(cols, rate_bits, cap_height, is_values)
    for c in range(W):
        coeffs = list(map(int, out["coeffs"][c]))
        if is_values:
            assert pyref.naive_ntt(coeffs) == cols[c]
        else:
            assert coeffs == cols[c]
        rows = pyref.naive_coset_lde_rows(coeffs, rate_bits)
        assert list(map(int, out["leaves"][:, c])) == rows
    d2, cap2 = ora.merkle_tree(out["leaves"], cap_height)
    assert np.array_equal(d2, out["digests"]) and np.array_equal(cap2, out["cap"])


def test_coset_decomposition_property(ora):
    # SURVEY 8e [probed]: block b of the committed order is the size-n coset NTT with shift
    # g * w_N^bitrev_rb(b), in bitrev_k order -- the identity the HIP LDE kernel is built on.
    rng = random.Random(11)
    log_n, rb = 4, 3
    n, N = 1 << log_n, 1 << (log_n + rb)
    coeffs = rnd_elems(rng, n)
    out = ora.commit([coeffs], rb, 0, is_values=False)["leaves"][:, 0]
    wN = pyref.root_of_unity(log_n + rb)
    for b in range(1 << rb):
        shift = G * pow(wN, pyref.bitrev(b, rb), P) % P
        blk = ora.coset_fft(coeffs, shift)
        blk = ora.reverse_index_bits(blk)
        assert np.array_equal(blk, out[b * n:(b + 1) * n])


# ---------------------------------------------------------------- Challenger
class PyChallenger:
    """plonky2/src/iop/challenger.rs:16-153 restated over the python Poseidon"""

    def __init__(self):
        self.state, self.inb, self.outb = [0] * 12, [], []

    def observe(self, xs):
        for x in xs:
            self.outb = []
            self.inb.append(x % P)
            if len(self.inb) == 8:
                self.duplex()

    def duplex(self):
        self.state[:len(self.inb)] = self.inb
        self.inb = []
        self.state = pyref.poseidon_naive(self.state)
        self.outb = self.state[:8]

    def get(self):
        if self.inb or not self.outb:
            self.duplex()
        return self.outb.pop()


def test_challenger_vs_python_twin(ora):
    rng = random.Random(13)
    a, b = ora.Challenger(), PyChallenger()
    for step in range(40):
        k = rng.randrange(0, 20)
        xs = rnd_elems(rng, k)
        a.observe_elements(xs)
        b.observe(xs)
        for _ in range(rng.randrange(0, 11)):
            assert a.get_challenge() == b.get()


# ---------------------------------------------------------------- FRI
def fri_verifier_chain(pyref, leaves, betas, final, arity_bits, log_N, x_index):
    """fri/verifier.rs:166-238 without the initial-oracle combination: old_eval starts as the
    round-0 committed value; each round interpolates the 2^arity_bits coset at beta
    (verifier.rs:22-47 compute_evaluation) and compares with the next round's committed value."""
    sub_x = G * pow(pyref.root_of_unity(log_N), pyref.bitrev(x_index, log_N), P) % P
    old = None
    for i, ab in enumerate(arity_bits):
        arity = 1 << ab
        row = leaves[i][x_index >> ab]
        evals = [(int(row[2 * t]), int(row[2 * t + 1])) for t in range(arity)]
        within = x_index & (arity - 1)
        if old is not None:
            assert evals[within] == old
        g = pyref.root_of_unity(ab)
        ev = [evals[pyref.bitrev(t, ab)] for t in range(arity)]  # reverse_index_bits_in_place
        start = sub_x * pow(g, arity - pyref.bitrev(within, ab), P) % P
        pts = [start * pow(g, t, P) % P for t in range(arity)]
        beta = (int(betas[i][0]), int(betas[i][1]))
        acc = (0, 0)  # Lagrange interpolation at beta
        for a in range(arity):
            num, den = (1, 0), 1
            for b in range(arity):
                if a != b:
                    num = pyref.ext_mul(num, ((beta[0] - pts[b]) % P, beta[1]))
                    den = den * (pts[a] - pts[b]) % P
            di = pow(den, P - 2, P)
            term = pyref.ext_mul(ev[a], (num[0] * di % P, num[1] * di % P))
            acc = pyref.ext_add(acc, term)
        old = acc
        sub_x = pow(sub_x, arity, P)
        x_index >>= ab
    assert pyref.ext_eval(final, (sub_x, 0)) == old


@pytest.mark.parametrize("log_n,rate_bits,arity_bits,cap_height", [(6, 3, [4], 2), (8, 1, [4, 1], 1), (7, 3, [2, 3], 0)])
def test_fri_commit_consistent_with_verifier(ora, log_n, rate_bits, arity_bits, cap_height):
    rng = random.Random(17 + log_n)
    n, N = 1 << log_n, 1 << (log_n + rate_bits)
    coeffs = np.zeros((N, 2), dtype=np.uint64)
    coeffs[:n] = np.array([[rng.randrange(P), rng.randrange(P)] for _ in range(n)], dtype=np.uint64)
    ch = ora.Challenger()
    ch.observe_elements([1, 2, 3])
    ch2 = ch.clone()
    out = ora.fri_commit(coeffs, rate_bits, cap_height, arity_bits, ch)
    # round-0 leaves are the bit-reversed coset evaluations of the input polynomial
    lg = log_n + rate_bits
    wN = pyref.root_of_unity(lg)
    flat0 = out["leaves"][0].reshape(-1, 2)
    for L in [0, 1, 5, N - 1]:
        x = G * pow(wN, pyref.bitrev(L, lg), P) % P
        assert pyref.ext_eval(coeffs[:n], (x, 0)) == (int(flat0[L][0]), int(flat0[L][1]))
    # trees, caps, transcript
    for i in range(len(arity_bits)):
        d, cap = ora.merkle_tree(out["leaves"][i], cap_height)
        assert np.array_equal(cap, out["caps"][i]) and np.array_equal(d, out["digests"][i])
        ch2.observe_cap(cap)
        assert ch2.get_extension_challenge() == list(map(int, out["betas"][i]))
    ch2.observe_elements(out["final"])
    assert ch2.get_challenge() == ch.get_challenge()
    # verifier chain on a few query indices
    for x_index in [0, 3, N // 2 + 1, N - 1]:
        fri_verifier_chain(pyref, out["leaves"], out["betas"], out["final"], arity_bits, lg, x_index)


def test_fri_pow(ora):
    ch = ora.Challenger()
    ch.observe_elements([5, 6, 7])
    ref = ch.clone()
    w = ora.fri_pow(ch, 8)
    # smallest witness, and valid per fri/verifier.rs:49-60
    for cand in range(w + 1):
        c = ref.clone()
        c.observe_elements([cand])
        resp = c.get_challenge()
        ok = resp < (1 << (64 - 8))
        assert ok == (cand == w)
    c = ref.clone()
    c.observe_elements([w])
    c.get_challenge()
    assert c.get_challenge() == ch.get_challenge()


def test_reduce_and_divide(ora):
    # util/reducing.rs:83-95, field/src/polynomial/division.rs:79-92
    rng = random.Random(23)
    n, k = 16, 5
    polys = [rnd_elems(rng, n) for _ in range(k)]
    alpha = (rng.randrange(P), rng.randrange(P))
    red = ora.reduce_polys_base(polys, alpha)
    for i in range(n):
        acc, pw = (0, 0), (1, 0)
        for j in range(k):
            acc = pyref.ext_add(acc, (pw[0] * polys[j][i] % P, pw[1] * polys[j][i] % P))
            pw = pyref.ext_mul(pw, alpha)
        assert acc == (int(red[i][0]), int(red[i][1]))
    z = (rng.randrange(P), rng.randrange(P))
    q = ora.divide_by_linear(red, z)
    assert (int(q[n - 1][0]), int(q[n - 1][1])) == (0, 0)
    # (p(X) - p(z)) == q(X) (X - z) at a random point
    x = (rng.randrange(P), rng.randrange(P))
    pz, px = pyref.ext_eval(red, z), pyref.ext_eval(red, x)
    lhs = ((px[0] - pz[0]) % P, (px[1] - pz[1]) % P)
    rhs = pyref.ext_mul(pyref.ext_eval(q, x), ((x[0] - z[0]) % P, (x[1] - z[1]) % P))
    assert lhs == rhs


def test_eval_polys_ext_vs_python_twin_and_the_ntt(ora):
    """OpeningSet::new's eval_commitment (plonk/proof.rs:314-327; PolynomialCoeffs::eval, field/src/polynomial/mod.rs:155-160):
    against Horner in Python integers, and -- the reference's own property for `eval` (fft.rs:230-239: the NTT is evaluation on
    the subgroup) -- at a base-field point w^i the value is entry i of the forward transform, with a zero second coordinate"""
    rng = random.Random(29)
    for lg in (0, 1, 4, 7):
        n = 1 << lg
        polys = [rnd_elems(rng, n) for _ in range(3)]
        z = (rng.randrange(P), rng.randrange(P))
        got = ora.eval_polys_ext(polys, z)
        for j in range(3):
            assert pyref.ext_eval([(c, 0) for c in polys[j]], z) == (int(got[j][0]), int(got[j][1]))
        w = ora.root_of_unity(lg)
        i = rng.randrange(n)
        at = ora.eval_polys_ext(polys, (pow(w, i, P), 0))
        for j in range(3):
            assert (int(at[j][0]), int(at[j][1])) == (int(ora.fft(polys[j])[i]) % P, 0)


# ---------------------------------------------------------------- the reference-run pin
def test_reference_run_comparer_on_the_oracles_own_dump(ora, tmp_path):
    """tools/reference_run.py: the comparison a `reference_run.json` from the Rust dumper goes through, exercised on a dump
    in the same schema written by the oracle itself (must pass) and on tampered copies (must name the field)"""
    import copy
    import json
    from tools import reference_run as rr
    p = tmp_path / "run.json"
    run = rr.emit(str(p), ["small_values", "small_coeffs", "small_fibonacci", "fri_small", "fri_starky_like"])
    done = dict(rr.check(str(p)))
    assert set(done) == {"small_values", "small_coeffs", "small_fibonacci", "fri_small", "fri_starky_like"}
    for name, field, mutate in (("small_values", "cap", lambda v: [[v[0][0] ^ 1] + v[0][1:]] + v[1:]),
                                ("small_coeffs", "sha256_lde", lambda v: v[:-1] + ("0" if v[-1] != "0" else "1")),
                                ("fri_small", "final_poly", lambda v: [[(v[0][0] + 1) % P, v[0][1]]] + v[1:]),
                                ("fri_starky_like", "pow_witness", lambda v: v + 1)):
        bad = copy.deepcopy(run)
        bad[name][field] = mutate(bad[name][field])
        q = tmp_path / "bad.json"
        json.dump(bad, open(q, "w"))
        with pytest.raises(AssertionError, match=name):
            rr.check(str(q))
    # a record of a golden shape is compared with tests/golden/commit_caps.json without recomputation
    g = json.load(open(rr.GOLDEN))
    json.dump({"c3_wires": {k: g["c3_wires"][k] for k in rr.COMMIT_FIELDS}}, open(p, "w"))
    assert rr.check(str(p)) == [("c3_wires", "tests/golden/commit_caps.json")]


def test_reference_run_pins_the_oracle_when_present(ora):
    """`tests/golden/reference_run.json` = the output of `cargo run --release --example p2hot_dump_goldens` on a box with the
    Rust toolchain (integration/plonky2_p2hot.patch).  When it is committed, the oracle -- and through
    tests/golden/commit_caps.json every GPU result -- is pinned to the real reference's bytes."""
    import os
    from tools import reference_run as rr
    # P2_REFERENCE_RUN: integration/first_contact.sh checks the dumper's file where it was written and copies it here only once it passed
    path = os.environ.get("P2_REFERENCE_RUN") or os.path.join(os.path.dirname(__file__), "golden", "reference_run.json")
    if not os.path.exists(path):
        if os.environ.get("P2_REFERENCE_RUN"):
            pytest.fail("P2_REFERENCE_RUN=%s does not exist" % path)
        pytest.skip("no reference run has been recorded yet (no cargo in the build image): parity stays pinned by KAT + property")
    done = rr.check(path)
    assert done, "empty reference run"
