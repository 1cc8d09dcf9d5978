"""the prove_openings prelude (reduce_polys_base, divide_by_linear, shift/accumulate),
the FRI commit from device-resident coefficients, proof of work and the query phase, against an
oracle-side composition that follows fri/oracle.rs:176-237 and fri/prover.rs:24-258 step by step."""
import numpy as np
import pytest

from tests.conftest import P, rand_field


def is_gpu(eng):
    return not bool(eng.lib.p2hot_is_emulated())


def _ext_mul(a, b):
    return [(a[0] * b[0] + 7 * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P]


def _ext_pow(a, e):
    r = [1, 0]
    for _ in range(e):
        r = _ext_mul(r, a)
    return r


def _oracle_final_poly(ora, batches, coeff_sets, alpha):
    n = coeff_sets[0].shape[1]
    final = np.zeros((n, 2), dtype=object)
    for bi, (point, polys) in enumerate(batches):
        ps = np.stack([coeff_sets[o][p] for (o, p) in polys])
        comp = ora.reduce_polys_base(ps, alpha)                 # util/reducing.rs:83-95
        quo = ora.divide_by_linear(comp, point)                 # division.rs:79-92 (+ zero pad)
        sh = _ext_pow([int(alpha[0]), int(alpha[1])], len(polys))  # shift_poly, reducing.rs:103-106
        for k in range(n):
            f = _ext_mul([int(final[k][0]), int(final[k][1])], sh) if bi else [0, 0]
            final[k][0] = (f[0] + int(quo[k][0])) % P
            final[k][1] = (f[1] + int(quo[k][1])) % P
    return np.array(final.tolist(), dtype=np.uint64)


# nxt = (oracle, count): the second opening batch.  The last case is the reference's own FRI instance in miniature
# (plonk/circuit_data.rs:530-548, :578-664): four oracles -- constants_sigmas, wires, Zs + partial products, quotient --
# every polynomial at zeta, and only the num_challenges = 2 Z polynomials (a SUB-RANGE of oracle 2) at g * zeta
@pytest.mark.parametrize("log_n,widths,rb,cap,arity,nxt", [(5, [3, 2], 3, 2, [2], None), (8, [7, 4, 2], 3, 4, [4], None), (0, [2], 1, 0, [], None),
                                                            (7, [2, 2], 1, 4, [1, 2], None), (6, [6, 9, 5, 4], 3, 4, [4], (2, 2))])
def test_final_poly_and_prove_openings_vs_oracle(eng, ora, log_n, widths, rb, cap, arity, nxt):
    from plonky2_amd.fri.oracle import FriBatchInfo, PolynomialBatch, final_poly_device, prove_openings
    from plonky2_amd.iop.challenger import Challenger
    rng = np.random.default_rng(log_n * 31 + len(widths))
    n = 1 << log_n
    cols = [rand_field(rng, w, n) for w in widths]
    # host arrays: p2hot_commit with host pointers only, exactly what the Rust shim calls (the batch lives behind a handle)
    oracles = [PolynomialBatch.from_coeffs(c, rb, False, cap, engine=eng) for c in cols]
    # device buffers: the *_dev flow wrapped into handles (p2hot_batch_wrap_dev)
    oracles_dev = [PolynomialBatch.from_coeffs(eng.dev(c), rb, False, cap, engine=eng) for c in cols]
    # two opening batches like plonky2's zeta / g*zeta (plonk_common.rs FRI_ORACLES): all polys at z0, the first oracle at z1
    all_polys = [(oi, pi) for oi, w in enumerate(widths) for pi in range(w)]
    first = [(nxt[0], pi) for pi in range(nxt[1])] if nxt else [(0, pi) for pi in range(widths[0])]
    z0, z1 = rand_field(rng, 2), rand_field(rng, 2)
    batches = [FriBatchInfo(z0, all_polys), FriBatchInfo(z1, first)]
    ob = [(z0, all_polys), (z1, first)]

    alpha = rand_field(rng, 2)
    got = eng.host(final_poly_device(batches, oracles_dev, alpha, eng))  # planes [2][n]
    exp = _oracle_final_poly(ora, ob, cols, alpha)
    assert (got.T == exp).all()

    # the whole prove_openings transcript
    c, oc = Challenger(eng), ora.Challenger()
    pre = rand_field(rng, 6)
    c.observe_elements(pre)
    oc.observe_elements(pre)
    pow_bits, n_queries = 5, 3
    c_dev = Challenger(eng)
    c_dev.observe_elements(pre)
    proof = prove_openings(batches, oracles, c, rb, cap, arity, pow_bits, n_queries, engine=eng)
    proof_dev = prove_openings(batches, oracles_dev, c_dev, rb, cap, arity, pow_bits, n_queries, engine=eng)
    assert proof["pow_witness"] == proof_dev["pow_witness"] and proof["query_indices"] == proof_dev["query_indices"]
    assert (proof["final_poly"] == proof_dev["final_poly"]).all()
    for qa, qb in zip(proof["query_round_proofs"], proof_dev["query_round_proofs"]):
        for (la, sa), (lb, sb) in zip(qa["initial_trees_proof"] + qa["steps"], qb["initial_trees_proof"] + qb["steps"]):
            assert (la == lb).all() and (sa == sb).all()
    a = oc.get_extension_challenge()
    fin = _oracle_final_poly(ora, ob, cols, np.array(a, dtype=np.uint64))
    pad = np.zeros((n << rb, 2), dtype=np.uint64)
    pad[:n] = fin
    o = ora.fri_commit(pad, rb, cap, arity, oc)
    assert (proof["final_poly"] == o["final"]).all()
    for i in range(len(arity)):
        assert (proof["commit_phase_merkle_caps"][i] == o["caps"][i]).all()
    assert proof["pow_witness"] == ora.fri_pow(oc, pow_bits)
    N = n << rb
    commits = [ora.commit(cc, rb, cap, False) for cc in cols]
    for q, rand in zip(proof["query_round_proofs"], oc.get_n_challenges(n_queries)):
        x = rand % N
        for (leaf, sib), cm in zip(q["initial_trees_proof"], commits):
            assert (leaf == cm["leaves"][x]).all()
            assert (sib == ora.merkle_prove(x, N, cap, cm["digests"])).all()
            assert ora.merkle_verify(leaf, x, cm["cap"], sib)
        for i, ab in enumerate(arity):
            evals, sib = q["steps"][i]
            nl = o["leaves"][i].shape[0]
            assert (evals.reshape(-1) == o["leaves"][i][x >> ab]).all()
            assert (sib == ora.merkle_prove(x >> ab, nl, cap, o["digests"][i])).all()
            x >>= ab
    assert c.get_n_challenges(2) == oc.get_n_challenges(2)


def test_final_poly_long_polynomials_two_level_carries(eng, ora):
    """divide_by_linear (division.rs:79-92) on polynomials long enough for the two-level carry scan (more than 4096 chunks:
    2^19 coefficients and up; the starky trace of C4 has 2^22): final_poly against the oracle's reduce_polys_base +
    divide_by_linear.  The CPU tier lowers the switch-over (P2HOT_HORNER_2L_MIN) on a context of its own to reach the same code."""
    import os
    from plonky2_amd.fri.oracle import FriBatchInfo, PolynomialBatch, final_poly_device
    rng = np.random.default_rng(19)
    if is_gpu(eng):
        e, sizes = eng, (19, 20)
    else:
        from tests.emu_backend import emu_engine
        old = os.environ.get("P2HOT_HORNER_2L_MIN")
        os.environ["P2HOT_HORNER_2L_MIN"] = "16"
        try:
            e = emu_engine()
        finally:
            if old is None:
                os.environ.pop("P2HOT_HORNER_2L_MIN")
            else:
                os.environ["P2HOT_HORNER_2L_MIN"] = old
        sizes = (9, 12, 14)      # 128 / 1024 / 1024 chunks -> 2 / 16 / 16 groups
    for log_n in sizes:
        widths = [2, 1]
        cols = [rand_field(rng, w, 1 << log_n) for w in widths]
        oracles = [PolynomialBatch.from_coeffs(e.dev(c), 1, False, 0, engine=e) for c in cols]
        all_polys = [(oi, pi) for oi, w in enumerate(widths) for pi in range(w)]
        z0, z1, alpha = rand_field(rng, 2), rand_field(rng, 2), rand_field(rng, 2)
        batches = [FriBatchInfo(z0, all_polys), FriBatchInfo(z1, [(0, 0), (0, 1)])]
        got = e.host(final_poly_device(batches, oracles, alpha, e))
        exp = _oracle_final_poly(ora, [(z0, all_polys), (z1, [(0, 0), (0, 1)])], cols, alpha)
        assert (got.T == exp).all(), log_n


def test_merkle_paths_device(eng, ora):
    """p2hot_merkle_paths_dev == merkle_tree_prove (hash/merkle_tree.rs:151-190)"""
    rng = np.random.default_rng(8)
    for (log_n, cap) in ((6, 0), (6, 3), (4, 4), (9, 4)):
        n = 1 << log_n
        leaves = rand_field(rng, n, 5)
        digests, capv = eng.merkle(eng.dev(leaves), 1, 5, log_n, cap)
        idx = rng.integers(0, n, size=7).astype(np.uint64)
        layers = log_n - cap
        out = eng.mem.zeros(7, max(layers, 1), 4)
        d_idx = eng.dev(idx)
        eng.check(eng.lib.p2hot_merkle_paths_dev(eng.ctx, eng.ptr(digests), log_n, cap, eng.ptr(d_idx), 7, eng.ptr(out)))
        got = eng.host(out)
        od, _ = ora.merkle_tree(leaves, cap)
        for q, i in enumerate(idx):
            assert (got[q][:layers] == ora.merkle_prove(int(i), n, cap, od)).all()


def test_eval_openings_vs_horner(eng):
    """OpeningSet::new (plonk/proof.rs:314-327): p(zeta) in F^2 by Horner on the host vs the device reduction"""
    from plonky2_amd.fri.oracle import PolynomialBatch, eval_openings
    rng = np.random.default_rng(33)
    for log_n in (0, 3, 8, 10, 13):  # 13: two 4096-coefficient segments per polynomial (stage 2 combines them)
        n = 1 << log_n
        co = rand_field(rng, 5 if log_n < 13 else 2, n)
        b = PolynomialBatch.from_coeffs(co, 1, False, 0, engine=eng)
        pts = rand_field(rng, 2, 2)
        got = eval_openings([b], pts, eng)[0]
        for pi, z in enumerate(pts):
            zz = [int(z[0]), int(z[1])]
            for j in range(co.shape[0]):
                acc = [0, 0]
                for c in reversed(co[j]):
                    acc = _ext_mul(acc, zz)
                    acc[0] = (acc[0] + int(c)) % P
                assert [int(got[pi][j][0]), int(got[pi][j][1])] == acc, (log_n, pi, j)


def test_remaining_device_and_accessor_entry_points(eng, ora):
    """Entry points nothing else calls directly: p2hot_eval_polys_dev (the device-pointer building block of OpeningSet::new,
    plonk/proof.rs:314-327) against a Horner evaluation in F^2; the batch accessors; p2hot_tune_overlap (leaf sponge of coset
    block b beside the LDE of block b+1 on a second stream -- identical commitment)."""
    from plonky2_amd.fri.oracle import PolynomialBatch
    from tests import pyref
    rng = np.random.default_rng(31)
    log_n, n_polys = 6, 5
    polys = rand_field(rng, n_polys, 1 << log_n, noncanonical=True)
    dev = eng.dev(polys)
    table = eng.dev(np.array([eng.mem.ptr(dev) + j * (8 << log_n) for j in range(n_polys)], dtype=np.uint64))
    pts = np.array([[3, 5], [P - 1, 7], [2**63, 0]], dtype=np.uint64)
    out = eng.mem.zeros(len(pts) * n_polys, 2)
    eng.check(eng.lib.p2hot_eval_polys_dev(eng.ctx, eng.ptr(table), n_polys, log_n, pts.ctypes.data, len(pts), eng.ptr(out)))
    got = eng.host(out).reshape(len(pts), n_polys, 2)
    for pi, pt in enumerate(pts):
        x = (int(pt[0]) % P, int(pt[1]) % P)
        for j in range(n_polys):
            want = pyref.ext_eval([(int(c) % P, 0) for c in polys[j]], x)
            assert tuple(int(v) for v in got[pi, j]) == tuple(want), (pi, j)
    # accessors of a commitment handle
    cols = rand_field(rng, 3, 1 << 4)
    b = PolynomialBatch.from_values(cols, 2, False, 1, engine=eng)
    assert eng.lib.p2hot_batch_width(b._owner.h) == 3 and eng.lib.p2hot_batch_degree_log(b._owner.h) == 4
    # the two-stream overlap knob changes nothing in the result
    o = ora.commit(cols, 2, 1, True)
    try:
        eng.check(eng.lib.p2hot_tune_overlap(eng.ctx, 1))
        r = eng.commit(eng.dev(cols), 4, 2, 1, True)
        assert (eng.host(r["cap"]) == o["cap"]).all() and (eng.host(r["digests"]) == o["digests"]).all()
    finally:
        eng.check(eng.lib.p2hot_tune_overlap(eng.ctx, 0))
    assert eng.lib.p2hot_tune_overlap(None, 1) != 0


def test_polynomial_batch_wire_format(eng):
    """write_polynomial_batch / write_merkle_tree byte layout (util/serialization/mod.rs:1417-1431, :1744-1763)"""
    import struct
    from plonky2_amd.fri.oracle import PolynomialBatch
    from tests.wire_format import read_polynomial_batch, write_polynomial_batch
    rng = np.random.default_rng(44)
    co = rand_field(rng, 3, 4, noncanonical=True)
    b = PolynomialBatch.from_coeffs(co, 1, False, 1, engine=eng)
    blob = write_polynomial_batch(b)
    # hand-assembled expectation, field by field
    exp = struct.pack("<Q", 3)
    for p in b.polynomials:
        exp += struct.pack("<Q", 4) + struct.pack("<4Q", *[int(x) for x in p])
    leaves = b.merkle_tree.leaves
    exp += struct.pack("<Q", 8)
    for row in leaves:
        exp += struct.pack("<Q", 3) + struct.pack("<3Q", *[int(x) for x in row])
    dig = np.asarray(b.merkle_tree.digests).reshape(-1, 4)
    exp += struct.pack("<Q", len(dig)) + b"".join(struct.pack("<4Q", *[int(x) for x in d]) for d in dig)
    exp += struct.pack("<Q", 1) + b"".join(struct.pack("<4Q", *[int(x) for x in c]) for c in b.merkle_tree.cap.entries)
    exp += struct.pack("<QQ", 2, 1) + b"\x00"
    assert blob == exp
    r = read_polynomial_batch(blob)
    assert (r["polynomials"] == b.polynomials).all() and (r["merkle_tree"].leaves == leaves).all()
    assert (r["merkle_tree"].digests == dig).all() and (r["merkle_tree"].cap.entries == b.merkle_tree.cap.entries).all()
    assert (r["degree_log"], r["rate_bits"], r["blinding"]) == (2, 1, False)
    assert (np.frombuffer(blob[8 + 8:8 + 8 + 32], dtype="<u8") < np.uint64(P)).all()  # canonical on the wire


@pytest.mark.parametrize("log_n,widths,rb,cap,arity,pow_bits,nq", [
    (6, [3, 2], 3, 2, [2, 1], 4, 5),      # mixed arities
    (8, [5, 4, 2], 3, 4, [4], 6, 4),      # plonky2-like: arity 16, cap 4
    (7, [2, 2], 1, 3, [1, 2, 1], 3, 6),   # starky-like rate 1/2
    (4, [3], 2, 0, [], 2, 3),             # no reduction rounds: the final polynomial is the whole codeword's polynomial
])
def test_fri_proof_passes_the_reference_verifier(eng, ora, log_n, widths, rb, cap, arity, pow_bits, nq):
    _fri_proof_verifies(eng, ora, log_n, widths, rb, cap, arity, pow_bits, nq, blinded=())


def test_fri_proof_of_the_plonky2_instance_shape_passes_the_reference_verifier(eng, ora):
    """get_fri_instance (plonk/circuit_data.rs:530-548): 4 oracles (constants_sigmas, wires, Zs + partial products, quotient), all
    polynomials at zeta, the 2 Z polynomials -- polynomials 0..1 of oracle 2 -- at g * zeta; four initial trees per query"""
    _fri_proof_verifies(eng, ora, 7, [8, 13, 6, 4], 3, 4, [4], 5, 4, blinded=(), nxt=(2, 2))


def test_fri_proof_over_blinded_oracles_passes_the_reference_verifier(eng, ora):
    """zero-knowledge configs (standard_recursion_zk_config): oracles 1 and 2 are salted (PlonkOracle::WIRES / ZS blinding,
    plonk/plonk_common.rs), oracle 0 is not (CONSTANTS_SIGMAS).  The opened leaves carry the salts, the Merkle paths
    authenticate the salted leaves, the verifier reads the polynomial evaluations in front of them (fri/proof.rs:45-52)"""
    _fri_proof_verifies(eng, ora, 6, [4, 5, 3], 3, 2, [2, 1], 4, 5, blinded=(1, 2))


def _fri_proof_verifies(eng, ora, log_n, widths, rb, cap, arity, pow_bits, nq, blinded, nxt=None):
    """The acceptance check of SURVEY 3.5: a proof produced on the device (commits, OpeningSet evaluations,
    prove_openings) verifies under a restatement of plonky2/src/fri/verifier.rs + challenges.rs that re-derives every
    challenge from the proof with the ORACLE's challenger; tampering with any part makes it fail."""
    import copy
    from oracle import fri_verifier as fv
    from plonky2_amd.fri.oracle import FriBatchInfo, PolynomialBatch, eval_openings, prove_openings
    from plonky2_amd.iop.challenger import Challenger
    rng = np.random.default_rng(log_n * 977 + len(widths))
    n = 1 << log_n
    oracles = [PolynomialBatch.from_coeffs(rand_field(rng, w, n), rb, oi in blinded, cap, engine=eng,
                                           salts=rand_field(rng, 4, n << rb) if oi in blinded else None)
               for oi, w in enumerate(widths)]
    leaf_widths = [w + (4 if oi in blinded else 0) for oi, w in enumerate(widths)]
    caps = [np.asarray(o.merkle_tree.cap.entries) for o in oracles]
    c, oc = Challenger(eng), ora.Challenger()
    for ch in (c, oc):
        for k in caps:
            ch.observe_cap(k)
    zeta = c.get_extension_challenge()
    assert list(zeta) == list(oc.get_extension_challenge())
    g = ora.root_of_unity(log_n)
    gz = [int(zeta[0]) * g % P, int(zeta[1]) * g % P]                      # zeta_next = g * zeta (plonk/prover.rs:338)
    all_polys = [(oi, pi) for oi, w in enumerate(widths) for pi in range(w)]
    first = [(nxt[0], pi) for pi in range(nxt[1])] if nxt else [(0, pi) for pi in range(widths[0])]
    inst = [(zeta, all_polys), (gz, first)]                                # FriInstanceInfo.batches
    ev = eval_openings(oracles, [zeta, gz], eng)                           # [oracle][point][poly][2]
    openings = [[ev[oi][bi][pi] for (oi, pi) in polys] for bi, (_, polys) in enumerate(inst)]   # FriOpenings
    for ch in (c, oc):                                                     # Challenger::observe_openings
        for vals in openings:
            ch.observe_elements(np.asarray(vals, dtype=np.uint64).reshape(-1))
    proof = prove_openings([FriBatchInfo(p, polys) for p, polys in inst], oracles, c, rb, cap, arity, pow_bits, nq, engine=eng)

    def verify(pf, op=openings, caps_=caps):
        vc = oc.clone()
        chal = fv.fri_challenges(vc, pf["commit_phase_merkle_caps"], pf["final_poly"], pf["pow_witness"], log_n, rb, cap, nq)
        fv.verify_fri_proof(inst, op, chal, caps_, pf, log_n, rb, arity, pow_bits, nq)
        return vc

    vc = verify(proof)
    assert vc.get_n_challenges(2) == c.get_n_challenges(2)                 # prover and verifier transcripts agree to the end

    # wire format (serialization/mod.rs:1595-1611): length as the reference lays it out, round trip, re-verification
    from tests.wire_format import read_fri_proof, write_fri_proof
    blob = write_fri_proof(proof)
    ncap, N = 1 << cap, n << rb
    expect, m = 32 * ncap * len(arity), N
    per_query = sum(8 * w + 1 + 32 * (log_n + rb - cap) for w in leaf_widths)
    for ab in arity:
        m >>= ab
        per_query += 16 * (1 << ab) + 1 + 32 * max(0, m.bit_length() - 1 - cap)
    expect += nq * per_query + 16 * (n >> sum(arity)) + 8
    assert len(blob) == expect
    back = read_fri_proof(blob, leaf_widths, log_n, rb, cap, arity, nq)
    assert write_fri_proof(back) == blob
    verify(back)

    bad = copy.deepcopy(proof)
    bad["final_poly"] = np.array(bad["final_poly"], dtype=np.uint64)
    bad["final_poly"][0][0] = (int(bad["final_poly"][0][0]) + 1) % P
    with pytest.raises(fv.VerificationError):
        verify(bad)
    bad_open = [[np.array(v, dtype=np.uint64) for v in vals] for vals in openings]
    bad_open[0][0][1] = (int(bad_open[0][0][1]) + 1) % P
    with pytest.raises(fv.VerificationError):
        verify(proof, op=bad_open)
    bad = copy.deepcopy(proof)
    leaf, sib = bad["query_round_proofs"][0]["initial_trees_proof"][0]
    leaf = np.array(leaf, dtype=np.uint64)
    leaf[0] = (int(leaf[0]) + 1) % P
    bad["query_round_proofs"][0]["initial_trees_proof"][0] = (leaf, sib)
    with pytest.raises(fv.VerificationError):
        verify(bad)
    if arity:
        bad = copy.deepcopy(proof)
        evals, sib = bad["query_round_proofs"][-1]["steps"][0]
        evals = np.array(evals, dtype=np.uint64)
        evals[0][0] = (int(evals[0][0]) + 1) % P
        bad["query_round_proofs"][-1]["steps"][0] = (evals, sib)
        with pytest.raises(fv.VerificationError):
            verify(bad)
    # another PoW witness: almost surely not a valid one, and it shifts the query indices.  "Almost": with pow_bits = 0 every witness
    # is valid, and on a 2^8-point domain two queries land on the same indices once in 65 536 (a randomized GPU session met it:
    # profiles/r06_fuzz.txt) -- then the tampered proof IS a valid proof, so the next witness is taken
    for step in range(1, 6):
        bad = copy.deepcopy(proof)
        bad["pow_witness"] = int(bad["pow_witness"]) + step
        chal = fv.fri_challenges(oc.clone(), bad["commit_phase_merkle_caps"], bad["final_poly"], bad["pow_witness"], log_n, rb, cap, nq)
        if pow_bits == 0 and chal["fri_query_indices"] == [int(x) for x in proof["query_indices"]]:
            continue
        with pytest.raises(fv.VerificationError):
            verify(bad)
        break
    else:
        raise AssertionError("five consecutive PoW witnesses reproduce the query indices")


def test_host_session_handles_and_errors(eng, ora):
    """The host-pointer session objects: kept values (P2HOT_KEEP_VALUES) feed the permutation argument, a committed
    column set is consumed, handles of another context / bad indices / hiding are refused with the reference's wording"""
    import ctypes as C
    from plonky2_amd import _lib
    from plonky2_amd.fri.oracle import DeviceColumns, FriBatchInfo, PolynomialBatch, eval_openings, prove_openings
    from plonky2_amd.iop.challenger import Challenger
    from plonky2_amd.plonk.prover import all_wires_permutation_partial_products
    rng = np.random.default_rng(1)
    n, W = 32, 6
    vals = rand_field(rng, W, n, noncanonical=True)
    b = PolynomialBatch.from_values(vals, 2, False, 1, engine=eng, keep_values=True)
    o = ora.commit(vals, 2, 1, True)
    assert (b.polynomials == o["coeffs"]).all() and (b.merkle_tree.cap.entries == o["cap"]).all()
    assert (b.merkle_tree.digests == o["digests"]).all() and (b.merkle_tree.leaves == o["leaves"]).all()
    assert (b.get_lde_values(3, 2) == o["leaves"][int(format(6, "07b")[::-1], 2)]).all()       # oracle.rs:142-147
    # the kept values are the inputs, and they drive p2hot_partial_products without another upload
    v = b.values()
    assert (v.host() == vals % np.uint64(P)).all() or (v.host() == vals).all()
    sig = DeviceColumns.upload(rand_field(rng, W, n), eng)
    k = np.array([pow(7, j, P) for j in range(W)], dtype=np.uint64)
    zs = all_wires_permutation_partial_products(v, sig, k, 2, [3], [5], eng)
    exp = ora.partial_products(vals, sig.host(), k, 2, 3, 5)
    got = zs.host()
    assert (got[0] == exp[-1]).all() and (got[1:] == exp[:-1]).all()
    zb = PolynomialBatch.from_values(zs, 2, False, 1, engine=eng)          # p2hot_commit_cols consumes the column set
    assert zs._h is None and (zb.merkle_tree.cap.entries == ora.commit(got, 2, 1, True)["cap"]).all()
    with pytest.raises(_lib.P2HotError, match="KEEP_VALUES"):
        zb.values()
    # argument validation of p2hot_prove_openings / p2hot_eval_openings
    ch = Challenger(eng)
    with pytest.raises(_lib.P2HotError, match="does not exist"):
        prove_openings([FriBatchInfo([1, 2], [(0, W)])], [b], ch, 2, 1, [1], 0, 1, engine=eng)
    with pytest.raises(_lib.P2HotError, match="another degree / rate / cap"):
        prove_openings([FriBatchInfo([1, 2], [(0, 0)])], [b], ch, 3, 1, [1], 0, 1, engine=eng)
    with pytest.raises(_lib.P2HotError):
        prove_openings([FriBatchInfo([1, 2], [(0, 0)])], [b], ch, 2, 1, [6], 0, 1, engine=eng)   # arity > degree
    other = PolynomialBatch.from_values(rand_field(rng, 2, 16), 2, False, 1, engine=eng)
    with pytest.raises(_lib.P2HotError, match="same degree"):
        eval_openings([b, other], [[1, 2]], eng)
    fp = _lib.FriParams(2, 1, 0, 1, (C.c_uint * 1)(1), 1, 1, 0, 0)                               # hiding = 1
    lay = _lib.FriProofLayout()
    h = (C.c_void_p * 1)(b._h)
    assert eng.lib.p2hot_fri_proof_sizes(h, 1, C.byref(fp), C.byref(lay)) == _lib.OK
    bufs = [np.zeros(max(1, x), dtype=np.uint64) for x in (lay.caps_words, lay.final_poly_words, lay.initial_leaves_words,
                                                          lay.initial_paths_words, lay.step_evals_words, lay.step_paths_words)]
    pr = _lib.FriProof(bufs[0].ctypes.data, bufs[1].ctypes.data, 0, None, bufs[2].ctypes.data, bufs[3].ctypes.data,
                       bufs[4].ctypes.data, bufs[5].ctypes.data)
    info = (_lib.FriBatchInfo * 1)()
    # FriParams::hiding is carried for the caller's transcript only (fri/mod.rs:148): the prover's FRI path does not depend on it
    st = _lib.ChallengerState()
    eng.check(eng.lib.p2hot_challenger_store(ch._h, C.byref(st)))
    rc1 = eng.lib.p2hot_prove_openings(eng.ctx, info, 0, h, 1, ch._h, C.byref(fp), C.byref(pr))
    assert rc1 != _lib.EUNSUPPORTED
    first = [x.copy() for x in bufs]
    eng.check(eng.lib.p2hot_challenger_load(ch._h, C.byref(st)))
    fp.hiding = 0
    assert eng.lib.p2hot_prove_openings(eng.ctx, info, 0, h, 1, ch._h, C.byref(fp), C.byref(pr)) == rc1
    assert all((a == b_).all() for a, b_ in zip(first, bufs))
    assert eng.lib.p2hot_ctx_trim(eng.ctx) == _lib.OK


def test_starky_padding_arguments_reach_the_transcript(eng, ora):
    """final_poly_coeff_len / max_num_query_steps (fri/prover.rs:89-90, :122-147) as arguments of p2hot_prove_openings:
    the transcript advances like the reference's (dummy zero caps + challenges, zero coefficients observed)"""
    from plonky2_amd.fri.oracle import FriBatchInfo, PolynomialBatch, prove_openings
    from plonky2_amd.iop.challenger import Challenger
    rng = np.random.default_rng(3)
    n, rb, cap, arity = 64, 1, 2, [2]
    co = rand_field(rng, 2, n)
    b = PolynomialBatch.from_coeffs(co, rb, False, cap, engine=eng)
    c, oc = Challenger(eng), ora.Challenger()
    z = [5, 9]
    proof = prove_openings([FriBatchInfo(z, [(0, 0), (0, 1)])], [b], c, rb, cap, arity, 0, 2, engine=eng,
                           final_poly_coeff_len=40, max_num_query_steps=3)
    # oracle side: the same steps with the padding spelled out
    alpha = oc.get_extension_challenge()
    comp = ora.reduce_polys_base(co, np.array(alpha, dtype=np.uint64))
    quo = ora.divide_by_linear(comp, z)
    pad = np.zeros((n << rb, 2), dtype=np.uint64)
    pad[:n] = quo
    o = ora.fri_commit(pad, rb, cap, arity, oc.clone())
    assert (proof["final_poly"] == o["final"]).all() and (proof["commit_phase_merkle_caps"][0] == o["caps"][0]).all()
    # replay the reference order by hand: round cap + beta, 2 dummy steps, final poly, zero padding
    oc.observe_cap(o["caps"][0])
    oc.get_extension_challenge()
    for _ in range(len(arity), 3):
        oc.observe_elements(np.zeros(4 << cap, dtype=np.uint64))
        oc.get_extension_challenge()
    oc.observe_elements(o["final"].reshape(-1))
    oc.observe_elements(np.zeros(2 * (40 - len(o["final"])), dtype=np.uint64))
    assert proof["pow_witness"] == ora.fri_pow(oc, 0)
    assert proof["query_indices"] == [r % (n << rb) for r in oc.get_n_challenges(2)]


def test_pow_witness_outside_the_first_search_range(eng, ora, monkeypatch):
    """The grind searches 2^(pow_bits + 5) candidates on the device without a host round trip; if none is valid
    (probability e^-32) p2hot_prove_openings rewinds the transcript and continues with host-checked chunks.  Forced here
    by shrinking the range (P2HOT_POW_RANGE_LOG): witness, query indices and the rest of the proof are unchanged."""
    from plonky2_amd.fri.oracle import FriBatchInfo, PolynomialBatch, prove_openings
    from plonky2_amd.iop.challenger import Challenger
    rng = np.random.default_rng(11)
    co = rand_field(rng, 3, 32)
    b = PolynomialBatch.from_coeffs(co, 2, False, 1, engine=eng)
    inst = [FriBatchInfo([3, 4], [(0, 0), (0, 1), (0, 2)])]
    proofs = []
    for rng_log in (None, "-6"):          # pow_bits 9: 2^3 candidates hold a witness with probability 1.5 %
        if rng_log is None:
            monkeypatch.delenv("P2HOT_POW_RANGE_LOG", raising=False)
        else:
            monkeypatch.setenv("P2HOT_POW_RANGE_LOG", rng_log)
        c = Challenger(eng)
        c.observe_elements(np.arange(5, dtype=np.uint64))
        proofs.append((prove_openings(inst, [b], c, 2, 1, [2], 9, 3, engine=eng), c.get_n_challenges(3)))
    (p0, t0), (p1, t1) = proofs
    assert p0["pow_witness"] == p1["pow_witness"] and p0["pow_witness"] >= 8
    assert p0["query_indices"] == p1["query_indices"] and t0 == t1
    for qa, qb in zip(p0["query_round_proofs"], p1["query_round_proofs"]):
        for (la, sa), (lb, sb) in zip(qa["initial_trees_proof"] + qa["steps"], qb["initial_trees_proof"] + qb["steps"]):
            assert (la == lb).all() and (sa == sb).all()
    oc = ora.Challenger()
    oc.observe_elements(np.arange(5, dtype=np.uint64))
    a = oc.get_extension_challenge()
    quo = ora.divide_by_linear(ora.reduce_polys_base(co, np.array(a, dtype=np.uint64)), [3, 4])
    pad = np.zeros((32 << 2, 2), dtype=np.uint64)
    pad[:32] = quo
    ora.fri_commit(pad, 2, 1, [2], oc)
    assert ora.fri_pow(oc, 9) == p0["pow_witness"]


@pytest.mark.parametrize("M", [1, 3, 11])
def test_prove_openings_many_equals_single_calls(eng, ora, M):
    """p2hot_prove_openings_many: M independent opening proofs on sibling contexts of the same GPU (own streams, host threads);
    every proof's buffers and the transcript afterwards equal a single p2hot_prove_openings call's (fri/oracle.rs:176-237)"""
    import ctypes as C
    from plonky2_amd import _lib
    from plonky2_amd.fri.oracle import PolynomialBatch
    from plonky2_amd.iop.challenger import Challenger
    rng = np.random.default_rng(500 + M)
    log_n, rb, cap, widths, arity, Q = 6, 3, 2, [5, 3], [2, 1], 4
    n = 1 << log_n
    ab = (C.c_uint * len(arity))(*arity)
    fp = _lib.FriParams(rb, cap, 3, Q, ab, len(arity), 0, 0, 0)
    oracles = [[PolynomialBatch.from_coeffs(rand_field(rng, w, n), rb, False, cap, engine=eng) for w in widths] for _ in range(M)]
    allp = [(oi, pi) for oi, w in enumerate(widths) for pi in range(w)]
    oi_arr = (C.c_uint32 * len(allp))(*[o for o, _ in allp])
    pi_arr = (C.c_uint32 * len(allp))(*[p for _, p in allp])
    points = [[int(x) for x in rand_field(rng, 2)] for _ in range(M)]
    lay = _lib.FriProofLayout()
    h0 = (C.c_void_p * len(widths))(*[o._h for o in oracles[0]])
    assert eng.lib.p2hot_fri_proof_sizes(h0, len(widths), C.byref(fp), C.byref(lay)) == _lib.OK
    names = ("caps", "final_poly", "initial_leaves", "initial_paths", "step_evals", "step_paths")

    def run(many):
        chs = [Challenger(eng) for _ in range(M)]
        for j, ch in enumerate(chs):
            ch.observe_elements(np.arange(j, j + 5, dtype=np.uint64))
        bufs = [{k: np.zeros(max(1, getattr(lay, k + "_words")), dtype=np.uint64) for k in names} for _ in range(M)]
        qidx = [np.zeros(Q, dtype=np.uint64) for _ in range(M)]
        infos = []
        for j in range(M):
            inf = (_lib.FriBatchInfo * 1)()
            inf[0].point[0], inf[0].point[1] = points[j]
            inf[0].oracle_index, inf[0].poly_index, inf[0].n_polys = oi_arr, pi_arr, len(allp)
            infos.append(inf)
        proofs = (_lib.FriProof * M)()
        for j in range(M):
            b = bufs[j]
            proofs[j] = _lib.FriProof(b["caps"].ctypes.data, b["final_poly"].ctypes.data, 0, qidx[j].ctypes.data, b["initial_leaves"].ctypes.data,
                                      b["initial_paths"].ctypes.data, b["step_evals"].ctypes.data, b["step_paths"].ctypes.data)
        if many:
            bp = (C.POINTER(_lib.FriBatchInfo) * M)(*[C.cast(inf, C.POINTER(_lib.FriBatchInfo)) for inf in infos])
            nb = (C.c_size_t * M)(*([1] * M))
            hs = (C.c_void_p * (M * len(widths)))(*[o._h for row in oracles for o in row])
            cp = (C.c_void_p * M)(*[ch._h for ch in chs])
            eng.check(eng.lib.p2hot_prove_openings_many(eng.ctx, M, bp, nb, hs, len(widths), cp, C.byref(fp), proofs))
        else:
            for j in range(M):
                hs = (C.c_void_p * len(widths))(*[o._h for o in oracles[j]])
                eng.check(eng.lib.p2hot_prove_openings(eng.ctx, infos[j], 1, hs, len(widths), chs[j]._h, C.byref(fp), C.byref(proofs[j])))
        return bufs, qidx, [int(proofs[j].pow_witness) for j in range(M)], [ch.get_n_challenges(3) for ch in chs]

    a, b = run(False), run(True)
    for j in range(M):
        for k in names:
            assert (a[0][j][k] == b[0][j][k]).all(), (j, k)
        assert (a[1][j] == b[1][j]).all() and a[2][j] == b[2][j] and a[3][j] == b[3][j], j
    assert len({tuple(x["final_poly"].tolist()) for x in b[0]}) == M  # the proofs are different proofs
    # argument errors name the proof
    bad = (C.c_void_p * (M * len(widths)))(*([None] * (M * len(widths))))
    cp = (C.c_void_p * M)(*[Challenger(eng)._h for _ in range(M)])
    assert eng.lib.p2hot_prove_openings_many(eng.ctx, M, None, None, bad, len(widths), cp, C.byref(fp), None) == _lib.EINVAL


def test_polynomial_batch_bytes_streamed_and_golden(eng):
    """the streamed writer (tests/wire_format.py iter_polynomial_batch_bytes) produces write_polynomial_batch's bytes,
    and a library-built batch (host pointers in, rows back through p2hot_batch_rows in committed order) serialises to the bytes
    the oracle-built batch does: `sha256_polynomial_batch` of tests/golden/commit_caps.json (tools/gen_golden_caps.py
    --batch-bytes); the full-size shapes of the same record are compared on the GPU (tests/test_gpu_fullsize.py)."""
    import hashlib
    import json
    import os
    from plonky2_amd.fri.oracle import PolynomialBatch
    from plonky2_amd.util.synthetic import splitmix_columns_numpy
    from tests.conftest import ROOT
    from tests.wire_format import iter_polynomial_batch_bytes, polynomial_batch_sha256, write_polynomial_batch
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "commit_caps.json")))["tiny_wires"]
    W, log_n, rb, cap = g["W"], g["log_n"], g["rate_bits"], g["cap_height"]
    b = PolynomialBatch.from_values(splitmix_columns_numpy(0, W, 1 << log_n), rb, False, cap, engine=eng)
    N = 1 << (log_n + rb)
    chunks = lambda: (b._owner.rows(np.arange(r, min(r + 300, N))) for r in range(0, N, 300))  # noqa: E731  (ragged on purpose)
    whole = write_polynomial_batch(b)
    assert b"".join(iter_polynomial_batch_bytes(b.polynomials, chunks(), b.merkle_tree.digests, b.merkle_tree.cap.entries, cap, log_n, rb, False)) == whole
    assert hashlib.sha256(whole).hexdigest() == g["sha256_polynomial_batch"]
    assert polynomial_batch_sha256(b.polynomials, chunks(), b.merkle_tree.digests, b.merkle_tree.cap.entries, cap, log_n, rb, False) == g["sha256_polynomial_batch"]
    with pytest.raises(ValueError):
        list(iter_polynomial_batch_bytes(b.polynomials, list(chunks())[:-1], b.merkle_tree.digests, b.merkle_tree.cap.entries, cap, log_n, rb, False))
