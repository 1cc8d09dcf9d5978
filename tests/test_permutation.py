"""the permutation argument's partial products and Z polynomials
(plonky2/src/plonk/prover.rs:356-449, util/partial_products.rs) against the oracle's sequential restatement,
plus the reference's own invariants: check_partial_products vanishes on every row (partial_products.rs:53-78,
test :151-176) and Z closes (Z(g^n) = Z(1) = 1) when the wires satisfy the copy constraints sigma encodes."""
import numpy as np
import pytest

from tests.conftest import P, rand_field

GEN = 14293326489335486720  # MULTIPLICATIVE_GROUP_GENERATOR, field/src/goldilocks_field.rs:80


def _k_is(num_shifts):
    """get_unique_coset_shifts (field/src/cosets.rs:9-24): g^0 .. g^(num_shifts-1)"""
    out, x = [], 1
    for _ in range(num_shifts):
        out.append(x)
        x = x * GEN % P
    return np.asarray(out, dtype=np.uint64)


def _subgroup(ora, log_n):
    g, x, out = ora.root_of_unity(log_n), 1, []
    for _ in range(1 << log_n):
        out.append(x)
        x = x * g % P
    return out


def _permutation_instance(ora, rng, num_routed, log_n):
    """wires constant on the cycles of a random permutation of the (wire, row) positions, and the sigma values
    k_{j'} * w^{i'} of the position each (j, i) maps to (circuit_builder.rs sigma_vecs / permutation_argument.rs)"""
    n = 1 << log_n
    sub, k = _subgroup(ora, log_n), _k_is(num_routed)
    m = num_routed * n
    perm = rng.permutation(m)
    label = np.arange(m)
    for s in range(m):  # cycle labels = smallest member
        if label[s] != s:
            continue
        t = perm[s]
        while t != s:
            label[t] = s
            t = perm[t]
    vals = rand_field(rng, m)
    wires = vals[label].reshape(num_routed, n)
    sig = np.zeros(m, dtype=np.uint64)
    for pos in range(m):
        j2, i2 = divmod(int(perm[pos]), n)
        sig[pos] = int(k[j2]) * sub[i2] % P
    return wires, sig.reshape(num_routed, n), k


@pytest.mark.parametrize("num_routed,degree,log_n,nc", [(80, 8, 5, 2), (9, 4, 3, 1), (10, 3, 7, 2), (4, 2, 0, 1), (33, 8, 10, 2)])
def test_partial_products_vs_oracle(eng, ora, num_routed, degree, log_n, nc):
    from plonky2_amd.plonk.prover import all_wires_permutation_partial_products, num_partial_products
    rng = np.random.default_rng(num_routed * 101 + log_n)
    n = 1 << log_n
    wires = rand_field(rng, num_routed, n, noncanonical=True)
    sigmas = rand_field(rng, num_routed, n, noncanonical=True)
    k = _k_is(num_routed)
    betas, gammas = rand_field(rng, nc), rand_field(rng, nc)
    got = eng.host(all_wires_permutation_partial_products(wires, sigmas, k, degree, betas, gammas, eng))
    num_prods = num_partial_products(num_routed, degree)
    assert got.shape == (nc * (num_prods + 1), n)
    for ch in range(nc):
        exp = ora.partial_products(wires, sigmas, k, degree, betas[ch], gammas[ch])   # [num_prods + 1][n], Z last
        assert (got[ch] == exp[num_prods]).all()                                       # Zs lead the batch (prover.rs:224-229)
        assert (got[nc + ch * num_prods: nc + (ch + 1) * num_prods] == exp[:num_prods]).all()
    assert (got < np.uint64(P)).all()


def test_partial_products_invariants_and_commit(eng, ora):
    """check_partial_products == 0 on every row, Z(1) = 1, Z closes on a satisfied permutation, and the batch commits"""
    from plonky2_amd.plonk.prover import all_wires_permutation_partial_products, partial_products_and_zs_commitment
    rng = np.random.default_rng(77)
    num_routed, degree, log_n = 12, 4, 6
    n = 1 << log_n
    wires, sigmas, k = _permutation_instance(ora, rng, num_routed, log_n)
    beta, gamma = rand_field(rng, 1), rand_field(rng, 1)
    got = eng.host(all_wires_permutation_partial_products(wires, sigmas, k, degree, beta, gamma, eng))
    z, pps = got[0], got[1:]
    sub = _subgroup(ora, log_n)
    b, g = int(beta[0]), int(gamma[0])
    assert int(z[0]) == 1
    for i in range(n):
        accs = [int(z[i])] + [int(p[i]) for p in pps] + [int(z[(i + 1) % n])]   # Z(x), partials, Z(gx); wraps: Z closes
        for c in range(num_routed // degree):
            num = den = 1
            for j in range(c * degree, (c + 1) * degree):
                w = int(wires[j][i])
                num = num * ((w + b * (int(k[j]) * sub[i] % P) + g) % P) % P
                den = den * ((w + b * int(sigmas[j][i]) + g) % P) % P
            assert (accs[c] * num - accs[c + 1] * den) % P == 0, (i, c)
    batch = partial_products_and_zs_commitment(wires, sigmas, k, degree, beta, gamma, 3, 2, engine=eng)
    exp = ora.commit(got, 3, 2, True)
    assert (batch.merkle_tree.cap.entries == exp["cap"]).all()


def test_partial_products_zero_denominator_is_an_error(eng):
    """the reference panics in batch_multiplicative_inverse ("Tried to invert zero"); the library reports EINVAL"""
    from plonky2_amd.plonk.prover import all_wires_permutation_partial_products
    num_routed, n = 4, 8
    wires = np.zeros((num_routed, n), dtype=np.uint64)
    sigmas = np.ones((num_routed, n), dtype=np.uint64)
    # wire + beta * sigma + gamma = 0 + 5 * 1 + (P - 5) = 0
    with pytest.raises(Exception, match="invert zero"):
        all_wires_permutation_partial_products(wires, sigmas, _k_is(num_routed), 2, [5], [P - 5], eng)


def test_partial_products_shape_errors(eng):
    from plonky2_amd.plonk.prover import all_wires_permutation_partial_products
    w = np.ones((4, 8), dtype=np.uint64)
    with pytest.raises(ValueError):
        all_wires_permutation_partial_products(w, w[:3], _k_is(4), 2, [1], [2], eng)
    with pytest.raises(ValueError):
        all_wires_permutation_partial_products(w, w, _k_is(4), 4, [1], [2], eng)   # degree must be < num_routed
    with pytest.raises(ValueError):
        all_wires_permutation_partial_products(w[:, :6], w[:, :6], _k_is(4), 2, [1], [2], eng)


@pytest.mark.parametrize("degree_bits,factor,nc", [(5, 8, 2), (4, 6, 1), (3, 2, 2)])
def test_quotient_poly_chunks(eng, ora, degree_bits, factor, nc):
    """the gate-independent tail of the quotient (prover.rs:274-289, :810-815): coset_ifft, trim, split into chunks,
    commit -- against the oracle's coset_ifft and from_coeffs; a quotient of too high degree is rejected like the
    reference's trim_to_len"""
    from plonky2_amd.fri.oracle import PolynomialBatch
    from plonky2_amd.plonk.prover import quotient_poly_chunks
    rng = np.random.default_rng(degree_bits * 10 + factor)
    n = 1 << degree_bits
    qbits = (factor - 1).bit_length()
    m = n << qbits
    coeffs = np.zeros((nc, m), dtype=np.uint64)
    coeffs[:, :n * factor] = rand_field(rng, nc, n * factor)                # degree < factor * n
    values = np.stack([ora.coset_fft(c.copy()) for c in coeffs])            # what compute_quotient_polys evaluates
    chunks = quotient_poly_chunks(values, degree_bits, factor, eng)
    got = eng.host(chunks)
    assert got.shape == (nc * factor, n)
    for ch in range(nc):
        back = ora.coset_ifft(values[ch].copy()) % np.uint64(P)
        assert (got[ch * factor:(ch + 1) * factor].reshape(-1) == back[:n * factor]).all()
    batch = PolynomialBatch.from_coeffs(chunks, 3, False, 1, engine=eng)
    assert (batch.merkle_tree.cap.entries == ora.commit(got, 3, 1, False)["cap"]).all()
    if n * factor < m:
        coeffs[0, m - 1] = 5                                                # not divisible by Z_H
        bad = np.stack([ora.coset_fft(c.copy()) for c in coeffs])
        with pytest.raises(ValueError, match="Quotient has failed"):
            quotient_poly_chunks(bad, degree_bits, factor, eng)


# ---------------------------------------------------------------- the permutation argument's share of the quotient
def _quotient_instance(ora, rng, num_routed, degree, log_n, rate_bits, nc, extra_wires=3, num_constants=2, cap=2):
    """a satisfied permutation argument as the prover holds it when compute_quotient_polys runs (plonk/prover.rs:231-260): the
    wires commitment (routed wires first), the constants_sigmas commitment, the Zs + partial products commitment (Z first,
    prover.rs:224-229), the challenges.  No gates: a circuit whose only constraints are its copy constraints"""
    n = 1 << log_n
    routed, sigmas, k = _permutation_instance(ora, rng, num_routed, log_n)
    wires = np.concatenate([routed, rand_field(rng, extra_wires, n)]) if extra_wires else routed
    cs = np.concatenate([rand_field(rng, num_constants, n), sigmas]) if num_constants else sigmas
    betas, gammas, alphas = rand_field(rng, nc), rand_field(rng, nc), rand_field(rng, nc)
    num_prods = -(-num_routed // degree) - 1
    pps = [ora.partial_products(routed, sigmas, k, degree, betas[c], gammas[c]) for c in range(nc)]   # [num_prods + 1][n], Z last
    zs = np.stack([pp[num_prods] for pp in pps] + [row for pp in pps for row in pp[:num_prods]])
    return dict(wires=wires, cs=cs, zs=zs, k=k, betas=betas, gammas=gammas, alphas=alphas, sigmas_first=num_constants, num_prods=num_prods,
                n=n, log_n=log_n, rate_bits=rate_bits, degree=degree, cap=cap, nc=nc)


def _oracle_quotient(ora, q, gate_sums=None, mutate=None):
    lv = {name: ora.commit(q[name], q["rate_bits"], q["cap"], True)["leaves"].copy() for name in ("wires", "cs", "zs")}
    if mutate:
        mutate(lv)
    return ora.quotient_permutation(lv["wires"], lv["cs"], q["sigmas_first"], lv["zs"], q["log_n"], q["rate_bits"], q["k"], q["degree"],
                                    q["betas"], q["gammas"], q["alphas"], gate_sums)


@pytest.mark.parametrize("num_routed,degree,log_n,rate_bits,nc", [(12, 5, 5, 3, 2), (7, 3, 4, 2, 1), (20, 6, 3, 3, 2)])
def test_oracle_quotient_of_a_satisfied_permutation_is_a_polynomial(ora, num_routed, degree, log_n, rate_bits, nc):
    """pins the ORACLE's restatement of compute_quotient_polys' gate-independent part (plonk/prover.rs:609-815,
    vanishing_poly.rs:167-330) by the property the reference itself asserts (`trim_to_len`: "Quotient has failed ..."): with a
    quotient degree factor that is not a power of two the quotient coset is larger than the kept degree, and for a satisfied
    permutation argument every coefficient beyond quotient_degree_factor * n vanishes -- row choice (get_lde_values + the next
    row), L_0, the chunk products, the order of the alpha powers and the division by Z_H all have to be right for that; one
    changed wire value breaks it"""
    rng = np.random.default_rng(num_routed * 7 + degree)
    q = _quotient_instance(ora, rng, num_routed, degree, log_n, rate_bits, nc)
    keep = degree << log_n
    vals = _oracle_quotient(ora, q)
    assert vals.shape[1] > keep, "pick a degree whose log2_ceil leaves a tail to check"
    for a in range(nc):
        co = ora.coset_ifft(vals[a])
        assert not co[keep:].any() and co[:keep].any()

    def spoil(lv):
        lv["wires"][3, 1] = (int(lv["wires"][3, 1]) + 1) % P
    bad = _oracle_quotient(ora, q, mutate=spoil)
    assert ora.coset_ifft(bad[0])[keep:].any()
    # the gate terms enter behind the permutation terms: alpha^K * gate_sums, divided by Z_H like the rest
    gs = rand_field(rng, nc, vals.shape[1])
    with_g = _oracle_quotient(ora, q, gate_sums=gs)
    K = nc + nc * (q["num_prods"] + 1)
    g_n = pow(GEN, q["n"], P)
    qbits = (degree - 1).bit_length()
    v = ora.root_of_unity(qbits)
    for a in range(nc):
        for i in (0, 1, vals.shape[1] - 1):
            zh = (g_n * pow(v, i % (1 << qbits), P) - 1) % P
            extra = pow(int(q["alphas"][a]), K, P) * int(gs[a, i]) % P * pow(zh, P - 2, P) % P
            assert int(with_g[a, i]) == (int(vals[a, i]) + extra) % P


@pytest.mark.parametrize("num_routed,degree,log_n,rate_bits,nc,with_gates", [
    (12, 5, 5, 3, 2, False), (7, 3, 4, 2, 1, True), (80, 8, 4, 3, 2, True),   # the last: standard_recursion_config's 80 routed wires, degree 8
    (20, 6, 3, 3, 3, False), (10, 4, 6, 3, 2, False),                         # rate above the quotient degree: step = 2 (prover.rs:640)
    (19, 8, 5, 3, 2, True), (9, 8, 3, 3, 2, False),                           # the pipelined <2, 8> instantiation with a partial last chunk
])
def test_quotient_polys_vs_oracle(eng, ora, num_routed, degree, log_n, rate_bits, nc, with_gates):
    """p2hot_quotient_polys: quotient values bit-exact against the oracle's restatement of the reference loop, the chunk polynomials
    = the oracle's coset_ifft of them, the commit of the chunks = the oracle's commit; with a wrong witness the call reports the
    reference's panic text"""
    from plonky2_amd.fri.oracle import PolynomialBatch
    from plonky2_amd.plonk.prover import compute_quotient_polys
    rng = np.random.default_rng(num_routed * 13 + degree + log_n)
    q = _quotient_instance(ora, rng, num_routed, degree, log_n, rate_bits, nc)
    n, qbits = q["n"], (degree - 1).bit_length()
    m = n << qbits
    gs = rand_field(rng, nc, m) if with_gates else None
    b_w, b_cs, b_z = (PolynomialBatch.from_values(q[name], rate_bits, False, q["cap"], engine=eng) for name in ("wires", "cs", "zs"))
    exp = _oracle_quotient(ora, q, gate_sums=gs)
    if with_gates:   # random "gate terms" are not divisible: only the values can be compared (and only when nothing is trimmed)
        if m == degree * n:
            cols, vals = compute_quotient_polys(b_w, b_cs, q["sigmas_first"], b_z, q["k"], degree, q["betas"], q["gammas"], q["alphas"], gate_sums=gs,
                                                want_values=True, engine=eng)
            assert (vals == exp).all()
            co = np.stack([ora.coset_ifft(exp[a]) for a in range(nc)]).reshape(nc * degree, n)
            assert (cols.host() == co).all()
        else:
            with pytest.raises(ValueError, match="Quotient has failed"):
                compute_quotient_polys(b_w, b_cs, q["sigmas_first"], b_z, q["k"], degree, q["betas"], q["gammas"], q["alphas"], gate_sums=gs, engine=eng)
        return
    cols, vals = compute_quotient_polys(b_w, b_cs, q["sigmas_first"], b_z, q["k"], degree, q["betas"], q["gammas"], q["alphas"], want_values=True, engine=eng)
    assert (vals == exp).all() and (vals < np.uint64(P)).all()
    co = np.stack([ora.coset_ifft(exp[a])[:degree * n] for a in range(nc)]).reshape(nc * degree, n)
    got = cols.host()
    assert got.shape == (nc * degree, n) and (got == co).all()
    batch = PolynomialBatch.from_coeffs(cols, rate_bits, False, q["cap"], engine=eng)     # quotient_polys_commitment (prover.rs:293-305)
    assert (batch.merkle_tree.cap.entries == ora.commit(co, rate_bits, q["cap"], False)["cap"]).all()
    # a witness that violates a copy constraint: the reference panics in trim_to_len
    if m > degree * n:
        bad = dict(q)
        bad["wires"] = q["wires"].copy()
        bad["wires"][1, 2] = (int(bad["wires"][1, 2]) + 1) % P
        b_bad = PolynomialBatch.from_values(bad["wires"], rate_bits, False, q["cap"], engine=eng)
        with pytest.raises(ValueError, match="Quotient has failed"):
            compute_quotient_polys(b_bad, b_cs, q["sigmas_first"], b_z, q["k"], degree, q["betas"], q["gammas"], q["alphas"], engine=eng)


def test_quotient_polys_argument_errors(eng, ora):
    from plonky2_amd import _lib
    from plonky2_amd.fri.oracle import PolynomialBatch
    from plonky2_amd.plonk.prover import compute_quotient_polys
    rng = np.random.default_rng(1)
    q = _quotient_instance(ora, rng, 6, 3, 3, 1, 1)          # quotient degree 2^2 above the rate 2^1
    b_w, b_cs, b_z = (PolynomialBatch.from_values(q[name], 1, False, 0, engine=eng) for name in ("wires", "cs", "zs"))
    with pytest.raises(_lib.P2HotError, match="above the rate"):
        compute_quotient_polys(b_w, b_cs, q["sigmas_first"], b_z, q["k"], 3, q["betas"], q["gammas"], q["alphas"], engine=eng)
    with pytest.raises(_lib.P2HotError, match="narrower"):
        compute_quotient_polys(b_w, b_cs, q["sigmas_first"] + 1, b_z, q["k"], 2, q["betas"], q["gammas"], q["alphas"], engine=eng)


@pytest.mark.parametrize("log_n,rate_bits,first,count,from_values", [(5, 3, 2, 7, True), (0, 1, 0, 3, False), (9, 1, 0, 12, False),
                                                                       (6, 2, 11, 1, True)])
def test_batch_subgroup_values_vs_oracle(eng, ora, log_n, rate_bits, first, count, from_values):
    """the sigma values the Rust shim feeds p2hot_partial_products: values on H of a committed batch's polynomials, recomputed from
    the device-resident coefficients (the reference keeps them in ProverOnlyCircuitData.sigmas, circuit_builder.rs:1177-1179)"""
    from plonky2_amd.fri.oracle import PolynomialBatch
    rng = np.random.default_rng(log_n * 13 + first)
    W, n = 12, 1 << log_n
    cols = rand_field(rng, W, n, noncanonical=True)
    build = PolynomialBatch.from_values if from_values else PolynomialBatch.from_coeffs
    batch = build(cols, rate_bits, False, 0, engine=eng)
    got = batch.subgroup_values(first, count)
    assert (got.width, got.degree_log) == (count, log_n)
    vals = got.host()
    exp = cols % np.uint64(P) if from_values else np.stack([ora.fft(c) for c in cols]) % np.uint64(P)
    assert (vals == exp[first:first + count]).all()
    assert (batch.polynomials == (np.stack([ora.ifft(c) for c in cols]) if from_values else cols % np.uint64(P))).all()   # untouched


def test_partial_products_from_a_committed_sigma_batch(eng, ora):
    """the shim's composition (integration/p2hot.rs all_wires_permutation_partial_products): wires uploaded, sigmas read off the
    constants_sigmas commitment at sigmas_range().start -- the same polynomials the CPU body computes from prover_data.sigmas"""
    from plonky2_amd.fri.oracle import DeviceColumns, PolynomialBatch
    from plonky2_amd.plonk.prover import all_wires_permutation_partial_products, num_partial_products
    rng = np.random.default_rng(4242)
    num_routed, degree, log_n, nc, num_constants = 10, 4, 6, 2, 3
    wires, sigmas, k = _permutation_instance(ora, rng, num_routed, log_n)
    constants = rand_field(rng, num_constants, 1 << log_n)
    cs = PolynomialBatch.from_values(np.concatenate([constants, sigmas]), 3, False, 1, engine=eng)
    betas, gammas = rand_field(rng, nc), rand_field(rng, nc)
    got = all_wires_permutation_partial_products(DeviceColumns.upload(wires, eng), cs.subgroup_values(num_constants, num_routed), k,
                                                 degree, betas, gammas, eng).host()
    num_prods = num_partial_products(num_routed, degree)
    for ch in range(nc):
        exp = ora.partial_products(wires, sigmas, k, degree, betas[ch], gammas[ch])
        assert (got[ch] == exp[num_prods]).all()
        assert (got[nc + ch * num_prods: nc + (ch + 1) * num_prods] == exp[:num_prods]).all()


def test_batch_subgroup_values_argument_errors(eng):
    from plonky2_amd.fri.oracle import PolynomialBatch
    batch = PolynomialBatch.from_coeffs(np.ones((3, 8), dtype=np.uint64), 1, False, 0, engine=eng)
    with pytest.raises(Exception, match="range|polynomial"):
        batch.subgroup_values(2, 2)
    with pytest.raises(Exception, match="range|polynomial"):
        batch.subgroup_values(4, 0)
    with pytest.raises(Exception, match="range|polynomial"):
        batch.subgroup_values(1, 0)   # an empty column set is refused, like a commit without polynomials
