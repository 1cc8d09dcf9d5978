"""TEST INFRASTRUCTURE ONLY -- a restatement of the reference's FRI verifier, the acceptance check for whatever the
GPU path produces ("FRI fold consistent with barycentric interpolation at beta on each coset
and with the final polynomial").  Pure-Python integers over the oracle's Poseidon / Merkle / Challenger primitives;
meant for small instances.  Follows, function by function:

  plonky2/src/fri/challenges.rs:28-87      Challenger::fri_challenges
  plonky2/src/fri/verifier.rs:20-46        compute_evaluation
  plonky2/src/fri/verifier.rs:48-60        fri_verify_proof_of_work
  plonky2/src/fri/verifier.rs:62-110       verify_fri_proof
  plonky2/src/fri/verifier.rs:112-122      fri_verify_initial_proof
  plonky2/src/fri/verifier.rs:124-161      fri_combine_initial
  plonky2/src/fri/verifier.rs:163-245      fri_verifier_query_round
  plonky2/src/fri/verifier.rs:254-267      PrecomputedReducedOpenings::from_os_and_alpha
  plonky2/src/util/reducing.rs:56-59, :91-95  ReducingFactor::reduce / shift
  field/src/interpolation.rs:40-65         interpolate / barycentric_weights

Nothing here is imported by the product; only tests/ use it.
"""
import numpy as np

from . import p2oracle as ora

P = 0xFFFFFFFF00000001
W = 7                          # F_p^2 = F_p[X] / (X^2 - 7), field/src/goldilocks_extensions.rs:19
MULTIPLICATIVE_GROUP_GENERATOR = 14293326489335486720   # field/src/goldilocks_field.rs:80


class VerificationError(Exception):
    """an `ensure!` of the reference verifier failed"""


# ------------------------------------------------------------------ F_p^2 on Python ints
def e_from_base(x):
    return (int(x) % P, 0)


def e_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def e_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def e_mul(a, b):
    return ((a[0] * b[0] + W * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def e_inv(a):
    norm = (a[0] * a[0] - W * a[1] * a[1]) % P
    if norm == 0:
        raise ZeroDivisionError("Tried to invert zero")
    ni = pow(norm, P - 2, P)
    return (a[0] * ni % P, (-a[1]) * ni % P)


def e_pow(a, e):
    r = (1, 0)
    while e:
        if e & 1:
            r = e_mul(r, a)
        a = e_mul(a, a)
        e >>= 1
    return r


def e_of(x):
    return (int(x[0]) % P, int(x[1]) % P)


def reverse_bits(x, bits):
    return int(format(x, "0%db" % bits)[::-1], 2) if bits else 0


class ReducingFactor:  # util/reducing.rs:25-106 (extension base)
    def __init__(self, base):
        self.base, self.count = base, 0

    def reduce(self, xs):
        acc = (0, 0)
        for x in reversed(list(xs)):
            self.count += 1
            acc = e_add(e_mul(self.base, acc), x)
        return acc

    def shift(self, x):
        t = e_mul(e_pow(self.base, self.count), x)
        self.count = 0
        return t


# ------------------------------------------------------------------ challenges (fri/challenges.rs:28-87)
def fri_challenges(challenger, commit_phase_merkle_caps, final_poly, pow_witness, degree_bits, rate_bits, cap_height,
                   num_query_rounds, final_poly_coeff_len=None, max_num_query_steps=None):
    """challenger: an oracle Challenger that has observed everything up to (and including) the openings"""
    lde_size = 1 << (degree_bits + rate_bits)
    fri_alpha = e_of(challenger.get_extension_challenge())
    fri_betas = []
    for cap in commit_phase_merkle_caps:
        challenger.observe_cap(np.asarray(cap, dtype=np.uint64))
        fri_betas.append(e_of(challenger.get_extension_challenge()))
    if max_num_query_steps is not None:
        zero_cap = np.zeros((1 << cap_height) * 4, dtype=np.uint64)
        for _ in range(len(commit_phase_merkle_caps), max_num_query_steps):
            challenger.observe_elements(zero_cap)
            challenger.get_extension_challenge()
    fp = np.asarray(final_poly, dtype=np.uint64).reshape(-1)
    if fp.size:
        challenger.observe_elements(fp)
    if final_poly_coeff_len is not None:
        for _ in range(fp.size // 2, final_poly_coeff_len):
            challenger.observe_elements(np.zeros(2, dtype=np.uint64))
    challenger.observe_elements(np.asarray([pow_witness], dtype=np.uint64))
    fri_pow_response = int(challenger.get_challenge())
    fri_query_indices = [int(challenger.get_challenge()) % lde_size for _ in range(num_query_rounds)]
    return {"fri_alpha": fri_alpha, "fri_betas": fri_betas, "fri_pow_response": fri_pow_response,
            "fri_query_indices": fri_query_indices}


# ------------------------------------------------------------------ verifier.rs
def barycentric_weights(points):
    out = []
    for i, (xi, _) in enumerate(points):
        prod = (1, 0)
        for j, (xj, _) in enumerate(points):
            if j != i:
                prod = e_mul(prod, e_sub(xi, xj))
        out.append(e_inv(prod))
    return out


def interpolate(points, x, weights):
    for xi, yi in points:
        if xi == x:
            return yi
    l_x = (1, 0)
    for xi, _ in points:
        l_x = e_mul(l_x, e_sub(x, xi))
    s = (0, 0)
    for (xi, yi), wi in zip(points, weights):
        s = e_add(s, e_mul(e_mul(wi, e_inv(e_sub(x, xi))), yi))
    return e_mul(l_x, s)


def compute_evaluation(x, x_index_within_coset, arity_bits, evals, beta):
    arity = 1 << arity_bits
    assert len(evals) == arity
    g = ora.root_of_unity(arity_bits)
    evals = [evals[reverse_bits(i, arity_bits)] for i in range(arity)]          # reverse_index_bits_in_place
    rev = reverse_bits(x_index_within_coset, arity_bits)
    coset_start = x * pow(g, arity - rev, P) % P
    points = [(e_from_base(coset_start * pow(g, i, P) % P), evals[i]) for i in range(arity)]
    return interpolate(points, beta, barycentric_weights(points))


def fri_verify_proof_of_work(fri_pow_response, proof_of_work_bits):
    leading_zeros = 64 - int(fri_pow_response).bit_length()
    if leading_zeros < proof_of_work_bits + (64 - P.bit_length()):
        raise VerificationError("Invalid proof of work witness.")


def _verify_merkle(leaf, index, cap, siblings):
    if not ora.merkle_verify(np.asarray(leaf, dtype=np.uint64).reshape(-1), int(index), np.asarray(cap, dtype=np.uint64),
                             np.asarray(siblings, dtype=np.uint64).reshape(-1, 4)):
        raise VerificationError("Invalid Merkle proof.")


def fri_combine_initial(batches, initial_trees_proof, alpha, subgroup_x, reduced_openings_at_point):
    sx = e_from_base(subgroup_x)
    rf = ReducingFactor(alpha)
    total = (0, 0)
    for (point, polynomials), reduced_openings in zip(batches, reduced_openings_at_point):
        evals = [e_from_base(initial_trees_proof[oi][0][pi]) for (oi, pi) in polynomials]   # proof.unsalted_eval
        reduced_evals = rf.reduce(evals)
        numerator = e_sub(reduced_evals, reduced_openings)
        denominator = e_sub(sx, e_of(point))
        total = rf.shift(total)
        total = e_add(total, e_mul(numerator, e_inv(denominator)))
    return total


def fri_verifier_query_round(batches, challenges, reduced_openings_at_point, initial_merkle_caps, proof, x_index, n,
                             round_proof, reduction_arity_bits):
    init = round_proof["initial_trees_proof"]
    if len(init) != len(initial_merkle_caps):
        raise VerificationError("shape: initial trees")
    for (evals, merkle_proof), cap in zip(init, initial_merkle_caps):                       # fri_verify_initial_proof
        _verify_merkle(evals, x_index, cap, merkle_proof)
    log_n = n.bit_length() - 1
    subgroup_x = MULTIPLICATIVE_GROUP_GENERATOR * pow(ora.root_of_unity(log_n), reverse_bits(x_index, log_n), P) % P
    old_eval = fri_combine_initial(batches, init, challenges["fri_alpha"], subgroup_x, reduced_openings_at_point)
    for i, arity_bits in enumerate(reduction_arity_bits):
        arity = 1 << arity_bits
        evals = [e_of(v) for v in np.asarray(round_proof["steps"][i][0], dtype=np.uint64).reshape(-1, 2)]
        if len(evals) != arity:
            raise VerificationError("shape: step evals")
        coset_index = x_index >> arity_bits
        x_index_within_coset = x_index & (arity - 1)
        if evals[x_index_within_coset] != old_eval:
            raise VerificationError("FRI step %d is inconsistent with the previous evaluation" % i)
        old_eval = compute_evaluation(subgroup_x, x_index_within_coset, arity_bits, evals, challenges["fri_betas"][i])
        _verify_merkle(np.asarray(round_proof["steps"][i][0], dtype=np.uint64).reshape(-1), coset_index,          # flatten(evals)
                       proof["commit_phase_merkle_caps"][i], round_proof["steps"][i][1])
        subgroup_x = pow(subgroup_x, arity, P)                                              # exp_power_of_2
        x_index = coset_index
    acc = (0, 0)                                                                            # final_poly.eval
    sx = e_from_base(subgroup_x)
    for c in reversed([e_of(c) for c in np.asarray(proof["final_poly"], dtype=np.uint64).reshape(-1, 2)]):
        acc = e_add(e_mul(acc, sx), c)
    if acc != old_eval:
        raise VerificationError("Final polynomial evaluation is invalid.")


def verify_fri_proof(batches, openings, challenges, initial_merkle_caps, proof, degree_bits, rate_bits,
                     reduction_arity_bits, proof_of_work_bits, num_query_rounds):
    """batches: [(point [2], [(oracle_index, polynomial_index), ...])] (FriInstanceInfo.batches);
    openings: per batch the opened values [[c0, c1], ...] (FriOpenings); proof: the dict prove_openings returns."""
    n = 1 << (degree_bits + rate_bits)
    if len(proof["commit_phase_merkle_caps"]) != len(reduction_arity_bits):                 # validate_fri_proof_shape
        raise VerificationError("shape: commit phase caps")
    if len(np.asarray(proof["final_poly"]).reshape(-1, 2)) != (1 << degree_bits) >> sum(reduction_arity_bits):
        raise VerificationError("shape: final polynomial length")
    fri_verify_proof_of_work(challenges["fri_pow_response"], proof_of_work_bits)
    if num_query_rounds != len(proof["query_round_proofs"]):
        raise VerificationError("Number of query rounds does not match config.")
    reduced = [ReducingFactor(challenges["fri_alpha"]).reduce([e_of(v) for v in vals]) for vals in openings]
    for x_index, round_proof in zip(challenges["fri_query_indices"], proof["query_round_proofs"]):
        fri_verifier_query_round(batches, challenges, reduced, initial_merkle_caps, proof, x_index, n, round_proof,
                                 reduction_arity_bits)
