"""The SURVEY section-8 stages of one proof, back to back, on a synthetic instance (plonky2_amd/util/synthetic.py) --
the composition `prove` performs around the stages that are out of scope (plonk/prover.rs:113-362; starky/src/prover.rs:41-155):

  plonk:   wires commit -> partial products + Zs -> Zs commit -> quotient polynomials (permutation terms) -> quotient commit
           -> OpeningSet evaluations -> prove_openings
  starky:  trace commit -> quotient commit -> StarkOpeningSet evaluations -> prove_openings

bench.py times `run_path` (per_proof_path_* lines) and, after the timed repetitions, runs it once more with keep=True and
compares every stage's output with the CPU oracle's bytes for the same instance (tests/golden/path_goldens.json, made by
tools/gen_golden_path.py) through `compare_with_golden`; tests/test_gpu_fullsize.py does the same and also hands the proof to
the restated reference verifier.  Nothing here touches oracle/: the golden file is data."""
import hashlib
import json
import os
import time

import numpy as np

from .synthetic import P, fibonacci_trace, second_point, splitmix_columns_numpy, splitmix_columns_torch

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "path_goldens.json")


def golden(name, path=GOLDEN):
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f).get(name)


def _columns(eng, seed, width, n):
    """splitmix columns as a device buffer: generated on the device under torch, uploaded otherwise (the CPU test tier)"""
    torch = getattr(eng.mem, "torch", None)
    if torch is not None:
        return splitmix_columns_torch(torch, eng.mem.device, seed, width, n)
    return eng.dev(splitmix_columns_numpy(seed, width, n))


def _sha(a):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.uint64))
    return hashlib.sha256(a.astype("<u8").tobytes()).hexdigest()


class PathInputs:
    """what exists before a proof starts: the witness columns and CircuitBuilder::build's constants_sigmas commitment
    (circuit_builder.rs:1182-1191: part of the circuit, made once, outside every timed path)"""

    def __init__(self, eng, inst):
        from ..fri.oracle import PolynomialBatch
        self.inst = inst
        n = 1 << inst["log_n"]
        if inst["kind"] == "plonk":
            self.wires = _columns(eng, inst["wires_seed"], inst["wires_width"], n)
            self.cs = _columns(eng, inst["cs_seed"], inst["cs_width"], n)
            f = inst["num_constants"]
            self.sigmas = self.cs[f:f + inst["num_routed"]]
            self.b_cs = PolynomialBatch.from_values(self.cs, inst["rate_bits"], False, inst["cap_height"], engine=eng)
        else:
            self.trace = eng.dev(fibonacci_trace(inst["log_n"]))
            self.quotient = _columns(eng, inst["quotient_seed"], 2, n)


def run_path(eng, inp, sync=None, keep=False):
    """One proof's stages.  Returns {"stage_ms": {label: ms}, ...}; with keep=True also every stage's output on the host
    (Zs matrix, quotient values, chunk polynomials, caps, opening values, the FriProof dict) for `compare_with_golden`.
    `sync`: called before every lap to make the stage times wall times (torch.cuda.synchronize on the GPU)."""
    from ..fri.oracle import FriBatchInfo, PolynomialBatch, eval_openings, prove_openings
    from ..iop.challenger import Challenger
    from ..plonk.prover import all_wires_permutation_partial_products, compute_quotient_polys
    inst = inp.inst
    rb, cap = inst["rate_bits"], inst["cap_height"]
    stage, out = {}, {}
    t = [time.perf_counter()]

    def lap(label):
        if sync:
            sync()
        now = time.perf_counter()
        stage[label] = (now - t[0]) * 1e3
        t[0] = now

    if inst["kind"] == "plonk":
        nr, qdf = inst["num_routed"], inst["quotient_degree_factor"]
        b_w = PolynomialBatch.from_values(inp.wires, rb, False, cap, engine=eng)
        lap("wires commit (W=135, from_values)")
        zs = all_wires_permutation_partial_products(inp.wires[:nr], inp.sigmas, inst["k_is"], qdf, inst["betas"], inst["gammas"], eng)
        lap("partial products + Zs (80 routed wires, 2 challenges)")
        b_z = PolynomialBatch.from_values(zs, rb, False, cap, engine=eng)
        lap("Zs + partial products commit (W=20, from_values)")
        # compute_quotient_polys without its gate evaluation (prover.rs:609-815; vanishing_poly.rs:167-330): the permutation
        # argument's terms at the 2^(k+3) points of the quotient coset from the three device-resident LDE matrices, / Z_H,
        # coset_ifft, 16 chunks -- the gate constraint terms are circuit specific and out of scope (gate_sums = None)
        q = compute_quotient_polys(b_w, inp.b_cs, inst["num_constants"], b_z, inst["k_is"], qdf, inst["betas"], inst["gammas"],
                                   inst["alphas"], want_values=keep, engine=eng)
        chunks, qvals = q if keep else (q, None)
        chunks_host = chunks.host() if keep else None     # from_coeffs consumes the handle (p2hot_commit_cols)
        lap("quotient polynomials: permutation terms on the quotient coset + coset_ifft + chunks (gate terms excluded)")
        b_q = PolynomialBatch.from_coeffs(chunks, rb, False, cap, engine=eng)
        lap("quotient chunks commit (W=16, from_coeffs)")
        oracles = [inp.b_cs, b_w, b_z, b_q]
    else:
        b_t = PolynomialBatch.from_values(inp.trace, rb, False, cap, engine=eng)
        lap("trace commit (W=2, from_values)")
        b_q = PolynomialBatch.from_coeffs(inp.quotient, rb, False, cap, engine=eng)
        lap("quotient commit (W=2, from_coeffs)")
        oracles = [b_t, b_q]
    ch = Challenger(eng)
    ch.observe_elements(np.asarray(inst["transcript_seed"], dtype=np.uint64))
    zeta = ch.get_extension_challenge()
    gz = second_point(zeta, inst["log_n"])
    if inst["kind"] == "plonk":
        # OpeningSet::new (plonk/proof.rs:314-345): the four commitments at zeta, the Zs commitment again at the second point
        ev_zeta = eval_openings(oracles, [zeta], eng)
        ev_next = eval_openings([oracles[2]], [gz], eng)
        lap("OpeningSet (255 polynomials at zeta; the 20 of the Zs oracle at g*zeta, of which the proof keeps the 2 Z)")
        ev_zeta, ev_next = [e[0] for e in ev_zeta], [e[0] for e in ev_next]
    else:
        ev = eval_openings(oracles, [zeta, gz], eng)
        lap("StarkOpeningSet (4 polynomials at 2 points)")
        ev_zeta, ev_next = [e[0] for e in ev], [e[1] for e in ev]
    proof = prove_openings([FriBatchInfo(zeta, inst["batch_zeta"]), FriBatchInfo(gz, inst["batch_next"])], oracles, ch, rb, cap,
                           inst["arity"], inst["pow_bits"], inst["num_queries"], engine=eng)
    lap("prove_openings (final_poly, FRI commit, PoW %d bits, %d queries x %d initial trees)" % (inst["pow_bits"], inst["num_queries"], len(oracles)))
    out["stage_ms"] = stage
    if keep:
        out.update({"zeta": [int(v) for v in zeta], "second_point": gz, "openings_zeta": ev_zeta, "openings_next": ev_next, "proof": proof,
                    "caps": [np.asarray(o.merkle_tree.cap.entries, dtype=np.uint64) for o in oracles],
                    "transcript_after": [int(x) for x in ch.get_n_challenges(2)], "oracles": oracles})
        if inst["kind"] == "plonk":
            out.update({"zs": eng.host(zs), "quotient_values": qvals, "chunks": chunks_host})
    return out


def run_group_path(group, inst, cs, wires, zs, chunks, sync=None, pipeline_chunks=8):
    """The per-proof path over a p2hot_group (one process, every GPU of the node; SURVEY 8e + 8f-1/2): the four commitments as
    coset-sharded commits from HOST columns (p2hot_group_commit: constants_sigmas and wires from values, the Zs / partial-products
    matrix from values, the quotient chunks from coefficients), the OpeningSet through p2hot_group_eval_openings and the opening
    proof through p2hot_group_prove_openings (rank 0 runs the polynomial side, the owners of the rows serve the initial trees).
    `zs` and `chunks` are the single-GPU stages' outputs (the partial products and the quotient need whole LDE matrices on one
    device; they are inputs here).  Returns the same dict run_path(keep=True) does, minus the single-GPU stage outputs, so
    compare_with_golden checks it against the same oracle record."""
    from ..fri.oracle import shape_fri_proof
    from ..iop.challenger import Challenger
    if inst["kind"] != "plonk":
        raise ValueError("the group path is the plonk instance (coset mode: world <= 2^rate_bits)")
    rb, cap, log_n = inst["rate_bits"], inst["cap_height"], inst["log_n"]
    stage, t = {}, [time.perf_counter()]

    def lap(label):
        if sync:
            sync()
        now = time.perf_counter()
        stage[label] = (now - t[0]) * 1e3
        t[0] = now
    commits = []
    for label, cols, is_values in (("constants_sigmas", cs, True), ("wires", wires, True), ("Zs + partial products", zs, True),
                                   ("quotient chunks", chunks, False)):
        commits.append(group.commit(cols, rb, cap, is_values=is_values, want_leaves=False, want_digests=False, pipeline_chunks=pipeline_chunks))
        lap("sharded commit: %s (W=%d, host columns in)" % (label, len(cols)))
    ch = Challenger(group.engine0())
    ch.observe_elements(np.asarray(inst["transcript_seed"], dtype=np.uint64))
    zeta = ch.get_extension_challenge()
    gz = second_point(zeta, log_n)
    ev_zeta = [e[0] for e in group.eval_openings(commits, [zeta])]
    ev_next = [e[0] for e in group.eval_openings([commits[2]], [gz])]
    lap("OpeningSet over the group")
    flat = group.prove_openings([(zeta, inst["batch_zeta"]), (gz, inst["batch_next"])], commits, ch, rb, cap, inst["arity"],
                                inst["pow_bits"], inst["num_queries"])
    lap("prove_openings over the group")
    proof = shape_fri_proof(flat, flat["pow_witness"], flat["query_indices"], [c["W"] for c in commits], log_n + rb, cap,
                            [int(a) for a in inst["arity"]], inst["num_queries"])
    out = {"stage_ms": stage, "zeta": [int(v) for v in zeta], "second_point": gz, "openings_zeta": ev_zeta, "openings_next": ev_next,
           "proof": proof, "caps": [c["cap"] for c in commits], "transcript_after": [int(x) for x in ch.get_n_challenges(2)]}
    for c in commits:
        c["free"]()
    return out


def serialize_fri_proof(proof):
    """write_fri_proof (plonky2/src/util/serialization/mod.rs:1595-1611, :1558-1573, :1476-1480): caps as bare hashes, per
    query round the (leaf, Merkle proof) pairs of the initial trees then the (evals, Merkle proof) steps -- vectors without
    length prefixes, Merkle proofs with a u8 length --, the final polynomial, the PoW witness; canonical little-endian u64s"""
    out = []

    def merkle(s):
        s = np.asarray(s, dtype="<u8").reshape(-1, 4)
        return bytes([s.shape[0]]) + s.tobytes()
    for c in proof["commit_phase_merkle_caps"]:
        out.append(np.asarray(c, dtype="<u8").tobytes())
    for qr in proof["query_round_proofs"]:
        for leaf, sib in qr["initial_trees_proof"] + qr["steps"]:
            out.append(np.asarray(leaf, dtype="<u8").tobytes())
            out.append(merkle(sib))
    out.append(np.asarray(proof["final_poly"], dtype="<u8").tobytes())
    out.append(np.asarray([int(proof["pow_witness"])], dtype="<u8").tobytes())
    return b"".join(out)


def compare_with_golden(res, g):
    """every kept stage output of `run_path(..., keep=True)` against the oracle's record `g`; returns the list of stages that
    differ (empty = bit-exact)"""
    bad = []

    def same(label, a, b):
        if not np.array_equal(np.asarray(a, dtype=np.uint64), np.asarray(b, dtype=np.uint64)):
            bad.append(label)
    for i, (c, gc) in enumerate(zip(res["caps"], g["caps"])):
        same("cap of oracle %d" % i, c, gc)
    if "zs" in res:
        if _sha(res["zs"]) != g["zs_sha256"]:
            bad.append("partial products + Zs")
        if _sha(res["quotient_values"]) != g["quotient_values_sha256"]:
            bad.append("quotient values")
        if _sha(res["chunks"]) != g["chunks_sha256"]:
            bad.append("quotient chunk polynomials")
    same("zeta", res["zeta"], g["zeta"])
    for i, (a, b) in enumerate(zip(res["openings_zeta"], g["openings_zeta"])):
        same("openings at zeta, oracle %d" % i, a, b)
    for i, (a, b) in enumerate(zip(res["openings_next"], g["openings_next"])):
        same("openings at the second point, %d" % i, a, b)
    pf = res["proof"]
    for i, (a, b) in enumerate(zip(pf["commit_phase_merkle_caps"], g["fri_caps"])):
        same("FRI commit-phase cap %d" % i, a, b)
    same("final_poly", pf["final_poly"], g["final_poly"])
    if int(pf["pow_witness"]) != int(g["pow_witness"]):
        bad.append("pow_witness")
    same("query indices", pf["query_indices"], g["query_indices"])
    blob = serialize_fri_proof(pf)
    if len(blob) != g["proof_len"] or hashlib.sha256(blob).hexdigest() != g["proof_sha256"]:
        bad.append("FriProof bytes (query rounds)")
    same("transcript after the proof", res["transcript_after"], g["transcript_after"])
    return bad
