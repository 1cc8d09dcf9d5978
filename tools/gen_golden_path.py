#!/usr/bin/env python3
"""Golden bytes of bench.py's per-proof paths, computed by the FAITHFUL CPU oracle (oracle/p2oracle.c) alone.

    python tools/gen_golden_path.py [--only NAME]      (build container: 62 GB of RAM, 8 cores; k20 takes ~25 minutes)

For each instance of plonky2_amd/util/synthetic.py:path_instance (per_proof_path_k20 / _k12 / _starky_k22: the inputs
bench.py times) the oracle walks the reference's stages one by one --

  plonk:   wires / constants_sigmas commits (fri/oracle.rs:57-112)  ->  partial products + Zs (plonk/prover.rs:392-449)
           ->  Zs commit  ->  the permutation terms of the quotient on the quotient coset (plonk/prover.rs:609-815,
           vanishing_poly.rs:167-330), coset_ifft, chunks  ->  quotient commit  ->  OpeningSet evaluations
           (plonk/proof.rs:314-345)  ->  prove_openings: alpha, final_poly (fri/oracle.rs:186-220), FRI commit phase
           (fri/prover.rs:84-150), proof of work (:153-202, smallest witness), query rounds (:204-258)
  starky:  trace commit, quotient commit, openings, prove_openings

-- and writes, into tests/golden/path_goldens.json: the caps of the commitments, SHA-256 of the Zs / partial-products matrix,
of the quotient values and of the chunk polynomials, the opening values, alpha, the commit-phase caps, the final
polynomial, the PoW witness, the query indices and SHA-256 of the whole FriProof in the reference's wire format
(serialization/mod.rs:1595-1611, tests/wire_format.py).  tests/test_gpu_fullsize.py and bench.py compare the device
results of the same instance with these bytes at FULL size without running the oracle on the GPU box.
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

from plonky2_amd.util.synthetic import (P, fibonacci_trace, path_instance, second_point,  # noqa: E402
                                        splitmix_columns_numpy)

OUT = os.path.join(ROOT, "tests", "golden", "path_goldens.json")
NAMES = ("per_proof_path_k12", "per_proof_path_starky_k22", "per_proof_path_k20")
# every file whose text decides a record's bytes: the oracle, its constants and binding, this script, the instance definitions and
# the wire-format writer `proof_sha256` hashes.  Each record carries sha256 over them (`source_sha256`);
# tests/test_proof_path.py::test_golden_records_carry_the_current_source_stamp fails in seconds when any of them changed
# without this script having been run again (the full-size records take minutes to recompute, so the CPU tier re-derives only k12).
STAMP_SOURCES = ("oracle/p2oracle.c", "oracle/p2oracle.h", "oracle/p2oracle.py", "oracle/poseidon_constants.h",
                 "tools/gen_golden_path.py", "plonky2_amd/util/synthetic.py", "tests/wire_format.py")


def source_stamp():
    h = hashlib.sha256()
    for rel in STAMP_SOURCES:
        h.update(rel.encode() + b"\0")
        h.update(open(os.path.join(ROOT, rel), "rb").read())
    return h.hexdigest()


def canon(a):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.uint64))
    return np.where(a >= np.uint64(P), a - np.uint64(P), a)


def sha(a):
    return hashlib.sha256(canon(a).astype("<u8").tobytes()).hexdigest()


def ext_mul(a, b):
    return [(a[0] * b[0] + 7 * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P]


def oracle_final_poly(ora, batches, coeff_sets, alpha):
    """fri/oracle.rs:186-213: final_poly = sum over the batches of alpha^(polys before) ... in ReducingFactor order:
    for each batch (point, polys): composition = reduce_polys_base(polys) (util/reducing.rs:83-95), quotient =
    divide_by_linear(composition, point) (division.rs:79-92), final = final * alpha^len(polys) + quotient (shift_poly :103-106)"""
    n = coeff_sets[0].shape[1]
    final = None
    al = [int(alpha[0]), int(alpha[1])]
    for point, polys in batches:
        ps = np.stack([coeff_sets[o][p] for (o, p) in polys])
        quo = ora.divide_by_linear(ora.reduce_polys_base(ps, alpha), point)
        if final is None:
            final = quo
            continue
        sh = [1, 0]
        for _ in range(len(polys)):
            sh = ext_mul(sh, al)
        # final * sh + quo, one extension multiplication per coefficient: the scalar's two words against the vector, by the
        # oracle's own reduce (sum_j alpha^j p_j with p = the planes of `final` is NOT it) -- plain python integers
        f0, f1 = final[:, 0].astype(object), final[:, 1].astype(object)
        r0 = (f0 * sh[0] + 7 * f1 * sh[1] + quo[:, 0].astype(object)) % P
        r1 = (f0 * sh[1] + f1 * sh[0] + quo[:, 1].astype(object)) % P
        final = np.stack([r0.astype(np.uint64), r1.astype(np.uint64)], axis=1)
        del ps
    return np.ascontiguousarray(final)


def oracle_prove_openings(ora, inst, commits, oc, zeta, gz):
    """prove_openings + fri_proof on the oracle's primitives; `commits` = ora.commit dicts (coeffs, leaves, digests, cap)"""
    log_n, rb, cap, arity = inst["log_n"], inst["rate_bits"], inst["cap_height"], inst["arity"]
    n, N = 1 << log_n, 1 << (log_n + rb)
    batches = [(zeta, inst["batch_zeta"]), (gz, inst["batch_next"])]
    alpha = oc.get_extension_challenge()
    fin = oracle_final_poly(ora, batches, [canon(c["coeffs"]) for c in commits], np.array(alpha, dtype=np.uint64))
    pad = np.zeros((N, 2), dtype=np.uint64)
    pad[:n] = fin
    o = ora.fri_commit(pad, rb, cap, arity, oc)
    pow_witness = ora.fri_pow(oc, inst["pow_bits"])
    indices = [int(r) % N for r in oc.get_n_challenges(inst["num_queries"])]
    queries = []
    for x0 in indices:
        x = x0
        initial = [(canon(c["leaves"][x]), ora.merkle_prove(x, N, cap, c["digests"])) for c in commits]
        steps = []
        for i, ab in enumerate(arity):
            nl = o["leaves"][i].shape[0]
            steps.append((canon(o["leaves"][i][x >> ab]).reshape(-1, 2), ora.merkle_prove(x >> ab, nl, cap, o["digests"][i])))
            x >>= ab
        queries.append({"initial_trees_proof": initial, "steps": steps})
    proof = {"commit_phase_merkle_caps": [canon(c) for c in o["caps"]], "query_round_proofs": queries,
             "final_poly": canon(o["final"]), "pow_witness": int(pow_witness)}
    return alpha, proof, indices


def opening_values(ora, commit, z):
    return ora.eval_polys_ext(canon(commit["coeffs"]), np.array(z, dtype=np.uint64))


def proof_record(ora, inst, commits, log=print):
    from tests.wire_format import write_fri_proof
    oc = ora.Challenger()
    oc.observe_elements(np.array(inst["transcript_seed"], dtype=np.uint64))
    zeta = [int(v) for v in oc.get_extension_challenge()]
    gz = second_point(zeta, inst["log_n"])
    t = time.time()
    rec = {"zeta": zeta, "second_point": gz,
           "openings_zeta": [opening_values(ora, c, zeta).tolist() for c in commits]}
    if inst["kind"] == "plonk":
        rec["openings_next"] = [opening_values(ora, commits[2], gz).tolist()]          # the Zs oracle again (proof.rs:332-333)
    else:
        rec["openings_next"] = [opening_values(ora, c, gz).tolist() for c in commits]   # StarkOpeningSet: both at both points
    log("  openings %.1f s" % (time.time() - t))
    t = time.time()
    alpha, proof, indices = oracle_prove_openings(ora, inst, commits, oc, zeta, gz)
    blob = write_fri_proof(proof)
    rec.update({"alpha": [int(a) for a in alpha], "fri_caps": [c.tolist() for c in proof["commit_phase_merkle_caps"]],
                "final_poly": proof["final_poly"].tolist(), "pow_witness": proof["pow_witness"], "query_indices": indices,
                "proof_sha256": hashlib.sha256(blob).hexdigest(), "proof_len": len(blob),
                "transcript_after": [int(x) for x in oc.get_n_challenges(2)]})
    log("  prove_openings %.1f s" % (time.time() - t))
    return rec


def plonk_golden(ora, inst, log=print):
    log_n, rb, cap = inst["log_n"], inst["rate_bits"], inst["cap_height"]
    n = 1 << log_n
    qdf, nc, nr, first = inst["quotient_degree_factor"], len(inst["betas"]), inst["num_routed"], inst["num_constants"]
    k = np.array(inst["k_is"], dtype=np.uint64)
    t = time.time()
    wires = splitmix_columns_numpy(inst["wires_seed"], inst["wires_width"], n)
    cs = splitmix_columns_numpy(inst["cs_seed"], inst["cs_width"], n)
    c_cs = ora.commit(cs, rb, cap, True)
    c_w = ora.commit(wires, rb, cap, True)
    log("  wires + constants_sigmas commits %.1f s" % (time.time() - t))
    t = time.time()
    num_prods = -(-nr // qdf) - 1
    pps = [ora.partial_products(wires[:nr], cs[first:first + nr], k, qdf, inst["betas"][c], inst["gammas"][c]) for c in range(nc)]
    zs = canon(np.stack([pp[num_prods] for pp in pps] + [row for pp in pps for row in pp[:num_prods]]))  # Z first (prover.rs:224-229)
    del pps
    log("  partial products %.1f s" % (time.time() - t))
    t = time.time()
    c_z = ora.commit(zs, rb, cap, True)
    qv = ora.quotient_permutation(c_w["leaves"], c_cs["leaves"], first, c_z["leaves"], log_n, rb, k, qdf,
                                  inst["betas"], inst["gammas"], inst["alphas"])
    qv = canon(qv)
    chunks = canon(np.stack([ora.coset_ifft(qv[a].copy())[:qdf * n] for a in range(nc)]).reshape(nc * qdf, n))
    log("  Zs commit + quotient %.1f s" % (time.time() - t))
    t = time.time()
    c_q = ora.commit(chunks, rb, cap, False)
    log("  quotient commit %.1f s" % (time.time() - t))
    commits = [c_cs, c_w, c_z, c_q]
    rec = {"params": {kk: (list(v) if isinstance(v, tuple) else v) for kk, v in inst.items() if kk not in ("k_is", "batch_zeta", "batch_next")},
           "caps": [canon(c["cap"]).tolist() for c in commits],
           "zs_sha256": sha(zs), "quotient_values_sha256": sha(qv), "chunks_sha256": sha(chunks),
           "zs_first_row": zs[:, 0].tolist(), "quotient_values_first": qv[:, :4].tolist()}
    rec.update(proof_record(ora, inst, commits, log))
    return rec


def starky_golden(ora, inst, log=print):
    log_n, rb, cap = inst["log_n"], inst["rate_bits"], inst["cap_height"]
    n = 1 << log_n
    t = time.time()
    c_t = ora.commit(fibonacci_trace(log_n), rb, cap, True)
    c_q = ora.commit(splitmix_columns_numpy(inst["quotient_seed"], 2, n), rb, cap, False)
    log("  trace + quotient commits %.1f s" % (time.time() - t))
    commits = [c_t, c_q]
    rec = {"params": {kk: (list(v) if isinstance(v, tuple) else v) for kk, v in inst.items() if kk not in ("batch_zeta", "batch_next")},
           "caps": [canon(c["cap"]).tolist() for c in commits]}
    rec.update(proof_record(ora, inst, commits, log))
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--out", default=OUT)
    args = ap.parse_args()
    from oracle import p2oracle as ora
    ora.set_num_threads(ora.usable_cores())
    data = json.load(open(args.out)) if os.path.exists(args.out) else {}
    for name in NAMES:
        if args.only and name != args.only:
            continue
        inst = path_instance(name)
        print(name, flush=True)
        t = time.time()
        data[name] = (plonk_golden if inst["kind"] == "plonk" else starky_golden)(ora, inst, log=lambda s: print(s, flush=True))
        data[name]["source_sha256"] = source_stamp()
        print("  total %.1f s" % (time.time() - t), flush=True)
        with open(args.out, "w") as f:
            json.dump(data, f, indent=0, separators=(",", ":"))
            f.write("\n")


if __name__ == "__main__":
    main()
