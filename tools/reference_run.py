#!/usr/bin/env python3
"""The oracle's side of the parity protocol of SURVEY 8c (test tooling, CPU only).

`examples/p2hot_dump_goldens.rs` (integration/plonky2_p2hot.patch) makes the REFERENCE prover write its commitments of
the synthetic inputs into a JSON file; this module recomputes the same records with the CPU oracle (oracle/p2oracle.c)
and compares.  It is what `tests/test_oracle.py::test_reference_run_*` call:

    python tools/reference_run.py check tests/golden/reference_run.json     # compare a dump with the oracle
    python tools/reference_run.py emit  /tmp/oracle_run.json [names...]      # write the ORACLE's records in the same schema

A record whose name is in tests/golden/commit_caps.json is compared with that golden entry directly (those were produced
by the same oracle on the same inputs, hours of CPU time for the big ones); others are recomputed if they are small."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from plonky2_amd.util.synthetic import fibonacci_trace, splitmix_columns_numpy  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden", "commit_caps.json")
COMMIT_FIELDS = ("W", "log_n", "rate_bits", "cap_height", "is_values", "input", "cap", "sha256_coeffs", "sha256_digests", "sha256_lde")
# the dumper's table (name -> W, log_n, rate_bits, cap_height, is_values, input); tests check it against the Rust source
DUMP_SHAPES = {
    "small_values": (7, 5, 3, 4, True, "splitmix"),
    "small_coeffs": (3, 8, 1, 0, False, "splitmix"),
    "small_wide": (135, 12, 3, 4, True, "splitmix"),
    "small_constants_sigmas": (84, 12, 3, 4, True, "splitmix"),
    "small_fibonacci": (2, 10, 1, 4, True, "fibonacci"),
    "c2_wires": (135, 16, 3, 4, True, "splitmix"),
    "c3_wires": (135, 20, 3, 4, True, "splitmix"),
    "c3_constants_sigmas": (84, 20, 3, 4, True, "splitmix"),
    "c3_zs_partial_products": (20, 20, 3, 4, True, "splitmix"),
    "c3_quotient_chunks": (16, 20, 3, 4, False, "splitmix"),
    "c4_fibonacci_trace": (2, 22, 1, 4, True, "fibonacci"),
}
DUMP_FRI = {  # name -> log_n, rate_bits, cap_height, arity_bits, proof_of_work_bits
    "fri_small": (8, 3, 2, [2, 1], 4),
    "fri_plonky2_like": (12, 3, 4, [4, 4], 8),
    "fri_starky_like": (10, 1, 3, [1, 2, 1], 6),
}
RECOMPUTE_LIMIT = 1 << 25  # W * N field elements the faithful oracle recomputes on the fly (a few seconds)


def oracle_commit_record(W, log_n, rb, cap, is_values, kind):
    from oracle import p2oracle as ora
    cols = fibonacci_trace(log_n) if kind == "fibonacci" else splitmix_columns_numpy(0, W, 1 << log_n)
    r = ora.commit(cols, rb, cap, is_values)
    sl = hashlib.sha256()
    for c in range(W):
        sl.update(np.ascontiguousarray(r["leaves"][:, c]).tobytes())
    return {"W": W, "log_n": log_n, "rate_bits": rb, "cap_height": cap, "is_values": is_values, "input": kind,
            "cap": r["cap"].tolist(), "sha256_coeffs": hashlib.sha256(r["coeffs"].tobytes()).hexdigest(),
            "sha256_digests": hashlib.sha256(r["digests"].tobytes()).hexdigest(), "sha256_lde": sl.hexdigest()}


def oracle_fri_record(log_n, rb, cap, arity, pow_bits):
    """fri_proof with no initial trees (fri/prover.rs:24-70): commit phase, then the grind (smallest witness)"""
    from oracle import p2oracle as ora
    n = 1 << log_n
    co = np.zeros((n << rb, 2), dtype=np.uint64)
    co[:n, 0] = splitmix_columns_numpy(0, 1, n)[0]
    co[:n, 1] = splitmix_columns_numpy(1, 1, n)[0]
    ch = ora.Challenger()
    ch.observe_elements(splitmix_columns_numpy(2, 1, 8)[0])
    r = ora.fri_commit(co, rb, cap, arity, ch)
    w = ora.fri_pow(ch, pow_bits)
    return {"log_n": log_n, "rate_bits": rb, "cap_height": cap, "arity_bits": list(arity), "proof_of_work_bits": pow_bits,
            "commit_phase_merkle_caps": [c.tolist() for c in r["caps"]], "final_poly": r["final"].tolist(), "pow_witness": int(w),
            "pow_witness_is_smallest": True}


def emit(path, names=None):
    out = {"source": "oracle/p2oracle.c through tools/reference_run.py (the schema examples/p2hot_dump_goldens.rs writes)"}
    for name, sh in DUMP_SHAPES.items():
        if (names and name not in names) or (not names and sh[0] << (sh[1] + sh[2]) > RECOMPUTE_LIMIT):
            continue
        out[name] = oracle_commit_record(*sh)
    for name, sh in DUMP_FRI.items():
        if names and name not in names:
            continue
        out[name] = oracle_fri_record(*sh)
    json.dump(out, open(path, "w"), indent=1)
    return out


def check(path):
    """-> list of (name, what was compared); raises AssertionError naming the first field that differs"""
    run = json.load(open(path))
    golden = json.load(open(GOLDEN))
    done = []
    for name, rec in run.items():
        if name == "source":
            continue
        if "sha256_coeffs" in rec:
            shape = (rec["W"], rec["log_n"], rec["rate_bits"], rec["cap_height"], bool(rec["is_values"]), rec["input"])
            if name in DUMP_SHAPES:
                assert shape == DUMP_SHAPES[name], "%s: the dump's shape %s is not the dumper's table entry %s" % (name, shape, DUMP_SHAPES[name])
            if name in golden:
                exp, how = golden[name], "tests/golden/commit_caps.json"
            elif rec["W"] << (rec["log_n"] + rec["rate_bits"]) <= RECOMPUTE_LIMIT:
                exp, how = oracle_commit_record(*shape), "oracle recomputed"
            else:
                done.append((name, "skipped: too large to recompute and not among the goldens"))
                continue
            for f in COMMIT_FIELDS:
                assert rec[f] == exp[f], "%s.%s: the reference run has %r, the oracle %r" % (name, f, str(rec[f])[:80], str(exp[f])[:80])
            done.append((name, how))
        elif "commit_phase_merkle_caps" in rec:
            exp = oracle_fri_record(rec["log_n"], rec["rate_bits"], rec["cap_height"], rec["arity_bits"], rec["proof_of_work_bits"])
            for f in ("commit_phase_merkle_caps", "final_poly"):
                assert rec[f] == exp[f], "%s.%s differs between the reference run and the oracle" % (name, f)
            if rec.get("pow_witness_is_smallest"):
                assert rec["pow_witness"] == exp["pow_witness"], "%s.pow_witness: %r vs the oracle's smallest %r" % (name, rec["pow_witness"], exp["pow_witness"])
            done.append((name, "oracle recomputed (FRI commit phase%s)" % (" + grind" if rec.get("pow_witness_is_smallest") else "")))
        else:
            raise AssertionError("%s: unknown record kind" % name)
    return done


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "emit":
        emit(sys.argv[2], sys.argv[3:] or None)
        print("wrote", sys.argv[2])
    elif len(sys.argv) == 3 and sys.argv[1] == "check":
        for name, how in check(sys.argv[2]):
            print("ok  %-28s %s" % (name, how))
    else:
        sys.exit(__doc__)
