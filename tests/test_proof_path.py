"""The per-proof paths bench.py times (per_proof_path_* lines), stage by stage against the CPU oracle's bytes for the same
synthetic instance (tests/golden/path_goldens.json, tools/gen_golden_path.py), and the device-made FriProof under the restated
reference verifier (oracle/fri_verifier.py: fri/verifier.rs:62, fri/challenges.rs:28-88).

k = 12 runs on both tiers (and re-derives its golden record from the oracle on the spot, so the committed file cannot drift from
oracle/p2oracle.c); the full-size instances -- k = 20 (2^20 rows, 2^23 LDE points: the headline size) and starky k = 22 -- run
on the MI355X only, against the committed records."""
import copy

import numpy as np
import pytest

from tests.conftest import P


def _verify(ora, inst, res, tamper=None):
    """verify_fri_proof over the path's own FRI instance; the verifier re-derives alpha, the betas, the PoW response and the
    query indices from the proof with the ORACLE's challenger"""
    from oracle import fri_verifier as fv
    oc = ora.Challenger()
    oc.observe_elements(np.asarray(inst["transcript_seed"], dtype=np.uint64))
    zeta = [int(v) for v in oc.get_extension_challenge()]
    assert zeta == res["zeta"]
    batches = [(zeta, inst["batch_zeta"]), (res["second_point"], inst["batch_next"])]
    at_zeta = {(oi, pi): res["openings_zeta"][oi][pi] for (oi, pi) in inst["batch_zeta"]}
    if inst["kind"] == "plonk":
        at_next = {(2, pi): res["openings_next"][0][pi] for (_, pi) in inst["batch_next"]}
    else:
        at_next = {(oi, pi): res["openings_next"][oi][pi] for (oi, pi) in inst["batch_next"]}
    openings = [[np.array(at_zeta[k], dtype=np.uint64) for k in inst["batch_zeta"]], [np.array(at_next[k], dtype=np.uint64) for k in inst["batch_next"]]]
    pf = copy.deepcopy(res["proof"])
    if tamper:
        tamper(pf, openings)
    chal = fv.fri_challenges(oc, pf["commit_phase_merkle_caps"], pf["final_poly"], pf["pow_witness"], inst["log_n"], inst["rate_bits"],
                             inst["cap_height"], inst["num_queries"])
    fv.verify_fri_proof(batches, openings, chal, res["caps"], pf, inst["log_n"], inst["rate_bits"], inst["arity"], inst["pow_bits"],
                        inst["num_queries"])
    return oc


def _tamperings():
    def final_poly(pf, op):
        pf["final_poly"] = np.array(pf["final_poly"], dtype=np.uint64)
        pf["final_poly"][0][0] = (int(pf["final_poly"][0][0]) + 1) % P

    def opening(pf, op):
        op[0][-2][1] = (int(op[0][-2][1]) + 1) % P

    def next_opening(pf, op):
        op[1][-1][0] = (int(op[1][-1][0]) + 1) % P

    def leaf(pf, op):
        q = pf["query_round_proofs"][3]
        lf, sib = q["initial_trees_proof"][1]
        lf = np.array(lf, dtype=np.uint64)
        lf[-1] = (int(lf[-1]) + 1) % P
        q["initial_trees_proof"][1] = (lf, sib)

    def step(pf, op):
        q = pf["query_round_proofs"][-1]
        ev, sib = q["steps"][-1]
        ev = np.array(ev, dtype=np.uint64)
        ev[3][1] = (int(ev[3][1]) + 1) % P
        q["steps"][-1] = (ev, sib)

    def pow_witness(pf, op):
        pf["pow_witness"] = int(pf["pow_witness"]) + 1
    return [final_poly, opening, next_opening, leaf, step, pow_witness]


def check_path(eng, ora, name, sync=None):
    """shared by the CPU tier (k12 on the emulator) and tests/test_gpu_fullsize.py: golden comparison, verification, tampering"""
    from oracle import fri_verifier as fv
    from plonky2_amd.util import proof_path as pp
    from plonky2_amd.util.synthetic import path_instance
    inst = path_instance(name)
    g = pp.golden(name)
    assert g is not None, "no oracle record for %s: run tools/gen_golden_path.py" % name
    inp = pp.PathInputs(eng, inst)
    res = pp.run_path(eng, inp, sync=sync, keep=True)
    assert pp.compare_with_golden(res, g) == []
    # the wire bytes the comparison hashed are the reference layout (tests/wire_format.py restates the serializer independently)
    from tests.wire_format import write_fri_proof
    assert write_fri_proof(res["proof"]) == pp.serialize_fri_proof(res["proof"])
    oc = _verify(ora, inst, res)
    assert [int(x) for x in oc.get_n_challenges(2)] == res["transcript_after"]     # prover and verifier transcripts agree to the end
    for t in _tamperings():
        with pytest.raises(fv.VerificationError):
            _verify(ora, inst, res, tamper=t)
    return inst, inp, res, g


def test_k12_golden_record_is_what_the_oracle_computes_now(ora):
    """the committed record of the cheap instance, recomputed: a change of oracle/p2oracle.c or of the instance definition that
    is not followed by tools/gen_golden_path.py fails here, in the CPU tier"""
    import json
    from plonky2_amd.util import proof_path as pp
    from plonky2_amd.util.synthetic import path_instance
    from tools.gen_golden_path import plonk_golden
    now = plonk_golden(ora, path_instance("per_proof_path_k12"), log=lambda s: None)
    rec = dict(pp.golden("per_proof_path_k12"))
    rec.pop("source_sha256", None)
    assert json.loads(json.dumps(now)) == rec


def test_golden_records_carry_the_current_source_stamp():
    """the full-size records (k20: ~15 minutes of oracle time) are trusted files; each carries sha256 over every source that
    decides its bytes (tools/gen_golden_path.py STAMP_SOURCES).  An edit of the oracle, the instance or the wire writer that is
    not followed by tools/gen_golden_path.py fails here without recomputing anything."""
    from plonky2_amd.util import proof_path as pp
    from tools.gen_golden_path import NAMES, source_stamp
    now = source_stamp()
    for name in NAMES:
        assert pp.golden(name).get("source_sha256") == now, "%s is stale: run tools/gen_golden_path.py" % name


def test_second_point_is_g_times_zeta():
    """plonk/circuit_data.rs:537-539: zeta_next = primitive_root_of_unity(degree_bits) * zeta (types.rs:268-272)"""
    from plonky2_amd.util.synthetic import P, primitive_root_of_unity, second_point
    for k in (12, 20, 22):
        g = primitive_root_of_unity(k)
        assert pow(g, 1 << k, P) == 1 and pow(g, 1 << (k - 1), P) == P - 1
        assert second_point([5, 9], k) == [5 * g % P, 9 * g % P]


def test_proof_path_k12_vs_oracle_record_and_verifier(eng, ora):
    check_path(eng, ora, "per_proof_path_k12")


def test_group_path_k12_vs_oracle_record(emu, ora):
    """The per-proof path over a p2hot_group of 2 ranks (both on the emulated device 0: the exchanges are copies) on the SAME oracle
    record the single-context path is checked against: four coset-sharded commits from host columns, p2hot_group_eval_openings,
    p2hot_group_prove_openings (plonky2_amd/util/proof_path.py run_group_path) -- caps, every opening value, alpha's consequences
    (FRI caps, final_poly), PoW witness, query indices, SHA-256 of the FriProof bytes and the transcript afterwards.  This is the
    line `bench.py --gpus N` adds under "group_per_proof_path" (tests/test_bench_launch.py parses it)."""
    from plonky2_amd.distributed import GroupCommit
    from plonky2_amd.util import proof_path as pp
    from plonky2_amd.util.synthetic import path_instance, splitmix_columns_numpy
    from tests.emu_backend import emu_lib
    name = "per_proof_path_k12"
    inst, g = path_instance(name), pp.golden(name)
    single = pp.run_path(emu, pp.PathInputs(emu, inst), keep=True)      # the Zs matrix and the quotient chunks are single-GPU stages
    assert pp.compare_with_golden(single, g) == []
    n = 1 << inst["log_n"]
    group = GroupCommit(emu_lib(), 2, [0, 0])
    try:
        res = pp.run_group_path(group, inst, splitmix_columns_numpy(inst["cs_seed"], inst["cs_width"], n),
                                splitmix_columns_numpy(inst["wires_seed"], inst["wires_width"], n), single["zs"], single["chunks"],
                                pipeline_chunks=2)
    finally:
        group.close()
    assert pp.compare_with_golden(res, g) == []
    assert pp.serialize_fri_proof(res["proof"]) == pp.serialize_fri_proof(single["proof"])
