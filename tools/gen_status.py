#!/usr/bin/env python3
"""STATUS.md: one page, SURVEY section-8 row -> the tests that pin it -> the number measured for it, generated from a full
bench.py JSON line (profiles/rNN_*_bench_full.json, the builder's run on an MI355X) and the driver's records when present
(BENCH_rNN.json, GPUTEST_rNN.json).  Nothing is typed by hand except the row -> test mapping below.

    python tools/gen_status.py [--bench profiles/r05_c_bench_full.json]
"""
import argparse
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ROWS = [
    ("a1", "Goldilocks add / sub / mul / reduce128", "test_parity.py::test_field_ops_edge_grid", None),
    ("a2-a5", "fft_root_table, fft_classic (+ zero tail), ifft, coset LDE", "test_parity.py::test_fft_ifft_vs_oracle, ::test_coset_ifft_vs_oracle, ::test_coset_lde_vs_oracle_and_naive", "ntt"),
    ("a6", "PolynomialBatch::lde_values (+ salts)", "test_parity.py::test_salted_commit_vs_oracle", None),
    ("a7-a8", "transpose, reverse_index_bits", "test_parity.py::test_transpose, ::test_reverse_index_bits_reference_table; test_async_leaves.py (natural-order host copy)", "bitrev"),
    ("a9-a12", "Poseidon permutation, sponge, hash_or_noop, two_to_one", "test_parity.py::test_poseidon_reference_kats (the 4 reference KATs), ::test_poseidon_random_vs_oracle, ::test_poseidon_edge_states_vs_oracle, ::test_hash_no_pad_and_two_to_one", "hash"),
    ("a13", "MerkleTree::new, reference digest layout", "test_gpu_fullsize.py::test_baseline_commits_bit_exact_vs_oracle_goldens (SHA-256 of the digest arrays at C2 / C3 / C4 / C5)", "merkle"),
    ("a14", "from_values / from_coeffs", "test_gpu_fullsize.py::test_baseline_commits_bit_exact_vs_oracle_goldens; bench.py: the cap of EVERY timed step", "headline"),
    ("a15", "fri_committed_trees", "test_gpu_fullsize.py::test_fri_commit_phase_full_size_vs_oracle (N = 2^23)", "fri"),
    ("a16", "Challenger", "test_parity.py::test_challenger_vs_oracle", None),
    ("b", "drop-in boundary (C ABI + Rust shim + patch)", "test_abi.py, test_integration_files.py (API linter, `git apply --check`, first_contact.sh --dry-run), test_async_leaves.py", "host"),
    ("c", "oracle", "test_oracle.py (KATs, reference properties, restated verifier), test_fast_oracle.py, test_proof_path.py::test_k12_golden_record_is_what_the_oracle_computes_now, ::test_golden_records_carry_the_current_source_stamp", None),
    ("d", "measurement", "bench.py: roofline (measured in the run), cpu_baseline, checked per-proof paths; profiles/", "roofline"),
    ("e", "multi-GPU", "test_distributed.py (gloo 2/4/8), test_emu_rccl_ranks.py, test_emu_devices.py, test_bench_launch.py (`bench.py --gpus 2` self-launched, incl. the per-proof path over the group), test_gpu_fullsize.py (C5 as 8 ranks on one GPU; the k20 proof over a group of 2 / 8 ranks; RCCL bound with and without torch)", "multi"),
    ("f1", "prove_openings prelude", "test_prove_openings.py::test_final_poly_and_prove_openings_vs_oracle; test_gpu_fullsize.py::test_proof_path_full_size_vs_oracle_records_and_verifier", "path"),
    ("f2", "PoW + query phase", "test_prove_openings.py::test_fri_proof_passes_the_reference_verifier; full size: the k20 / starky k22 FriProof bytes = the oracle's, verified + 6 tamperings", "path"),
    ("f3", "partial products / Z, quotient permutation terms, OpeningSet", "test_permutation.py; full size: SHA-256 of the Zs matrix, the 2^23 quotient values, the 16 chunks, all 275 openings vs the oracle record; gate sums on 4096 points + linearity", "path"),
    ("f4", "wire formats", "tests/wire_format.py checker; the FriProof bytes of the full-size proofs hash to the oracle's; test_gpu_fullsize.py::test_polynomial_batch_wire_bytes_vs_oracle_at_full_size (SHA-256 of the whole write_polynomial_batch stream, C2 wires + C3 quotient chunks); Rust serializer: never compiled", None),
    ("g1", "patched CircuitBuilder::build / prove vs the Rust prover", "integration/first_contact.sh (dry-run only: no cargo in the image)", None),
]


def newest(pattern):
    files = sorted(glob.glob(os.path.join(ROOT, pattern)))
    return files[-1] if files else None


def load_line(path):
    txt = open(path).read().strip()
    try:
        return json.loads(txt)
    except Exception:
        return json.loads(txt.splitlines()[-1])


def numbers(d):
    oc = d.get("other_configs", {})
    k = d.get("kernels", {})
    ms = lambda name: k.get(name, {}).get("ms_per_launch")  # noqa: E731
    rn = d.get("roofline_ntt") or {}
    r = d.get("roofline") or {}
    out = {}
    out["headline"] = "C3 wires commit %.2f ms = %.2f GFE/s, %d of %d caps = golden" % (d["ms_per_step"], d["value"], d.get("caps_checked", 0), d["steps"])
    if rn:
        out["ntt"] = "family (4 passes%s) %.2f ms per step = %.2f of 8 TB/s by SURVEY 8(d)'s bytes%s (LDE contiguous %.2f ms, strided %.2f ms)" % (
            " + bit reversal" if "bitrev_permute" in rn["passes"] else "", rn["family_ms_per_step"], rn["family_frac"],
            ", traffic %.2f x algorithmic" % rn["traffic_ratio"] if rn.get("traffic_ratio") else "",
            rn["passes"]["ntt_lde_contig"]["ms"], rn["passes"]["ntt_lde_strided"]["ms"])
    if ms("bitrev_permute"):
        out["bitrev"] = "bit reversal %.2f ms" % ms("bitrev_permute")
    if ms("hash_leaves"):
        out["hash"] = "leaf sponge %.2f ms per 1.43e8 permutations" % ms("hash_leaves")
    if ms("merkle_levels"):
        out["merkle"] = "tree levels %.2f ms" % ms("merkle_levels")
    if "c3_fri_commit_phase" in oc:
        out["fri"] = "C3 / C4 FRI commit phase %.2f / %.2f ms" % (oc["c3_fri_commit_phase"]["ms"], oc["c4_fri_commit_phase"]["ms"])
    if r:
        emp = (r.get("valu_empirical") or {}).get("frac_raw")
        out["roofline"] = "leaf sponge: VALU nominal %.3f, VALU empirical %s, HBM %.4f (all from the run's own kernel time)" % (
            r["frac"], "%.3f" % emp if emp else "n/a", r["hbm_frac"])
    paths = [(n, oc[n]) for n in ("per_proof_path_k20", "per_proof_path_k12", "per_proof_path_starky_k22") if n in oc]
    if paths:
        out["path"] = "; ".join("%s %.2f ms checked=%s" % (n.replace("per_proof_path_", ""), v["ms"], v.get("checked")) for n, v in paths)
    if "host_c3_wires_leaves_async" in oc:
        a = oc["host_c3_wires_leaves_async"]
        out["host"] = "p2hot_commit C3 wires over PCIe: %.0f ms without leaves; with the 9.1 GB leaf matrix: synchronous %.0f ms, asynchronous call %.0f ms / last row %.0f ms" % (
            oc["host_c3_wires_coeffs_digests"]["ms"], oc["host_c3_wires_leaves_back_pinned"]["ms"], a["ms"], a["ms_last_row"])
    out["multi"] = "no N > 1 measurement exists (one-GPU boxes); C5's shape on one GPU: %s" % (
        "%.0f ms" % oc["c5_wires"]["ms"] if "c5_wires" in oc else "n/a")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bench", default=None)
    a = ap.parse_args()
    bench = a.bench or newest("profiles/r*_bench_full.json")
    d = load_line(bench)
    nums = numbers(d)
    lines = ["# STATUS -- SURVEY section-8 row -> tests -> measured number", "",
             "Generated by `tools/gen_status.py` from `%s`%s.  Tests live under `tests/`; numbers are one MI355X, inputs resident in HBM unless the line says PCIe." % (
                 os.path.relpath(bench, ROOT), ""), ""]
    gt = newest("GPUTEST_r*.json")
    if gt:
        g = json.load(open(gt))
        lines += ["Driver record `%s` (head `%s`): `pytest -m gpu` %s passed, rc %s; smoke rc %s; native libraries loaded: %s." % (
            os.path.basename(gt), g.get("head"), g.get("gpu_tests_passed"), g.get("gpu_tests_rc"), g.get("smoke_rc"), ", ".join(g.get("native_so_loaded", []))), ""]
    bn = newest("BENCH_r*.json")
    if bn:
        b = json.load(open(bn))
        p = b.get("parsed", {})
        lines += ["Driver record `%s` (head `%s`): %.2f %s, %.2f ms per step, driver wall %.1f s." % (
            os.path.basename(bn), b.get("head"), p.get("value", float("nan")), p.get("unit", ""), p.get("ms_per_step", float("nan")), b.get("driver_run_s", float("nan"))), ""]
    lines += ["| row | what | pinned by | measured |", "|---|---|---|---|"]
    for row, what, tests, key in ROWS:
        lines.append("| %s | %s | %s | %s |" % (row, what, tests, nums.get(key, "--") if key else "--"))
    cb = d.get("cpu_baseline")
    if cb:
        lines += ["", "CPU baseline in the same run (`cpu_baseline`, kind `%s`, %d cores): %.3f GFE/s; the Rust / rayon prover has never run (no cargo)." % (
            cb["kind"], cb["cores"], cb["value"])]
    open(os.path.join(ROOT, "STATUS.md"), "w").write("\n".join(lines) + "\n")
    print("wrote STATUS.md from", bench)


if __name__ == "__main__":
    main()
