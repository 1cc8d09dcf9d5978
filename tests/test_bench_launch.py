"""`python bench.py --gpus N` as the driver may run it: no launcher, no WORLD_SIZE / RANK / MASTER_* in the environment.
bench.py starts its own ranks (torch.distributed.run on 127.0.0.1), forwards rank 0's JSON line and the exit code.  Here the
ranks sit on the kernel-source emulator and its fake RCCL (P2HOT_BENCH_EMU=1: the test tier's switch, refused nowhere else),
so the whole front door -- launcher, rank plumbing, in-library RCCL communicator, preflight, sharded commit, the cap check
against the oracle's golden of the 2-rank shape, the strong-scaling companion, the JSON contract -- runs in the CPU tier."""
import json
import os
import subprocess
import sys

import pytest

from tests.conftest import ROOT


def _bench(*args, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                             "P2HOT_EXCHANGE", "P2HOT_TRANSPORT")}
    env["P2HOT_BENCH_EMU"] = "1"
    env["P2HOT_BENCH_NO_GROUP_PATH"] = "1"   # (the per-proof path over a group: only the test below pays for it)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), env=env, capture_output=True, text=True,
                          timeout=timeout, cwd="/tmp")


def _one_json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_bench_gpus_2_launches_its_own_ranks():
    r = _bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--log-n", "6")
    assert r.returncode == 0, r.stderr[-2000:]
    d = _one_json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["emulated"] is True
    assert d["metric"] == "LDE+Poseidon-commit GFE/s" and d["unit"] == "GFE/s" and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["cap_checked"] is True and d["caps_checked"] == 2          # the 2-rank weak shape: 2^7 rows, golden "tiny_wires"
    assert [row["rank"] for row in d["ranks"]] == [0, 1]
    assert all(row["transport"] == "rccl" and row["preflight"]["selftest"] == "ok" for row in d["ranks"])
    assert "RCCL inside libp2hot" in d["config"]["transport"] and d["config"]["exchange"] in ("allgather", "broadcast")
    assert "2^7 rows" in d["config"]["workload"]
    s = d["strong_scaling"]                                             # the same total size split over the ranks: 2^6 rows
    assert s["scaling"] == "strong" and s["cap_checked"] is True and s["caps_checked"] == 2 and s["value"] > 0
    for key in ("value", "ms_per_step", "roofline", "kernels", "algorithmic_bytes_per_step"):
        assert key in d


def test_bench_gpus_2_adds_the_per_proof_path_over_the_group():
    """behind world > 1: one `group_per_proof_path` record -- the per-proof path as ONE process driving both (emulated) devices
    (p2hot_group: four sharded commits from host columns, p2hot_group_eval_openings, p2hot_group_prove_openings), on the k12
    instance here, checked against the same oracle record the single-context path is (tests/golden/path_goldens.json)"""
    r = _bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--log-n", "6", timeout=1500,
               env_extra={"P2HOT_BENCH_NO_GROUP_PATH": "0", "P2HOT_BENCH_GROUP_PATH": "per_proof_path_k12", "P2HOT_BENCH_NO_STRONG": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    d = _one_json_line(r.stdout)
    gp = d["group_per_proof_path"]
    assert gp.get("skipped") is None, gp
    assert gp["checked"] is True and gp["single_gpu_checked"] is True and gp["differs"] is None
    assert "per_proof_path_k12 over a p2hot_group of 2 devices" in gp["workload"] and gp["uses_rccl"] is True and gp["ms"] > 0
    assert any(k.startswith("sharded commit: wires") for k in gp["stage_ms"]) and "prove_openings over the group" in gp["stage_ms"]
    assert d["cap_checked"] is True


def test_bench_gpus_8_launches_eight_ranks():
    """the driver's `python bench.py --gpus 8` on the day an 8-GPU node exists: eight ranks on eight emulated devices, the same
    2^7-row shape as eight LDE cosets (one per rank: C5's arrangement), the cap of every step = the oracle's golden"""
    r = _bench("--gpus", "8", "--steps", "1", "--warmup", "1", "--log-n", "4", timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _one_json_line(r.stdout)
    assert d["n_gpus"] == 8 and d["cap_checked"] is True and [row["rank"] for row in d["ranks"]] == list(range(8))
    assert all(row["transport"] == "rccl" and row["preflight"]["selftest"] == "ok" for row in d["ranks"])
    assert len({row["device"] for row in d["ranks"]}) == 8
    assert d["strong_scaling"]["value"] > 0 and "split over 8 ranks" in d["strong_scaling"]["workload"]


def test_a_failed_rccl_preflight_falls_back_to_the_launchers_communicator():
    """first contact with a real multi-GPU node happens once, in the driver's run: when the in-library RCCL communicator fails its
    preflight (here: rank 1 never reaches the collective, the fake RCCL reports the lonely rank where the real one would hang),
    every rank agrees on it through the launcher's process group, the exchange moves to that group's transport, and the line
    says so instead of the run ending without a record"""
    r = _bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--log-n", "6",
               env_extra={"P2HOT_BENCH_EMU_FAIL_PREFLIGHT": "1", "P2HOT_EMU_RCCL_TIMEOUT_MS": "1500"})
    assert r.returncode == 0, r.stderr[-2000:]
    d = _one_json_line(r.stdout)
    assert d["cap_checked"] is True and d["n_gpus"] == 2
    assert all(row["transport"] == "gloo" and "failed its preflight" in row["preflight"]["fallback"] and row["preflight"]["selftest"] == "ok" for row in d["ranks"])
    # with the fallback switched off the same failure ends the run, by name
    r = _bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--log-n", "6",
               env_extra={"P2HOT_BENCH_EMU_FAIL_PREFLIGHT": "1", "P2HOT_EMU_RCCL_TIMEOUT_MS": "1500", "P2HOT_BENCH_NO_FALLBACK": "1"})
    assert r.returncode != 0 and "bench preflight" in (r.stderr + r.stdout)


def test_bench_gpus_1_needs_no_launcher_either():
    r = _bench("--gpus", "1", "--steps", "1", "--warmup", "0", "--log-n", "7", "--no-extra", "--no-cpu-baseline")
    assert r.returncode == 0, r.stderr[-2000:]
    d = _one_json_line(r.stdout)
    assert d["n_gpus"] == 1 and d["cap_checked"] is True and "ranks" not in d and "strong_scaling" not in d


def test_a_launcher_and_gpus_that_disagree_is_an_error_not_a_hang():
    r = _bench("--gpus", "2", "--steps", "1", "--log-n", "6", env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "must agree" in (r.stderr + r.stdout)


def test_a_wrong_cap_fails_the_bench(tmp_path):
    """what is timed is checked: with the golden of the shape replaced by another cap the run exits non-zero"""
    import shutil
    gold = os.path.join(ROOT, "tests", "golden", "commit_caps.json")
    d = json.load(open(gold))
    d["tiny_wires"]["cap"][3][1] += 1
    alt = tmp_path / "caps.json"
    alt.write_text(json.dumps(d))
    r = _bench("--gpus", "1", "--steps", "1", "--warmup", "0", "--log-n", "7", "--no-extra", "--no-cpu-baseline",
               env_extra={"P2HOT_BENCH_GOLDEN_CAPS": str(alt)})
    assert r.returncode != 0 and "differs from the oracle's golden cap" in (r.stderr + r.stdout)
    shutil.rmtree(tmp_path, ignore_errors=True)
