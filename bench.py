#!/usr/bin/env python3
"""bench.py -- LDE + Poseidon-commit throughput of the HIP hot path (BASELINE.json's metric).

A "step" is one PolynomialBatch::from_values (iNTT -> coset LDE -> Poseidon leaf hash -> Merkle
tree) over one synthetic batch of trace columns, inputs already resident in HBM:
  --gpus 1 : W = 135 wire columns, 2^20 rows, rate 1/8 (N = 2^23), cap_height 4 -- the wires commit
             of the 2^20-gate standard_recursion_config circuit the metric is quoted on (config C3).
  --gpus G : the same per-GPU work, weak scaling: 2^(20 + log2 G) rows, LDE cosets sharded over the
             ranks (plonky2_amd.distributed), iNTT column-sharded + RCCL all-gather of coefficients,
             RCCL all-gather of digests/cap.
value = W * N_total / t / 1e9 (GFE/s, whole job).  One JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--log-n 20] [--width 135] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

P = 0xFFFFFFFF00000001
from plonky2_amd.util.chip import (CO_ISSUED_MOVES_PER_SBOX_PRODUCT, HBM_PEAK_GBS, NUM_SIMDS, POSEIDON_SBOX_PRODUCTS_PER_PERMUTATION,  # noqa: E402
                                   VALU_NOMINAL_GWAVE_INST_PER_S, box_report)  # (constants cite the guide)


from plonky2_amd.util.synthetic import splitmix_columns_numpy, splitmix_columns_torch  # noqa: E402,F401


def algorithmic_bytes(W, log_n, rate_bits, is_values=True):
    """each logical array touched once per logical stage, 8-byte elements."""
    n, N = 1 << log_n, 1 << (log_n + rate_bits)
    b = {
        "intt": 16 * W * n if is_values else 0,
        "lde": 8 * W * n + 8 * W * N,
        "hash_leaves": 8 * W * N + 32 * N,
        "merkle_levels": 96 * N,
    }
    b["total"] = sum(b.values())
    return b


def _pmc_file():
    """profiles/pmc_traffic.json (tools/profile_all.sh + tools/prof_summarize.py): one entry per kernel instantiation, stamped with
    the hash of plonky2_amd/csrc it was collected on.  Counters cannot be read inside this run."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except Exception:
        return None


def pmc_entry(W, log_n, rb, cap, world, *needles):
    """the entry of the kernel instantiation whose key contains every needle (same workload only), else None"""
    d = _pmc_file()
    if not d:
        return None
    w = d.get("workload", {})
    if (w.get("W"), w.get("log_n"), w.get("rate_bits"), w.get("cap_height"), w.get("n_gpus")) != (W, log_n, rb, cap, world):
        return None
    hits = [v for k, v in d["kernels"].items() if all(n in k for n in needles)]
    return hits[0] if len(hits) == 1 else None


def pmc_stale():
    """True when the committed counters were collected on other kernel sources than the ones being timed"""
    d = _pmc_file()
    if not d:
        return None
    from tools.csrc_hash import csrc_hash
    return d.get("csrc_sha256_16") != csrc_hash()


def ubench_json():
    """profiles/ubench.json (tools/ubench_pmc.sh + tools/ubench_summarize.py on the MI355X): per probe and per occupancy
    (4 / 8 waves per SIMD) the TRUE shader cycles per wave64 instruction per SIMD (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs /
    SQ_INSTS_VALU) and the clock the probe ran at"""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "ubench.json")))
    except Exception:
        return None


def cpu_baseline(W, log_n, rate_bits, cap_height, budget_s=45.0, golden_cap=None):
    """The tuned CPU implementation of the same step (oracle/p2fast.c, kind "port", tuned: branch-free reduce128, the
    reference's fast-partial Poseidon, eight rows per AVX-512 instruction stream in the leaf sponge / tree levels / NTT
    layers when the CPU has AVX-512, cached root tables, coset-by-coset LDE, OpenMP over the host cores this job may
    use), bit-exact against the faithful oracle (tests/test_fast_oracle.py).
    The WHOLE 2^log_n-row step is timed when a 1/16 sample predicts it fits the budget, else the largest power-of-two
    row count that does.  NOT the Rust prover (no cargo in the image)."""
    from oracle import p2fast as fast
    from oracle import p2oracle as ora
    cores = ora.usable_cores()  # min(affinity, cgroup CPU quota): oversubscribing a quota-limited job throttles it
    fast.set_num_threads(cores)
    k = max(0, min(log_n, log_n - 4))
    t0 = time.perf_counter()
    fast.commit(splitmix_columns_numpy(0, W, 1 << k), rate_bits, cap_height, True, want_coeffs=False, want_digests=False)
    probe = time.perf_counter() - t0
    kk = k
    while kk < log_n and probe * (1 << (kk + 1 - k)) * 1.1 <= budget_s:
        kk += 1
    cols = splitmix_columns_numpy(0, W, 1 << kk)
    tm = {}
    t0 = time.perf_counter()
    r = fast.commit(cols, rate_bits, cap_height, True, want_coeffs=False, want_digests=False, timed=tm)
    dt = time.perf_counter() - t0
    fe = W * (1 << (kk + rate_bits))
    whole = kk == log_n
    out = {"value": fe / dt / 1e9, "unit": "GFE/s", "cores": cores, "kind": "port", "variant": "a port (oracle/p2fast.c), tuned: AVX-512 + OpenMP", "seconds": dt,
           "sample": ("the whole step" if whole else "1/%d of the GPU step's rows" % (1 << (log_n - kk)))
                     + ": from_values W=%d, 2^%d rows, rate 1/%d, cap %d, %.2f s on %d cores; oracle/p2fast.c (tuned C + OpenMP + AVX-512 "
                       "restatement of the reference algorithm), not the Rust prover (no cargo in the image)" % (W, kk, 1 << rate_bits, cap_height, dt, cores),
           "stage_seconds": {k_: round(v, 4) for k_, v in tm.items()},  # the reference's TimingTree scopes, fri/oracle.rs:65-103
           "us_per_permutation_per_core": tm["build Merkle tree"] * cores / (((W + 7) // 8 + 1) * (1 << (kk + rate_bits))) * 1e6}
    if whole and golden_cap is not None:
        out["cap_matches_golden"] = r["cap"].tolist() == golden_cap
    out["host"] = "%d logical CPUs visible, %d usable by this job (cgroup quota / affinity)" % (os.cpu_count(), cores)
    try:
        out["cpu_model"] = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    return out


def golden(name):
    """the oracle's record of a commit shape (tools/gen_golden_caps.py).  P2HOT_BENCH_GOLDEN_CAPS: another file (the test of the
    check itself, tests/test_bench_launch.py)"""
    try:
        return json.load(open(os.environ.get("P2HOT_BENCH_GOLDEN_CAPS") or os.path.join(ROOT, "tests", "golden", "commit_caps.json")))[name]
    except Exception:
        return None


def host_pointer_lines(eng, out, reps):
    """SURVEY 8d's second and third figures, driver-timed: p2hot_commit with PAGEABLE host columns (what the Rust shim passes:
    W separate Vec<F>), PCIe included -- (i) coefficients + digests + cap back, (ii) digests kept on the device (paths served
    from the handle), (iii) the 9 GB row-major leaf matrix copied back too; and the recursion-size (2^12 rows) call."""
    import ctypes as C
    rb, cap = 3, 4

    def run(name, W, log_n, what, want_coeffs, want_digests, want_leaves, gname, reps=reps, is_values=1, per_column=False, leaves_mem="touched"):
        """leaves_mem: where the leaf matrix lands -- "touched": pageable memory written before (no page fault in the timed calls);
        "fresh": a NEW pageable buffer per call, as a Vec::with_capacity would be (first-touch faults while the copy runs);
        "pinned": a block of the context's pinned cache (p2hot_host_alloc: what the Rust shim's flat leaf store uses), allocated and
        freed around every call, i.e. reused from the cache after the first"""
        n, N = 1 << log_n, 1 << (log_n + rb)
        cols = [np.ascontiguousarray(c) for c in splitmix_columns_numpy(0, W, n)]
        ptrs = (C.c_void_p * W)(*[c.ctypes.data for c in cols])
        coeffs = np.zeros((W, n), dtype=np.uint64) if want_coeffs else None
        # per_column: one destination per polynomial (what the Rust shim passes: W separate Vec<F>, P2HOT_COEFFS_PER_COLUMN)
        table = (C.c_void_p * W)(*[coeffs[c].ctypes.data for c in range(W)]) if (per_column and want_coeffs) else None
        digests = np.zeros((eng.num_digests(log_n + rb, cap), 4), dtype=np.uint64) if want_digests else None
        leaves = np.zeros((N, W), dtype=np.uint64) if (want_leaves and leaves_mem == "touched") else None   # touched here: page faults are not timed
        capv = np.zeros((1 << cap, 4), dtype=np.uint64)

        def ptr(a):
            return a.ctypes.data if a is not None else None

        def once():
            h, blk, lv = C.c_void_p(), C.c_void_p(), ptr(leaves)
            if want_leaves and leaves_mem == "fresh":
                fresh = np.empty((N, W), dtype=np.uint64)  # untouched: mmap'd by the allocator, unmapped again when it is dropped
                lv = fresh.ctypes.data
            elif want_leaves and leaves_mem == "pinned":
                eng.check(eng.lib.p2hot_host_alloc(eng.ctx, N * W * 8, C.byref(blk)))
                lv = blk
            eng.check(eng.lib.p2hot_commit(eng.ctx, ptrs, W, log_n, rb, cap, is_values, 2 if table is not None else 0,
                                           C.cast(table, C.c_void_p) if table is not None else ptr(coeffs), lv, ptr(digests),
                                           capv.ctypes.data, C.byref(h)))
            eng.lib.p2hot_batch_free(h)
            if blk.value:
                eng.lib.p2hot_host_free(eng.ctx, blk)
        first_pin_ms = None
        if want_leaves and leaves_mem == "pinned":  # what the FIRST commitment of a process pays: pinning the block (later ones reuse it)
            eng.check(eng.lib.p2hot_ctx_trim(eng.ctx))
            blk0 = C.c_void_p()
            t0 = time.perf_counter()
            eng.check(eng.lib.p2hot_host_alloc(eng.ctx, N * W * 8, C.byref(blk0)))
            first_pin_ms = (time.perf_counter() - t0) * 1e3
            eng.lib.p2hot_host_free(eng.ctx, blk0)
        once()
        t0 = time.perf_counter()
        for _ in range(reps):
            once()  # synchronous: the call returns when the caller's buffers are filled
        ms = (time.perf_counter() - t0) / reps * 1e3
        g = golden(gname) if gname else None
        out[name] = {"workload": what, "ms": ms, "GFE/s": W * N / ms / 1e6, "pcie_inclusive": True,
                     **({"first_pin_ms": first_pin_ms} if first_pin_ms is not None else {}),
                     **({"cap_checked": capv.tolist() == g["cap"]} if g else {})}

    base = "p2hot_commit (host pointers, pageable memory) C3 wires: W=135, 2^20 rows, rate 1/8, cap 4; 1.1 GB of columns in, "
    run("host_c3_wires_coeffs_digests", 135, 20, base + "1.1 GB of coefficients + 0.54 GB of digests + cap out (leaf matrix stays on the device)",
        True, True, False, "c3_wires")
    run("host_c3_wires_digests_on_device", 135, 20, base + "coefficients + cap out; digests and leaves stay behind the handle (p2hot_batch_paths / _rows)",
        True, False, False, "c3_wires")
    run("host_c3_wires_leaves_back", 135, 20, base + "coefficients (one destination per polynomial, P2HOT_COEFFS_PER_COLUMN) + digests + cap + the "
        "9.1 GB row-major leaf matrix out: exactly the Rust shim's default call (P2HOT_LEAVES=host)",
        True, True, True, "c3_wires", reps=2, per_column=True)
    run("host_c3_wires_leaves_back_fresh_pages", 135, 20, base + "as host_c3_wires_leaves_back, but the 9.1 GB leaf matrix lands in a NEW pageable "
        "buffer every call (what a fresh Vec is: a page fault per 4 KiB while the copy runs)", True, True, True, "c3_wires", reps=2, per_column=True, leaves_mem="fresh")
    run("host_c3_wires_leaves_back_pinned", 135, 20, base + "as host_c3_wires_leaves_back, but the leaf matrix lands in a block of the context's PINNED "
        "cache (p2hot_host_alloc / _free around every call: the Rust shim's flat leaf store)", True, True, True, "c3_wires", reps=2, per_column=True, leaves_mem="pinned")
    # P2HOT_LEAVES_ASYNC | P2HOT_LEAVES_NATURAL (the shim's mode from round 5 on): the call returns with cap + coefficients + digests,
    # the leaf matrix keeps travelling into the pinned block in 64 row blocks behind fences (p2hot_batch_leaves_wait)
    def run_async(name, W, log_n, what, gname, reps=2):
        n, N = 1 << log_n, 1 << (log_n + rb)
        cols = [np.ascontiguousarray(c) for c in splitmix_columns_numpy(0, W, n)]
        ptrs = (C.c_void_p * W)(*[c.ctypes.data for c in cols])
        coeffs = np.zeros((W, n), dtype=np.uint64)
        table = (C.c_void_p * W)(*[coeffs[c].ctypes.data for c in range(W)])
        digests = np.zeros((eng.num_digests(log_n + rb, cap), 4), dtype=np.uint64)
        capv = np.zeros((1 << cap, 4), dtype=np.uint64)
        t_call, t_first, t_last = [], [], []

        def once(timed):
            h, blk = C.c_void_p(), C.c_void_p()
            eng.check(eng.lib.p2hot_host_alloc(eng.ctx, N * W * 8, C.byref(blk)))
            t0 = time.perf_counter()
            eng.check(eng.lib.p2hot_commit(eng.ctx, ptrs, W, log_n, rb, cap, 1, 2 | 4 | 8, C.cast(table, C.c_void_p), blk, digests.ctypes.data,
                                           capv.ctypes.data, C.byref(h)))
            t1 = time.perf_counter()
            eng.check(eng.lib.p2hot_batch_leaves_wait(h, 0, 1))
            t2 = time.perf_counter()
            eng.check(eng.lib.p2hot_batch_leaves_wait(h, 0, N))
            t3 = time.perf_counter()
            if timed:
                t_call.append((t1 - t0) * 1e3), t_first.append((t2 - t0) * 1e3), t_last.append((t3 - t0) * 1e3)
            # the natural-order buffer: row 1 is committed row N / 2 (reverse_bits(1)), served by the handle in the committed indexing
            row = np.frombuffer((C.c_uint64 * W).from_address(blk.value + 8 * W), dtype=np.uint64).copy()
            idx = np.array([N // 2], dtype=np.uint64)
            got = np.zeros((1, W), dtype=np.uint64)
            eng.check(eng.lib.p2hot_batch_rows(h, idx.ctypes.data, 1, got.ctypes.data))
            eng.lib.p2hot_batch_free(h)
            eng.lib.p2hot_host_free(eng.ctx, blk)
            return bool((row == got[0]).all())
        once(False)
        rows_ok = all([once(True) for _ in range(reps)])
        g = golden(gname) if gname else None
        out[name] = {"workload": what, "ms": sum(t_call) / reps, "ms_first_block": sum(t_first) / reps, "ms_last_row": sum(t_last) / reps,
                     "pcie_inclusive": True, "natural_order_row_checked": rows_ok, **({"cap_checked": capv.tolist() == g["cap"]} if g else {})}

    run_async("host_c3_wires_leaves_async", 135, 20, base + "as host_c3_wires_leaves_back_pinned with P2HOT_LEAVES_ASYNC | P2HOT_LEAVES_NATURAL: `ms` = the call "
              "(cap + coefficients + digests back), `ms_first_block` = the first 1/64 of the rows fenced, `ms_last_row` = the whole 9.1 GB matrix landed", "c3_wires")
    # the other two commitments of a proof in the shim's default mode (leaves back): one column block each
    run("host_c3_zs_leaves_back", 20, 20, "p2hot_commit (host pointers, pageable memory) C3 Zs + partial products: from_values W=20, 2^20 rows; "
        "coefficients + digests + cap + the 1.3 GB leaf matrix out", True, True, True, "c3_zs_partial_products", reps=3)
    run("host_c3_quotient_leaves_back", 16, 20, "p2hot_commit (host pointers, pageable memory) C3 quotient chunks: from_coeffs W=16, 2^20 rows; "
        "digests + cap + the 1.1 GB leaf matrix out", False, True, True, "c3_quotient_chunks", reps=3, is_values=0)
    run("host_k12_wires", 135, 12, "p2hot_commit (host pointers) at recursion size: W=135, 2^12 rows, rate 1/8, cap 4; coefficients + digests + cap out",
        True, True, False, None, reps=20)


def host_tail_line(out):
    """What the Rust shim (integration/p2hot.rs) does on the HOST after p2hot_commit_salted returns, priced on this box's cores
    by tools/host_tail (C++: the same allocations and copies as the shim's Rust, one for one) next to the GPU call it follows.
    Default mode (P2HOT_LEAVES=host): the leaf matrix stays the ONE flat buffer the library filled (MerkleTree::get serves
    slices of it) and every polynomial is written into its own Vec by the library (P2HOT_COEFFS_PER_COLUMN): no host work
    beyond W allocations.  `rows_*` are the alternatives: P2HOT_LEAVES=vec (Vec<Vec<F>> rebuilt in parallel) and the
    round-3 shim's serial rebuild."""
    import subprocess
    exe = os.path.join(ROOT, "tools", "host_tail")
    try:
        if not os.path.exists(exe):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-o", exe, os.path.join(ROOT, "tools", "host_tail.cpp")])
        r = json.loads(subprocess.run([exe, "135", "20", "3", "21"], capture_output=True, text=True, timeout=300).stdout)
    except Exception as ex:  # noqa: BLE001
        out["host_tail_c3_wires"] = {"error": repr(ex)}
        return
    gpu = (out.get("host_c3_wires_leaves_back_pinned") or out.get("host_c3_wires_leaves_back", {})).get("ms")
    rec = {"workload": "host work of the Rust shim after the C3 wires commit (W=135, 2^20 rows, leaves back), tools/host_tail.cpp on %d threads" % r["threads"],
           "default_mode_ms": 0.0,
           "default_mode": "P2HOT_LEAVES=host + P2HOT_COEFFS_PER_COLUMN: W Vec allocations, no copy (the flat leaf buffer moves into the DeviceTree)",
           "split_coeffs_ms_if_flat": r["split_coeffs_ms"], "leaves_vec_parallel_ms": r["rows_parallel_ms"], "leaves_vec_serial_ms_round3": r["rows_serial_ms"],
           "rows_sample": "2^%d of 2^23 rows, scaled" % r["rows_sample_log"], "gpu_call_ms": gpu}
    if gpu:
        rec["default_tail_over_gpu_call"] = 0.0
        rec["vec_parallel_tail_over_gpu_call"] = r["rows_parallel_ms"] / gpu
        rec["round3_serial_tail_over_gpu_call"] = r["rows_serial_ms"] / gpu
    out["host_tail_c3_wires"] = rec


def recursion_lines(eng, torch, out):
    """Recursion-size throughput (the two 2^12-row recursive proofs of bench_recursion's chain, examples/bench_recursion.rs:317-345):
    one proof at a time is latency-bound, so (a) p2hot_commit_many commits M same-shape proofs in one set of launches and
    (b) K host threads, each with its own context and stream, prove independent proofs concurrently."""
    import threading
    from plonky2_amd import Engine
    from plonky2_amd.fri.oracle import FriBatchInfo, PolynomialBatch, eval_openings, prove_openings
    from plonky2_amd.iop.challenger import Challenger
    dev = eng.mem.device
    W, log_n, rb, cap = 135, 12, 3, 4
    n, N = 1 << log_n, 1 << (log_n + rb)
    nd = eng.num_digests(log_n + rb, cap)
    ref_cap = None
    rec = {"workload": "p2hot_commit_many_dev: M wires commitments (W=135, 2^12 rows, rate 1/8, cap 4) per call, device resident", "M": {}}
    for M in (1, 8, 64):
        inter = splitmix_columns_torch(torch, dev, 0, W, n).unsqueeze(1).repeat(1, M, 1).contiguous().reshape(W * M, n)  # [W][M][n]
        d_lde, d_dig, d_cap = eng.mem.empty(W * M, N), eng.mem.empty(M * nd, 4), eng.mem.empty(M << cap, 4)

        def run():
            work = inter.clone()  # the call transforms its input in place
            eng.check(eng.lib.p2hot_commit_many_dev(eng.ctx, eng.ptr(work), M, W, log_n, rb, cap, 1, eng.ptr(d_lde), eng.ptr(d_dig), eng.ptr(d_cap)))
        run()
        torch.cuda.synchronize()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            run()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        caps = eng.host(d_cap).reshape(M, 1 << cap, 4)
        if ref_cap is None:
            ref_cap = eng.host(eng.commit(splitmix_columns_torch(torch, dev, 0, W, n), log_n, rb, cap, True)["cap"])
        rec["M"][str(M)] = {"ms": ms, "commits_per_s": M / ms * 1e3, "caps_equal_single_commit": bool((caps == ref_cap[None]).all())}
        del inter, d_lde, d_dig, d_cap
    out["recursion_commit_many"] = rec

    # (a') M whole opening-proof pipelines per call: three p2hot_commit_many (wires 135, Zs 20 from values, quotient 16 from
    # coefficients: host columns in, handles out) + p2hot_prove_openings_many (4 oracles, 255 polynomials at zeta, the 2 Z at g*zeta,
    # arity 16 x2, PoW 16 bits, 28 queries) -- the FRI side of M recursion-size proofs; the proofs of one call are checked to be equal
    import ctypes as C
    from plonky2_amd import _lib
    cols_w, cols_z, cols_q, cols_cs = (splitmix_columns_numpy(b0, w, n) for b0, w in ((0, 135), (1000, 20), (2000, 16), (3000, 84)))
    # get_fri_instance (plonk/circuit_data.rs:530-548): oracles [constants_sigmas 84, wires 135, Zs + partial products 20, quotient 16],
    # all 255 polynomials at zeta, the 2 Z polynomials at g * zeta
    widths = (84, 135, 20, 16)
    allp = [(oi, pi) for oi, w in enumerate(widths) for pi in range(w)]
    nxt = [(2, pi) for pi in range(2)]
    arrs = [((C.c_uint32 * len(pl))(*[o for o, _ in pl]), (C.c_uint32 * len(pl))(*[q for _, q in pl])) for pl in (allp, nxt)]
    arity = (C.c_uint * 2)(4, 4)
    fp = _lib.FriParams(rb, cap, 16, 28, arity, 2, 0, 0, 0)
    rec = {"workload": "M x (3 p2hot_commit_many from host columns + p2hot_prove_openings_many) at 2^12 rows: the commitments and the "
                       "opening proof of M recursion-size proofs per call on the reference's FRI instance (4 oracles, 255 polynomials at zeta, "
                       "the 2 Z at g*zeta; the constants_sigmas commitments are the circuits', made once outside the timed calls; OpeningSet "
                       "evaluation and the permutation argument excluded)", "M": {}}
    for M in (1, 8, 64):
        cs_ptrs = (C.c_void_p * (M * 84))(*([cols_cs[e].ctypes.data for e in range(84)] * M))
        cs_handles = (C.c_void_p * M)()
        eng.check(eng.lib.p2hot_commit_many(eng.ctx, cs_ptrs, M, 84, log_n, rb, cap, 1, None, None, None, cs_handles))

        def pipeline():
            handles = [cs_handles]
            for cols_, w, isv in ((cols_w, 135, 1), (cols_z, 20, 1), (cols_q, 16, 0)):
                ptrs = (C.c_void_p * (M * w))(*([cols_[e].ctypes.data for e in range(w)] * M))
                hs = (C.c_void_p * M)()
                eng.check(eng.lib.p2hot_commit_many(eng.ctx, ptrs, M, w, log_n, rb, cap, isv, None, None, None, hs))
                handles.append(hs)
            chs = []
            for _ in range(M):
                h = C.c_void_p()
                eng.check(eng.lib.p2hot_challenger_create(eng.ctx, C.byref(h)))
                chs.append(h)
            lay = _lib.FriProofLayout()
            h0 = (C.c_void_p * 4)(handles[0][0], handles[1][0], handles[2][0], handles[3][0])
            eng.check(eng.lib.p2hot_fri_proof_sizes(h0, 4, C.byref(fp), C.byref(lay)))
            bufs = [[np.zeros(max(1, getattr(lay, k + "_words")), dtype=np.uint64) for k in ("caps", "final_poly", "initial_leaves", "initial_paths", "step_evals", "step_paths")]
                    for _ in range(M)]
            proofs = (_lib.FriProof * M)()
            infos = (_lib.FriBatchInfo * 2)()
            for k, (pt, (oi, pi)) in enumerate(zip(((3, 5), (21, 5)), arrs)):
                infos[k].point[0], infos[k].point[1] = pt
                infos[k].oracle_index, infos[k].poly_index, infos[k].n_polys = oi, pi, len(oi)
            bp = (C.POINTER(_lib.FriBatchInfo) * M)(*([C.cast(infos, C.POINTER(_lib.FriBatchInfo))] * M))
            nb = (C.c_size_t * M)(*([2] * M))
            hs_all = (C.c_void_p * (4 * M))(*[handles[o][m] for m in range(M) for o in range(4)])
            for m in range(M):
                b = bufs[m]
                proofs[m] = _lib.FriProof(b[0].ctypes.data, b[1].ctypes.data, 0, None, b[2].ctypes.data, b[3].ctypes.data, b[4].ctypes.data, b[5].ctypes.data)
            cp = (C.c_void_p * M)(*chs)
            eng.check(eng.lib.p2hot_prove_openings_many(eng.ctx, M, bp, nb, hs_all, 4, cp, C.byref(fp), proofs))
            same = all(int(proofs[m].pow_witness) == int(proofs[0].pow_witness) and (bufs[m][1] == bufs[0][1]).all() for m in range(M))
            for h in chs:
                eng.lib.p2hot_challenger_destroy(h)
            for hs in handles[1:]:
                for m in range(M):
                    eng.lib.p2hot_batch_free(hs[m])
            return same
        pipeline()
        reps = 10 if M < 64 else 4
        t0 = time.perf_counter()
        ok = all([pipeline() for _ in range(reps)])
        ms = (time.perf_counter() - t0) / reps * 1e3
        for m in range(M):
            eng.lib.p2hot_batch_free(cs_handles[m])
        rec["M"][str(M)] = {"ms": ms, "proofs_per_s": M / ms * 1e3, "proofs_equal": bool(ok)}
    out["recursion_pipeline_many"] = rec

    # (b) whole per-proof paths through the host-pointer entry points, K contexts side by side
    cols = splitmix_columns_numpy(0, W, n)
    zs = splitmix_columns_numpy(1000, 20, n)
    quo = splitmix_columns_numpy(2000, 16, n)
    cs = splitmix_columns_numpy(3000, 84, n)

    def one_proof(e):
        if not hasattr(e, "_bench_cs"):  # the circuit's own commitment (CircuitBuilder::build): once per context, outside the proofs
            e._bench_cs = PolynomialBatch.from_values(cs, rb, False, cap, engine=e)
        b_w = PolynomialBatch.from_values(cols, rb, False, cap, engine=e)
        b_z = PolynomialBatch.from_values(zs, rb, False, cap, engine=e)
        b_q = PolynomialBatch.from_coeffs(quo, rb, False, cap, engine=e)
        ch = Challenger(e)
        ch.observe_elements(np.arange(8, dtype=np.uint64))
        zeta = ch.get_extension_challenge()
        gz = [(zeta[0] * 7) % P, zeta[1]]
        oracles = [e._bench_cs, b_w, b_z, b_q]
        eval_openings(oracles, [zeta], e)   # OpeningSet::new (plonk/proof.rs:314-345): everything at zeta ...
        eval_openings([b_z], [gz], e)       # ... and the Zs commitment at g * zeta
        allp = [(oi, pi) for oi, w in enumerate((84, 135, 20, 16)) for pi in range(w)]
        nxt = [(2, pi) for pi in range(2)]
        return prove_openings([FriBatchInfo(zeta, allp), FriBatchInfo(gz, nxt)], oracles, ch, rb, cap, [4, 4], 16, 28, engine=e)["pow_witness"]

    rec = {"workload": "K host threads x (3 commits from host columns + OpeningSet + prove_openings on the reference's FRI instance: 4 oracles, 255 "
                       "polynomials at zeta, the 2 Z at g*zeta) at 2^12 rows, one context and stream per thread, host-pointer entry points (PCIe included)", "K": {}}
    for K in (1, 4, 8):
        per, results, errs = 12, [None] * K, []

        def worker(k):
            try:
                with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                    e = Engine(eng.mem.device_index if hasattr(eng.mem, "device_index") else 0)
                    one_proof(e)
                    barrier.wait()
                    for _ in range(per):
                        results[k] = one_proof(e)
                    e.sync()
                    barrier.wait()
                    del e._bench_cs
                    e.close()
            except Exception as ex:  # noqa: BLE001
                errs.append(repr(ex))
                barrier.abort()
        barrier = threading.Barrier(K + 1)
        ths = [threading.Thread(target=worker, args=(k,)) for k in range(K)]
        for t in ths:
            t.start()
        try:
            barrier.wait()
            t0 = time.perf_counter()
            barrier.wait()
            dt = time.perf_counter() - t0
        except threading.BrokenBarrierError:
            dt = float("nan")
        for t in ths:
            t.join()
        rec["K"][str(K)] = {"proofs_per_s": K * per / dt, "ms_per_proof_per_thread": dt / per * 1e3,
                            "same_witness_on_every_thread": len(set(results)) == 1, **({"errors": errs[:2]} if errs else {})}
    out["recursion_proofs_concurrent_contexts"] = rec


def other_configs(eng, torch, reps=3, only=None):
    """Driver-timed lines for the other BASELINE shapes (extra keys of the JSON line; the headline is unchanged):
    each is `reps` timed repetitions after one warm-up, inputs resident in HBM, synchronised wall time."""
    from plonky2_amd.fri.prover import fri_committed_trees_device
    from plonky2_amd.iop.challenger import Challenger
    from plonky2_amd.util.synthetic import fibonacci_trace
    dev = eng.mem.device
    out = {}

    def timed(fn, reps=reps):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    def commit_line(name, W, log_n, rb, cap, is_values, cols, what, reps=reps):
        ms = timed(lambda: eng.commit(cols, log_n, rb, cap, is_values), reps)
        g = golden(name)
        rec = {"workload": what, "ms": ms, "GFE/s": W * (1 << (log_n + rb)) / ms / 1e6}
        if g is not None:
            rec["cap_checked"] = eng.host(eng.commit(cols, log_n, rb, cap, is_values)["cap"]).tolist() == g["cap"]
        out[name] = rec

    if only == "host":  # tooling: the PCIe-inclusive lines alone
        host_pointer_lines(eng, out, reps)
        host_tail_line(out)
        return out
    if only == "path":  # tooling: the per-proof paths alone
        _only_paths = True
    else:
        _only_paths = False
    if not _only_paths:
        commit_line("c2_wires", 135, 16, 3, 4, True, splitmix_columns_torch(torch, dev, 0, 135, 1 << 16),
                    "C2: from_values W=135, 2^16 rows, rate 1/8, cap 4")
        commit_line("c3_constants_sigmas", 84, 20, 3, 4, True, splitmix_columns_torch(torch, dev, 0, 84, 1 << 20),
                    "C3: from_values W=84 (constants + sigmas: the commitment CircuitBuilder::build makes, circuit_builder.rs:1182-1191), "
                    "2^20 rows, rate 1/8, cap 4")
        commit_line("c3_zs_partial_products", 20, 20, 3, 4, True, splitmix_columns_torch(torch, dev, 0, 20, 1 << 20),
                    "C3: from_values W=20 (Zs + partial products), 2^20 rows, rate 1/8, cap 4")
        commit_line("c3_quotient_chunks", 16, 20, 3, 4, False, splitmix_columns_torch(torch, dev, 0, 16, 1 << 20),
                    "C3: from_coeffs W=16 (quotient chunks), 2^20 rows, rate 1/8, cap 4")
        commit_line("c4_fibonacci_trace", 2, 22, 1, 4, True, eng.dev(fibonacci_trace(22)),
                    "C4: from_values W=2 (Fibonacci trace), 2^22 rows, rate 1/2, cap 4 (hash_or_noop leaves)")
        host_pointer_lines(eng, out, reps)
        host_tail_line(out)
        recursion_lines(eng, torch, out)
        for name, log_n, rb in (("c3_fri_commit_phase", 20, 3), ("c4_fri_commit_phase", 22, 1)):
            planes = splitmix_columns_torch(torch, dev, 500, 2, 1 << log_n)
            ch = Challenger(eng)
            ms = timed(lambda: fri_committed_trees_device(planes, log_n, ch, rb, 4, [4, 4, 4, 4], eng))
            out[name] = {"workload": "fri_committed_trees, N=2^%d, arity 16 x4, cap 4 (final FFT + 4 round trees + folds, device resident)"
                                     % (log_n + rb), "ms": ms}
    # the per-proof path of a standard_recursion_config proof: every stage of SURVEY section 8 back to back, at the
    # headline size (2^20 gates) and at recursion size (2^12 gates: the two recursive proofs of bench_recursion's chain), and C4's
    # starky path (StarkConfig::standard_fast_config: rate 1/2, cap 4, 84 queries, PoW 16 bits).  The stages live in
    # plonky2_amd/util/proof_path.py (shared with tests/test_gpu_fullsize.py); the instances in plonky2_amd/util/synthetic.py.
    # WHAT IS TIMED IS CHECKED: after the timed repetitions the same path runs once more keeping every stage's output, and each
    # is compared with the CPU oracle's bytes for the same instance (tests/golden/path_goldens.json, tools/gen_golden_path.py):
    # the four caps, SHA-256 of the Zs / partial-products matrix, of the quotient values at all 2^(k+3) points and of the 16 chunk
    # polynomials, every opening value, alpha (through final_poly), the commit-phase caps, final_poly, the PoW witness, the query
    # indices and SHA-256 of the whole FriProof in the reference's wire format.  A mismatch ends the bench with a non-zero exit.
    from plonky2_amd.util import proof_path as pp
    from plonky2_amd.util.synthetic import path_instance

    def path_line(name, what, warmups):
        inst = path_instance(name)
        inp = pp.PathInputs(eng, inst)
        for _ in range(warmups):  # the block cache alternates between two sets while the previous proof's commitments are still alive
            pp.run_path(eng, inp, sync=torch.cuda.synchronize)
        stages = [pp.run_path(eng, inp, sync=torch.cuda.synchronize)["stage_ms"] for _ in range(reps)]
        mean = {k: sum(s_[k] for s_ in stages) / reps for k in stages[0]}
        rec = {"workload": what, "ms": sum(mean.values()), "stage_ms": {k: round(v, 3) for k, v in mean.items()},
               "ms_per_repetition": [round(sum(s_.values()), 3) for s_ in stages]}
        g = pp.golden(name)
        if g is not None:
            res = pp.run_path(eng, inp, sync=torch.cuda.synchronize, keep=True)
            bad = pp.compare_with_golden(res, g)
            if bad:
                raise SystemExit("bench: %s differs from the oracle's record in: %s" % (name, "; ".join(bad)))
            rec["checked"] = True
            rec["checked_against"] = ("tests/golden/path_goldens.json (oracle/p2oracle.c on the same instance): caps, Zs + partial products, quotient "
                                      "values, chunk polynomials, openings, FRI caps, final_poly, PoW witness, query indices, FriProof bytes"
                                      if inst["kind"] == "plonk" else
                                      "tests/golden/path_goldens.json (oracle/p2oracle.c on the same instance): caps, openings, FRI caps, final_poly, "
                                      "PoW witness, query indices, FriProof bytes")
            del res
        else:
            rec["checked"] = False
        out[name] = rec
        del inp
        torch.cuda.empty_cache()

    plonk_what = ("the SURVEY section-8 stages of one 2^%d-gate standard_recursion_config proof back to back, on the reference's own "
                  "FRI instance (get_fri_instance, plonk/circuit_data.rs:530-548: 4 oracles, 255 polynomials at zeta, the 2 Z at g*zeta); "
                  "the constants_sigmas commitment (W=84) belongs to CircuitBuilder::build and is made outside the timed path; "
                  "gate evaluation / witness generation excluded (out of scope)")
    path_line("per_proof_path_k20", plonk_what % 20, 2)
    path_line("per_proof_path_k12", plonk_what % 12, 2)
    if _only_paths:
        return out
    path_line("per_proof_path_starky_k22", "C4: every section-8 stage of one starky proof of a 2-column, 2^22-row trace (Fibonacci), rate 1/2, back to "
              "back (constraint evaluation excluded: out of scope)", 1)

    # C5 (2^23 rows, W = 135: the 8-GPU configuration, 72.5 GB of LDE values) as ONE MI355X's commit: HBM holds it whole
    torch.cuda.empty_cache()
    if torch.cuda.mem_get_info()[0] > 130 * (1 << 30):
        commit_line("c5_wires", 135, 23, 3, 4, True, splitmix_columns_torch(torch, dev, 0, 135, 1 << 23),
                    "C5 on one GPU: from_values W=135, 2^23 rows, rate 1/8, cap 4 (the shape bench.py --gpus 8 shards over 8 ranks)", reps=2)
        torch.cuda.empty_cache()
    return out


def valu_line(e, launches_per_step, h, perms=None):
    """Issue-rate view of the dominant kernel, MEASURED IN THIS RUN: achieved = (wave64 VALU instructions one launch retires) /
    (the kernel's average duration in this run's timed region, HIP events on its launch stream).  The instruction count is a
    property of the kernel binary and the shape (SQ_INSTS_VALU of the committed PMC pass, profiles/pmc_traffic.json: it does not
    vary from run to run); the time is this run's.  Priced against two ceilings, neither capped:
      valu_nominal    1024 SIMDs x 2.4 GHz / 4 cycles per single-issue wave64 instruction = 614.4 Gwave-inst/s
                      (plonky2_amd/util/chip.py; spec clock, so a chip that clocks to its power budget shows below 1).  Since round 6
                      the S-box products carry three v_mov_b32 each that CO-ISSUE with the multiply-adds (they take no slot of their
                      own: profiles/r06_ubench_cheap.txt, r06_sbox_carryfree_ab.txt): this view counts the FULL-PRICE instructions,
                      SQ_INSTS_VALU minus 3 x 472 moves per permutation (`co_issued_moves_per_launch`)
      valu_empirical  the kernel's OWN instruction mix issued as dependency-free streams (tools/ubench.hip `mix hash_leaves r06b`:
                      60 % multiply-adds, 15 % carry adds / subtracts, 6 % selects, 12 % moves, 6 % plain adds), as measured on an
                      MI355X under rocprofv3 --pmc (profiles/ubench.json): 1024 x probe clock / probe cycles per instruction,
                      against ALL of the kernel's instructions; `frac_raw` only -- a self-measured ceiling that a kernel can tie
                      or beat by scatter, so it is never clamped
    The committed PMC pass's own cycles per instruction (true shader cycles, clock-free) stay under `pmc_pass`."""
    if not e or not e.get("sq_insts_valu_per_launch") or not h.get("ms_per_launch") or h["ms_per_launch"] != h["ms_per_launch"]:
        return None
    n_all = e["sq_insts_valu_per_launch"] / launches_per_step
    # the S-box products' v_mov_b32 co-issue with the multiply-adds (plonky2_amd/util/chip.py): the 4-cycle-slot ceiling prices the rest
    n_mov = (perms / 64.0) * POSEIDON_SBOX_PRODUCTS_PER_PERMUTATION * CO_ISSUED_MOVES_PER_SBOX_PRODUCT if perms else 0.0
    n = n_all - n_mov
    live = n / (h["ms_per_launch"] * 1e-3) / 1e9
    out = {"kernel": "hash_leaves", "bound": "valu-issue", "wave_insts_per_launch": n_all, "co_issued_moves_per_launch": n_mov,
           "full_price_wave_insts_per_launch": n, "live_ms_per_launch": h["ms_per_launch"],
           "achieved_Gwave_inst_per_s": live, "all_insts_Gwave_inst_per_s": n_all / (h["ms_per_launch"] * 1e-3) / 1e9,
           "valu_nominal": {"peak": VALU_NOMINAL_GWAVE_INST_PER_S, "unit": "Gwave-inst/s", "frac": live / VALU_NOMINAL_GWAVE_INST_PER_S,
                            "counts": "full-price instructions: SQ_INSTS_VALU minus the S-box products' co-issued v_mov_b32 (3 x 472 per permutation)",
                            "peak_source": "plonky2_amd/util/chip.py: 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 VALU instruction"},
           "source": "instruction count: profiles/pmc_traffic.json (SQ_INSTS_VALU per launch, committed PMC pass); time: this run's HIP events"}
    if e.get("clock_ghz") and e.get("ms_per_launch_under_pmc"):
        clock = e["clock_ghz"]
        cyc = e["ms_per_launch_under_pmc"] * 1e-3 * clock * 1e9 * NUM_SIMDS / e["sq_insts_valu_per_launch"]
        out["pmc_pass"] = {"clock_ghz": clock, "ms_per_launch": e["ms_per_launch_under_pmc"], "cycles_per_inst": cyc,
                           # 4 cycles x ALL instructions / the pass's cycles: above 1 since the moves share slots; the share of the
                           # 4-cycle slots that full-price instructions take is the line below it
                           "valu_busy_frac": e.get("valu_busy_frac"),
                           "full_price_slot_frac": 4.0 * (n / n_all) / cyc,
                           # the clock this run's kernel time implies if it retired its instructions at the PMC pass's cycles each
                           "implied_live_clock_ghz": out["all_insts_Gwave_inst_per_s"] * cyc / NUM_SIMDS}
    ub = ubench_json()
    if ub:
        occ = ub.get("occupancy", {})
        # the mix with the S-box products' moves, if it was probed (r06b: the 14-instruction products that ship; r06: the 16-instruction ones)
        r06 = next((nm for nm in ("mix hash_leaves r06b", "mix hash_leaves r06") if any(nm in o.get("probes", {}) for o in occ.values())), None)
        names = (r06, r06 + " x4") if r06 else ("mix hash_leaves", "mix hash_leaves x4")
        probes = {w + ("" if not nm.endswith("x4") else " x4"): o["probes"].get(nm, {}) for w, o in occ.items() for nm in names}
        rated = [(NUM_SIMDS * p_["clock_ghz"] / p_["cyc_per_inst"], w) for w, p_ in probes.items() if p_.get("cyc_per_inst") and p_.get("clock_ghz")]
        if rated:
            peak, w = max(rated)
            live_all = out["all_insts_Gwave_inst_per_s"] if r06 else live  # the r06 mix contains the moves: it is compared with ALL instructions
            out["valu_empirical"] = {"peak": peak, "unit": "Gwave-inst/s", "frac_raw": live_all / peak,
                                     "probe": "tools/ubench.hip `%s` (%s: waves per SIMD, x4 = 128 instructions per loop trip): %.3f cycles per "
                                              "instruction at %.2f GHz" % (names[0], w[1:], probes[w]["cyc_per_inst"], probes[w]["clock_ghz"]),
                                     "cycles_per_inst_by_occupancy": {w_: round(p_["cyc_per_inst"], 3) for w_, p_ in probes.items() if p_.get("cyc_per_inst")},
                                     "peak_source": "profiles/ubench.json (tools/ubench_pmc.sh: every probe under rocprofv3 --pmc, true cycles)"}
        out["classes_cycles_per_inst"] = {w: {c: round(v["cyc_per_inst_median"], 2) for c, v in o.get("classes", {}).items()} for w, o in occ.items()}
    return out


def roofline_line(vl, hbm_achieved, hash_bytes, perms, h, launches_per_step, e):
    """The dominant kernel (the Poseidon leaf sponge), every figure from THIS run's kernel time.  Three fractions side by side,
    none capped: `hbm` (the contract's view: algorithmic bytes per launch / live duration / 8 TB/s; also the scalar `hbm_frac`),
    `valu_nominal` and `valu_empirical` (valu_line).  The kernel is integer-VALU-issue bound (PMC traffic = 1.000 x its
    algorithmic bytes, 2-3 % of HBM peak), so the top-level bound / achieved / peak / frac are the nominal VALU view."""
    hbm = {"bound": "hbm", "achieved": hbm_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_achieved / HBM_PEAK_GBS,
           "algorithmic_bytes_per_launch": hash_bytes, "peak_source": "plonky2_amd/util/chip.py (the guide: 8.0 TB/s HBM3E)"}
    common = {"kernel": "hash_leaves_kernel<ColMajorReader> (Poseidon leaf sponge)",
              "traffic": e["hbm_bytes_per_launch"] / launches_per_step if e else None, "traffic_stale": pmc_stale(),
              "launches_per_step": launches_per_step, "live_ms_per_launch": h["ms_per_launch"],
              "traffic_source": "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; "
                                "traffic_stale = the kernel sources changed since they were collected)",
              **({"overlap": "launched on a second stream beside the next coset block's LDE; durations are wall "
                             "time while sharing the GPU"} if launches_per_step > 1 else {}),
              "note": "integer-VALU-issue bound (%.3g permutations per launch, %.2f Gperm/s, %s VALU instructions each); "
                      "algorithmic bytes per launch = 8*W*rows + 32*rows = %d"
                      % (perms, perms / (h["ms_per_launch"] * 1e-3) / 1e9,
                         "%.2f k" % (e["sq_insts_valu_per_launch"] / launches_per_step * 64 / perms / 1e3) if e and e.get("sq_insts_valu_per_launch") else "~13 k", hash_bytes),
              "hbm_frac": hbm["frac"], "hbm": hbm}
    if vl:
        nom = vl["valu_nominal"]
        return {**common, "bound": "valu", "achieved": vl["achieved_Gwave_inst_per_s"], "peak": nom["peak"], "unit": "Gwave-inst/s",
                "frac": nom["frac"], "valu_nominal": nom, "valu_empirical": vl.get("valu_empirical")}
    return {**common, **{k: hbm[k] for k in ("bound", "achieved", "peak", "unit", "frac")}}  # no committed counters for this shape


def ntt_roofline(kern, W, n_local, rows_local, steps, entry):
    """The NTT family of a step (iNTT strided + contiguous, the bit reversal between them and the LDE -- its own launch, or since
    round 6 folded into the LDE's strided pass --, LDE strided + contiguous)
    on SURVEY section 8(d)'s contract: each LOGICAL array counted once per LOGICAL stage, no credit for a multi-pass transform --
    `algorithmic_bytes` = iNTT 16*W*n + coset LDE 8*W*n + 8*W*N; `family_frac` = that / the five launches' time / 8 TB/s;
    `traffic` = the PMC bytes of the same five launches (profiles/pmc_traffic.json), `traffic_ratio` = traffic / algorithmic.
    The per-launch view (what each launch itself reads and writes: 16*W*n per iNTT pass, 16*W*n for the bit reversal,
    8*W*n + 8*W*N for the strided LDE pass, 16*W*N for the contiguous one) is kept under `passes` as `pass_bytes` / `pass_frac`."""
    per = {"ntt_intt_strided": (16 * W * n_local, ("ntt_limbpass_kernel<true", ",4,")),
           "ntt_intt_contig": (16 * W * n_local, ("ntt_limbpass_kernel<true,12,0",)),
           "bitrev_permute": (16 * W * n_local, ("bitrev_tiled_kernel",)),
           "ntt_lde_strided": (8 * W * n_local + 8 * W * rows_local, ("ntt_limbpass_kernel<false", ",4,")),
           "ntt_lde_contig": (16 * W * rows_local, ("ntt_limbpass_kernel<false,12,0",))}
    if "bitrev_permute" not in kern:  # the bit reversal is folded into the LDE's strided pass (nttl.hpp BRIN), which also writes the
        b_, n_ = per["ntt_lde_strided"]  # natural-order coefficients: 8*W*n more bytes for that launch, 16*W*n fewer for the family
        per["ntt_lde_strided"] = (b_ + 8 * W * n_local, n_)
    stage_bytes = {"intt": 16 * W * n_local, "coset_lde": 8 * W * n_local + 8 * W * rows_local}
    alg = sum(stage_bytes.values())
    passes, tot_ms, traffic, traffic_complete = {}, 0.0, 0.0, True
    for k, (b, needles) in per.items():
        if k not in kern:
            continue
        ms = kern[k]["ms_per_launch"]
        e = entry(*needles)
        per_step = kern[k]["launches"] / steps
        passes[k] = {"ms": ms, "launches_per_step": per_step, "pass_bytes": b, "pass_achieved": b / (ms * 1e-3) / 1e9,
                     "pass_frac": b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": e["hbm_bytes_per_launch"] if e else None,
                     "valu_insts_per_element": (e["sq_insts_valu_per_launch"] * 64 / (b / 16)) if e and e.get("sq_insts_valu_per_launch") else None}
        tot_ms += ms * per_step
        if e:
            traffic += e["hbm_bytes_per_launch"] * per_step
        else:
            traffic_complete = False
    if not passes:
        return None
    fam = alg / (tot_ms * 1e-3) / 1e9
    vpe = [p["valu_insts_per_element"] for k, p in passes.items() if k.endswith("_contig") and p.get("valu_insts_per_element")]
    return {"kernel": "ntt_limbpass_kernel (24-bit-limb radix-8 passes, nttl.hpp; + bitrev_tiled_kernel where the bit reversal is not folded): iNTT, bit reversal and coset LDE of one from_values commit",
            "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "achieved": fam, "frac": fam / HBM_PEAK_GBS,
            "algorithmic_bytes": alg, "algorithmic_bytes_by_stage": stage_bytes,
            "family_ms_per_step": tot_ms, "family_achieved": fam, "family_frac": fam / HBM_PEAK_GBS,
            "traffic": traffic if traffic_complete and traffic else None, "traffic_ratio": traffic / alg if traffic_complete and traffic else None,
            "traffic_stale": pmc_stale(),
            "passes": passes,
            "note": "SURVEY 8(d): iNTT 16*W*n + LDE 8*W*n + 8*W*N counted once each; a two-pass transform plus the bit reversal (folded into the LDE's strided pass where that pass is a limb pass) moves "
                    "%s that (`traffic_ratio`), so `pass_frac` (each launch against the bytes it moves itself) is higher than `family_frac`.  The passes "
                    "retire ~%s VALU instructions per element-pass (carry-free 24-bit limbs: half of them plain 32-bit adds) with VALU, LDS "
                    "and HBM all busy (profiles/*_pmc_sq.txt)"
                    % ("%.2f x" % (traffic / alg) if traffic_complete and traffic else "about 2.8 x", "%.0f" % (sum(vpe) / len(vpe)) if vpe else "110")}


def group_per_proof_path(eng, torch, world, emu, device_sync):
    """`--gpus N`, rank 0 only, after the timed commit: the per-proof path of `other_configs.per_proof_path_k20` over a p2hot_group of
    all N devices -- ONE process driving every GPU, the deployment a patched plonky2 uses -- as plonky2_amd/util/proof_path.py
    run_group_path: four coset-sharded commits from host columns, p2hot_group_eval_openings, p2hot_group_prove_openings, every
    output compared with the same oracle record the single-GPU line is checked against.  The Zs matrix and the quotient chunks
    come from the single-GPU stages (run once, untimed, on this rank's device).  Never fatal: a failure is reported in the record.
    P2HOT_BENCH_GROUP_PATH names the instance (default per_proof_path_k20; the emulated tier runs per_proof_path_k12)."""
    name = os.environ.get("P2HOT_BENCH_GROUP_PATH", "per_proof_path_k20")
    rec = {"workload": name + " over a p2hot_group of %d devices (one process): 4 sharded commits from host columns + OpeningSet + prove_openings" % world}
    try:
        from plonky2_amd.distributed import GroupCommit
        from plonky2_amd.util import proof_path as pp
        from plonky2_amd.util.synthetic import path_instance, splitmix_columns_numpy
        inst, g = path_instance(name), pp.golden(name)
        if g is None:
            raise RuntimeError("no oracle record for " + name)
        if (1 << inst["rate_bits"]) < world:
            raise RuntimeError("the group proof is coset mode: world %d > 2^rate_bits" % world)
        if not emu and torch.cuda.device_count() < world:
            raise RuntimeError("rank 0 sees %d devices, the group needs %d" % (torch.cuda.device_count(), world))
        single = pp.run_path(eng, pp.PathInputs(eng, inst), sync=device_sync, keep=True)
        zs, chunks = single["zs"], single["chunks"]
        rec["single_gpu_ms"] = sum(single["stage_ms"].values())
        rec["single_gpu_checked"] = pp.compare_with_golden(single, g) == []
        del single
        if not emu:
            torch.cuda.empty_cache()
        n = 1 << inst["log_n"]
        cs = splitmix_columns_numpy(inst["cs_seed"], inst["cs_width"], n)
        wires = splitmix_columns_numpy(inst["wires_seed"], inst["wires_width"], n)
        group = GroupCommit(eng.lib, world, list(range(world)))
        try:
            rec["uses_rccl"], rec["exchange"] = group.uses_rccl, group.exchange
            best, res = None, None
            for rep in range(1 if emu else 3):   # the first is the warm-up (tables, pools) unless it is the only one
                device_sync()
                t0 = time.perf_counter()
                res = pp.run_group_path(group, inst, cs, wires, zs, chunks, sync=None)
                dt_ = (time.perf_counter() - t0) * 1e3
                if emu or rep > 0:
                    best = dt_ if best is None else min(best, dt_)
            bad = pp.compare_with_golden(res, g)
            rec.update({"ms": best, "stage_ms": res["stage_ms"], "checked": not bad, "differs": bad or None,
                        "note": "host columns in (PCIe-inclusive), results on the host; stage_ms are un-synchronised issue times"})
        finally:
            group.close()
    except Exception as ex:  # noqa: BLE001
        rec.update({"skipped": "%s: %s" % (type(ex).__name__, ex), "checked": False})
    return rec


def self_launch(n, argv):
    """`python bench.py --gpus N` outside a launcher: re-run this file as N ranks (one process per GPU) under
    torch.distributed.run on 127.0.0.1 and a free port; the ranks inherit stdout / stderr, so rank 0's JSON line is this
    process's JSON line, and the launcher's exit code is returned."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env, cwd=ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-n", type=int, default=20, help="rows per GPU = 2^log_n (weak scaling)")
    ap.add_argument("--width", type=int, default=135)
    ap.add_argument("--rate-bits", type=int, default=3)
    ap.add_argument("--cap-height", type=int, default=4)
    ap.add_argument("--strong", action="store_true", help="strong scaling: 2^log_n rows in TOTAL (the C3 commit split over the GPUs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra driver-timed lines of the other BASELINE shapes")
    ap.add_argument("--extra-only", default=None, help="tooling: only this group of extra lines (host)")
    args = ap.parse_args()

    # `python bench.py --gpus N` with no launcher around it: start the N ranks ourselves (one process per GPU under
    # torch.distributed.run, rendezvous on 127.0.0.1), pass the child's JSON line and exit code through
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))

    # P2HOT_BENCH_EMU=1 (CPU TEST TIER ONLY, tests/test_bench_launch.py): the same main() over the kernel-source emulator and its
    # fake RCCL (tests/emu) -- launcher, rank plumbing, preflight, sharded commit, cap check and the JSON line run where there is
    # no GPU; the line says "emulated": true and its timings mean nothing.  Without it, no GPU = no bench.
    emu = os.environ.get("P2HOT_BENCH_EMU") == "1"
    import torch
    if not emu and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: the launcher's world size and --gpus must agree" % (args.gpus, world))
    # P2HOT_BENCH_BACKEND=gloo (tooling): the ranks may share a GPU and exchange through the host -- the whole multi-process
    # flow, cap check included, on a one-GPU box; the numbers of such a run are not scaling numbers
    backend = "gloo" if emu else os.environ.get("P2HOT_BENCH_BACKEND", "nccl")
    if not emu:
        if backend != "nccl":
            local_rank %= torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from plonky2_amd.distributed import ShardedCommit
    if emu:
        import ctypes as C
        from plonky2_amd.engine import Engine
        from tests.emu_backend import HostMemory, emu_lib
        lib = emu_lib()
        lib.p2hot_emu_set_device.argtypes = [C.c_int]
        lib.p2hot_emu_set_device(local_rank)
        eng = Engine(local_rank, lib=lib, memory=HostMemory())
        device_sync = eng.sync
    else:
        import __graft_entry__ as ge
        if rank == 0:
            ge.build_product()
        if dist:
            dist.barrier()
        from plonky2_amd import Engine
        eng = Engine(local_rank)
        device_sync = torch.cuda.synchronize
    W, rb, cap = args.width, args.rate_bits, args.cap_height
    log_g = (world - 1).bit_length()
    if world != 1 << log_g:
        raise SystemExit("--gpus must be a power of two")
    log_n = args.log_n + (0 if args.strong else log_g)
    n, N = 1 << log_n, 1 << (log_n + rb)

    # digests stay with the rank that owns the rows (its Merkle paths never leave its cap subtrees); only the cap is
    # all-gathered (SURVEY 8e collective 2).  P2HOT_GATHER_DIGESTS=1 also reassembles the full digest array everywhere.
    gather_digests = os.environ.get("P2HOT_GATHER_DIGESTS") == "1"
    def make_job(log_rows, comm=None, transport=None):
        """the sharded commit of 2^log_rows rows in total + this rank's synthetic columns (generated on its device)"""
        j = ShardedCommit(eng, W, log_rows, rb, cap, is_values=True, rank=rank, world=world, dist=dist, gather_digests=gather_digests,
                          transport=transport or ("rccl" if (emu and world > 1) else None), comm=comm)
        a, b = j.column_range
        if emu:
            return j, eng.dev(splitmix_columns_numpy(a, b - a, 1 << log_rows))
        return j, splitmix_columns_torch(torch, eng.mem.device, a, b - a, 1 << log_rows)

    job, cols = make_job(log_n)

    # preflight (before anything is timed): peer access between the node's GPUs, the transport that was bound, and a 1 MB
    # all-gather through the library's own exchange path checked on every rank -- a broken fabric fails here, by name
    preflight = None
    if world > 1:
        preflight = {"transport": job.comm.transport}
        try:
            nd_ = torch.cuda.device_count() if not emu else 0
            preflight["peer_access"] = [bool(torch.cuda.can_device_access_peer(local_rank, j)) for j in range(nd_) if j != local_rank] if backend == "nccl" else None
        except Exception as ex:  # noqa: BLE001
            preflight["peer_access"] = "query failed: %r" % (ex,)
        t_pf = time.perf_counter()
        flag_dev = eng.mem.device if backend == "nccl" else "cpu"

        def any_rank(bad):
            """every rank learns whether ANY rank says `bad` (through the launcher's process group, not the transport under test)"""
            f = torch.tensor([1 if bad else 0], dtype=torch.int32, device=flag_dev)
            dist.all_reduce(f, op=dist.ReduceOp.MAX)
            return bool(int(f.item()))

        def guarded_selftest(job_, not_ready=None):
            """The 1 MB all-gather through the transport under test, entered by ALL ranks or by none: the ranks first tell each other
            (over torch's process group) that they got as far as the collective -- a rank that failed earlier would otherwise leave the
            healthy ones alone inside an all-gather that never completes -- and the collective itself runs under a watchdog
            (P2HOT_BENCH_PREFLIGHT_TIMEOUT seconds, default 120), so a rank that dies INSIDE it ends the run with a message instead of
            a hang.  Returns (error or None, stuck): stuck = a library call is still inside the collective, nothing can be reused."""
            import threading
            if any_rank(not_ready is not None):
                return not_ready or RuntimeError("another rank did not reach the preflight collective"), False
            box = {}

            def run():
                try:
                    if not emu:
                        torch.cuda.set_device(local_rank)  # (the current device is per thread)
                    job_.comm.selftest(1 << 20)
                except Exception as ex_:  # noqa: BLE001
                    box["err"] = ex_
            th = threading.Thread(target=run, daemon=True)
            th.start()
            th.join(float(os.environ.get("P2HOT_BENCH_PREFLIGHT_TIMEOUT", "120")))
            if th.is_alive():
                return TimeoutError("the preflight all-gather did not complete on rank %d" % rank), True
            return box.get("err"), False

        injected = None
        if emu and os.environ.get("P2HOT_BENCH_EMU_FAIL_PREFLIGHT") == str(rank):  # (test tier: this rank never reaches the collective)
            injected = RuntimeError("injected preflight failure on rank %d" % rank)
        err, stuck = guarded_selftest(job, injected)
        failed = any_rank(err is not None)
        if any_rank(stuck):  # some rank is still inside the library's collective: no transport can be swapped in under it
            print("bench preflight: the 1 MB all-gather through the %s transport hung (rank %d: %s)" % (job.comm.transport, rank, err), file=sys.stderr, flush=True)
            os._exit(3)
        if failed and job.comm.transport == "rccl" and os.environ.get("P2HOT_BENCH_NO_FALLBACK") != "1":
            # first contact with a real multi-GPU node happens in the driver's run, once: if the in-library RCCL communicator fails its
            # preflight, the exchange moves to torch.distributed's own communicator (device buffers, synchronous) instead of ending the
            # run -- the line says so (`transport`, `preflight.fallback`) and the scaling numbers are then those of that transport
            preflight["fallback"] = "the in-library RCCL transport failed its preflight on at least one rank (%s); torch.distributed's communicator carries the exchange" % (err,)
            job.comm.close()
            del job, cols
            job, cols = make_job(log_n, transport="torch" if backend == "nccl" else "gloo")  # (gloo: the emulated tier's process group)
            preflight["transport"] = job.comm.transport
            err, stuck = guarded_selftest(job)
            failed = any_rank(err is not None)
            if any_rank(stuck):
                print("bench preflight: the fallback transport's all-gather hung (rank %d: %s)" % (rank, err), file=sys.stderr, flush=True)
                os._exit(3)
        if failed:
            raise SystemExit("bench preflight: the 1 MB all-gather through the %s transport failed (on rank %d: %s)" % (job.comm.transport, rank, err))
        preflight["selftest"] = "ok"
        preflight["selftest_ms"] = (time.perf_counter() - t_pf) * 1e3
        preflight["exchange"] = job.comm.exchange  # "allgather" / "broadcast": what the selftest's micro-timing of both kept
        preflight["rccl"] = job.comm.rccl_info() if job.comm.transport == "rccl" else None  # the file and version the library bound

    gnames = {(135, 16, 3, 4): "c2_wires", (135, 20, 3, 4): "c3_wires", (135, 21, 3, 4): "scale2_wires", (135, 22, 3, 4): "scale4_wires",
              (135, 23, 3, 4): "c5_wires", (135, 7, 3, 4): "tiny_wires", (135, 6, 3, 4): "tiny_strong_wires"}

    def timed_steps(job, cols, log_rows):
        """W untimed warm-up steps, then exactly K steps between barrier + device synchronisation on both sides; the cap of EVERY
        timed step is kept (512 bytes each, an asynchronous device copy) and compared, after the timed region, with the oracle's
        golden cap of the same synthetic columns (tests/golden/commit_caps.json: the headline shape and the 2 / 4 / 8-GPU
        weak-scaling shapes, C5 at 8).  Returns (seconds, kernel profile, cap_checked, golden record)."""
        cap_log = eng.mem.zeros(max(args.steps, 1), (1 << cap) * 4)

        def step(i=None):
            job.run(cols)
            if i is not None:
                if emu:
                    cap_log[i][...] = eng.host(job.cap).reshape(-1)
                else:
                    cap_log[i].copy_(job.cap.reshape(-1))

        for _ in range(args.warmup):
            step()
        eng.profile(True)
        eng.profile_results(reset=True)
        if dist:
            dist.barrier()
        device_sync()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        device_sync()
        if dist:
            dist.barrier()
        device_sync()
        dt = time.perf_counter() - t0
        prof = eng.profile_results(reset=True)
        eng.profile(False)
        gname = gnames.get((W, log_rows, rb, cap))
        g = golden(gname) if gname else None
        cap_checked = None
        if g is not None:
            caps_host = eng.host(cap_log).reshape(args.steps, 1 << cap, 4).tolist()
            bad = [i for i, c in enumerate(caps_host) if c != g["cap"]]
            cap_checked = not bad and args.steps > 0
            if bad:
                raise SystemExit("bench: the Merkle cap of timed step(s) %s differs from the oracle's golden cap" % bad[:8])
        return dt, prof, cap_checked, g

    def max_over_ranks(dt):
        if not dist:
            return dt
        t = torch.tensor([dt], dtype=torch.float64, device=eng.mem.device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    dt, prof, cap_checked, g = timed_steps(job, cols, log_n)
    rank_rows = None
    if dist:
        dt_local, dt = dt, max_over_ranks(dt)
        # one row per rank: its own wall time, the span of its exchanges on the communication stream, the sum of its compute
        # kernels and the kernel table -- a slow curve is diagnosed from ONE driver run
        ex_ms = prof.get("exchange", {"ms": 0.0})["ms"] / args.steps
        comp_ms = sum(v["ms"] for k, v in prof.items() if k != "exchange") / args.steps
        mine = {"rank": rank, "device": "emulated device %d" % local_rank if emu else torch.cuda.get_device_name(local_rank), "transport": job.comm.transport, "ms_per_step": dt_local / args.steps * 1e3,
                "exchange_ms": ex_ms, "compute_ms": comp_ms, "preflight": preflight,
                "kernels": {k: round(v["ms"] / max(v["launches"], 1), 4) for k, v in prof.items()}}
        rank_rows = [None] * world
        dist.all_gather_object(rank_rows, mine)

    # the strong-scaling companion of a weak-scaling run: the SAME total size at every --gpus (2^log_n rows in all: the C3 wires
    # commit split over the ranks), timed and cap-checked the same way on the same communicator, reported under "strong_scaling"
    strong = None
    if world > 1 and not args.strong and os.environ.get("P2HOT_BENCH_NO_STRONG") != "1":
        del cols
        try:  # (a shape this world size cannot shard is the same ValueError on every rank: the companion is dropped, the line stays)
            job2, cols2 = make_job(args.log_n, comm=job.comm)
        except ValueError as ex:
            strong, job2 = {"scaling": "strong", "skipped": str(ex)}, None
        if job2 is not None:
            dt2, _prof2, cap2, _g2 = timed_steps(job2, cols2, args.log_n)
            dt2 = max_over_ranks(dt2)
            strong = {"scaling": "strong", "workload": "PolynomialBatch::from_values, W=%d, 2^%d rows IN TOTAL split over %d ranks, rate 1/%d, cap_height %d"
                                                       % (W, args.log_n, world, 1 << rb, cap),
                      "ms_per_step": dt2 / args.steps * 1e3, "value": W * (1 << (args.log_n + rb)) / (dt2 / args.steps) / 1e9, "unit": "GFE/s",
                      "steps": args.steps, "cap_checked": cap2, "caps_checked": args.steps if cap2 else 0}
            del job2, cols2

    if rank == 0:
        ms = dt / args.steps * 1e3
        fe = W * N
        # dominant kernel: the Poseidon leaf sponge over this rank's rows
        rows_local = N // world
        ab = algorithmic_bytes(W, log_n, rb)
        # (the emulated runtime's events carry no time: a floor keeps the emulated line's arithmetic finite)
        kern = {k: {"ms_per_launch": max(v["ms"] / max(v["launches"], 1), 1e-9 if emu else 0.0), "launches": v["launches"]} for k, v in prof.items()}
        for agg in ("strided", "contig"):  # the aggregated keys of rounds 1-2: all passes of that kind in a step
            parts = [v for k, v in kern.items() if k.startswith("ntt_") and k.endswith("_" + agg)]
            if parts:
                ln = sum(v["launches"] for v in parts)
                kern["ntt_pass_" + agg] = {"ms_per_launch": sum(v["ms_per_launch"] * v["launches"] for v in parts) / ln, "launches": ln}
        h = kern.get("hash_leaves", {"ms_per_launch": float("nan"), "launches": args.steps})
        # the sponge runs once per coset block when it is overlapped with the next block's LDE
        launches_per_step = max(1, h["launches"] // args.steps)
        rows_per_launch = rows_local // launches_per_step
        hash_bytes = 8 * W * rows_per_launch + 32 * rows_per_launch
        achieved = hash_bytes / (h["ms_per_launch"] * 1e-3) / 1e9
        perms = rows_per_launch * ((W + 7) // 8)
        vl = valu_line(pmc_entry(W, log_n, rb, cap, world, "hash_leaves_kernel", "ColMajorReader"), launches_per_step, h, perms)
        out = {
            "metric": "LDE+Poseidon-commit GFE/s", "value": fe / (dt / args.steps) / 1e9, "unit": "GFE/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "u64 (Goldilocks)",
            "data": "synthetic (splitmix64 columns generated on device)" if not emu else "synthetic; EMULATED devices (CPU test tier): timings mean nothing",
            **({"emulated": True} if emu else {}),
            "config": {"workload": "PolynomialBatch::from_values, W=%d, 2^%d rows, rate 1/%d (N=2^%d), cap_height %d, "
                                   "PoseidonGoldilocksConfig (%s)"
                                   % (W, log_n, 1 << rb, log_n + rb, cap, "--strong: the same total size at every --gpus" if args.strong else
                                      "C3 wires commit at --gpus 1; +1 bit of rows per doubling of GPUs"),
                       "exchange": None if world == 1 else job.comm.exchange,
                       "transport": None if world == 1 else {"rccl": "RCCL inside libp2hot on the library's communication stream: %s, chosen by the preflight's micro-timing of both forms (P2HOT_EXCHANGE pins one)"
                                                                    % ("ncclAllGather (pipelined column chunks through a chunk-major staging block)" if job.comm.exchange == "allgather" else "one grouped ncclBroadcast per slice"),
                                                                    "torch": "torch.distributed all_gather on device buffers (fallback: librccl could not be bound)",
                                                                    "gloo": "gloo through host staging (ranks sharing a GPU: a functional run, not a scaling number)"}.get(job.comm.transport, job.comm.transport),
                       "sharding": "none" if world == 1 else "LDE cosets over %d ranks; iNTT column-sharded, coefficients all-gathered in "
                                   "async column chunks overlapped with the NTTs; RCCL all-gather of %s" % (world, "digests + cap" if gather_digests else "the cap (digests stay with the row owner)")},
            "roofline": roofline_line(vl, achieved, hash_bytes, perms, h, launches_per_step,
                                      pmc_entry(W, log_n, rb, cap, world, "hash_leaves_kernel", "ColMajorReader")),
            "valu": vl,
            "roofline_ntt": ntt_roofline(kern, W, n if world == 1 else n // world, rows_local, args.steps,
                                         lambda *needles: pmc_entry(W, log_n, rb, cap, world, *needles)),
            "cap_checked": cap_checked,
            "caps_checked": args.steps if cap_checked else 0,
            "box": None if emu else box_report(torch, local_rank),
            **({"ranks": rank_rows} if rank_rows else {}),
            **({"strong_scaling": strong} if strong else {}),
            "kernels": kern,
            "algorithmic_bytes_per_step": ab,
            "commit_hbm_frac": ab["total"] / world / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
        }
        if world == 1 and not args.no_extra and not emu:
            del job, cols
            torch.cuda.empty_cache()
            out["other_configs"] = other_configs(eng, torch, only=args.extra_only)
        if world == 1 and not args.no_cpu_baseline and not emu:
            out["cpu_baseline"] = cpu_baseline(W, log_n, rb, cap, golden_cap=g["cap"] if g else None)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        if world > 1 and os.environ.get("P2HOT_BENCH_NO_GROUP_PATH") != "1":
            # (the other ranks have released their devices and wait on the host for the store key set below)
            job.comm.close()
            del job
            if not emu:
                torch.cuda.empty_cache()
            out["group_per_proof_path"] = group_per_proof_path(eng, torch, world, emu, device_sync)
        print(json.dumps(out))
    if dist and world > 1 and os.environ.get("P2HOT_BENCH_NO_GROUP_PATH") != "1":
        # the per-proof path over a p2hot_group is ONE process driving every GPU (rank 0 above): the other ranks free their device
        # memory and wait on the HOST (the rendezvous store; a device-side barrier would spin on the GPUs rank 0 is using)
        store = torch.distributed.distributed_c10d._get_default_store()
        if rank == 0:
            store.set("p2hot_group_path_done", "1")
        else:
            if "job" in dir():
                job.comm.close()
                del job
            if not emu:
                torch.cuda.empty_cache()
            import datetime
            store.wait(["p2hot_group_path_done"], datetime.timedelta(minutes=30))  # (rank 0: a single-GPU proof, the group's RCCL init, three group proofs)
    if dist:
        dist.barrier()
        if "job" in dir():
            job.comm.close()  # the library's RCCL communicator, before torch tears its own down
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
