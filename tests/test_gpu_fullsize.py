"""BASELINE.json's full-size configurations on the MI355X (-m gpu): bit-exact against the oracle where
the oracle finishes in seconds (C2, the FRI commit phases), and through size-independent properties
at C3 / C4 size (Merkle paths verify to the cap, LDE rows equal direct polynomial evaluation,
coefficients interpolate the inputs, coset-sharded assembly equals the monolithic tree)."""
import numpy as np
import pytest

from tests import pyref
from tests.conftest import P, rand_field

pytestmark = pytest.mark.gpu

G = pyref.G


def _check_properties(gpu, ora, cols_dev, r, W, log_n, rb, cap, rng, n_paths=24, n_evals=3):
    N, n = 1 << (log_n + rb), 1 << log_n
    digests, capv = gpu.host(r["digests"]), gpu.host(r["cap"])
    # (a) every sampled Merkle path verifies against the cap (merkle_tree.rs:253-267)
    idx = np.unique(np.concatenate([[0, N - 1], rng.integers(0, N, size=n_paths)])).astype(np.uint64)
    rows = gpu.host(gpu.gather_rows(r["lde"], idx))
    for i, row in zip(idx, rows):
        proof = ora.merkle_prove(int(i), N, cap, digests)
        assert ora.merkle_verify(row, int(i), capv, proof), int(i)
    # (b) leaf row L, column c is p_c(g * w_N^bitrev(L)) (oracle.rs:142-147; polynomial/mod.rs:486-491)
    wN = pyref.root_of_unity(log_n + rb)
    for _ in range(n_evals):
        q = int(rng.integers(0, len(idx)))
        c = int(rng.integers(0, W))
        L = int(idx[q])
        coeffs = [int(x) for x in gpu.host(r["coeffs"][c])]
        x = G * pow(wN, pyref.bitrev(L, log_n + rb), P) % P
        assert pyref.eval_poly(coeffs, x) == int(rows[q][c]), (L, c)
        # (c) the coefficients interpolate the input values on H_n (fft.rs:233-239)
        i = int(rng.integers(0, n))
        assert pyref.eval_poly(coeffs, pow(pyref.root_of_unity(log_n), i, P)) == int(gpu.host(cols_dev[c, i:i + 1])[0]) % P


def _sha_device_matrix(gpu, mat):
    """SHA-256 of a device matrix [rows][cols] in row order, copied back one row (= one column of the LDE) at a time"""
    import hashlib
    h = hashlib.sha256()
    for r in range(mat.shape[0]):
        h.update(gpu.host(mat[r]).tobytes())
    return h.hexdigest()


@pytest.mark.parametrize("name", ["c2_wires", "c3_wires", "c3_constants_sigmas", "c3_zs_partial_products", "c3_quotient_chunks", "c4_fibonacci_trace"])
def test_baseline_commits_bit_exact_vs_oracle_goldens(gpu, name):
    """Every full-size commit of BASELINE.json's configs -- C2, the three C3 commits (W = 135 / 20 from_values, 16
    from_coeffs; 2^20 rows, rate 1/8, cap 4: W = 135 is the bench workload) and the C4 trace commit (W = 2, 2^22 rows,
    rate 1/2) -- against the bytes the FAITHFUL CPU oracle produced for the same inputs (tests/golden/commit_caps.json,
    tools/gen_golden_caps.py): the cap, and SHA-256 of the coefficient matrix, of the whole digest array (reference
    layout) and of the whole LDE matrix (= MerkleTree::leaves, column-major committed order).
    Reference: fri/oracle.rs:57-112, hash/merkle_tree.rs:193-224."""
    import hashlib
    import json
    import os
    import torch
    from plonky2_amd.util.synthetic import fibonacci_trace, splitmix_columns_torch
    from tests.conftest import ROOT
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "commit_caps.json")))[name]
    W, log_n, rb, cap, is_values = g["W"], g["log_n"], g["rate_bits"], g["cap_height"], g["is_values"]
    if g["input"] == "splitmix":
        cols = splitmix_columns_torch(torch, gpu.mem.device, 0, W, 1 << log_n)
    else:
        cols = gpu.dev(fibonacci_trace(log_n))
    r = gpu.commit(cols, log_n, rb, cap, is_values)
    assert gpu.host(r["cap"]).tolist() == g["cap"], "Merkle cap differs from the oracle"
    assert hashlib.sha256(gpu.host(r["digests"]).tobytes()).hexdigest() == g["sha256_digests"], "digest array differs"
    co = gpu.host(r["coeffs"])
    assert hashlib.sha256(np.where(co >= np.uint64(P), co - np.uint64(P), co).tobytes()).hexdigest() == g["sha256_coeffs"]
    assert _sha_device_matrix(gpu, r["lde"]) == g["sha256_lde"], "LDE matrix (leaves) differs"
    del r, cols
    torch.cuda.empty_cache()


@pytest.mark.parametrize("name", ["c2_wires", "c3_quotient_chunks"])
def test_polynomial_batch_wire_bytes_vs_oracle_at_full_size(gpu, name):
    """SURVEY 8f-4 at real sizes: the reference's `write_polynomial_batch` byte stream (util/serialization/mod.rs:1744-1763 around
    write_merkle_tree :1417-1431: W, per polynomial its length and canonical coefficients, the leaf count, per leaf row its
    length and words IN COMMITTED ORDER, the digest array in the reference layout, the cap, degree_log, rate_bits, blinding) of
    the DEVICE-built batch -- host column pointers in through p2hot_commit (what the Rust shim calls), leaf rows back through
    p2hot_batch_rows, digests through p2hot_batch_digests, coefficients through p2hot_batch_coeffs -- against SHA-256 of the
    stream the ORACLE-built batch serialises to (tests/golden/commit_caps.json `sha256_polynomial_batch`, tools/gen_golden_caps.py
    --batch-bytes).  C2 wires (from_values, W = 135, 2^16 rows: 0.64 GB of stream) and the C3 quotient chunks (from_coeffs,
    W = 16, 2^20 rows: 1.5 GB).  What the Rust serializer will emit for these batches is thereby pinned; that it runs on the
    shim's structs needs rustc (integration/first_contact.sh)."""
    from plonky2_amd.fri.oracle import PolynomialBatch
    from plonky2_amd.util.synthetic import splitmix_columns_numpy
    from tests.wire_format import polynomial_batch_sha256
    g = _golden(name)
    assert g and g.get("sha256_polynomial_batch"), "run tools/gen_golden_caps.py --batch-bytes"
    W, log_n, rb, cap = g["W"], g["log_n"], g["rate_bits"], g["cap_height"]
    cols = splitmix_columns_numpy(0, W, 1 << log_n)
    build = PolynomialBatch.from_values if g["is_values"] else PolynomialBatch.from_coeffs
    b = build(cols, rb, False, cap, engine=gpu)
    assert (b.degree_log, b.rate_bits, b.blinding) == (log_n, rb, False)
    assert gpu.lib.p2hot_batch_width(b._owner.h) == W and gpu.lib.p2hot_batch_degree_log(b._owner.h) == log_n
    N, step = 1 << (log_n + rb), 1 << 16
    rows = (b._owner.rows(np.arange(r, r + step, dtype=np.uint64)) for r in range(0, N, step))
    got = polynomial_batch_sha256(b.polynomials, rows, b.merkle_tree.digests, b.merkle_tree.cap.entries, cap, b.degree_log, b.rate_bits, b.blinding)
    assert got == g["sha256_polynomial_batch"], "write_polynomial_batch bytes of the device-built batch differ from the oracle-built one"
    del b


def _golden(name):
    import json
    import os
    from tests.conftest import ROOT
    return json.load(open(os.path.join(ROOT, "tests", "golden", "commit_caps.json"))).get(name)


@pytest.mark.parametrize("name,lde_blocks", [("scale2_wires", (0, 7)), ("scale4_wires", (3,)), ("c5_wires", (5,))])
def test_weak_scaling_shapes_bit_exact_on_one_gpu(gpu, name, lde_blocks):
    """The shapes bench.py commits at --gpus 2 / 4 / 8 (W = 135 at 2^21 / 2^22 / 2^23 rows; the last is BASELINE C5:
    72.5 GB of LDE values, which one MI355X holds) as ONE context's commit: cap, SHA-256 of the coefficient matrix and of the
    whole digest array, and of sampled leaf blocks of the LDE matrix, against the oracle's golden bytes
    (tools/gen_golden_caps.py --streamed)."""
    import hashlib
    import torch
    from plonky2_amd.util.synthetic import splitmix_columns_torch
    g = _golden(name)
    if g is None:
        pytest.skip("no golden for " + name)
    W, log_n, rb, cap = g["W"], g["log_n"], g["rate_bits"], g["cap_height"]
    n = 1 << log_n
    need = (2 * W * n + W * (n << rb) + 8 * (n << rb)) * 8 * 1.15      # columns, coefficients, LDE, digests + slack
    torch.cuda.empty_cache()
    if torch.cuda.mem_get_info()[0] < need:
        pytest.skip("needs %.0f GB of free HBM" % (need / 2**30))
    cols = splitmix_columns_torch(torch, gpu.mem.device, 0, W, n)
    r = gpu.commit(cols, log_n, rb, cap, True)
    del cols
    assert gpu.host(r["cap"]).tolist() == g["cap"], "Merkle cap differs from the oracle"
    h = hashlib.sha256()
    nd = r["digests"].shape[0]
    for lo in range(0, nd, 1 << 24):     # 512 MB at a time
        h.update(gpu.host(r["digests"][lo:lo + (1 << 24)]).tobytes())
    assert h.hexdigest() == g["sha256_digests"], "digest array differs"
    h = hashlib.sha256()
    for c in range(W):
        co = gpu.host(r["coeffs"][c])
        h.update(np.where(co >= np.uint64(P), co - np.uint64(P), co).tobytes())
    assert h.hexdigest() == g["sha256_coeffs"]
    for b in lde_blocks:                 # leaf block b = LDE coset bitrev(b): rows [b n, (b + 1) n) of every column
        h = hashlib.sha256()
        for c in range(W):
            h.update(gpu.host(r["lde"][c, b * n:(b + 1) * n]).tobytes())
        assert h.hexdigest() == g["sha256_lde_blocks"][b], ("LDE block", b)
    del r
    torch.cuda.empty_cache()
    gpu.check(gpu.lib.p2hot_ctx_trim(gpu.ctx))


@pytest.mark.parametrize("name,world,by_columns", [("scale2_wires", 2, False), ("c5_wires", 8, False), ("scale2_wires", 4, True)])
def test_weak_scaling_shapes_as_rank_groups_on_one_gpu(gpu, name, world, by_columns):
    """The same shapes the way the scaling runs shard them -- `world` ranks (all on device 0: the box has one GPU; C5 as
    its 8 ranks), coset-sharded (and once column-sharded), host pointers in, through p2hot_group_commit: the assembled cap,
    digest array and coefficients equal the oracle's golden bytes, and an owner serves rows + Merkle paths that verify"""
    import hashlib
    import torch
    from oracle import p2oracle as ora
    from plonky2_amd.distributed import GroupCommit
    from plonky2_amd.util.synthetic import splitmix_columns_numpy
    g = _golden(name)
    if g is None:
        pytest.skip("no golden for " + name)
    W, log_n, rb, cap = g["W"], g["log_n"], g["rate_bits"], g["cap_height"]
    torch.cuda.empty_cache()
    gpu.check(gpu.lib.p2hot_ctx_trim(gpu.ctx))
    n = 1 << log_n
    # every rank holds all coefficients and its row block of the LDE (column mode: its columns' whole LDE and the row block)
    need = (world * 2 * W * n + (2 if by_columns else 1) * W * (n << rb) + 8 * (n << rb)) * 8 * 1.1
    if torch.cuda.mem_get_info()[0] < need:
        pytest.skip("needs %.0f GB of free HBM" % (need / 2**30))
    cols = splitmix_columns_numpy(0, W, 1 << log_n)
    grp = GroupCommit(gpu.lib, world, [0] * world)
    r = grp.commit(cols, rb, cap, True, want_leaves=False, want_digests=True, by_columns=by_columns)
    del cols
    assert r["cap"].tolist() == g["cap"]
    assert hashlib.sha256(r["digests"].tobytes()).hexdigest() == g["sha256_digests"]
    assert hashlib.sha256(r["coeffs"].tobytes()).hexdigest() == g["sha256_coeffs"]
    N = 1 << (log_n + rb)
    xs = [0, N - 1, N // world, N // world - 1, N // 2 + 12345]
    rows, paths = r["open"](xs)
    for x, row, path in zip(xs, rows, paths):
        assert ora.merkle_verify(row, x, r["cap"], path), x
    r["free"]()
    grp.close()


def test_c3_wires_host_pointer_commit_equals_golden(gpu):
    """the same C3 wires commit through the host-pointer entry point the Rust shim calls (p2hot_commit, pipelined
    PCIe copies): cap and digest array equal the oracle's golden bytes"""
    import ctypes as C
    import hashlib
    import json
    import os
    from plonky2_amd.util.synthetic import splitmix_columns_numpy
    from tests.conftest import ROOT
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "commit_caps.json")))["c3_wires"]
    W, log_n, rb, cap = g["W"], g["log_n"], g["rate_bits"], g["cap_height"]
    cols = splitmix_columns_numpy(0, W, 1 << log_n)
    ptrs = (C.c_void_p * W)(*[cols[c].ctypes.data for c in range(W)])
    nd = gpu.num_digests(log_n + rb, cap)
    digests = np.zeros((nd, 4), dtype=np.uint64)
    capv = np.zeros((1 << cap, 4), dtype=np.uint64)
    coeffs = np.zeros((W, 1 << log_n), dtype=np.uint64)
    gpu.check(gpu.lib.p2hot_commit(gpu.ctx, ptrs, W, log_n, rb, cap, 1, 0, coeffs.ctypes.data, None, digests.ctypes.data,
                                   capv.ctypes.data, None))
    assert capv.tolist() == g["cap"]
    assert hashlib.sha256(digests.tobytes()).hexdigest() == g["sha256_digests"]
    assert hashlib.sha256(coeffs.tobytes()).hexdigest() == g["sha256_coeffs"]
    gpu.check(gpu.lib.p2hot_ctx_trim(gpu.ctx))


def test_c2_commit_bit_exact_vs_oracle(gpu, ora):
    """C2: W = 135, 2^16 rows, rate 1/8, cap 4 -- everything compared with the oracle"""
    rng = np.random.default_rng(2)
    W, log_n, rb, cap = 135, 16, 3, 4
    cols = rand_field(rng, W, 1 << log_n, noncanonical=True)
    r = gpu.commit(gpu.dev(cols), log_n, rb, cap, True, want_leaves=True)
    o = ora.commit(cols, rb, cap, True)
    assert (gpu.host(r["coeffs"]) == o["coeffs"]).all()
    assert (gpu.host(r["cap"]) == o["cap"]).all()
    assert (gpu.host(r["digests"]) == o["digests"]).all()
    assert (gpu.host(r["leaves"]) == o["leaves"]).all()


@pytest.mark.parametrize("W,is_values", [(135, True), (20, True), (16, False)])
def test_c3_commit_properties(gpu, ora, W, is_values):
    """C3: the three commits of a 2^20-gate proof (wires, Zs+partial products, quotient chunks)"""
    rng = np.random.default_rng(3 + W)
    log_n, rb, cap = 20, 3, 4
    cols = gpu.dev(rand_field(rng, W, 1 << log_n))
    r = gpu.commit(cols, log_n, rb, cap, is_values)
    _check_properties(gpu, ora, cols if is_values else r["coeffs"], r, W, log_n, rb, cap, rng,
                      n_evals=3 if is_values else 0)
    if W == 20:
        # coset-sharded assembly (the multi-GPU unit) gives the same digests and cap
        n = 1 << log_n
        nd = gpu.num_digests(log_n + rb, cap)
        dig, capbuf = gpu.mem.zeros(nd, 4), gpu.mem.zeros(1 << cap, 4)
        for part in range(8):
            gpu.commit(cols, log_n, rb, cap, is_values, row_begin=part * n, row_count=n, digests=dig, cap=capbuf)
        assert (gpu.host(dig) == gpu.host(r["digests"])).all() and (gpu.host(capbuf) == gpu.host(r["cap"])).all()


def test_c4_starky_trace_commit_properties(gpu, ora):
    """C4: Fibonacci trace, 2 columns x 2^22 rows, rate 1/2, cap 4 (leaves are hash_or_noop copies)"""
    rng = np.random.default_rng(4)
    log_n, rb, cap = 22, 1, 4
    n = 1 << log_n
    # x0 = 0, x1 = 1; row i = (F_i, F_{i+1}) mod P (starky/src/fibonacci_stark.rs:47-57), generated on the host
    fib = np.zeros((2, n), dtype=np.uint64)
    a, b = 0, 1
    for i in range(n):
        fib[0, i], fib[1, i] = a, b
        a, b = b, (a + b) % P
    cols = gpu.dev(fib)
    r = gpu.commit(cols, log_n, rb, cap, True)
    _check_properties(gpu, ora, cols, r, 2, log_n, rb, cap, rng, n_paths=16, n_evals=2)


@pytest.mark.parametrize("log_n,rb,arity", [(20, 3, [4, 4, 4, 4]), (22, 1, [4, 4, 4, 4])])
def test_fri_commit_phase_full_size_vs_oracle(gpu, ora, log_n, rb, arity):
    """C3 / C4 FRI commit phases (N = 2^23, arity 16): caps, betas, final polynomial and trees vs the oracle"""
    from plonky2_amd.fri.prover import fri_committed_trees
    from plonky2_amd.iop.challenger import Challenger
    rng = np.random.default_rng(50 + log_n)
    n = 1 << log_n
    co = rand_field(rng, n, 2)
    pad = np.zeros((n << rb, 2), dtype=np.uint64)
    pad[:n] = co
    c, oc = Challenger(gpu), ora.Challenger()
    trees, final, betas = fri_committed_trees(co, c, rb, 4, arity, engine=gpu)
    o = ora.fri_commit(pad, rb, 4, arity, oc)
    assert (betas == o["betas"]).all() and (final == o["final"]).all()
    for i, t in enumerate(trees):
        assert (t.cap.entries == o["caps"][i]).all(), i
        assert (np.asarray(t.digests).reshape(-1, 4) == o["digests"][i]).all(), i
        assert (t.leaves == o["leaves"][i]).all(), i
    assert c.get_n_challenges(2) == oc.get_n_challenges(2)


@pytest.mark.parametrize("world,W,log_n,chunks,by_columns,rb", [(2, 9, 12, 3, False, 3), (4, 20, 14, 8, False, 3), (8, 135, 12, 4, False, 3),
                                                                (4, 20, 13, 1, True, 3), (8, 135, 11, 1, True, 3),
                                                                # more ranks than cosets (starky's rate 1/2): sub-cosets of H_n
                                                                (8, 2, 16, 1, False, 1), (4, 135, 13, 4, False, 1), (8, 20, 12, 2, False, 0)])
def test_group_commit_ranks_on_one_gpu(gpu, ora, world, W, log_n, chunks, by_columns, rb):
    """p2hot_group_commit on the real GPU: `world` ranks of ONE process, all on device 0 (the box has one GPU), exchanging
    by copies on their communicator streams; coefficients, leaves, the whole digest array and the cap equal the oracle's,
    and the owner of a row serves its Merkle path"""
    from plonky2_amd.distributed import GroupCommit
    rng = np.random.default_rng(9 + world)
    cap = 4
    cols = rand_field(rng, W, 1 << log_n, noncanonical=True)
    o = ora.commit(cols, rb, cap, True)
    g = GroupCommit(gpu.lib, world, [0] * world)
    assert not g.uses_rccl          # a repeated device cannot carry an RCCL communicator
    for _ in range(2):
        r = g.commit(cols, rb, cap, True, want_leaves=True, pipeline_chunks=chunks, by_columns=by_columns)
        assert (r["coeffs"] == o["coeffs"]).all() and (r["cap"] == o["cap"]).all()
        assert (r["digests"] == o["digests"]).all() and (r["leaves"] == o["leaves"]).all()
        N = 1 << (log_n + rb)
        xs = [0, N - 1, N // world, N // world - 1] + [int(x) for x in rng.integers(0, N, 8)]
        rows, paths = r["open"](xs)
        for x, row, path in zip(xs, rows, paths):
            assert (row == o["leaves"][x]).all() and ora.merkle_verify(row, x, o["cap"], path)
        r["free"]()
    g.close()


@pytest.mark.parametrize("world,widths,log_n,arity", [(2, [9, 4], 10, [4]), (8, [20, 16], 12, [4, 4])])
def test_group_prove_openings_on_one_gpu(gpu, world, widths, log_n, arity):
    """p2hot_group_eval_openings / p2hot_group_prove_openings on the real GPU (all ranks on device 0): the proof over the
    sharded oracles equals the single-context proof buffer by buffer, transcripts included"""
    from tests.test_distributed import _group_proof_equals_single_context
    _group_proof_equals_single_context(gpu.lib, gpu, world, widths, log_n, 3, 4, arity)


@pytest.mark.parametrize("chunks,gather", [(1, True), (5, False)])
def test_rccl_communicator_one_rank(gpu, ora, chunks, gather):
    """The in-library RCCL transport on the real GPU with a one-rank communicator (all this box can host):
    ncclGetUniqueId / ncclCommInitRank / grouped ncclBroadcast on the communicator stream, ordered against the compute
    stream by events -- the same calls p2hot_commit_sharded_dev issues at 2 / 4 / 8 ranks; result vs the oracle"""
    from plonky2_amd.distributed import ShardedCommit
    rng = np.random.default_rng(77)
    W, log_n, rb, cap = 11, 12, 3, 4
    cols = rand_field(rng, W, 1 << log_n)
    o = ora.commit(cols, rb, cap, True)
    job = ShardedCommit(gpu, W, log_n, rb, cap, is_values=True, rank=0, world=1, dist=None, want_leaves=True,
                        pipeline_chunks=chunks, gather_digests=gather, transport="rccl")
    for _ in range(2):
        out = job.run(gpu.dev(cols))
        gpu.sync()
        assert (gpu.host(out["coeffs"]) == o["coeffs"]).all() and (gpu.host(out["cap"]) == o["cap"]).all()
        assert (gpu.host(out["digests"]) == o["digests"]).all() and (gpu.host(out["leaves"]) == o["leaves"]).all()
    job.comm.close()


def test_rccl_binding_without_torch(gpu):
    """The OTHER deployment mode of the in-library RCCL transport: a patched plonky2 is a process without torch, so libp2hot binds
    /opt/rocm's librccl by path instead of the copy PyTorch ships (host_multi.hpp rccl::api: RTLD_NOLOAD first).  A subprocess that
    never imports torch (tests/helpers/rccl_no_torch.py: ctypes + numpy) creates the one-rank communicator, runs the preflight
    all-gather and a sharded commit against the oracle, and reports the file and ncclGetVersion it bound; this process (torch
    loaded first: the `gpu` fixture) reports its own binding through the same call.  Both must bind and work; the two paths are
    printed so that GPUTEST's log says which RCCL build each mode used."""
    import json
    import os
    import subprocess
    import sys
    from plonky2_amd.distributed import rccl_info
    from tests.conftest import ROOT
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "rccl_no_torch.py")], capture_output=True, text=True,
                       timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, "no-torch RCCL process failed:\n" + r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]   # (RCCL prints its version banner on stdout too)
    assert lines, "no JSON line from the no-torch RCCL process:\n" + r.stdout[-2000:] + r.stderr[-2000:]
    rec = json.loads(lines[-1])
    assert rec["checked"] and rec["torch_imported"] is False and rec["exchange_mode"] in (0, 1)
    assert rec["rccl"]["path"] and rec["rccl"]["version"] > 0
    assert "torch" not in rec["rccl"]["path"], "a process without torch must not have bound PyTorch's bundled RCCL: %r" % rec
    here = rccl_info(gpu.lib)            # this process: torch was imported before the library looked for RCCL
    assert here is not None and here["path"] and here["version"] > 0
    print("RCCL bound without torch: %s (version %d); with torch loaded first: %s (version %d)"
          % (rec["rccl"]["path"], rec["rccl"]["version"], here["path"], here["version"]))


def test_device_buffer_transport_hooks(gpu):
    """The two caller-side p2hot_allgather_fn hooks that take DEVICE buffers -- over torch.distributed's own RCCL communicator
    (the fallback when the library cannot bind librccl) and over gloo with host staging (ranks sharing a GPU) -- wrap raw
    device ranges without copies and leave a one-rank buffer intact; the multi-rank flow through the second one is what
    tools/gpu_multiproc.sh runs on this box"""
    import ctypes as C
    import torch
    import torch.distributed as dist
    from plonky2_amd import distributed as D
    buf = torch.arange(1 << 12, dtype=torch.int64, device=gpu.mem.device)
    want = buf.clone()
    offs = (C.c_size_t * 1)(0)
    for backend, make, port in (("nccl", D._torch_transport, 29561), ("gloo", D._gloo_device_transport, 29562)):
        kw = {"device_id": gpu.mem.device} if backend == "nccl" else {}
        dist.init_process_group(backend, init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, **kw)
        try:
            cb = make(dist, 0)
            assert cb(None, buf.data_ptr() + 64, offs, 1, buf.numel() * 8 - 64, None) == 0
            torch.cuda.synchronize()
            assert torch.equal(buf, want)
        finally:
            dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("is_values,want_leaves", [(True, False), (False, True)])
def test_host_pointer_commit_pipelined_path(gpu, ora, is_values, want_leaves):
    """p2hot_commit above its pipelining threshold (column blocks uploaded on the side stream beside the transforms,
    coefficient blocks downloaded beside the leaf sponge): same bytes as the oracle, and as the small-batch path"""
    import ctypes as C
    rng = np.random.default_rng(4242)
    W, log_n, rb, cap = 37, 17, 3, 4          # W * n = 2^22.2 elements: 3 column blocks, the last one partial
    n, N = 1 << log_n, 1 << (log_n + rb)
    cols = [rand_field(rng, n, noncanonical=True) for _ in range(W)]
    ptrs = (C.c_void_p * W)(*[c.ctypes.data for c in cols])
    coeffs = np.zeros((W, n), dtype=np.uint64)
    leaves = np.zeros((N, W), dtype=np.uint64) if want_leaves else None
    nd = gpu.num_digests(log_n + rb, cap)
    digests = np.zeros((nd, 4), dtype=np.uint64)
    capv = np.zeros((1 << cap, 4), dtype=np.uint64)
    handle = C.c_void_p()
    gpu.check(gpu.lib.p2hot_commit(gpu.ctx, ptrs, W, log_n, rb, cap, 1 if is_values else 0, 0, coeffs.ctypes.data,
                                   leaves.ctypes.data if want_leaves else None, digests.ctypes.data, capv.ctypes.data,
                                   C.byref(handle)))
    o = ora.commit(np.stack(cols), rb, cap, is_values)
    assert (coeffs == o["coeffs"] % np.uint64(P)).all()
    assert (digests == o["digests"]).all() and (capv == o["cap"]).all()
    if want_leaves:
        assert (leaves == o["leaves"]).all()
    idx = np.array([0, 77, N - 1], dtype=np.uint64)
    rows = np.zeros((3, W), dtype=np.uint64)
    gpu.check(gpu.lib.p2hot_batch_rows(handle, idx.ctypes.data, 3, rows.ctypes.data))
    assert (rows == o["leaves"][idx.astype(np.int64)]).all()
    gpu.lib.p2hot_batch_free(handle)


@pytest.mark.gpu
def test_commit_many_upload_in_staging_slices(gpu):
    """p2hot_commit_many of more short host columns than the 64 MB pinned staging block holds (32 recursion-size proofs:
    4320 vectors of 32 KB = 135 MB): the block's halves take slices in turn; every proof's cap equals the single-proof
    commitment of the same columns and distinct proofs stay distinct (fri/oracle.rs:57-79)"""
    import ctypes as C
    import numpy as np
    from plonky2_amd.util.synthetic import splitmix_columns_numpy
    eng = gpu
    M, W, log_n, rb, cap = 32, 135, 12, 3, 4
    n = 1 << log_n
    sets = [splitmix_columns_numpy(1000 * (m % 3), W, n) for m in range(M)]  # three different proofs, repeated
    ptrs = (C.c_void_p * (M * W))(*[sets[m][e].ctypes.data for m in range(M) for e in range(W)])
    caps = np.zeros((M, 1 << cap, 4), dtype=np.uint64)
    coeffs = np.zeros((M, W, n), dtype=np.uint64)
    eng.check(eng.lib.p2hot_commit_many(eng.ctx, ptrs, M, W, log_n, rb, cap, 1, coeffs.ctypes.data, None, caps.ctypes.data, None))
    for k in range(3):
        one = (C.c_void_p * W)(*[sets[k][e].ctypes.data for e in range(W)])
        c1, co1 = np.zeros((1 << cap, 4), dtype=np.uint64), np.zeros((W, n), dtype=np.uint64)
        eng.check(eng.lib.p2hot_commit(eng.ctx, one, W, log_n, rb, cap, 1, 0, co1.ctypes.data, None, None, c1.ctypes.data, None))
        for m in range(k, M, 3):
            assert (caps[m] == c1).all() and (coeffs[m] == co1).all(), (k, m)
    assert (caps[0] != caps[1]).any() and (caps[1] != caps[2]).any()


# ------------------------------------------------------------------ the widened rows at the sizes bench.py times them
@pytest.mark.parametrize("world", [2, 8])
def test_group_proof_path_k20_on_one_gpu(gpu, world):
    """`bench.py --gpus N`'s group_per_proof_path at FULL size with the ranks of the group all on device 0 (the box has one GPU;
    the exchanges are copies instead of RCCL): the four coset-sharded commits of per_proof_path_k20 from host columns (W = 84,
    135, 20, 16 at 2^20 rows), p2hot_group_eval_openings and p2hot_group_prove_openings against the same oracle record as the
    single-context path -- caps, 275 opening values, FRI caps, final_poly, PoW witness, query indices, FriProof bytes, transcript.
    World 8 = one LDE coset per rank (C5's arrangement)."""
    import torch
    from plonky2_amd.distributed import GroupCommit
    from plonky2_amd.util import proof_path as pp
    from plonky2_amd.util.synthetic import path_instance, splitmix_columns_numpy
    name = "per_proof_path_k20"
    inst, g = path_instance(name), pp.golden(name)
    single = pp.run_path(gpu, pp.PathInputs(gpu, inst), sync=torch.cuda.synchronize, keep=True)
    zs, chunks = single["zs"], single["chunks"]
    assert pp.compare_with_golden(single, g) == []
    del single
    torch.cuda.empty_cache()
    n = 1 << inst["log_n"]
    group = GroupCommit(gpu.lib, world, [0] * world)
    try:
        res = pp.run_group_path(group, inst, splitmix_columns_numpy(inst["cs_seed"], inst["cs_width"], n),
                                splitmix_columns_numpy(inst["wires_seed"], inst["wires_width"], n), zs, chunks)
    finally:
        group.close()
    assert pp.compare_with_golden(res, g) == []
    torch.cuda.empty_cache()
    gpu.check(gpu.lib.p2hot_ctx_trim(gpu.ctx))


@pytest.mark.parametrize("name", ["per_proof_path_k20", "per_proof_path_starky_k22"])
def test_proof_path_full_size_vs_oracle_records_and_verifier(gpu, ora, name):
    """bench.py's per_proof_path_k20 (2^20 rows, 80 routed wires, degree 8, 2 challenges; 4 oracles, 255 polynomials at zeta, 2 at
    the second point, arity [4,4,4,4], 28 queries) and per_proof_path_starky_k22 (C4: 2^22 rows, rate 1/2, 84 queries), every stage
    against the CPU oracle's bytes for the same instance (tests/golden/path_goldens.json, tools/gen_golden_path.py):
      p2hot_partial_products     SHA-256 of the whole [20][2^20] Zs + partial-products matrix   (plonk/prover.rs:392-449)
      p2hot_quotient_polys       SHA-256 of the quotient values at all 2^23 points and of the 16 chunk polynomials
                                 (plonk/prover.rs:609-815, vanishing_poly.rs:167-330)
      p2hot_eval_openings        every opening value (255 at zeta, 20 at the second point)      (plonk/proof.rs:314-345)
      p2hot_prove_openings       commit-phase caps, final_poly, PoW witness, query indices, SHA-256 of the FriProof's wire bytes
                                 (fri/oracle.rs:176-237, fri/prover.rs:24-258)
    and the device-made proof VERIFIES under the restated reference verifier (oracle/fri_verifier.py) and fails under six kinds of
    tampering.  For k20 the gate-sum input of the quotient is exercised at full size as well: see below."""
    import torch
    from tests.test_proof_path import check_path
    inst, inp, res, g = check_path(gpu, ora, name, sync=torch.cuda.synchronize)
    if inst["kind"] != "plonk":
        return
    # with gate sums (the caller's reduce_with_powers of the gate constraint terms): the values are the golden-checked permutation
    # quotient plus alpha^K * gate_sums / Z_H (vanishing_poly.rs:326-330, prover.rs:733-754) -- on 4096 sampled points in python
    # integers --, and the chunk polynomials follow by linearity: chunks(with) - chunks(without) = coset_ifft(values(with) - values(without))
    from plonky2_amd.plonk.prover import compute_quotient_polys
    from plonky2_amd.util.synthetic import GENERATOR, splitmix_columns_numpy
    n, qdf, nc = 1 << inst["log_n"], inst["quotient_degree_factor"], len(inst["alphas"])
    qbits = (qdf - 1).bit_length()
    m = n << qbits
    gs = splitmix_columns_numpy(7000, nc, m)
    b_cs, b_w, b_z, _ = res["oracles"]
    cols_g, vals_g = compute_quotient_polys(b_w, b_cs, inst["num_constants"], b_z, inst["k_is"], qdf, inst["betas"], inst["gammas"],
                                            inst["alphas"], gate_sums=gs, want_values=True, engine=gpu)
    vals = res["quotient_values"]
    K = nc + nc * (-(-inst["num_routed"] // qdf))        # L_0 terms + (num_prods + 1) partial-product terms per challenge
    g_n, v = pow(GENERATOR, n, P), ora.root_of_unity(qbits)
    rng = np.random.default_rng(20)
    pts = np.unique(np.concatenate([[0, 1, m - 1], rng.integers(0, m, 4096)]))
    for a in range(nc):
        aK = pow(int(inst["alphas"][a]), K, P)
        for i in pts:
            zh = (g_n * pow(v, int(i) % (1 << qbits), P) - 1) % P
            extra = aK * int(gs[a, i]) % P * pow(zh, P - 2, P) % P
            assert int(vals_g[a, i]) == (int(vals[a, i]) + extra) % P, (a, int(i))
    got = cols_g.host()
    pp_ = np.uint64(P)
    for a in range(nc):
        d = np.where(vals_g[a] >= vals[a], vals_g[a] - vals[a], vals_g[a] + (pp_ - vals[a]))
        co = ora.coset_ifft(d) % pp_
        want = res["chunks"][a * qdf:(a + 1) * qdf].reshape(-1)
        s = want + co[:qdf * n]                                  # mod-P addition on u64s: wrapped sums are below P - 2^32 + ... ; redo in two steps
        s = np.where(s < want, s + np.uint64(0xFFFFFFFF), s)     # 2^64 = 2^32 - 1 (mod P)
        s = np.where(s >= pp_, s - pp_, s)
        assert (got[a * qdf:(a + 1) * qdf].reshape(-1) == s).all(), a
