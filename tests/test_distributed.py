"""The N>1 path: coset-sharded commit with all-gathers, world_size 2 and 4 over gloo on CPU.

Each rank drives the real library code path (plonky2_amd.distributed + libp2hot's C ABI) with the
kernels running under the test-only emulator; the assembled tree must equal the oracle's."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from tests.conftest import rand_field


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, log_n, rb, cap, is_values, chunks, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import p2oracle as ora
        from plonky2_amd.distributed import ShardedCommit
        from tests.emu_backend import emu_engine
        eng = emu_engine()
        rng = np.random.default_rng(1234)  # every rank derives the same full input
        cols = rand_field(rng, W, 1 << log_n)
        job = ShardedCommit(eng, W, log_n, rb, cap, is_values=is_values, rank=rank, world=world, dist=dist, want_leaves=True,
                            pipeline_chunks=chunks)
        c0, c1 = job.column_range
        job.comm.selftest(4096)  # the bench's preflight: a pattern slice per rank through the same exchange path
        for _ in range(2):  # buffers are reused across steps
            r = job.run(eng.dev(cols[c0:c1]))
        o = ora.commit(cols, rb, cap, is_values)
        r0, rc = job.plan.rows(rank)
        ok = bool((eng.host(r["coeffs"]) == o["coeffs"]).all() and (eng.host(r["digests"]) == o["digests"]).all()
                  and (eng.host(r["cap"]) == o["cap"]).all() and (eng.host(r["leaves"]) == o["leaves"][r0:r0 + rc]).all())
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,W,log_n,rb,cap,is_values,chunks", [
    (2, 5, 5, 3, 4, True, 1), (4, 7, 4, 3, 4, True, 1), (2, 3, 6, 1, 2, False, 1),
    # pipelined coefficient exchange: async chunked all-gathers overlapped with the iNTT / LDE of the other chunks
    (2, 11, 5, 3, 4, True, 4), (4, 135, 3, 3, 4, True, 4), (2, 7, 6, 1, 2, False, 3), (8, 20, 3, 3, 4, True, 2),
    # more ranks than LDE cosets (starky's rate 1/2, C4's shape): one sub-coset of H_n per rank
    (4, 2, 6, 1, 4, True, 1), (8, 3, 5, 1, 3, False, 2)])
def test_sharded_commit_gloo(world, W, log_n, rb, cap, is_values, chunks):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, W, log_n, rb, cap, is_values, chunks, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(world))
    assert res == [(r, True) for r in range(world)]


def _worker_sync_env(rank, world, port, sync, q):
    """P2HOT_SYNC_COLLECTIVES read from the environment (the switch bench.py's users flip): one synchronous coefficient
    all-gather (1) or the pipelined chunks (unset); and a transport that delivers nothing is caught by the preflight"""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if sync:
        os.environ["P2HOT_SYNC_COLLECTIVES"] = "1"
    else:
        os.environ.pop("P2HOT_SYNC_COLLECTIVES", None)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import p2oracle as ora
        from plonky2_amd import _lib
        from plonky2_amd.distributed import Communicator, ShardedCommit
        from tests.emu_backend import emu_engine
        eng = emu_engine()
        W, log_n, rb, cap = 13, 5, 3, 4
        cols = rand_field(np.random.default_rng(4321), W, 1 << log_n)
        job = ShardedCommit(eng, W, log_n, rb, cap, is_values=True, rank=rank, world=world, dist=dist)
        ok = job.pipeline_chunks == (1 if sync else 8)
        job.comm.selftest()
        c0, c1 = job.column_range
        r = job.run(eng.dev(cols[c0:c1]))
        o = ora.commit(cols, rb, cap, True)
        ok = ok and bool((eng.host(r["cap"]) == o["cap"]).all() and (eng.host(r["coeffs"]) == o["coeffs"]).all())
        broken = Communicator(eng, rank, world, dist, transport="null")  # a hook that returns success without moving anything
        try:
            broken.selftest(64)
            ok = False
        except _lib.P2HotError as e:
            ok = ok and "did not receive" in str(e)
        broken.close()
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("sync", [False, True])
def test_sharded_commit_gloo_sync_and_pipelined_env(sync):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sync_env, args=(r, world, port, sync, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(world)) == [(r, True) for r in range(world)]


def _worker_sharded_digests(rank, world, port, W, log_n, rb, cap, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import p2oracle as ora
        from plonky2_amd.distributed import ShardedCommit
        from tests.emu_backend import emu_engine
        eng = emu_engine()
        rng = np.random.default_rng(99)
        cols = rand_field(rng, W, 1 << log_n)
        job = ShardedCommit(eng, W, log_n, rb, cap, is_values=True, rank=rank, world=world, dist=dist, gather_digests=False)
        c0, c1 = job.column_range
        r = job.run(eng.dev(cols[c0:c1]))
        o = ora.commit(cols, rb, cap, True)
        p = job.plan
        d0, d1 = rank * p.digests_per_rank, (rank + 1) * p.digests_per_rank
        ok = bool((eng.host(r["cap"]) == o["cap"]).all() and (eng.host(r["digests"])[d0:d1] == o["digests"][d0:d1]).all())
        # every rank answers the queries that fall into its rows; paths verify against the all-gathered cap
        xs = [int(x) for x in np.random.default_rng(5).integers(0, p.N, 12)] + [0, p.N - 1]
        mine = [x for x in xs if job.owner(x) == rank]
        rows, paths = job.prove_local(mine)
        for x, row, path in zip(mine, rows, paths):
            ok = ok and bool((row == o["leaves"][x]).all()) and bool((path == ora.merkle_prove(x, p.N, cap, o["digests"])).all())
            ok = ok and ora.merkle_verify(row, x, eng.host(r["cap"]), path)
        try:
            other = next(x for x in range(p.N) if job.owner(x) != rank)
            job.prove_local([other])
            ok = False
        except ValueError:
            pass
        q.put((rank, ok, len(mine)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,W,log_n,rb,cap", [(2, 9, 5, 3, 4), (4, 6, 4, 3, 3)])
def test_sharded_digests_owner_serves_paths_gloo(world, W, log_n, rb, cap):
    """gather_digests=False (SURVEY 8e collective 2, cap-only): digests stay with the rank that owns the rows"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sharded_digests, args=(r, world, port, W, log_n, rb, cap, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(world))
    assert [(r, ok) for r, ok, _ in res] == [(r, True) for r in range(world)]
    assert sum(m for _, _, m in res) == 14  # every query was answered by exactly one owner


def test_shard_plan():
    from plonky2_amd.distributed import ShardPlan
    p = ShardPlan(135, 23, 3, 4, 8)  # config C5
    assert p.rows(3) == (3 << 23, 1 << 23)
    assert p.cols_per_rank == 17 and p.columns(7) == (119, 135)
    assert sorted(sum((p.cosets(r) for r in range(8)), [])) == list(range(8))
    assert p.cosets(1) == [4]  # row block 1 is coset bitrev_3(1) = 4
    assert p.digests_per_rank * 8 == p.num_digests and p.cap_per_rank == 2
    q = ShardPlan(2, 22, 1, 4, 8)  # starky rate 1/2 (C4) on 8 GPUs: two cosets, each split into four sub-cosets of H_n
    assert q.sub_bits == 2 and q.rows_per_rank == 1 << 20 and q.cosets(1) == [4] and q.cosets(6) == [3]
    assert sorted(sum((q.cosets(r) for r in range(8)), [])) == list(range(8))
    with pytest.raises(ValueError):
        ShardPlan(2, 10, 3, 1, 4)  # fewer cap subtrees than ranks


@pytest.mark.parametrize("world,W,log_n,rb,cap,is_values,chunks", [
    (2, 5, 5, 3, 4, True, 1), (4, 135, 3, 3, 4, True, 4), (8, 20, 4, 3, 4, True, 2), (2, 3, 6, 1, 2, False, 3),
    (8, 9, 3, 3, 3, False, 8), (1, 6, 4, 2, 1, True, 2),
    (4, 2, 6, 1, 4, True, 1), (8, 2, 5, 1, 3, False, 2), (8, 5, 4, 0, 3, True, 3), (4, 135, 2, 1, 2, True, 4)])   # more ranks than cosets: sub-cosets
def test_group_commit_single_process(ora, world, W, log_n, rb, cap, is_values, chunks):
    """p2hot_group_commit -- ONE process driving `world` ranks (a patched plonky2's mode), host pointers in and out:
    coefficients, leaves, the full digest array and the cap equal the oracle's; rows / paths are served by the owning
    rank.  The emulator's ranks share one "device", so the exchange runs as copies (on a node: RCCL over xGMI)."""
    from plonky2_amd.distributed import GroupCommit
    from tests.emu_backend import emu_lib
    rng = np.random.default_rng(world * 1000 + W)
    cols = rand_field(rng, W, 1 << log_n, noncanonical=True)
    g = GroupCommit(emu_lib(), world, [0] * world)
    assert not g.uses_rccl
    for _ in range(2):  # pool blocks are reused by the second call
        r = g.commit(cols, rb, cap, is_values, want_leaves=True, pipeline_chunks=chunks)
        o = ora.commit(cols, rb, cap, is_values)
        assert (r["coeffs"] == o["coeffs"] % np.uint64(0xFFFFFFFF00000001)).all()
        assert (r["cap"] == o["cap"]).all()
        assert (r["digests"] == o["digests"]).all()
        assert (r["leaves"] == o["leaves"]).all()
        N = 1 << (log_n + rb)
        xs = [0, N - 1] + [int(x) for x in rng.integers(0, N, 6)]
        rows, paths = r["open"](xs)
        for x, row, path in zip(xs, rows, paths):
            assert (row == o["leaves"][x]).all()
            assert (path == ora.merkle_prove(x, N, cap, o["digests"])).all()
        with pytest.raises(Exception):
            r["open"]([N])
        r["free"]()
    g.close()


def test_group_and_comm_argument_errors():
    import ctypes as C
    from plonky2_amd import _lib
    from plonky2_amd.distributed import GroupCommit
    from tests.emu_backend import emu_engine, emu_lib
    lib = emu_lib()
    with pytest.raises(_lib.P2HotError):
        GroupCommit(lib, 3, [0, 0, 0])                      # not a power of two
    g = GroupCommit(lib, 4, [0] * 4)
    assert lib.p2hot_group_size(g._h) == 4 and lib.p2hot_group_size(None) == 0
    cols = np.ones((2, 16), dtype=np.uint64)
    r = g.commit(cols, 1, 4)                                # starky rate 1/2: two cosets on four ranks = sub-cosets, not an error
    r["free"]()
    with pytest.raises(_lib.P2HotError, match="LDE rows|cap subtrees"):
        g.commit(cols[:, :1], 0, 0)                         # one row: nothing to shard
    with pytest.raises(_lib.P2HotError, match="cap subtrees"):
        g.commit(cols, 3, 1)
    g.close()
    eng = emu_engine()
    h = C.c_void_p()
    # the emulator's fake RCCL makes one-rank communicators (as the GPU tier does on the real library); a multi-rank
    # ncclCommInitRank waits for the other ranks' processes (tests/test_emu_rccl_ranks.py) -- alone it times out, not hangs
    uid = np.zeros(128, dtype=np.uint8)
    assert lib.p2hot_comm_unique_id(uid.ctypes.data) == 0 and uid.any()
    os.environ["P2HOT_EMU_RCCL_TIMEOUT_MS"] = "300"
    try:
        assert lib.p2hot_comm_create_rccl(eng.ctx, 0, 2, uid.ctypes.data, C.byref(h)) == _lib.ECOMM
        assert b"hangs here" in lib.p2hot_last_error(eng.ctx)
    finally:
        del os.environ["P2HOT_EMU_RCCL_TIMEOUT_MS"]
    assert lib.p2hot_comm_unique_id(uid.ctypes.data) == 0
    assert lib.p2hot_comm_create_rccl(eng.ctx, 0, 1, uid.ctypes.data, C.byref(h)) == 0
    assert lib.p2hot_comm_selftest(h, 4096) == 0
    lib.p2hot_comm_destroy(h)
    cb = _lib.ALLGATHER_FN(lambda *a: 0)
    assert lib.p2hot_comm_create_callback(eng.ctx, 3, 8, cb, None, C.byref(h)) == 0
    assert (lib.p2hot_comm_rank(h), lib.p2hot_comm_world(h)) == (3, 8)
    lib.p2hot_comm_destroy(h)
    assert lib.p2hot_comm_create_callback(eng.ctx, 8, 8, cb, None, C.byref(h)) != 0   # rank outside the world
    first, count = C.c_size_t(), C.c_size_t()
    from plonky2_amd.distributed import ShardPlan
    for (W, world) in ((135, 8), (20, 4), (3, 8), (0, 2)):
        p = ShardPlan(W, 10, 3, 4, world)
        for r in range(world):
            assert lib.p2hot_shard_columns(W, world, r, C.byref(first), C.byref(count)) == 0
            assert (first.value, first.value + count.value) == p.columns(r)


@pytest.mark.parametrize("world,W,log_n,rb,cap,is_values", [
    (2, 5, 5, 3, 4, True), (4, 135, 3, 3, 4, True), (8, 20, 4, 3, 4, False),
    (4, 2, 6, 1, 4, True), (8, 2, 5, 1, 3, True),     # starky: rate 1/2 has two cosets, columns still shard over 4 / 8 ranks
    (8, 3, 4, 3, 3, True)])                            # fewer columns than ranks: some ranks transform nothing, all hash
def test_group_commit_column_sharded_fallback(ora, world, W, log_n, rb, cap, is_values):
    """P2HOT_SHARD_COLUMNS (last row): whole-column LDEs per rank, an all-to-all of the LDE matrix to row
    blocks, then the same per-rank hashing -- bit-identical to the oracle and therefore to the coset scheme"""
    from plonky2_amd.distributed import GroupCommit
    from tests.emu_backend import emu_lib
    rng = np.random.default_rng(world * 77 + W)
    cols = rand_field(rng, W, 1 << log_n, noncanonical=True)
    o = ora.commit(cols, rb, cap, is_values)
    g = GroupCommit(emu_lib(), world, [0] * world)
    r = g.commit(cols, rb, cap, is_values, want_leaves=True, by_columns=True)
    assert (r["coeffs"] == o["coeffs"] % np.uint64(0xFFFFFFFF00000001)).all()
    assert (r["cap"] == o["cap"]).all() and (r["digests"] == o["digests"]).all() and (r["leaves"] == o["leaves"]).all()
    N = 1 << (log_n + rb)
    rows, paths = r["open"]([0, N - 1, N // 2])
    for x, row, path in zip([0, N - 1, N // 2], rows, paths):
        assert (row == o["leaves"][x]).all() and ora.merkle_verify(row, x, o["cap"], path)
    r["free"]()
    g.close()


@pytest.mark.parametrize("world,widths,log_n,rb,cap,arity", [(2, [5, 3], 6, 3, 4, [2, 1]), (4, [7, 2, 2], 5, 3, 3, [2]), (8, [3], 4, 3, 3, []),
                                                               (4, [2, 2], 7, 1, 4, [2, 1])])   # starky's shape on sub-cosets
def test_group_prove_openings_equals_single_context_proof(ora, world, widths, log_n, rb, cap, arity):
    from tests.emu_backend import emu_engine, emu_lib
    _group_proof_equals_single_context(emu_lib(), emu_engine(), world, widths, log_n, rb, cap, arity)


def _group_proof_equals_single_context(lib, eng, world, widths, log_n, rb, cap, arity):
    """A whole opening proof over sharded oracles (p2hot_group_eval_openings / p2hot_group_prove_openings): rank 0 runs the
    polynomial side, the owners of the rows serve the initial trees' openings; every buffer of the proof equals the one
    the single-context p2hot_prove_openings produces for the same polynomials, and the transcripts end in the same state"""
    import ctypes as C
    from plonky2_amd import _lib
    from plonky2_amd.distributed import GroupCommit
    from plonky2_amd.fri.oracle import FriBatchInfo, PolynomialBatch, eval_openings, prove_openings
    from plonky2_amd.iop.challenger import Challenger
    rng = np.random.default_rng(world * 31 + len(widths))
    cols = [rand_field(rng, w, 1 << log_n) for w in widths]
    inst = [([3, 4], [(oi, pi) for oi, w in enumerate(widths) for pi in range(w)]), ([9, 1], [(0, pi) for pi in range(widths[0])])]
    pre = rand_field(rng, 5)
    # single context
    oracles = [PolynomialBatch.from_coeffs(c, rb, False, cap, engine=eng) for c in cols]
    ch = Challenger(eng)
    ch.observe_elements(pre)
    ev1 = eval_openings(oracles, [p for p, _ in inst], eng)
    p1 = prove_openings([FriBatchInfo(p, polys) for p, polys in inst], oracles, ch, rb, cap, arity, 3, 4, engine=eng)
    # the group
    g = GroupCommit(lib, world, [0] * world)
    commits = [g.commit(c, rb, cap, is_values=False, pipeline_chunks=2) for c in cols]
    for cm, o in zip(commits, oracles):
        assert (cm["cap"] == o.merkle_tree.cap.entries).all()
    ch2 = Challenger(g.engine0())
    ch2.observe_elements(pre)
    ev2 = g.eval_openings(commits, [p for p, _ in inst])
    for a, b in zip(ev1, ev2):
        assert (a == b).all()
    p2 = g.prove_openings(inst, commits, ch2, rb, cap, arity, 3, 4)
    assert p2["pow_witness"] == p1["pow_witness"] and p2["query_indices"] == p1["query_indices"]
    assert (p2["final_poly"].reshape(-1, 2) == p1["final_poly"]).all()
    flat_caps = np.concatenate([c.reshape(-1) for c in p1["commit_phase_merkle_caps"]]) if arity else np.zeros(0, dtype=np.uint64)
    assert (p2["caps"] == flat_caps).all()
    Q, wsum = 4, sum(widths)
    il = p2["initial_leaves"].reshape(Q, wsum)
    layers0 = log_n + rb - cap
    ip = p2["initial_paths"].reshape(Q, len(widths), layers0, 4)
    for q, qr in enumerate(p1["query_round_proofs"]):
        wo = 0
        for oi, (leaf, sib) in enumerate(qr["initial_trees_proof"]):
            assert (il[q, wo:wo + widths[oi]] == leaf).all() and (ip[q, oi] == sib).all()
            wo += widths[oi]
        ev = np.concatenate([e.reshape(-1) for e, _ in qr["steps"]]) if arity else np.zeros(0, dtype=np.uint64)
        sp = np.concatenate([s_.reshape(-1) for _, s_ in qr["steps"]]) if arity else np.zeros(0, dtype=np.uint64)
        assert (p2["step_evals"].reshape(Q, -1)[q] == ev).all() and (p2["step_paths"].reshape(Q, -1)[q] == sp).all()
    assert ch.get_n_challenges(3) == ch2.get_n_challenges(3)
    # column-sharded oracles cannot serve it: no rank holds all coefficients
    cm = g.commit(cols[0], rb, cap, is_values=False, by_columns=True)
    with pytest.raises(_lib.P2HotError, match="column-sharded"):
        g.eval_openings([cm], [[1, 2]])
    for c in commits + [cm]:
        c["free"]()
    del ch2
    g.close()
