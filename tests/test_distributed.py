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
    (2, 11, 5, 3, 4, True, 4), (4, 135, 3, 3, 4, True, 4), (2, 7, 6, 1, 2, False, 3), (8, 20, 3, 3, 4, True, 2)])
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
    with pytest.raises(ValueError):
        ShardPlan(2, 22, 1, 4, 4)  # starky rate 1/2: only two cosets
    with pytest.raises(ValueError):
        ShardPlan(2, 10, 3, 1, 4)  # fewer cap subtrees than ranks
