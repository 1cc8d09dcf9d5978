"""One process per GPU over RCCL -- what `bench.py --gpus N` and `plonky2_amd.distributed.Communicator(transport="rccl")` run on
an 8-GPU node -- in the CPU tier: `world` PROCESSES, rank r on emulated device r, the unique id handed out over a gloo process
group (the launcher's job), `p2hot_comm_create_rccl` -> ncclCommInitRank, the preflight's micro-timing of both exchange forms
with rank 0's verdict broadcast, and `p2hot_commit_sharded_dev` exchanging through grouped ncclBroadcast / ncclAllGather on the
communication stream.  The emulator's fake RCCL joins the processes through a shared-memory segment and announces every
collective before it moves a byte: a rank that posts another sequence, another size, sits on another rank's device or never
arrives is an ERROR with a message here (a timeout where the real library hangs); the last tests show that it is."""
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


def _setup(rank, world, port, device, env):
    import ctypes as C
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.pop("P2HOT_EXCHANGE", None)
    os.environ.update(env)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from plonky2_amd.engine import Engine
    from tests.emu_backend import HostMemory, emu_lib
    lib = emu_lib()
    lib.p2hot_emu_set_device.argtypes = [C.c_int]
    lib.p2hot_emu_stats.argtypes = [C.POINTER(C.c_ulonglong)]
    lib.p2hot_emu_stats.restype = None
    lib.p2hot_emu_set_device(device)   # LOCAL_RANK -> torch.cuda.set_device in bench.py
    return dist, lib, Engine(device, lib=lib, memory=HostMemory())


def _stats(lib):
    import ctypes as C
    a = (C.c_ulonglong * 8)()
    lib.p2hot_emu_stats(a)
    return [int(x) for x in a]


def _worker_commit(rank, world, port, W, log_n, rb, cap, is_values, chunks, exchange, q):
    dist, lib, eng = _setup(rank, world, port, rank, {"P2HOT_EXCHANGE": exchange} if exchange else {})
    try:
        from oracle import p2oracle as ora
        from plonky2_amd.distributed import ShardedCommit
        cols = rand_field(np.random.default_rng(99), W, 1 << log_n)
        job = ShardedCommit(eng, W, log_n, rb, cap, is_values=is_values, rank=rank, world=world, dist=dist, want_leaves=True,
                            pipeline_chunks=chunks, transport="rccl")
        before = _stats(lib)
        job.comm.selftest(4096)
        mode = job.comm.exchange
        c0, c1 = job.column_range
        for _ in range(2):
            r = job.run(eng.dev(cols[c0:c1]))
        eng.sync() if hasattr(eng, "sync") else None
        o = ora.commit(cols, rb, cap, is_values)
        r0, rc = job.plan.rows(rank)
        ok = bool((eng.host(r["coeffs"]) == o["coeffs"]).all() and (eng.host(r["digests"]) == o["digests"]).all()
                  and (eng.host(r["cap"]) == o["cap"]).all() and (eng.host(r["leaves"]) == o["leaves"][r0:r0 + rc]).all())
        after = _stats(lib)
        # [2] broadcasts, [3] all-gathers, [6] violations, [7] current device
        q.put((rank, ok, mode, after[2] - before[2], after[3] - before[3], after[6] - before[6], after[7]))
        job.comm.close()
    finally:
        dist.destroy_process_group()


def _run(target, world, *args, timeout=300):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + args + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout)
        assert p.exitcode == 0
    return sorted(q.get(timeout=5) for _ in range(world))


@pytest.mark.parametrize("world,W,log_n,rb,cap,is_values,chunks,exchange", [
    (2, 5, 5, 3, 4, True, 1, None), (2, 11, 5, 3, 4, True, 4, "allgather"), (2, 11, 5, 3, 4, False, 4, "broadcast"),
    (4, 135, 3, 3, 4, True, 4, "allgather"), (4, 7, 4, 3, 4, True, 2, "broadcast"), (8, 20, 3, 3, 4, True, 2, None),
    (4, 2, 6, 1, 4, True, 1, None)])   # the last one: C4's shape, more ranks than cosets
def test_sharded_commit_over_rccl_one_process_per_device(world, W, log_n, rb, cap, is_values, chunks, exchange):
    res = _run(_worker_commit, world, W, log_n, rb, cap, is_values, chunks, exchange)
    assert [(r, ok) for r, ok, *_ in res] == [(r, True) for r in range(world)]
    modes = {m for _, _, m, *_ in res}
    assert len(modes) == 1 and modes <= {"allgather", "broadcast"}      # every rank runs rank 0's verdict
    if exchange:
        assert modes == {exchange}
    for rank, _, mode, n_bc, n_ag, violations, cur in res:
        assert violations == 0 and cur == rank
        assert (n_ag > 0) if mode == "allgather" else (n_bc > 0 and n_ag == 0)


def _worker_rules(rank, world, port, case, q):
    """what must NOT hang or pass silently; every rank reports the error text it got (None = no error)"""
    env = {"P2HOT_EMU_RCCL_TIMEOUT_MS": "1500", "P2HOT_EXCHANGE": "broadcast"}
    device = 0 if case == "same_device" else rank
    dist, lib, eng = _setup(rank, world, port, device, env)
    try:
        from plonky2_amd import _lib
        from plonky2_amd.distributed import Communicator, ShardedCommit
        err = None
        try:
            if case == "same_device":
                Communicator(eng, rank, world, dist, transport="rccl")
            elif case == "lonely_preflight":       # rank 1 skips the collective preflight
                comm = Communicator(eng, rank, world, dist, transport="rccl")
                if rank == 0:
                    comm.selftest(256)
            elif case == "different_shapes":       # the ranks disagree on the job: different slice sizes in the first exchange
                job = ShardedCommit(eng, 4 if rank == 0 else 6, 4, 3, 4, is_values=True, rank=rank, world=world, dist=dist, transport="rccl",
                                    pipeline_chunks=1)
                c0, c1 = job.column_range
                job.run(eng.dev(np.ones((c1 - c0, 16), dtype=np.uint64)))
        except _lib.P2HotError as e:
            err = str(e)
        q.put((rank, err))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case,needle", [("same_device", "Duplicate GPU detected"), ("lonely_preflight", "the real library hangs here"),
                                         ("different_shapes", "differs between ranks")])
def test_fake_rccl_turns_multi_process_hangs_into_errors(case, needle):
    res = _run(_worker_rules, 2, case, timeout=120)
    errs = dict(res)
    if case == "lonely_preflight":
        assert errs[1] is None and needle in errs[0]
    else:
        assert all(e is not None and needle in e for e in errs.values()), errs
