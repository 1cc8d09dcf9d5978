"""The single-process multi-GPU path on DISTINCT devices, in the CPU tier.

The kernel emulator models the part of the HIP runtime the multi-GPU layer depends on (tests/emu/hip_emu.h): eight devices, a
per-thread current device, device-tagged allocations behind page protection, streams and events with device identity, and a
fake RCCL (ncclCommInitAll, grouped ncclBroadcast / ncclAllGather) that enforces per-rank device, stream and buffer identity.
So `p2hot_group_create(8, [0..7])` -- the path that had only ever run with every rank on device 0 -- runs here as it will on an
8-GPU node: a wrong current device, a buffer on the wrong GPU, an event recorded on another device's stream or a collective
that would hang all fail loudly.  The last tests show that they do, with a deliberately mis-guarded entry point."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.conftest import P, ROOT, rand_field


def _lib_emu():
    from tests.emu_backend import emu_lib
    lib = emu_lib()
    lib.p2hot_emu_stats.argtypes = [C.POINTER(C.c_ulonglong)]
    lib.p2hot_emu_stats.restype = None
    lib.p2hot_emu_fault.argtypes = [C.c_char_p, C.c_int]
    lib.p2hot_emu_set_device.argtypes = [C.c_int]
    return lib


def stats(lib):
    a = (C.c_ulonglong * 8)()
    lib.p2hot_emu_stats(a)
    return dict(zip(("peer_copies", "peer_bytes", "nccl_broadcasts", "nccl_allgathers", "nccl_bytes", "device_switches", "violations",
                     "current_device"), [int(x) for x in a]))


def delta(lib, before):
    now = stats(lib)
    return {k: now[k] - before[k] for k in now if k != "current_device"}, now["current_device"]


@pytest.mark.parametrize("exchange", ["allgather", "broadcast", None])
@pytest.mark.parametrize("world,W,log_n,rb,cap,is_values", [(8, 20, 5, 3, 4, True), (4, 7, 6, 3, 2, False), (2, 135, 4, 1, 4, True),
                                                                 (8, 2, 6, 1, 4, True)])   # C4's shape: 2 cosets on 8 GPUs = sub-cosets
def test_group_commit_on_distinct_devices_over_rccl(ora, monkeypatch, world, W, log_n, rb, cap, is_values, exchange):
    """p2hot_group_create(G, [0..G-1]) -> ncclCommInitAll; the coset-sharded commit over G emulated GPUs: same bytes as the oracle,
    every exchange through the collectives library (no raw peer copy), not one device / stream / buffer violation, and the calling
    thread's current device restored after every call"""
    from plonky2_amd.distributed import GroupCommit
    lib = _lib_emu()
    assert lib.p2hot_emu_set_device(0) == 0
    # both exchange forms: ncclAllGather (pipelined column chunks through the chunk-major staging block) and one grouped
    # ncclBroadcast per slice; None = the group times both at creation and keeps the faster
    if exchange:
        monkeypatch.setenv("P2HOT_EXCHANGE", exchange)
    else:
        monkeypatch.delenv("P2HOT_EXCHANGE", raising=False)
    before = stats(lib)
    g = GroupCommit(lib, world, list(range(world)))
    assert g.uses_rccl and g.exchange == (exchange or g.exchange) and g.exchange in ("allgather", "broadcast")
    for i in range(world):  # profiling on: the exchange scope records events on the communication stream
        lib.p2hot_profile_enable(lib.p2hot_group_ctx(g._h, i), 1)
    rng = np.random.default_rng(world * 100 + W)
    cols = rand_field(rng, W, 1 << log_n)
    r = g.commit(cols, rb, cap, is_values=is_values, want_leaves=True, want_digests=True)
    o = ora.commit(cols, rb, cap, is_values)
    assert (r["coeffs"] == o["coeffs"] % np.uint64(P)).all() and (r["cap"] == o["cap"]).all()
    assert (r["digests"] == o["digests"]).all() and (r["leaves"] == o["leaves"]).all()
    N = 1 << (log_n + rb)
    idx = [0, N - 1, N // 2 + 1, N // world]  # leaves of several owners
    rows, paths = r["open"](idx)
    for q, i in enumerate(idx):
        assert (rows[q] == o["leaves"][i]).all() and ora.merkle_verify(rows[q], i, o["cap"], paths[q])
    d, cur = delta(lib, before)
    assert d["violations"] == 0, d
    assert d["nccl_broadcasts"] + d["nccl_allgathers"] > 0 and d["peer_copies"] == 0, d   # the exchange went through (fake) RCCL
    if exchange == "allgather":  # coefficient chunks, cap (the selftest at creation included): all-gathers only
        assert d["nccl_allgathers"] > 0 and d["nccl_broadcasts"] == 0, d
    if exchange == "broadcast":
        assert d["nccl_allgathers"] == 0, d
    assert d["device_switches"] >= world and cur == 0                                      # ... on G devices; the caller's device is back
    prof = lib.p2hot_profile_json(lib.p2hot_group_ctx(g._h, 0), 1).decode()
    assert '"exchange"' in prof
    r["free"]()
    g.close()
    assert stats(lib)["violations"] == before["violations"] and stats(lib)["current_device"] == 0


def test_group_prove_openings_on_distinct_devices_equals_single_context_proof(ora):
    """a whole opening proof over oracles sharded across 4 emulated GPUs: rank 0 runs what needs the polynomials, the rows and
    paths come from the owning GPUs -- buffer for buffer the single-context proof"""
    from plonky2_amd.distributed import GroupCommit
    from plonky2_amd.fri.oracle import FriBatchInfo, PolynomialBatch, prove_openings
    from plonky2_amd.iop.challenger import Challenger
    from tests.emu_backend import emu_engine
    lib = _lib_emu()
    lib.p2hot_emu_set_device(0)
    before = stats(lib)
    world, log_n, rb, cap, widths = 4, 5, 3, 4, [5, 3]
    rng = np.random.default_rng(17)
    cols = [rand_field(rng, w, 1 << log_n) for w in widths]
    g = GroupCommit(lib, world, [0, 1, 2, 3])
    commits = [g.commit(c, rb, cap, is_values=False) for c in cols]
    e0 = g.engine0()
    ch = Challenger(e0)
    pre = rand_field(rng, 5)
    ch.observe_elements(pre)
    z0, z1 = rand_field(rng, 2), rand_field(rng, 2)
    allp = [(oi, pi) for oi, w in enumerate(widths) for pi in range(w)]
    batches = [(z0, allp), (z1, [(1, 0), (1, 1)])]
    got = g.prove_openings(batches, commits, ch, rb, cap, [2, 1], 4, 5)
    eng = emu_engine()
    oracles = [PolynomialBatch.from_coeffs(c, rb, False, cap, engine=eng) for c in cols]
    ch1 = Challenger(eng)
    ch1.observe_elements(pre)
    ref = prove_openings([FriBatchInfo(p, polys) for p, polys in batches], oracles, ch1, rb, cap, [2, 1], 4, 5, engine=eng)
    assert got["pow_witness"] == ref["pow_witness"] and got["query_indices"] == ref["query_indices"]
    assert (got["final_poly"].reshape(-1, 2) == ref["final_poly"]).all()
    il = np.concatenate([np.concatenate([leaf for leaf, _ in q["initial_trees_proof"]]) for q in ref["query_round_proofs"]])
    assert (got["initial_leaves"] == il).all()
    ev = g.eval_openings(commits, [z0])
    assert all(e.shape == (1, w, 2) for e, w in zip(ev, widths))
    d, cur = delta(lib, before)
    assert d["violations"] == 0 and cur == 0, d
    for c in commits:
        c["free"]()
    ch.close() if hasattr(ch, "close") else None
    del ch
    g.close()


@pytest.mark.parametrize("mode", ["columns", "peer_copy"])
def test_peer_copy_transports_on_distinct_devices(ora, monkeypatch, mode):
    """the two paths that move data with explicit peer copies instead of collectives -- the column-sharded fallback's all-to-all
    and P2HOT_GROUP_PEER_COPY=1 -- on distinct devices: identical results, the copies counted as peer copies, no violation"""
    from plonky2_amd.distributed import GroupCommit
    lib = _lib_emu()
    lib.p2hot_emu_set_device(0)
    if mode == "peer_copy":
        monkeypatch.setenv("P2HOT_GROUP_PEER_COPY", "1")
    before = stats(lib)
    world, W, log_n, rb, cap = 4, 6, 5, (1 if mode == "columns" else 3), 3
    g = GroupCommit(lib, world, [4, 5, 6, 7])
    assert g.uses_rccl == (mode == "columns")
    cols = rand_field(np.random.default_rng(5), W, 1 << log_n)
    r = g.commit(cols, rb, cap, want_leaves=True, by_columns=(mode == "columns"))
    o = ora.commit(cols, rb, cap, True)
    assert (r["cap"] == o["cap"]).all() and (r["leaves"] == o["leaves"]).all() and (r["digests"] == o["digests"]).all()
    d, cur = delta(lib, before)
    assert d["violations"] == 0 and d["peer_copies"] > 0 and cur == 0, d
    r["free"]()
    g.close()


def test_single_context_calls_on_a_group_context_follow_its_device(ora):
    """a single-context entry point on p2hot_group_ctx(group, 3) called from a thread whose current device is 0 (after any
    p2hot_group_* call it is): the DeviceGuard makes device 3 current for the call and restores 0 -- ADVICE round 2, now executed
    on distinct devices.  With the guard switched off (the emulator build's test hook) the same call FAILS: its launches go to a
    stream of device 3 while device 0 is current"""
    from plonky2_amd import _lib
    from plonky2_amd.distributed import GroupCommit
    lib = _lib_emu()
    lib.p2hot_emu_set_device(0)
    g = GroupCommit(lib, 4, [0, 1, 2, 3])
    ctx3 = C.c_void_p(lib.p2hot_group_ctx(g._h, 3))
    W, log_n = 3, 4
    cols = rand_field(np.random.default_rng(9), W, 1 << log_n)
    ptrs = (C.c_void_p * W)(*[cols[c].ctypes.data for c in range(W)])
    cap = np.zeros((4, 4), dtype=np.uint64)
    before = stats(lib)
    assert lib.p2hot_commit(ctx3, ptrs, W, log_n, 3, 2, 1, 0, None, None, None, cap.ctypes.data, None) == _lib.OK
    assert (cap == ora.commit(cols, 3, 2, True)["cap"]).all()
    d, cur = delta(lib, before)
    assert d["violations"] == 0 and d["device_switches"] >= 2 and cur == 0
    try:
        assert lib.p2hot_emu_fault(b"no_device_guard", 1) == 0
        rc = lib.p2hot_commit(ctx3, ptrs, W, log_n, 3, 2, 1, 0, None, None, None, cap.ctypes.data, None)
        msg = lib.p2hot_last_error(ctx3).decode()
    finally:
        lib.p2hot_emu_fault(b"no_device_guard", 0)
    assert rc == _lib.EHIP and "stream of device 3 while device 0 is current" in msg, (rc, msg)
    assert stats(lib)["violations"] > before["violations"]
    lib.p2hot_emu_set_device(0)
    g.close()


def test_kernel_touching_another_gpus_memory_is_a_named_fault():
    """contexts on the null stream (what the Rust shim creates): without the guard a call on device 5's context from a thread
    sitting on device 0 launches on device 0's default stream and its kernels read device 5's tables -- on the emulated node
    that is a page fault, reported with both devices and the allocation, and the process aborts (so: a subprocess)"""
    code = r"""
import ctypes as C, numpy as np, sys
sys.path.insert(0, %r)
from tests.emu_backend import emu_lib
lib = emu_lib()
lib.p2hot_emu_fault.argtypes = [C.c_char_p, C.c_int]
ctx = C.c_void_p()
assert lib.p2hot_ctx_create(5, None, C.byref(ctx)) == 0
lib.p2hot_emu_set_device(0)
st = np.zeros((4, 12), dtype=np.uint64)
dev = C.c_void_p()
print("guarded", lib.p2hot_challenger_create(ctx, C.byref(dev)), flush=True)
lib.p2hot_emu_fault(b"no_device_guard", 1)
cols = np.ones((2, 16), dtype=np.uint64)
ptrs = (C.c_void_p * 2)(cols[0].ctypes.data, cols[1].ctypes.data)
cap = np.zeros((4, 4), dtype=np.uint64)
rc = lib.p2hot_commit(ctx, ptrs, 2, 4, 3, 2, 1, 0, None, None, None, cap.ctypes.data, None)
print("survived", rc, flush=True)
""" % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=ROOT,
                       env={**os.environ, "PYTHONFAULTHANDLER": "0"})
    assert "guarded 0" in r.stdout, r.stdout + r.stderr
    assert "survived" not in r.stdout and r.returncode != 0, r.stdout + r.stderr
    assert "DEVICE MEMORY FAULT" in r.stderr and "of device 5" in r.stderr and "device 0 is current" in r.stderr, r.stderr


def test_fake_rccl_refuses_what_hangs_or_corrupts_the_real_library():
    """the fake RCCL's own rules, exercised directly: duplicate devices in ncclCommInitAll, a multi-rank collective outside a
    group, ranks posting different sequences, a stream or a buffer of another device -- each is an error with a cause, and a
    well-formed grouped broadcast / all-gather delivers"""
    lib = _lib_emu()
    lib.p2hot_emu_set_device(0)
    for f in ("emu_ncclBroadcast", "emu_ncclAllGather"):
        getattr(lib, f).restype = C.c_int
    lib.emu_ncclBroadcast.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.emu_ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    lib.emu_ncclCommInitAll.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int)]
    lib.emu_ncclCommDestroy.argtypes = [C.c_void_p]
    lib.emu_ncclGetErrorString.restype = C.c_char_p
    err = lambda rc: lib.emu_ncclGetErrorString(rc).decode()
    comms = (C.c_void_p * 2)()
    rc = lib.emu_ncclCommInitAll(comms, 2, (C.c_int * 2)(3, 3))
    assert rc != 0 and "Duplicate GPU" in err(rc)
    assert lib.emu_ncclCommInitAll(comms, 2, (C.c_int * 2)(1, 2)) == 0
    a, b = np.arange(8, dtype=np.uint64), np.zeros(8, dtype=np.uint64)   # untracked test memory: allowed on any rank
    # outside a group, one thread, two ranks: the real call never returns
    lib.p2hot_emu_set_device(1)
    rc = lib.emu_ncclBroadcast(a.ctypes.data, a.ctypes.data, 64, 1, 0, comms[0], None)
    assert rc != 0 and "blocks here for ever" in err(rc)
    # rank 1's stream must belong to device 2: the null stream of device 1 does not
    lib.emu_ncclGroupStart()
    rc = lib.emu_ncclBroadcast(b.ctypes.data, b.ctypes.data, 64, 1, 0, comms[1], None)
    assert rc != 0 and "its stream belongs to device 1" in err(rc)
    # only one of the two ranks posts: the group would hang
    assert lib.emu_ncclBroadcast(a.ctypes.data, a.ctypes.data, 64, 1, 0, comms[0], None) == 0
    rc = lib.emu_ncclGroupEnd()
    assert rc != 0 and "would hang" in err(rc)
    # well formed: both ranks, each from its own device's (null) stream
    lib.emu_ncclGroupStart()
    lib.p2hot_emu_set_device(1)
    assert lib.emu_ncclBroadcast(a.ctypes.data, a.ctypes.data, 64, 1, 0, comms[0], None) == 0
    lib.p2hot_emu_set_device(2)
    assert lib.emu_ncclBroadcast(a.ctypes.data, b.ctypes.data, 64, 1, 0, comms[1], None) == 0
    assert lib.emu_ncclGroupEnd() == 0
    assert os.environ.get("P2HOT_EMU_ASYNC", "1") != "1" or not (b == a).all()   # posted, not run: a collective is stream work like any other ...
    ssync = _runtime(lib)("hipStreamSynchronize", [C.c_void_p])
    assert ssync(None) == 0 and (b == a).all()  # ... waiting for ONE rank's stream pulls every rank's stream up to the collective
    # in-place all-gather of two 32-byte slices
    x, y = np.zeros(8, dtype=np.uint64), np.zeros(8, dtype=np.uint64)
    x[:4], y[4:] = 7, 9
    lib.emu_ncclGroupStart()
    lib.p2hot_emu_set_device(1)
    assert lib.emu_ncclAllGather(x.ctypes.data, x.ctypes.data, 32, 1, comms[0], None) == 0
    lib.p2hot_emu_set_device(2)
    assert lib.emu_ncclAllGather(y.ctypes.data + 32, y.ctypes.data, 32, 1, comms[1], None) == 0
    assert lib.emu_ncclGroupEnd() == 0 and ssync(None) == 0
    assert (x == y).all() and (x[:4] == 7).all() and (x[4:] == 9).all()
    # mismatched sequences
    lib.emu_ncclGroupStart()
    lib.p2hot_emu_set_device(1)
    lib.emu_ncclBroadcast(a.ctypes.data, a.ctypes.data, 64, 1, 0, comms[0], None)
    lib.p2hot_emu_set_device(2)
    lib.emu_ncclBroadcast(b.ctypes.data, b.ctypes.data, 32, 1, 0, comms[1], None)
    rc = lib.emu_ncclGroupEnd()
    assert rc != 0 and "mismatched collectives" in err(rc)
    for c in comms:
        assert lib.emu_ncclCommDestroy(c) == 0
    lib.p2hot_emu_set_device(0)


def test_runtime_refuses_cross_device_handles():
    """the emulated runtime itself: an event of one device recorded on a stream of another, a launch-type call on a foreign
    stream, a destroyed stream -- hipErrorInvalidResourceHandle with a cause; a wait on ANOTHER device's event is legal"""
    lib = _lib_emu()
    import re
    syms = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "tests", "emu", "libp2hot_emu.so")], capture_output=True, text=True).stdout

    def fn(name, argtypes):
        m = re.search(r"\b(_Z%d%s\w*)\b" % (len(name), name), syms)
        assert m, name
        f = getattr(lib, m.group(1))
        f.argtypes, f.restype = argtypes, C.c_int
        return f
    screate, ecreate = fn("hipStreamCreateWithFlags", [C.POINTER(C.c_void_p), C.c_uint]), fn("hipEventCreateWithFlags", [C.POINTER(C.c_void_p), C.c_uint])
    erecord, swait = fn("hipEventRecord", [C.c_void_p, C.c_void_p]), fn("hipStreamWaitEvent", [C.c_void_p, C.c_void_p, C.c_uint])
    sdestroy, edestroy = fn("hipStreamDestroy", [C.c_void_p]), fn("hipEventDestroy", [C.c_void_p])
    malloc_, free_, memset_ = fn("hipMalloc", [C.POINTER(C.c_void_p), C.c_size_t]), fn("hipFree", [C.c_void_p]), fn("hipMemsetAsync", [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p])
    lib.p2hot_emu_device_of.argtypes = [C.c_void_p]
    lib.p2hot_emu_set_device(6)
    s6, e6, m6 = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert screate(C.byref(s6), 1) == 0 and ecreate(C.byref(e6), 2) == 0 and malloc_(C.byref(m6), 4096) == 0
    assert lib.p2hot_emu_device_of(m6) == 6
    assert erecord(e6, s6) == 0
    lib.p2hot_emu_set_device(7)
    s7, e7 = C.c_void_p(), C.c_void_p()
    assert screate(C.byref(s7), 1) == 0 and ecreate(C.byref(e7), 2) == 0
    before = stats(lib)["violations"]
    assert erecord(e7, s6) == 400          # a stream of device 6 while device 7 is current
    assert swait(s7, e6, 0) == 0           # waiting for another device's event orders the two GPUs: legal
    assert memset_(m6, 0, 64, s7) != 0     # device 6's memory from device 7
    lib.p2hot_emu_set_device(6)
    assert erecord(e7, s6) == 400          # an event of device 7 on a stream of device 6
    assert memset_(m6, 1, 64, s6) == 0
    assert sdestroy(s6) == 0 and erecord(e6, s6) == 400   # destroyed handle
    assert stats(lib)["violations"] == before + 4
    assert free_(m6) == 0 and edestroy(e6) == 0
    lib.p2hot_emu_set_device(7)
    assert sdestroy(s7) == 0 and edestroy(e7) == 0
    lib.p2hot_emu_set_device(0)


def test_index_past_the_end_of_a_device_buffer_is_a_named_fault():
    """emulated device allocations END at a guard page: a kernel whose index runs off the end of a buffer (here: a batch NTT told
    about one polynomial more than the buffer holds) faults on the first word past the end, and the handler says how far past the end
    of which allocation -- in every kernel, copy and host loop of the CPU tier; in-bounds work on the same buffer passes"""
    code = r"""
import ctypes as C, re, subprocess, sys, os
sys.path.insert(0, %r)
from tests.emu_backend import emu_lib
lib = emu_lib()
syms = subprocess.run(["nm", "-D", "--defined-only", os.path.join(%r, "tests", "emu", "libp2hot_emu.so")], capture_output=True, text=True).stdout
hip_malloc = getattr(lib, re.search(r"\b(_Z9hipMalloc\w*)\b", syms).group(1))
hip_malloc.argtypes, hip_malloc.restype = [C.POINTER(C.c_void_p), C.c_size_t], C.c_int
ctx = C.c_void_p()
assert lib.p2hot_ctx_create(0, None, C.byref(ctx)) == 0
n, batch = 64, 3
buf = C.c_void_p()
assert hip_malloc(C.byref(buf), batch * n * 8) == 0
C.memset(buf, 0, batch * n * 8)
lib.p2hot_fft_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint]
print("in bounds", lib.p2hot_fft_dev(ctx, buf, batch, n, 6), flush=True)
hip_sync = getattr(lib, re.search(r"\b(_Z20hipStreamSynchronize\w*)\b", syms).group(1))
hip_sync.argtypes, hip_sync.restype = [C.c_void_p], C.c_int
rc = lib.p2hot_fft_dev(ctx, buf, batch + 1, n, 6)     # queued ...
print("queued", rc, flush=True)
rc = hip_sync(None)                                   # ... runs when somebody waits for it
print("survived", rc, flush=True)
""" % (ROOT, ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=ROOT,
                       env={**os.environ, "PYTHONFAULTHANDLER": "0", "P2HOT_EMU_ASYNC": "1"})
    assert "in bounds 0" in r.stdout and "queued 0" in r.stdout, r.stdout + r.stderr
    assert "survived" not in r.stdout and r.returncode != 0, r.stdout + r.stderr
    assert "DEVICE MEMORY OVERRUN" in r.stderr and "past the end of a 1536-byte allocation of device 0" in r.stderr, r.stderr


def _runtime(lib):
    import re
    syms = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "tests", "emu", "libp2hot_emu.so")], capture_output=True, text=True).stdout

    def fn(name, argtypes, restype=C.c_int):
        m = re.search(r"\b(_Z%d%s\w*)\b" % (len(name), name), syms)
        assert m, name
        f = getattr(lib, m.group(1))
        f.argtypes, f.restype = argtypes, restype
        return f
    return fn


def test_streams_are_queues_a_missing_wait_reads_stale_memory():
    """the emulated streams execute as late as is legal: work runs when somebody waits for it.  A consumer stream WITHOUT a
    hipStreamWaitEvent on its producer's event runs before the producer and copies stale bytes; with the wait it copies the
    produced bytes; a host read of pinned memory before the synchronisation is stale; a copy into pageable memory is synchronous;
    a pageable source is captured when the call returns; a wait placed before its record does not wait"""
    if os.environ.get("P2HOT_EMU_ASYNC", "1") != "1":
        pytest.skip("this test pins the as-late-as-legal schedule")
    lib = _lib_emu()
    fn = _runtime(lib)
    vp, sz = C.c_void_p, C.c_size_t
    malloc_, free_ = fn("hipMalloc", [C.POINTER(vp), sz]), fn("hipFree", [vp])
    hmalloc, hfree = fn("hipHostMalloc", [C.POINTER(vp), sz, C.c_uint]), fn("hipHostFree", [vp])
    memset_, memcpy_ = fn("hipMemsetAsync", [vp, C.c_int, sz, vp]), fn("hipMemcpyAsync", [vp, vp, sz, C.c_int, vp])
    screate, sdestroy, ssync = fn("hipStreamCreateWithFlags", [C.POINTER(vp), C.c_uint]), fn("hipStreamDestroy", [vp]), fn("hipStreamSynchronize", [vp])
    ecreate, edestroy = fn("hipEventCreateWithFlags", [C.POINTER(vp), C.c_uint]), fn("hipEventDestroy", [vp])
    erecord, swait, esync = fn("hipEventRecord", [vp, vp]), fn("hipStreamWaitEvent", [vp, vp, C.c_uint]), fn("hipEventSynchronize", [vp])
    errstr = fn("hipGetErrorString", [C.c_int], C.c_char_p)
    lib.p2hot_emu_set_device(0)
    a, b, pin = vp(), vp(), vp()
    assert malloc_(C.byref(a), 64) == 0 and malloc_(C.byref(b), 64) == 0 and hmalloc(C.byref(pin), 64, 0) == 0
    prod, cons, ev = vp(), vp(), vp()
    assert screate(C.byref(prod), 1) == 0 and screate(C.byref(cons), 1) == 0 and ecreate(C.byref(ev), 2) == 0
    pinned = (C.c_ubyte * 64).from_address(pin.value)

    def round_trip(with_wait):
        C.memset(pin, 0xEE, 64)
        assert memset_(a, 0x11, 64, None) == 0 and ssync(None) == 0               # a = 0x11 everywhere, done
        assert memset_(a, 0x77, 64, prod) == 0 and erecord(ev, prod) == 0           # the producer overwrites a ...
        if with_wait:
            assert swait(cons, ev, 0) == 0
        assert memcpy_(pin, a, 64, 2, cons) == 0                                    # ... the consumer copies a into pinned host memory
        assert pinned[0] == 0xEE                                                    # nothing has run: the host sees the old bytes
        assert ssync(cons) == 0
        got = pinned[0]
        assert ssync(prod) == 0
        return got
    assert round_trip(False) == 0x11      # no wait: the consumer ran first and saw the value from before the producer
    assert round_trip(True) == 0x77       # with the wait the producer was pulled in front of it
    # pageable destination: synchronous (and everything queued before it on that stream has run)
    host = (C.c_ubyte * 64)()
    assert memset_(b, 0x42, 64, prod) == 0 and memcpy_(host, b, 64, 2, prod) == 0 and host[63] == 0x42
    # pageable source: captured when the call returns
    src = (C.c_ubyte * 64)(*([9] * 64))
    assert memcpy_(b, src, 64, 1, prod) == 0
    C.memset(src, 0, 64)
    assert memcpy_(host, b, 64, 2, prod) == 0 and host[0] == 9
    # a wait placed BEFORE the record it is meant for does not wait (the event's state at the time of the call counts)
    fresh = vp()
    assert ecreate(C.byref(fresh), 2) == 0
    assert memset_(a, 0x55, 64, None) == 0 and ssync(None) == 0
    assert swait(cons, fresh, 0) == 0                                               # never recorded: no-op
    assert memset_(a, 0x66, 64, prod) == 0 and erecord(fresh, prod) == 0
    C.memset(pin, 0, 64)
    assert memcpy_(pin, a, 64, 2, cons) == 0 and ssync(cons) == 0 and pinned[0] == 0x55
    assert esync(fresh) == 0 and memcpy_(pin, a, 64, 2, cons) == 0 and ssync(cons) == 0 and pinned[0] == 0x66
    for h in (prod, cons):
        assert sdestroy(h) == 0
    for h in (ev, fresh):
        assert edestroy(h) == 0
    assert free_(a) == 0 and free_(b) == 0 and hfree(pin) == 0


def test_forgotten_stream_waits_are_visible_as_wrong_results(ora, monkeypatch):
    """what the queue model is for: with every hipStreamWaitEvent of the program dropped (the emulator build's test hook) the
    pipelined host-pointer commit -- uploads on a side stream, leaf copies beside the sponge -- and the group commit -- exchange on
    the communication streams -- no longer produce the oracle's trees under the as-late-as-legal schedule; with the waits in
    place they do (every other test of the tier)"""
    if os.environ.get("P2HOT_EMU_ASYNC", "1") != "1":
        pytest.skip("this test pins the as-late-as-legal schedule")
    from plonky2_amd import _lib
    from plonky2_amd.distributed import GroupCommit
    lib = _lib_emu()
    lib.p2hot_emu_set_device(0)
    rng = np.random.default_rng(3)
    W, log_n, rb, cap = 70, 5, 2, 2
    monkeypatch.setenv("P2HOT_HOST_BLOCK_COLS", "16")   # column blocks as at 2^20 rows: uploads on the side stream, the sponge absorbing beside them
    cols = rand_field(rng, W, 1 << log_n)
    o = ora.commit(cols, rb, cap, True)
    ptrs = (C.c_void_p * W)(*[cols[c].ctypes.data for c in range(W)])
    N = 1 << (log_n + rb)

    def host_commit():
        ctx = C.c_void_p()
        assert lib.p2hot_ctx_create(0, None, C.byref(ctx)) == 0
        capv, leaves, coeffs = np.zeros((1 << cap, 4), dtype=np.uint64), np.zeros((N, W), dtype=np.uint64), np.zeros((W, 1 << log_n), dtype=np.uint64)
        digests = np.zeros((max(1, 2 * (N - (1 << cap))), 4), dtype=np.uint64)
        rc = lib.p2hot_commit(ctx, ptrs, W, log_n, rb, cap, 1, 0, coeffs.ctypes.data, leaves.ctypes.data, digests.ctypes.data, capv.ctypes.data, None)
        lib.p2hot_ctx_destroy(ctx)
        return rc == _lib.OK and (capv == o["cap"]).all() and (leaves == o["leaves"]).all() and (coeffs == o["coeffs"]).all()

    def group_commit():
        try:
            g = GroupCommit(lib, 4, [0, 1, 2, 3])
        except _lib.P2HotError:
            return False          # the group's own preflight (a pattern exchange checked on every rank) already notices
        try:
            r = g.commit(cols, rb, cap, want_leaves=True, pipeline_chunks=3)
            ok = bool((r["cap"] == o["cap"]).all() and (r["leaves"] == o["leaves"]).all())
            r["free"]()
            return ok
        except _lib.P2HotError:
            return False
        finally:
            g.close()
    assert host_commit() and group_commit()
    assert lib.p2hot_emu_fault(b"drop_stream_waits", 1) == 0
    try:
        broken = (host_commit(), group_commit())
    finally:
        assert lib.p2hot_emu_fault(b"drop_stream_waits", 0) == 0
        lib.p2hot_emu_set_device(0)
    assert broken == (False, False), broken
    assert host_commit() and group_commit()


def test_rccl_info_reports_the_bound_library(emu):
    """p2hot_rccl_info: the emulator build is bound to the fake RCCL of tests/emu and says so (the product build reports the real
    file and ncclGetVersion: tests/test_gpu_fullsize.py::test_rccl_binding_without_torch); NULL outputs are allowed"""
    from plonky2_amd.distributed import rccl_info
    r = rccl_info(emu.lib)
    assert r == {"path": "tests/emu fake RCCL", "version": 0}
    assert emu.lib.p2hot_rccl_info(None, 0, None) == 0
