"""p2hot_commit with P2HOT_LEAVES_ASYNC / P2HOT_LEAVES_NATURAL (include/p2hot.h): the call returns with the cap, the coefficients
and the digests; the row-major leaf matrix (merkle_tree.leaves, fri/oracle.rs:97-98) keeps travelling into the caller's buffer in
row blocks, and p2hot_batch_leaves_wait is the fence a reader of MerkleTree::get (hash/merkle_tree.rs:227) sits behind.
NATURAL: the buffer holds transpose(lde_values) WITHOUT reverse_index_bits, so get_lde_values(i, step) (oracle.rs:142-147) is
row i * step -- the quotient loop (plonk/prover.rs:712-722) walks it forward, in the order the blocks arrive."""
import ctypes as C

import numpy as np
import pytest

from plonky2_amd import _lib
from tests.conftest import P, rand_field


def _bitrev_perm(bits):
    return np.array([int(format(i, "0%db" % bits)[::-1], 2) if bits else 0 for i in range(1 << bits)], dtype=np.int64)


def _commit(eng, cols, rb, cap, flags, leaves, is_values=1):
    W, n = cols.shape
    log_n = n.bit_length() - 1
    ptrs = (C.c_void_p * W)(*[cols[c].ctypes.data for c in range(W)])
    coeffs = np.zeros((W, n), dtype=np.uint64)
    digests = np.zeros((eng.num_digests(log_n + rb, cap), 4), dtype=np.uint64)
    capv = np.zeros((1 << cap, 4), dtype=np.uint64)
    h = C.c_void_p()
    rc = eng.lib.p2hot_commit(eng.ctx, ptrs, W, log_n, rb, cap, is_values, flags, coeffs.ctypes.data,
                              leaves.ctypes.data if leaves is not None else None, digests.ctypes.data, capv.ctypes.data, C.byref(h))
    return rc, h, coeffs, digests, capv


@pytest.mark.parametrize("W,log_n,rb,cap,flags", [(11, 7, 3, 4, _lib.LEAVES_ASYNC), (5, 9, 3, 2, _lib.LEAVES_ASYNC | _lib.LEAVES_NATURAL),
                                                   (37, 13, 3, 4, _lib.LEAVES_ASYNC | _lib.LEAVES_NATURAL),   # 2^16 rows: 64 row blocks
                                                   (3, 4, 1, 0, _lib.LEAVES_NATURAL), (2, 0, 2, 0, _lib.LEAVES_ASYNC | _lib.LEAVES_NATURAL)])
def test_async_and_natural_leaves_vs_oracle(eng, ora, W, log_n, rb, cap, flags):
    rng = np.random.default_rng(W * 31 + log_n)
    cols = rand_field(rng, W, 1 << log_n, noncanonical=True)
    N = 1 << (log_n + rb)
    o = ora.commit(cols, rb, cap, True)
    want = o["leaves"][_bitrev_perm(log_n + rb)] if flags & _lib.LEAVES_NATURAL else o["leaves"]
    leaves = np.zeros((N, W), dtype=np.uint64)
    rc, h, coeffs, digests, capv = _commit(eng, cols, rb, cap, flags, leaves)
    eng.check(rc)
    # everything but the leaf matrix is complete when the call returns
    assert (coeffs == o["coeffs"] % np.uint64(P)).all() and (digests == o["digests"]).all() and (capv == o["cap"]).all()
    # fences: a sub-range first, then everything (a second wait over landed rows returns at once)
    eng.check(eng.lib.p2hot_batch_leaves_wait(h, 0, min(N, 3)))
    assert (leaves[:min(N, 3)] == want[:min(N, 3)]).all()
    eng.check(eng.lib.p2hot_batch_leaves_wait(h, N // 2, N))
    assert (leaves[N // 2:] == want[N // 2:]).all()
    eng.check(eng.lib.p2hot_batch_leaves_wait(h, 0, N))
    assert (leaves == want).all()
    # the handle serves rows / paths in the COMMITTED indexing whatever the host copy's order
    idx = np.array([0, N - 1, N // 3], dtype=np.uint64)
    rows = np.zeros((3, W), dtype=np.uint64)
    eng.check(eng.lib.p2hot_batch_rows(h, idx.ctypes.data, 3, rows.ctypes.data))
    assert (rows == o["leaves"][idx.astype(np.int64)]).all()
    assert eng.lib.p2hot_batch_leaves_wait(h, 0, N + 1) == _lib.EINVAL and eng.lib.p2hot_batch_leaves_wait(h, 2, 1) == _lib.EINVAL
    eng.lib.p2hot_batch_free(h)


def test_a_read_before_the_fence_is_a_wrong_answer(emu, ora):
    """the emulated streams are queues that run when somebody waits for them (tests/emu/hip_emu_rt.cpp): after the call returns the
    leaf stream has not moved a byte, so an unfenced reader sees the buffer as it was; the fence of ONE block delivers that block
    and not the ones behind it; p2hot_batch_free delivers the rest (the buffer may be freed after it)"""
    rng = np.random.default_rng(5)
    W, log_n, rb, cap = 9, 13, 3, 4
    N = 1 << (log_n + rb)
    cols = rand_field(rng, W, 1 << log_n)
    o = ora.commit(cols, rb, cap, True)
    # a PINNED destination (the shim's flat leaf store, p2hot_host_alloc): into pageable memory a copy is synchronous by the
    # runtime's rules, the emulator's too
    blk = C.c_void_p()
    emu.check(emu.lib.p2hot_host_alloc(emu.ctx, N * W * 8, C.byref(blk)))
    leaves = np.frombuffer((C.c_uint64 * (N * W)).from_address(blk.value), dtype=np.uint64).reshape(N, W)
    leaves[...] = 0xDEAD
    rc, h, *_ = _commit(emu, cols, rb, cap, _lib.LEAVES_ASYNC, leaves)
    emu.check(rc)
    assert (leaves == 0xDEAD).all(), "no fence yet: nothing may have landed under the lazy schedule"
    rpb = N // 64
    assert emu.lib.p2hot_batch_leaves_block_rows(h) == rpb   # min(64, N / 1024) blocks: what the shim rounds its landed mark up to
    emu.check(emu.lib.p2hot_batch_leaves_wait(h, 5 * rpb + 1, 5 * rpb + 2))       # block 5: blocks 0..5 have landed (stream order)
    assert (leaves[:6 * rpb] == o["leaves"][:6 * rpb]).all()
    assert (leaves[6 * rpb:] == 0xDEAD).all(), "blocks behind the fenced one are still in flight"
    emu.lib.p2hot_batch_free(h)
    assert (leaves == o["leaves"]).all()
    del leaves
    emu.lib.p2hot_host_free(emu.ctx, blk)


def test_async_leaves_argument_errors(eng):
    cols = np.ones((2, 8), dtype=np.uint64)
    leaves = np.zeros((16, 2), dtype=np.uint64)
    ptrs = (C.c_void_p * 2)(cols[0].ctypes.data, cols[1].ctypes.data)
    capv = np.zeros((1, 4), dtype=np.uint64)
    # ASYNC without a handle: who would own the copy in flight?
    assert eng.lib.p2hot_commit(eng.ctx, ptrs, 2, 3, 1, 0, 1, _lib.LEAVES_ASYNC, None, leaves.ctypes.data, None, capv.ctypes.data, None) == _lib.EINVAL
    h = C.c_void_p()
    assert eng.lib.p2hot_commit(eng.ctx, ptrs, 2, 3, 1, 0, 1, _lib.LEAVES_ASYNC, None, None, None, capv.ctypes.data, C.byref(h)) == _lib.EINVAL
    assert eng.lib.p2hot_commit(eng.ctx, ptrs, 2, 3, 1, 0, 1, _lib.LEAVES_NATURAL, None, None, None, capv.ctypes.data, C.byref(h)) == _lib.EINVAL
    assert eng.lib.p2hot_commit(eng.ctx, ptrs, 2, 3, 1, 0, 1, 16, None, None, None, capv.ctypes.data, C.byref(h)) == _lib.EINVAL
    assert eng.lib.p2hot_batch_leaves_wait(None, 0, 0) == _lib.EINVAL
    assert eng.lib.p2hot_batch_leaves_block_rows(None) == 0
    # a batch without a copy in flight: the fence is a no-op
    eng.check(eng.lib.p2hot_commit(eng.ctx, ptrs, 2, 3, 1, 0, 1, 0, None, leaves.ctypes.data, None, capv.ctypes.data, C.byref(h)))
    eng.check(eng.lib.p2hot_batch_leaves_wait(h, 0, 16))
    assert eng.lib.p2hot_batch_leaves_block_rows(h) == 0
    # the fence is lock-free and therefore reports by code alone: a bad range does not touch the context's error text
    eng.lib.p2hot_last_error.restype = C.c_char_p
    before = eng.lib.p2hot_last_error(eng.ctx)
    assert eng.lib.p2hot_batch_leaves_wait(h, 3, 2) == _lib.EINVAL and eng.lib.p2hot_last_error(eng.ctx) == before
    eng.lib.p2hot_batch_free(h)


def test_salted_leaves_in_natural_order_and_asynchronously(eng, ora):
    """blinding = true (fri/oracle.rs:123-137): the salt columns travel with the leaf rows whatever the host copy's order"""
    rng = np.random.default_rng(17)
    W, S, log_n, rb, cap = 6, 4, 7, 3, 2
    n, N = 1 << log_n, 1 << (log_n + rb)
    cols, salts = rand_field(rng, W, n), rand_field(rng, S, N)
    o = ora.commit_salted(cols, salts, rb, cap, True)
    want = o["leaves"][_bitrev_perm(log_n + rb)]
    ptrs = (C.c_void_p * W)(*[cols[c].ctypes.data for c in range(W)])
    sp = (C.c_void_p * S)(*[salts[j].ctypes.data for j in range(S)])
    leaves = np.zeros((N, W + S), dtype=np.uint64)
    capv = np.zeros((1 << cap, 4), dtype=np.uint64)
    h = C.c_void_p()
    eng.check(eng.lib.p2hot_commit_salted(eng.ctx, ptrs, W, log_n, rb, cap, 1, _lib.LEAVES_ASYNC | _lib.LEAVES_NATURAL, sp, S, None, leaves.ctypes.data,
                                          None, capv.ctypes.data, C.byref(h)))
    eng.check(eng.lib.p2hot_batch_leaves_wait(h, 0, N))
    assert (capv == o["cap"]).all() and (leaves == want).all()
    eng.lib.p2hot_batch_free(h)


@pytest.mark.parametrize("W,log_n,rb,cap,S,block,grouped", [(37, 6, 3, 2, 0, 16, False), (37, 6, 3, 4, 4, 16, True), (135, 5, 3, 4, 0, 16, True),
                                                             (24, 4, 2, 3, 0, 8, False), (41, 7, 3, 4, 0, 8, True)])
def test_async_leaves_with_split_lanes(eng, ora, monkeypatch, W, log_n, rb, cap, S, block, grouped):
    """P2HOT_LEAVES_ASYNC with several column blocks (every commit of >= 2^22 values): the transforms and the transposition run on a
    lane of their own, the sponge absorbs the blocks' columns beside them on the context's stream, the leaf blocks leave on the leaf
    stream (csrc/host_prover.hpp, xsplit).  Every output against the oracle -- with the tail per group of cap subtrees and without,
    salted and not, and with P2HOT_HOST_ASYNC_SPLIT=0 (the single-stream leaves-first order) producing the same bytes"""
    from plonky2_amd.engine import Engine
    monkeypatch.setenv("P2HOT_HOST_BLOCK_COLS", str(block))
    monkeypatch.setenv("P2HOT_HOST_TAIL_MIN_LEAVES", "1" if grouped else str(1 << 40))
    rng = np.random.default_rng(W * 5 + log_n)
    n, N = 1 << log_n, 1 << (log_n + rb)
    cols = rand_field(rng, W, n, noncanonical=True)
    salts = rand_field(rng, max(S, 1), N)
    o = ora.commit_salted(cols, salts[:S], rb, cap, True) if S else ora.commit(cols, rb, cap, True)
    want = o["leaves"][_bitrev_perm(log_n + rb)]
    ptrs = (C.c_void_p * W)(*[cols[c].ctypes.data for c in range(W)])
    sp = (C.c_void_p * max(S, 1))(*[salts[j].ctypes.data for j in range(max(S, 1))])
    for split in ("1", "0"):
        monkeypatch.setenv("P2HOT_HOST_ASYNC_SPLIT", split)
        e2 = Engine(0, lib=eng.lib, memory=eng.mem)
        try:
            for _ in range(2):   # twice: the lanes are reused
                coeffs = np.zeros((W, n), dtype=np.uint64)
                leaves = np.zeros((N, W + S), dtype=np.uint64)
                digests = np.zeros((max(e2.num_digests(log_n + rb, cap), 1), 4), dtype=np.uint64)
                capv = np.zeros((1 << cap, 4), dtype=np.uint64)
                h = C.c_void_p()
                e2.check(e2.lib.p2hot_commit_salted(e2.ctx, ptrs, W, log_n, rb, cap, 1, _lib.LEAVES_ASYNC | _lib.LEAVES_NATURAL, sp, S, coeffs.ctypes.data,
                                                    leaves.ctypes.data, digests.ctypes.data, capv.ctypes.data, C.byref(h)))
                nd = e2.num_digests(log_n + rb, cap)
                assert (coeffs == o["coeffs"] % np.uint64(P)).all() and (capv == o["cap"]).all() and (digests[:nd] == o["digests"]).all(), split
                e2.check(e2.lib.p2hot_batch_leaves_wait(h, 0, N))
                assert (leaves == want).all(), split
                idx = np.array([1, N - 1], dtype=np.uint64)
                rows = np.zeros((2, W + S), dtype=np.uint64)
                e2.check(e2.lib.p2hot_batch_rows(h, idx.ctypes.data, 2, rows.ctypes.data))
                assert (rows == o["leaves"][idx.astype(np.int64)]).all()
                e2.lib.p2hot_batch_free(h)
        finally:
            e2.close()
