"""Allocation-failure injection (CPU tier: the emulated runtime's `fail_malloc_at`): for every allocation the host-pointer
entry points make -- p2hot_commit, p2hot_quotient_polys, p2hot_prove_openings on one context, p2hot_group_commit on eight --
the call in which that allocation fails returns P2HOT_ENOMEM with a text, leaves no block checked out (after p2hot_ctx_trim the
runtime's live-allocation count is what it was before the call), and the SAME context then completes the same call with the
oracle's bytes.  That is the reference's contract for a failed call -- `anyhow` / panic unwinding leaves no state behind --
on an error path that neither the GPU nor the plain emulator can reach (a rank of C5 holds 27 GB + caches; hipMalloc does
not fail on a 288 GB part until it does)."""
import ctypes as C

import numpy as np
import pytest

from plonky2_amd import _lib
from tests.conftest import rand_field


def _hooks(lib):
    lib.p2hot_emu_fault.argtypes = [C.c_char_p, C.c_int]
    lib.p2hot_emu_fault.restype = C.c_int
    return lambda what, v=0: lib.p2hot_emu_fault(what.encode(), v)


def _sweep(eng, trim, call, check_ok, max_allocs=400, once_every=1):
    """call() -> rc (and leaves its outputs where check_ok() reads them).  Returns the number of allocation sites swept."""
    fault = _hooks(eng.lib)
    trim()
    assert call() == _lib.OK          # warm: one-time tables and caches of the context exist from here on
    check_ok()
    trim()
    base = fault("live_allocs")
    fault("fail_malloc_at", 1 << 30)  # count the allocations of one call that starts from an empty block cache
    assert call() == _lib.OK
    n = fault("malloc_calls")
    fault("fail_malloc_at", 0)
    trim()
    assert fault("live_allocs") == base
    assert 0 < n <= max_allocs, n
    for k in range(1, n + 1):
        # the device is full from allocation k on: the call fails as a whole, cleanly
        fault("fail_malloc_at", k)
        rc = call()
        fired = fault("malloc_calls") >= k
        fault("fail_malloc_at", 0)
        assert fired
        assert rc == _lib.ENOMEM, "allocations from %d of %d failed and the call returned %r" % (k, n, rc)
        trim()
        assert fault("live_allocs") == base, "allocation %d of %d failed: %d blocks leaked" % (k, n, fault("live_allocs") - base)
        # a transient failure of allocation k alone: recovered (the block cache gives its free blocks back and retries) or reported,
        # never a wrong answer
        if k % once_every:
            continue
        fault("fail_malloc_once", k)
        rc = call()
        fault("fail_malloc_once", 0)
        assert rc in (_lib.OK, _lib.ENOMEM), rc
        if rc == _lib.OK:
            check_ok()
        trim()
        assert fault("live_allocs") == base
    assert call() == _lib.OK          # the same context, after every failure: still right
    check_ok()
    trim()
    assert fault("live_allocs") == base
    return n


def test_every_allocation_of_a_host_pointer_commit_may_fail(emu, ora):
    rng = np.random.default_rng(3)
    W, log_n, rb, cap = 37, 9, 3, 4
    n, N = 1 << log_n, 1 << (log_n + rb)
    cols = rand_field(rng, W, n)
    o = ora.commit(cols, rb, cap, True)
    ptrs = (C.c_void_p * W)(*[cols[c].ctypes.data for c in range(W)])
    coeffs, leaves = np.zeros((W, n), dtype=np.uint64), np.zeros((N, W), dtype=np.uint64)
    digests, capv = np.zeros((emu.num_digests(log_n + rb, cap), 4), dtype=np.uint64), np.zeros((1 << cap, 4), dtype=np.uint64)
    last = {}

    def call():
        for a in (coeffs, leaves, digests, capv):
            a[...] = 0
        h = C.c_void_p()
        rc = emu.lib.p2hot_commit(emu.ctx, ptrs, W, log_n, rb, cap, 1, 0, coeffs.ctypes.data, leaves.ctypes.data, digests.ctypes.data,
                                  capv.ctypes.data, C.byref(h))
        last["msg"] = emu.lib.p2hot_last_error(emu._ctx).decode()
        if rc == _lib.OK:
            emu.lib.p2hot_batch_free(h)
        else:
            assert not h.value and last["msg"], "a failed commit returns no handle and says why"
        return rc

    def ok():
        assert (coeffs == o["coeffs"]).all() and (leaves == o["leaves"]).all() and (digests == o["digests"]).all() and (capv == o["cap"]).all()
    n_sites = _sweep(emu, lambda: emu.check(emu.lib.p2hot_ctx_trim(emu.ctx)), call, ok)
    assert n_sites >= 5   # columns, LDE, leaf matrix, digests, cap at least


def test_every_allocation_of_an_asynchronous_leaf_copy_may_fail(emu, ora):
    """P2HOT_LEAVES_ASYNC: a failed call leaves no copy in flight and no staging block behind"""
    rng = np.random.default_rng(4)
    W, log_n, rb, cap = 9, 10, 3, 4
    n, N = 1 << log_n, 1 << (log_n + rb)
    cols = rand_field(rng, W, n)
    o = ora.commit(cols, rb, cap, True)
    ptrs = (C.c_void_p * W)(*[cols[c].ctypes.data for c in range(W)])
    leaves, capv = np.zeros((N, W), dtype=np.uint64), np.zeros((1 << cap, 4), dtype=np.uint64)

    def call():
        leaves[...] = 0
        h = C.c_void_p()
        rc = emu.lib.p2hot_commit(emu.ctx, ptrs, W, log_n, rb, cap, 1, _lib.LEAVES_ASYNC, None, leaves.ctypes.data, None, capv.ctypes.data, C.byref(h))
        if rc == _lib.OK:
            emu.lib.p2hot_batch_free(h)
        return rc

    def ok():
        assert (leaves == o["leaves"]).all() and (capv == o["cap"]).all()
    _sweep(emu, lambda: emu.check(emu.lib.p2hot_ctx_trim(emu.ctx)), call, ok)


def test_every_allocation_of_quotient_polys_and_prove_openings_may_fail(emu, ora):
    from plonky2_amd.fri.oracle import FriBatchInfo, PolynomialBatch, prove_openings
    from plonky2_amd.iop.challenger import Challenger
    from plonky2_amd.plonk.prover import compute_quotient_polys
    from tests.test_permutation import _oracle_quotient, _quotient_instance
    rng = np.random.default_rng(5)
    q = _quotient_instance(ora, rng, 12, 4, 5, 3, 2)
    rb, cap = q["rate_bits"], q["cap"]
    b_w, b_cs, b_z = (PolynomialBatch.from_values(q[name], rb, False, cap, engine=emu) for name in ("wires", "cs", "zs"))
    exp = _oracle_quotient(ora, q)
    want = np.stack([ora.coset_ifft(exp[a])[:q["degree"] * q["n"]] for a in range(q["nc"])]).reshape(q["nc"] * q["degree"], q["n"])
    out = {}

    def call_q():
        try:
            out["cols"] = compute_quotient_polys(b_w, b_cs, q["sigmas_first"], b_z, q["k"], q["degree"], q["betas"], q["gammas"], q["alphas"], engine=emu).host()
            return _lib.OK
        except _lib.P2HotError as e:
            assert str(e)
            return e.code
    trim = lambda: emu.check(emu.lib.p2hot_ctx_trim(emu.ctx))   # noqa: E731
    _sweep(emu, trim, call_q, lambda: np.testing.assert_array_equal(out["cols"], want))

    # prove_openings over the three commitments: every polynomial at one point, the first oracle at a second
    widths = [b._W for b in (b_w, b_cs, b_z)]
    allp = [(oi, pi) for oi, w in enumerate(widths) for pi in range(w)]
    z0, z1 = rand_field(rng, 2), rand_field(rng, 2)
    batches = [FriBatchInfo(z0, allp), FriBatchInfo(z1, [(0, pi) for pi in range(widths[0])])]
    ref = {}

    def call_p():
        try:
            ch = Challenger(emu)
            ch.observe_elements(np.arange(5, dtype=np.uint64))
            pf = prove_openings(batches, [b_w, b_cs, b_z], ch, rb, cap, [2, 1], 3, 4, engine=emu)
        except _lib.P2HotError as e:
            assert str(e)
            return e.code
        out["proof"] = (pf["pow_witness"], pf["query_indices"], np.asarray(pf["final_poly"]).tolist(),
                        [np.asarray(c).tolist() for c in pf["commit_phase_merkle_caps"]])
        ref.setdefault("proof", out["proof"])
        return _lib.OK
    _sweep(emu, trim, call_p, lambda: (out["proof"] == ref["proof"]) or pytest.fail("the proof changed after the failures"))


def test_every_allocation_of_an_eight_rank_group_commit_may_fail(ora):
    """p2hot_group_commit on eight DISTINCT emulated devices (the C5 arrangement): whichever rank's allocation fails, the call fails as
    a whole with ENOMEM, no rank keeps a block, no rank waits for a peer that gave up, and the group commits correctly afterwards"""
    from plonky2_amd.distributed import GroupCommit
    from tests.emu_backend import emu_lib
    lib = emu_lib()
    fault = _hooks(lib)
    rng = np.random.default_rng(6)
    W, log_n, rb, cap, world = 9, 5, 3, 4, 8
    cols = rand_field(rng, W, 1 << log_n)
    o = ora.commit(cols, rb, cap, True)
    g = GroupCommit(lib, world, list(range(world)))
    last = {}

    def call():
        try:
            r = g.commit(cols, rb, cap, True, want_leaves=True, pipeline_chunks=2)
        except _lib.P2HotError as e:
            assert str(e)
            return e.code
        last["r"] = {k: r[k] for k in ("coeffs", "leaves", "digests", "cap")}
        r["free"]()
        return _lib.OK

    def ok():
        r = last["r"]
        assert (r["coeffs"] == o["coeffs"]).all() and (r["leaves"] == o["leaves"]).all() and (r["digests"] == o["digests"]).all() and (r["cap"] == o["cap"]).all()

    def trim():
        for i in range(world):
            assert lib.p2hot_ctx_trim(lib.p2hot_group_ctx(g._h, i)) == _lib.OK
    try:
        n_sites = _sweep(type("E", (), {"lib": lib})(), trim, call, ok, max_allocs=1500, once_every=3)
        assert n_sites >= world
    finally:
        fault("fail_malloc_at", 0)
        g.close()
