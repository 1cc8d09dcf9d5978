"""The C-ABI library builds for gfx950, loads without a GPU and exports every symbol include/p2hot.h
declares; the host-side argument validation mirrors the reference's panics."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from tests.conftest import P as P_
from tests.conftest import ROOT


def _declared():
    h = open(os.path.join(ROOT, "include", "p2hot.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(p2hot_[a-z0-9_]+)\s*\(", h)))


def test_header_and_binding_agree():
    from plonky2_amd import _lib
    assert _declared() == sorted(_lib.SIGNATURES), "include/p2hot.h and plonky2_amd/_lib.py disagree"


def test_every_entry_point_is_exercised():
    """each function of include/p2hot.h is called by a test, or by the host-side mirror the parity tests drive"""
    import glob
    text = ""
    for f in glob.glob(os.path.join(ROOT, "tests", "*.py")) + glob.glob(os.path.join(ROOT, "plonky2_amd", "**", "*.py"), recursive=True):
        if not f.endswith("_lib.py"):
            text += open(f).read()
    missing = [s for s in _declared() if not re.search(r"\b%s\b" % s, text) and s != "p2hot_version"]
    assert not missing, missing


def test_product_library_builds_and_exports_every_symbol():
    import __graft_entry__ as ge
    so = ge.build_product()
    lib = C.CDLL(so)  # loads without a GPU (no compute calls here)
    for name in _declared():
        assert hasattr(lib, name), name
    lib.p2hot_is_emulated.restype = C.c_int
    assert lib.p2hot_is_emulated() == 0
    lib.p2hot_num_digests.restype = C.c_size_t
    assert lib.p2hot_num_digests(10, 4) == 2 * (1024 - 16)
    assert lib.p2hot_num_digests(4, 4) == 0


def test_product_has_gfx950_code_object():
    import subprocess
    import __graft_entry__ as ge
    so = ge.build_product()
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", so], capture_output=True, text=True).stdout
    assert "gfx950" in out


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from plonky2_amd import Engine
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Engine(0)


def test_product_never_imports_oracle_or_emulator():
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "plonky2_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                t = open(os.path.join(d, f)).read()
                if re.search(r"^\s*(from|import)\s+(oracle|tests)\b", t, re.M) or "p2oracle" in t or "hip_emu.cpp" in t:
                    bad.append(f)
    assert not bad, bad


def test_shape_errors_match_reference_panics(emu):
    from plonky2_amd import _lib
    # cap_height > log2(leaves): merkle_tree.rs:195-200 asserts
    leaves = np.zeros((4, 5), dtype=np.uint64)
    with pytest.raises(_lib.P2HotError, match="cap_height"):
        emu.merkle(leaves, 1, 5, 2, 3)
    # rows that are not whole coset blocks
    co = np.zeros((1, 8), dtype=np.uint64)
    with pytest.raises(_lib.P2HotError, match="coset blocks"):
        emu.coset_lde(co, 3, 1, row_begin=4, row_count=8)
    # blinding needs the caller's salts (the library draws no random numbers), of the right shape
    from plonky2_amd.fri.oracle import PolynomialBatch
    with pytest.raises(ValueError, match="salt"):
        PolynomialBatch.from_values(co, 1, True, 0, engine=emu)
    with pytest.raises(ValueError, match="salt"):
        PolynomialBatch.from_values(co, 1, True, 0, engine=emu, salts=np.zeros((4, 8), dtype=np.uint64))
    with pytest.raises(ValueError):
        PolynomialBatch.from_values(np.zeros((1, 6), dtype=np.uint64), 1, False, 0, engine=emu)


def test_memory_backends_have_the_same_surface():
    """the product's TorchMemory and the test emulator's HostMemory must expose the same methods (the multi-GPU
    path calls them); checked by name so it runs without a GPU"""
    from plonky2_amd.engine import TorchMemory
    from tests.emu_backend import HostMemory
    pub = lambda c: {m for m in dir(c) if not m.startswith("_")}
    assert pub(HostMemory) <= pub(TorchMemory) | {"DevArray"}, pub(HostMemory) - pub(TorchMemory)
    assert pub(TorchMemory) <= pub(HostMemory), pub(TorchMemory) - pub(HostMemory)


def test_one_host_call_at_a_time_per_context(emu):
    """A context runs one host-pointer call at a time: a second thread entering while a commit runs gets P2HOT_EBUSY
    (no data race on the context's scratch blocks), and the context works normally afterwards."""
    import threading
    import time
    from plonky2_amd import _lib
    rng = np.random.default_rng(0)
    W, log_n = 48, 9
    cols = rng.integers(0, 2**63, size=(W, 1 << log_n), dtype=np.uint64)
    ptrs = (C.c_void_p * W)(*[cols[c].ctypes.data for c in range(W)])
    cap = np.zeros((4, 4), dtype=np.uint64)
    rc_a = []

    def worker():
        rc_a.append(emu.lib.p2hot_commit(emu.ctx, ptrs, W, log_n, 3, 2, 1, 0, None, None, None, cap.ctypes.data, None))

    t = threading.Thread(target=worker)
    t.start()
    seen_busy = False
    deadline = time.time() + 60
    while t.is_alive() and time.time() < deadline:
        rc = emu.lib.p2hot_ctx_trim(emu.ctx)          # ctypes releases the GIL: this really runs beside the commit
        if rc == _lib.EBUSY:
            seen_busy = True
            break
        time.sleep(0.001)
    t.join(120)
    assert rc_a == [_lib.OK]
    assert seen_busy, "the second caller was never refused while the commit was running"
    assert emu.lib.p2hot_ctx_trim(emu.ctx) == _lib.OK
    assert cap.any()


def test_commit_without_polynomials_is_einval(eng):
    """W = 0: the reference panics on polynomials[0] (fri/oracle.rs:90); every commit entry point returns P2HOT_EINVAL with a
    message of its own (round-3 review: p2hot_commit returned P2HOT_OK and left a stale error text)"""
    from plonky2_amd import _lib
    cap = np.zeros((4, 4), dtype=np.uint64)
    h = C.c_void_p()
    rc = eng.lib.p2hot_commit(eng.ctx, None, 0, 4, 3, 2, 1, 0, None, None, None, cap.ctypes.data, C.byref(h))
    assert rc == _lib.EINVAL and not h.value
    assert b"no polynomials" in eng.lib.p2hot_last_error(eng.ctx)
    rc = eng.lib.p2hot_commit_salted(eng.ctx, None, 0, 4, 3, 2, 1, 0, None, 0, None, None, None, cap.ctypes.data, None)
    assert rc == _lib.EINVAL
    rc = eng.lib.p2hot_commit_many(eng.ctx, None, 2, 0, 4, 3, 2, 1, None, None, None, None)
    assert rc == _lib.EINVAL
    assert eng.lib.p2hot_commit_many(eng.ctx, None, 0, 3, 4, 3, 2, 1, None, None, None, None) == _lib.OK  # no proofs: nothing to do
    rc = eng.lib.p2hot_commit_dev(eng.ctx, None, 16, 0, 4, 3, 2, 1, 0, 128, None, 16, None, 128, None, None, None)
    assert rc == _lib.EINVAL


def test_commit_cols_einval_never_consumes_the_set(eng):
    """every P2HOT_EINVAL of p2hot_commit_cols (bad rate / cap height / flags, a borrowed view) is raised
    before the set is touched, so the caller still owns the handle -- and can commit it afterwards"""
    from plonky2_amd import _lib
    rng = np.random.default_rng(3)
    W, log_n = 3, 5
    cols = rng.integers(0, P_, size=(W, 1 << log_n), dtype=np.uint64)
    ptrs = (C.c_void_p * W)(*[cols[c].ctypes.data for c in range(W)])
    h = C.c_void_p()
    assert eng.lib.p2hot_cols_upload(eng.ctx, ptrs, W, log_n, C.byref(h)) == _lib.OK
    cap = np.zeros((4, 4), dtype=np.uint64)
    out = C.c_void_p()
    assert eng.lib.p2hot_commit_cols(eng.ctx, h, 3, 2, 1, 0xF0, None, None, None, cap.ctypes.data, C.byref(out)) == _lib.EINVAL  # flags
    assert eng.lib.p2hot_commit_cols(eng.ctx, h, 3, 99, 1, 0, None, None, None, cap.ctypes.data, C.byref(out)) == _lib.EINVAL    # cap height
    assert eng.lib.p2hot_commit_cols(eng.ctx, h, 60, 2, 1, 0, None, None, None, cap.ctypes.data, C.byref(out)) == _lib.EINVAL    # log N > 32
    assert eng.lib.p2hot_cols_width(h) == W  # still alive, still ours
    assert eng.lib.p2hot_commit_cols(eng.ctx, h, 3, 2, 1, 0, None, None, None, cap.ctypes.data, C.byref(out)) == _lib.OK
    ref = np.zeros((4, 4), dtype=np.uint64)
    assert eng.lib.p2hot_commit(eng.ctx, ptrs, W, log_n, 3, 2, 1, 0, None, None, None, ref.ctypes.data, None) == _lib.OK
    assert (cap == ref).all() and cap.any()
    eng.lib.p2hot_batch_free(out)


def test_pinned_host_blocks_are_cached_per_context(eng, ora):
    """p2hot_host_alloc / p2hot_host_free: the flat leaf matrix the Rust shim keeps behind MerkleTree::get lives in a pinned block
    of the context's cache -- a commit writes it, a freed block is handed out again for the next request of that size,
    p2hot_ctx_trim gives cached blocks back"""
    from plonky2_amd import _lib
    W, log_n, rb, cap = 5, 5, 3, 2
    n, N = 1 << log_n, 1 << (log_n + rb)
    cols = np.random.default_rng(11).integers(0, P_, size=(W, n), dtype=np.uint64)
    ptrs = (C.c_void_p * W)(*[cols[c].ctypes.data for c in range(W)])
    blk = C.c_void_p()
    assert eng.lib.p2hot_host_alloc(eng.ctx, N * W * 8, C.byref(blk)) == _lib.OK and blk.value
    capv = np.zeros((1 << cap, 4), dtype=np.uint64)
    assert eng.lib.p2hot_commit(eng.ctx, ptrs, W, log_n, rb, cap, 1, 0, None, blk, None, capv.ctypes.data, None) == _lib.OK
    leaves = np.ctypeslib.as_array(C.cast(blk, C.POINTER(C.c_uint64)), shape=(N * W,)).reshape(N, W).copy()
    o = ora.commit(cols, rb, cap, True)
    assert (leaves == o["leaves"]).all() and (capv == o["cap"]).all()
    first = blk.value
    eng.lib.p2hot_host_free(eng.ctx, blk)
    eng.lib.p2hot_host_free(eng.ctx, blk)            # a second free of the same block is ignored
    again = C.c_void_p()
    assert eng.lib.p2hot_host_alloc(eng.ctx, N * W * 8 - 64, C.byref(again)) == _lib.OK
    assert again.value == first, "the cached block was not reused"
    eng.lib.p2hot_host_free(eng.ctx, again)
    assert eng.lib.p2hot_ctx_trim(eng.ctx) == _lib.OK
    assert eng.lib.p2hot_host_alloc(None, 8, C.byref(again)) == _lib.EINVAL
