#!/usr/bin/env python3
"""TEST HELPER (run as a subprocess by tests/test_gpu_fullsize.py::test_rccl_binding_without_torch).

The patched-plonky2 deployment mode: a process that has NOT imported torch binds RCCL through libp2hot's own dlopen -- no copy is
loaded yet, so it is /opt/rocm/lib/librccl.so.1 by path, not PyTorch's bundled one -- creates a one-rank communicator
(ncclGetUniqueId / ncclCommInitRank), runs the preflight all-gather (p2hot_comm_selftest: both exchange forms, checked and
timed) and one coset-sharded commit with DEVICE pointers (p2hot_commit_sharded_dev) whose cap, digests, leaves and coefficients it
compares with the CPU oracle.  ctypes + numpy only; device memory through the HIP runtime libp2hot.so already links.
Prints one JSON line: {"rccl": {"path", "version"}, "exchange_mode", "torch_imported": false, "checked": true}."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    assert "torch" not in sys.modules
    lib = C.CDLL(os.path.join(ROOT, "plonky2_amd", "libp2hot.so"))
    hip = C.CDLL("libamdhip64.so")           # the runtime libp2hot.so is linked against (already mapped)
    vp, sz = C.c_void_p, C.c_size_t
    lib.p2hot_last_error.restype = C.c_char_p
    lib.p2hot_last_error.argtypes = [vp]

    def check(rc, what, ctx=None):
        if rc != 0:
            raise SystemExit("%s failed (%d): %s" % (what, rc, lib.p2hot_last_error(ctx).decode() if ctx else ""))

    ctx = vp()
    lib.p2hot_ctx_create.argtypes = [C.c_int, vp, C.POINTER(vp)]
    check(lib.p2hot_ctx_create(0, None, C.byref(ctx)), "p2hot_ctx_create")
    uid = (C.c_uint8 * 128)()
    check(lib.p2hot_comm_unique_id(uid), "p2hot_comm_unique_id", ctx)
    comm = vp()
    lib.p2hot_comm_create_rccl.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_uint8), C.POINTER(vp)]
    check(lib.p2hot_comm_create_rccl(ctx, 0, 1, uid, C.byref(comm)), "p2hot_comm_create_rccl", ctx)
    lib.p2hot_comm_selftest.argtypes = [vp, sz]
    check(lib.p2hot_comm_selftest(comm, 1 << 20), "p2hot_comm_selftest", ctx)
    buf, ver = C.create_string_buffer(1024), C.c_int(0)
    lib.p2hot_rccl_info.argtypes = [C.c_char_p, sz, C.POINTER(C.c_int)]
    check(lib.p2hot_rccl_info(buf, len(buf), C.byref(ver)), "p2hot_rccl_info", ctx)
    lib.p2hot_comm_exchange_mode.argtypes = [vp]

    # one sharded commit through the communicator, device pointers, against the oracle
    from oracle import p2oracle as ora
    rng = np.random.default_rng(77)
    W, log_n, rb, cap = 11, 12, 3, 4
    n, N = 1 << log_n, 1 << (log_n + rb)
    cols = rng.integers(0, ora.P, size=(W, n), dtype=np.uint64)
    o = ora.commit(cols, rb, cap, True)
    hip.hipMalloc.argtypes = [C.POINTER(vp), sz]
    hip.hipMemcpy.argtypes = [vp, vp, sz, C.c_int]
    hip.hipFree.argtypes = [vp]

    def dmalloc(words):
        p = vp()
        assert hip.hipMalloc(C.byref(p), words * 8) == 0
        return p

    nd = 2 * (N - (1 << cap))
    d_cols, d_coeffs, d_lde, d_leaves, d_dig, d_cap = (dmalloc(w) for w in (W * n, W * n, W * N, N * W, nd * 4, (1 << cap) * 4))
    assert hip.hipMemcpy(d_cols, cols.ctypes.data, W * n * 8, 1) == 0            # hipMemcpyHostToDevice
    lib.p2hot_commit_sharded_dev.argtypes = [vp, vp, vp, sz, sz, C.c_uint, C.c_uint, C.c_uint, C.c_int, C.c_int, C.c_uint,
                                             vp, vp, sz, vp, vp, vp]
    check(lib.p2hot_commit_sharded_dev(ctx, comm, d_cols, n, W, log_n, rb, cap, 1, 1, 2, d_coeffs, d_lde, N, d_leaves, d_dig, d_cap),
          "p2hot_commit_sharded_dev", ctx)
    lib.p2hot_ctx_sync.argtypes = [vp]
    check(lib.p2hot_ctx_sync(ctx), "p2hot_ctx_sync", ctx)

    def back(p, shape):
        a = np.empty(shape, dtype=np.uint64)
        assert hip.hipMemcpy(a.ctypes.data, p, a.nbytes, 2) == 0                # hipMemcpyDeviceToHost
        return a

    assert (back(d_cap, (1 << cap, 4)) == o["cap"]).all(), "cap"
    assert (back(d_dig, (nd, 4)) == o["digests"]).all(), "digests"
    assert (back(d_leaves, (N, W)) == o["leaves"]).all(), "leaves"
    assert (back(d_coeffs, (W, n)) % np.uint64(ora.P) == o["coeffs"] % np.uint64(ora.P)).all(), "coefficients"
    for p in (d_cols, d_coeffs, d_lde, d_leaves, d_dig, d_cap):
        hip.hipFree(p)
    mode = lib.p2hot_comm_exchange_mode(comm)
    lib.p2hot_comm_destroy.argtypes = [vp]
    lib.p2hot_comm_destroy(comm)
    lib.p2hot_ctx_destroy.argtypes = [vp]
    lib.p2hot_ctx_destroy(ctx)
    assert "torch" not in sys.modules
    print(json.dumps({"rccl": {"path": buf.value.decode(), "version": ver.value}, "exchange_mode": mode,
                      "torch_imported": False, "checked": True}))


if __name__ == "__main__":
    main()
