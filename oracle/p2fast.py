"""ctypes binding of the tuned CPU implementation (oracle/libp2fast.so, see p2fast.c).

TEST INFRASTRUCTURE ONLY, like everything under oracle/: bench.py's cpu_baseline leg times it
("port-tuned"), and the full-size GPU parity tests use it as the fast checker.  It is pinned
bit-for-bit to the faithful restatement (p2oracle.c) by tests/test_fast_oracle.py.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libp2fast.so")
P = 0xFFFFFFFF00000001
SCOPES = ("IFFT", "FFT + blinding", "transpose LDEs", "build Merkle tree")  # fri/oracle.rs:65-103

u64p = C.POINTER(C.c_uint64)


def build(force=False):
    src = os.path.join(_HERE, "p2fast.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libp2fast.so"] + (["-B"] if force else []))
    return _SO


def _load():
    build()
    lib = C.CDLL(_SO)
    lib.fast_commit.restype = C.c_int
    lib.fast_commit.argtypes = [u64p, C.c_size_t, C.c_uint, C.c_uint, C.c_uint, C.c_int, u64p, u64p, u64p, u64p,
                                C.POINTER(C.c_double)]
    lib.fast_poseidon.argtypes = [u64p, C.c_size_t]
    lib.fast_hash_rows.argtypes = [u64p, C.c_size_t, C.c_size_t, u64p]
    lib.fast_num_threads.restype = C.c_int
    return lib


lib = _load()


def _p(a):
    return a.ctypes.data_as(u64p) if a is not None else None


def set_num_threads(n):
    lib.fast_set_num_threads(C.c_int(int(n)))


def poseidon(states):
    """states: [count][12] -> permuted, canonical"""
    s = np.ascontiguousarray(np.array(states, dtype=np.uint64)).reshape(-1, 12).copy()
    lib.fast_poseidon(_p(s), s.shape[0])
    return s


def hash_rows(rows):
    """hash_or_noop of every row of [m][w] -> [m][4]"""
    rows = np.ascontiguousarray(np.asarray(rows, dtype=np.uint64))
    m, w = rows.shape
    out = np.zeros((m, 4), dtype=np.uint64)
    lib.fast_hash_rows(_p(rows), m, w, _p(out))
    return out


def commit(cols, rate_bits, cap_height, is_values=True, want_coeffs=True, want_leaves=False, want_digests=True,
           timed=None):
    """cols: [W][n] host.  Returns dict(coeffs, leaves, digests, cap); the optional arrays are None when not
    requested (a C3-sized leaf matrix is 9 GB).  `timed` accumulates seconds per TimingTree scope."""
    cols = np.ascontiguousarray(np.asarray(cols, dtype=np.uint64))
    W, n = cols.shape
    log_n = n.bit_length() - 1
    assert n == 1 << log_n
    N = n << rate_bits
    ncap = 1 << cap_height
    coeffs = np.zeros((W, n), dtype=np.uint64) if want_coeffs else None
    leaves = np.zeros((N, W), dtype=np.uint64) if want_leaves else None
    digests = np.zeros((2 * (N - ncap), 4), dtype=np.uint64) if want_digests else None
    cap = np.zeros((ncap, 4), dtype=np.uint64)
    secs = (C.c_double * 4)(0, 0, 0, 0)
    rc = lib.fast_commit(_p(cols), W, log_n, rate_bits, cap_height, 1 if is_values else 0, _p(coeffs), _p(leaves),
                         _p(digests), _p(cap), secs)
    if rc:
        raise RuntimeError("fast_commit failed (%d)" % rc)
    if timed is not None:
        for i, k in enumerate(SCOPES):
            timed[k] = timed.get(k, 0.0) + secs[i]
    return dict(coeffs=coeffs, leaves=leaves, digests=digests, cap=cap)
