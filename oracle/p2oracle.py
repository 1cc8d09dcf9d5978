"""ctypes binding of the CPU oracle (oracle/libp2oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never from plonky2_amd/ (the product path).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libp2oracle.so")
P = 0xFFFFFFFF00000001
COSET_SHIFT = 14293326489335486720

u64p = C.POINTER(C.c_uint64)


def build(force=False):
    src = os.path.join(_HERE, "p2oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


def _load():
    if not os.path.exists(_SO):
        build()
    lib = C.CDLL(_SO)
    lib.ora_gl_mul.restype = C.c_uint64
    lib.ora_gl_mul.argtypes = [C.c_uint64, C.c_uint64]
    lib.ora_gl_add.restype = C.c_uint64
    lib.ora_gl_add.argtypes = [C.c_uint64, C.c_uint64]
    lib.ora_gl_sub.restype = C.c_uint64
    lib.ora_gl_sub.argtypes = [C.c_uint64, C.c_uint64]
    lib.ora_gl_inv.restype = C.c_uint64
    lib.ora_gl_inv.argtypes = [C.c_uint64]
    lib.ora_gl_pow.restype = C.c_uint64
    lib.ora_gl_pow.argtypes = [C.c_uint64, C.c_uint64]
    lib.ora_gl_root_of_unity.restype = C.c_uint64
    lib.ora_gl_root_of_unity.argtypes = [C.c_uint]
    lib.ora_challenger_get.restype = C.c_uint64
    lib.ora_fri_pow.restype = C.c_uint64
    lib.ora_num_threads.restype = C.c_int
    return lib


lib = _load()


def _p(a):
    return a.ctypes.data_as(u64p) if a is not None else None


def arr(x, shape=None):
    a = np.ascontiguousarray(np.array(x, dtype=np.uint64))
    return a.reshape(shape) if shape is not None else a


def set_num_threads(n):
    lib.ora_set_num_threads(C.c_int(int(n)))


def usable_cores():
    """host cores this job may really use: min(affinity mask, cgroup CPU quota)"""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return n


def num_threads():
    return lib.ora_num_threads()


def gl_mul(a, b):
    return lib.ora_gl_mul(a, b)


def gl_inv(a):
    return lib.ora_gl_inv(a)


def gl_pow(a, e):
    return lib.ora_gl_pow(a, e)


def root_of_unity(log_n):
    return lib.ora_gl_root_of_unity(log_n)


def reverse_index_bits(a, elem_words=1):
    a = arr(a).copy()
    n = a.size // elem_words
    lib.ora_reverse_index_bits(_p(a), C.c_size_t(n), C.c_size_t(elem_words))
    return a


def fft(v, r=0):
    v = arr(v).copy()
    lib.ora_fft(_p(v), C.c_uint(v.size.bit_length() - 1), C.c_uint(r))
    return v


def ifft(v):
    v = arr(v).copy()
    lib.ora_ifft(_p(v), C.c_uint(v.size.bit_length() - 1))
    return v


def coset_fft(c, shift=COSET_SHIFT, zero_factor=0):
    c = arr(c).copy()
    lib.ora_coset_fft(_p(c), C.c_uint(c.size.bit_length() - 1), C.c_uint64(shift), C.c_uint(zero_factor))
    return c


def coset_ifft(v, shift=COSET_SHIFT):
    v = arr(v).copy()
    lib.ora_coset_ifft(_p(v), C.c_uint(v.size.bit_length() - 1), C.c_uint64(shift))
    return v


def poseidon(state, naive=False):
    s = arr(state).copy()
    assert s.size == 12
    (lib.ora_poseidon_naive if naive else lib.ora_poseidon)(_p(s))
    return s


def hash_no_pad(x):
    x = arr(x)
    out = np.zeros(4, dtype=np.uint64)
    lib.ora_hash_no_pad(_p(x), C.c_size_t(x.size), _p(out))
    return out


def hash_or_noop(x):
    x = arr(x)
    out = np.zeros(4, dtype=np.uint64)
    lib.ora_hash_or_noop(_p(x), C.c_size_t(x.size), _p(out))
    return out


def two_to_one(l, r):
    l, r = arr(l), arr(r)
    out = np.zeros(4, dtype=np.uint64)
    lib.ora_two_to_one(_p(l), _p(r), _p(out))
    return out


def merkle_tree(leaves, cap_height):
    """leaves: [n][w] -> (digests [2(n-2^cap)][4], cap [2^cap][4])"""
    leaves = arr(leaves)
    n, w = leaves.shape
    ncap = 1 << cap_height
    digests = np.zeros((2 * (n - ncap), 4), dtype=np.uint64)
    cap = np.zeros((ncap, 4), dtype=np.uint64)
    lib.ora_merkle_tree(_p(leaves), C.c_size_t(n), C.c_size_t(w), C.c_uint(cap_height), _p(digests), _p(cap))
    return digests, cap


def merkle_prove(leaf_index, n, cap_height, digests):
    digests = arr(digests)
    nl = (n.bit_length() - 1) - cap_height
    sib = np.zeros((nl, 4), dtype=np.uint64)
    lib.ora_merkle_prove(C.c_size_t(leaf_index), C.c_size_t(n), C.c_uint(cap_height), _p(digests), _p(sib))
    return sib


def merkle_verify(leaf, leaf_index, cap, siblings):
    leaf, cap, siblings = arr(leaf), arr(cap), arr(siblings)
    cap_height = (cap.shape[0]).bit_length() - 1
    return bool(lib.ora_merkle_verify(_p(leaf), C.c_size_t(leaf.size), C.c_size_t(leaf_index), _p(cap),
                                      C.c_uint(cap_height), _p(siblings), C.c_uint(siblings.shape[0] if siblings.size else 0)))


def commit(cols, rate_bits, cap_height, is_values=True, timed=None):
    """cols: [W][n].  Returns dict(coeffs [W][n], leaves [N][W], digests, cap)."""
    cols = arr(cols)
    W, n = cols.shape
    log_n = n.bit_length() - 1
    N = n << rate_bits
    ncap = 1 << cap_height
    coeffs = np.zeros((W, n), dtype=np.uint64)
    leaves = np.zeros((N, W), dtype=np.uint64)
    digests = np.zeros((2 * (N - ncap), 4), dtype=np.uint64)
    cap = np.zeros((ncap, 4), dtype=np.uint64)
    secs = (C.c_double * 4)(0, 0, 0, 0)
    lib.ora_commit_timed(_p(cols), C.c_size_t(W), C.c_uint(log_n), C.c_uint(rate_bits), C.c_uint(cap_height),
                         C.c_int(1 if is_values else 0), _p(coeffs), _p(leaves), _p(digests), _p(cap), secs)
    if timed is not None:
        for i, k in enumerate(("IFFT", "FFT + blinding", "transpose LDEs", "build Merkle tree")):
            timed[k] = timed.get(k, 0.0) + secs[i]
    return dict(coeffs=coeffs, leaves=leaves, digests=digests, cap=cap)


def commit_salted(cols, salts, rate_bits, cap_height, is_values=True):
    """blinding = true (oracle.rs:114-139): the SALT_SIZE salt vectors (length N, in the order F::rand_vec produced them) are
    appended to the LDE-value vectors BEFORE transpose + reverse_index_bits (:97-98), so leaf L ends in salt_j[reverse_bits(L)];
    the tree is built over the widened leaves.  Returns dict(coeffs, leaves [N][W + S], digests, cap)."""
    r = commit(cols, rate_bits, cap_height, is_values)
    salts = arr(salts) % np.uint64(P)
    N = r["leaves"].shape[0]
    bits = N.bit_length() - 1
    rev = np.array([int(format(i, "0%db" % bits)[::-1], 2) if bits else 0 for i in range(N)], dtype=np.int64)
    leaves = np.ascontiguousarray(np.concatenate([r["leaves"], salts[:, rev].T], axis=1))
    digests, cap = merkle_tree(leaves, cap_height)
    return dict(coeffs=r["coeffs"], leaves=leaves, digests=digests, cap=cap)


class ChallengerState(C.Structure):
    _fields_ = [("state", C.c_uint64 * 12), ("inb", C.c_uint64 * 8), ("n_in", C.c_uint32),
                ("outb", C.c_uint64 * 8), ("n_out", C.c_uint32)]


class Challenger:
    """plonky2/src/iop/challenger.rs:16-153 (oracle side)."""

    def __init__(self):
        self.s = ChallengerState()
        lib.ora_challenger_init(C.byref(self.s))

    def observe_elements(self, e):
        e = arr(e).reshape(-1)
        lib.ora_challenger_observe(C.byref(self.s), _p(e), C.c_size_t(e.size))

    def observe_cap(self, cap):
        self.observe_elements(cap)

    def get_challenge(self):
        return int(lib.ora_challenger_get(C.byref(self.s)))

    def get_n_challenges(self, n):
        return [self.get_challenge() for _ in range(n)]

    def get_extension_challenge(self):
        return self.get_n_challenges(2)

    def clone(self):
        c = Challenger()
        C.memmove(C.byref(c.s), C.byref(self.s), C.sizeof(ChallengerState))
        return c


def fri_commit(coeffs, rate_bits, cap_height, arity_bits, challenger):
    """coeffs: [N][2] extension coefficients.  Returns dict(leaves, digests, caps (lists per round),
    betas [[b0,b1]..], final [[c0,c1]..])."""
    coeffs = arr(coeffs)
    N = coeffs.shape[0]
    log_N = N.bit_length() - 1
    ncap = 1 << cap_height
    m = N
    sizes = []
    for ab in arity_bits:
        nl = m >> ab
        sizes.append((m, nl, 2 * (nl - ncap)))
        m >>= ab
    n_final = m >> rate_bits
    leaves = np.zeros(sum(s[0] for s in sizes) * 2, dtype=np.uint64)
    digests = np.zeros(max(1, sum(s[2] for s in sizes) * 4), dtype=np.uint64)
    caps = np.zeros(len(sizes) * ncap * 4, dtype=np.uint64)
    betas = np.zeros((len(sizes), 2), dtype=np.uint64)
    final = np.zeros((n_final, 2), dtype=np.uint64)
    ab = (C.c_uint * len(arity_bits))(*arity_bits)
    lib.ora_fri_commit(_p(coeffs), C.c_uint(log_N), C.c_uint(rate_bits), C.c_uint(cap_height), ab,
                       C.c_uint(len(arity_bits)), C.byref(challenger.s), _p(leaves), _p(digests), _p(caps),
                       _p(betas), _p(final))
    out = dict(leaves=[], digests=[], caps=[], betas=betas, final=final)
    lo = do = 0
    for i, (mi, nl, nd) in enumerate(sizes):
        out["leaves"].append(leaves[lo:lo + 2 * mi].reshape(nl, -1))
        out["digests"].append(digests[do:do + 4 * nd].reshape(nd, 4))
        out["caps"].append(caps[i * ncap * 4:(i + 1) * ncap * 4].reshape(ncap, 4))
        lo += 2 * mi
        do += 4 * nd
    return out


def fri_pow(challenger, pow_bits):
    return int(lib.ora_fri_pow(C.byref(challenger.s), C.c_uint(pow_bits)))


def reduce_polys_base(polys, alpha):
    polys = arr(polys)
    k, n = polys.shape
    ptrs = (u64p * k)(*[polys[i].ctypes.data_as(u64p) for i in range(k)])
    out = np.zeros((n, 2), dtype=np.uint64)
    a = arr(alpha)
    lib.ora_reduce_polys_base(ptrs, C.c_size_t(k), C.c_size_t(n), _p(a), _p(out))
    return out


def divide_by_linear(poly, z):
    poly = arr(poly)
    n = poly.shape[0]
    out = np.zeros((n, 2), dtype=np.uint64)
    zz = arr(z)
    lib.ora_divide_by_linear(_p(poly), C.c_size_t(n), _p(zz), _p(out))
    return out


def eval_polys_ext(polys, z):
    """OpeningSet::new's eval_commitment (plonk/proof.rs:314-327): [n_polys][2] = every polynomial at the extension point z"""
    polys = arr(polys)
    k, n = polys.shape
    ptrs = (u64p * k)(*[polys[i].ctypes.data_as(u64p) for i in range(k)])
    out = np.zeros((k, 2), dtype=np.uint64)
    zz = arr(z)
    lib.ora_eval_polys_ext(ptrs, C.c_size_t(k), C.c_size_t(n), _p(zz), _p(out))
    return out


def partial_products(wires, sigmas, k_is, degree, beta, gamma):
    """wires_permutation_partial_products_and_zs (plonk/prover.rs:392-449): [num_prods + 1][n], Z last"""
    wires, sigmas, k = arr(wires), arr(sigmas), arr(k_is)
    r, n = wires.shape
    num_chunks = -(-r // degree)
    out = np.zeros((num_chunks, n), dtype=np.uint64)
    rc = lib.ora_partial_products(_p(wires), _p(sigmas), _p(k), C.c_size_t(r), C.c_uint(n.bit_length() - 1),
                                  C.c_size_t(degree), C.c_uint64(int(beta)), C.c_uint64(int(gamma)), _p(out))
    if rc:
        raise ZeroDivisionError("Tried to invert zero")
    return out



def quotient_permutation(wires_leaves, cs_leaves, sigmas_first, zs_leaves, log_n, rate_bits, k_is, qdf, betas, gammas, alphas, gate_sums=None):
    """the permutation argument's share of compute_quotient_polys (plonk/prover.rs:609-815, plonk/vanishing_poly.rs:167-330):
    quotient values [nc][n << log2_ceil(qdf)] in natural order, from the three commitments' leaf matrices (committed order)"""
    ww, cc, zz, k = arr(wires_leaves), arr(cs_leaves), arr(zs_leaves), arr(k_is)
    b, g, a = arr(betas), arr(gammas), arr(alphas)
    nc = len(b)
    qbits = (int(qdf) - 1).bit_length()
    out = np.zeros((nc, (1 << log_n) << qbits), dtype=np.uint64)
    gs = arr(gate_sums) if gate_sums is not None else None
    rc = lib.ora_quotient_permutation(_p(ww), C.c_size_t(ww.shape[1]), _p(cc), C.c_size_t(cc.shape[1]), C.c_size_t(sigmas_first), _p(zz),
                                      C.c_size_t(zz.shape[1]), C.c_uint(log_n), C.c_uint(rate_bits), _p(k), C.c_size_t(len(k)), C.c_size_t(qdf),
                                      C.c_uint(nc), _p(b), _p(g), _p(a), _p(gs) if gs is not None else None, _p(out))
    if rc:
        raise ValueError("unsupported quotient shape")
    return out
