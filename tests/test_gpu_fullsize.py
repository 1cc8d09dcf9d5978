"""BASELINE.json's full-size configurations on the MI355X (-m gpu): bit-exact against the oracle where
the oracle finishes in seconds (C2, the FRI commit phases), and through size-independent properties
at C3 / C4 size (Merkle paths verify to the cap, LDE rows equal direct polynomial evaluation,
coefficients interpolate the inputs, coset-sharded assembly equals the monolithic tree)."""
import numpy as np
import pytest

from tests import pyref
from tests.conftest import P, rand_field

pytestmark = pytest.mark.gpu

G = pyref.G


def _check_properties(gpu, ora, cols_dev, r, W, log_n, rb, cap, rng, n_paths=24, n_evals=3):
    N, n = 1 << (log_n + rb), 1 << log_n
    digests, capv = gpu.host(r["digests"]), gpu.host(r["cap"])
    # (a) every sampled Merkle path verifies against the cap (merkle_tree.rs:253-267)
    idx = np.unique(np.concatenate([[0, N - 1], rng.integers(0, N, size=n_paths)])).astype(np.uint64)
    rows = gpu.host(gpu.gather_rows(r["lde"], idx))
    for i, row in zip(idx, rows):
        proof = ora.merkle_prove(int(i), N, cap, digests)
        assert ora.merkle_verify(row, int(i), capv, proof), int(i)
    # (b) leaf row L, column c is p_c(g * w_N^bitrev(L)) (oracle.rs:142-147; polynomial/mod.rs:486-491)
    wN = pyref.root_of_unity(log_n + rb)
    for _ in range(n_evals):
        q = int(rng.integers(0, len(idx)))
        c = int(rng.integers(0, W))
        L = int(idx[q])
        coeffs = [int(x) for x in gpu.host(r["coeffs"][c])]
        x = G * pow(wN, pyref.bitrev(L, log_n + rb), P) % P
        assert pyref.eval_poly(coeffs, x) == int(rows[q][c]), (L, c)
        # (c) the coefficients interpolate the input values on H_n (fft.rs:233-239)
        i = int(rng.integers(0, n))
        assert pyref.eval_poly(coeffs, pow(pyref.root_of_unity(log_n), i, P)) == int(gpu.host(cols_dev[c, i:i + 1])[0]) % P


def test_c2_commit_bit_exact_vs_oracle(gpu, ora):
    """C2: W = 135, 2^16 rows, rate 1/8, cap 4 -- everything compared with the oracle"""
    rng = np.random.default_rng(2)
    W, log_n, rb, cap = 135, 16, 3, 4
    cols = rand_field(rng, W, 1 << log_n, noncanonical=True)
    r = gpu.commit(gpu.dev(cols), log_n, rb, cap, True, want_leaves=True)
    o = ora.commit(cols, rb, cap, True)
    assert (gpu.host(r["coeffs"]) == o["coeffs"]).all()
    assert (gpu.host(r["cap"]) == o["cap"]).all()
    assert (gpu.host(r["digests"]) == o["digests"]).all()
    assert (gpu.host(r["leaves"]) == o["leaves"]).all()


@pytest.mark.parametrize("W,is_values", [(135, True), (20, True), (16, False)])
def test_c3_commit_properties(gpu, ora, W, is_values):
    """C3: the three commits of a 2^20-gate proof (wires, Zs+partial products, quotient chunks)"""
    rng = np.random.default_rng(3 + W)
    log_n, rb, cap = 20, 3, 4
    cols = gpu.dev(rand_field(rng, W, 1 << log_n))
    r = gpu.commit(cols, log_n, rb, cap, is_values)
    _check_properties(gpu, ora, cols if is_values else r["coeffs"], r, W, log_n, rb, cap, rng,
                      n_evals=3 if is_values else 0)
    if W == 20:
        # coset-sharded assembly (the multi-GPU unit) gives the same digests and cap
        n = 1 << log_n
        nd = gpu.num_digests(log_n + rb, cap)
        dig, capbuf = gpu.mem.zeros(nd, 4), gpu.mem.zeros(1 << cap, 4)
        for part in range(8):
            gpu.commit(cols, log_n, rb, cap, is_values, row_begin=part * n, row_count=n, digests=dig, cap=capbuf)
        assert (gpu.host(dig) == gpu.host(r["digests"])).all() and (gpu.host(capbuf) == gpu.host(r["cap"])).all()


def test_c4_starky_trace_commit_properties(gpu, ora):
    """C4: Fibonacci trace, 2 columns x 2^22 rows, rate 1/2, cap 4 (leaves are hash_or_noop copies)"""
    rng = np.random.default_rng(4)
    log_n, rb, cap = 22, 1, 4
    n = 1 << log_n
    # x0 = 0, x1 = 1; row i = (F_i, F_{i+1}) mod P (starky/src/fibonacci_stark.rs:47-57), generated on the host
    fib = np.zeros((2, n), dtype=np.uint64)
    a, b = 0, 1
    for i in range(n):
        fib[0, i], fib[1, i] = a, b
        a, b = b, (a + b) % P
    cols = gpu.dev(fib)
    r = gpu.commit(cols, log_n, rb, cap, True)
    _check_properties(gpu, ora, cols, r, 2, log_n, rb, cap, rng, n_paths=16, n_evals=2)


@pytest.mark.parametrize("log_n,rb,arity", [(20, 3, [4, 4, 4, 4]), (22, 1, [4, 4, 4, 4])])
def test_fri_commit_phase_full_size_vs_oracle(gpu, ora, log_n, rb, arity):
    """C3 / C4 FRI commit phases (N = 2^23, arity 16): caps, betas, final polynomial and trees vs the oracle"""
    from plonky2_amd.fri.prover import fri_committed_trees
    from plonky2_amd.iop.challenger import Challenger
    rng = np.random.default_rng(50 + log_n)
    n = 1 << log_n
    co = rand_field(rng, n, 2)
    pad = np.zeros((n << rb, 2), dtype=np.uint64)
    pad[:n] = co
    c, oc = Challenger(gpu), ora.Challenger()
    trees, final, betas = fri_committed_trees(co, c, rb, 4, arity, engine=gpu)
    o = ora.fri_commit(pad, rb, 4, arity, oc)
    assert (betas == o["betas"]).all() and (final == o["final"]).all()
    for i, t in enumerate(trees):
        assert (t.cap.entries == o["caps"][i]).all(), i
        assert (np.asarray(t.digests).reshape(-1, 4) == o["digests"][i]).all(), i
        assert (t.leaves == o["leaves"][i]).all(), i
    assert c.get_n_challenges(2) == oc.get_n_challenges(2)


class _LoopbackDist:
    """single-process stand-in for torch.distributed: rank r's all-gather input is written into slot r of the
    output and the other slots are filled from what the other simulated ranks deposited earlier"""

    def __init__(self, world):
        self.world, self.rank, self.store = world, 0, {}
        self.calls = 0

    def all_gather_into_tensor(self, out, inp):
        key = self.calls
        self.calls += 1
        self.store.setdefault(key, {})[self.rank] = inp.clone()
        chunk = inp.numel()
        for r, t in self.store[key].items():
            out[r * chunk:(r + 1) * chunk] = t

    def all_gather(self, outs, inp, async_op=False):
        key = self.calls
        self.calls += 1
        self.store.setdefault(key, {})[self.rank] = inp.clone()
        for r, t in self.store[key].items():
            outs[r].copy_(t)

        class _Done:
            def wait(self):
                return True
        return _Done()


@pytest.mark.parametrize("chunks,gather", [(1, True), (3, True), (3, False)])
def test_sharded_commit_ranks_on_one_gpu(gpu, ora, chunks, gather):
    """the world > 1 branch of plonky2_amd.distributed on the real GPU: both ranks of a 2-rank job are
    run one after the other with a loopback all-gather; rank 1 then holds the full tree"""
    from plonky2_amd.distributed import ShardedCommit
    rng = np.random.default_rng(9)
    W, log_n, rb, cap = 9, 12, 3, 4
    cols = rand_field(rng, W, 1 << log_n)
    o = ora.commit(cols, rb, cap, True)
    dist = _LoopbackDist(2)
    jobs = [ShardedCommit(gpu, W, log_n, rb, cap, is_values=True, rank=r, world=2, dist=dist, want_leaves=True,
                          pipeline_chunks=chunks, gather_digests=gather) for r in range(2)]
    out = None
    for _pass in range(2):  # second sweep: every rank's deposits are present, like a real collective
        dist.calls = 0
        for r, job in enumerate(jobs):
            dist.rank = r
            base = dist.calls
            c0, c1 = job.column_range
            out = job.run(gpu.dev(cols[c0:c1]))
            if r == 0:
                dist.calls = base  # rank 1 replays the same sequence of collectives
    r0, rc = jobs[1].plan.rows(1)
    assert (gpu.host(out["coeffs"]) == o["coeffs"]).all()
    assert (gpu.host(out["leaves"]) == o["leaves"][r0:r0 + rc]).all()
    assert (gpu.host(out["cap"]) == o["cap"]).all()
    if gather:
        assert (gpu.host(out["digests"]) == o["digests"]).all()
    else:  # digests stay with the row owner: rank 1's slice is filled, and it answers the queries that fall into its rows
        p = jobs[1].plan
        d0, d1 = p.digests_per_rank, 2 * p.digests_per_rank
        assert (gpu.host(out["digests"])[d0:d1] == o["digests"][d0:d1]).all()
        mine = [r0, r0 + 5, r0 + rc - 1]
        rows, paths = jobs[1].prove_local(mine)
        for x, row, path in zip(mine, rows, paths):
            assert (row == o["leaves"][x]).all()
            assert (path == ora.merkle_prove(x, p.N, cap, o["digests"])).all()
            assert ora.merkle_verify(row, x, o["cap"], path)


@pytest.mark.gpu
@pytest.mark.parametrize("is_values,want_leaves", [(True, False), (False, True)])
def test_host_pointer_commit_pipelined_path(gpu, ora, is_values, want_leaves):
    """p2hot_commit above its pipelining threshold (column blocks uploaded on the side stream beside the transforms,
    coefficient blocks downloaded beside the leaf sponge): same bytes as the oracle, and as the small-batch path"""
    import ctypes as C
    rng = np.random.default_rng(4242)
    W, log_n, rb, cap = 37, 17, 3, 4          # W * n = 2^22.2 elements: 3 column blocks, the last one partial
    n, N = 1 << log_n, 1 << (log_n + rb)
    cols = [rand_field(rng, n, noncanonical=True) for _ in range(W)]
    ptrs = (C.c_void_p * W)(*[c.ctypes.data for c in cols])
    coeffs = np.zeros((W, n), dtype=np.uint64)
    leaves = np.zeros((N, W), dtype=np.uint64) if want_leaves else None
    nd = gpu.num_digests(log_n + rb, cap)
    digests = np.zeros((nd, 4), dtype=np.uint64)
    capv = np.zeros((1 << cap, 4), dtype=np.uint64)
    handle = C.c_void_p()
    gpu.check(gpu.lib.p2hot_commit(gpu.ctx, ptrs, W, log_n, rb, cap, 1 if is_values else 0, 0, coeffs.ctypes.data,
                                   leaves.ctypes.data if want_leaves else None, digests.ctypes.data, capv.ctypes.data,
                                   C.byref(handle)))
    o = ora.commit(np.stack(cols), rb, cap, is_values)
    assert (coeffs == o["coeffs"] % np.uint64(P)).all()
    assert (digests == o["digests"]).all() and (capv == o["cap"]).all()
    if want_leaves:
        assert (leaves == o["leaves"]).all()
    idx = np.array([0, 77, N - 1], dtype=np.uint64)
    rows = np.zeros((3, W), dtype=np.uint64)
    gpu.check(gpu.lib.p2hot_batch_rows(handle, idx.ctypes.data, 3, rows.ctypes.data))
    assert (rows == o["leaves"][idx.astype(np.int64)]).all()
    gpu.lib.p2hot_batch_free(handle)
