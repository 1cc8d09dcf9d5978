"""Coset-sharded PolynomialBatch commit across the GPUs of one node (one process per GPU).

The reference has no multi-device mode; this is the MI355X design of SURVEY.md 8(e):

  * the rate-1/B LDE is B independent coset transforms, and in the committed (bit-reversed) leaf
    order coset j is the contiguous row block bitrev(j); with B = 8 and cap_height = 4 a block is
    two whole cap subtrees.  Rank r therefore owns rows [r*N/G, (r+1)*N/G): it runs the LDE, the
    Poseidon leaf sponge and the Merkle levels of its rows with NO data-path exchange;
  * every rank needs all W*n coefficients: the iNTT is column-sharded (ceil(W/G) columns per
    rank) and followed by ONE all-gather of coefficients (W*n*8 bytes in total);
  * the digests of a rank's subtrees are a contiguous slice of the reference digest array, so ONE
    all-gather of digests (+ cap entries) reassembles MerkleTree::digests / ::cap on every rank.  That digest
    exchange is optional (`gather_digests=False`): a Merkle path below the cap never leaves the cap subtree of its
    leaf, i.e. the slice of the rank that owns the row, so with only the 2^cap_height cap entries all-gathered every
    query can still be answered by the owner (`ShardedCommit.owner` / `.prove_local`), SURVEY 8(e) collective (2).

  * the coefficient all-gather is pipelined in column chunks: chunk k is gathered asynchronously (RCCL runs on its
    own stream) while chunk k+1 goes through the iNTT and chunk k-1 through the LDE, so most of the W*n*8-byte
    exchange hides behind the NTT work; the leaf sponge starts when the last chunk's LDE is done.

Collectives are torch.distributed (backend "nccl" = RCCL over xGMI on the GPUs; "gloo" in the
CPU tests).  There is no reduction anywhere, only all-gathers.  P2HOT_SYNC_COLLECTIVES=1 selects the
unpipelined variant (one blocking all-gather of all coefficients).
"""
import os

import numpy as np

COSET_SHIFT = 14293326489335486720


class ShardPlan:
    """Pure arithmetic of the sharding (testable without devices)."""

    def __init__(self, W, log_n, rate_bits, cap_height, world):
        if world < 1 or world & (world - 1):
            raise ValueError("world size must be a power of two")
        self.W, self.log_n, self.rate_bits, self.cap_height, self.world = W, log_n, rate_bits, cap_height, world
        self.n = 1 << log_n
        self.N = self.n << rate_bits
        self.log_N = log_n + rate_bits
        if cap_height > self.log_N:
            raise ValueError("cap_height > log2(N) (merkle_tree.rs:195-200)")
        self.rows_per_rank = self.N // world
        sub_leaves = self.N >> cap_height
        if world > (1 << rate_bits) or self.rows_per_rank % self.n:
            raise ValueError("world size %d exceeds the %d LDE cosets" % (world, 1 << rate_bits))
        if world > (1 << cap_height) or self.rows_per_rank % sub_leaves:
            raise ValueError("world size %d exceeds the %d cap subtrees" % (world, 1 << cap_height))
        self.cols_per_rank = -(-W // world) if W else 0
        self.num_digests = 2 * (self.N - (1 << cap_height))
        self.digests_per_rank = self.num_digests // world
        self.cap_per_rank = (1 << cap_height) // world

    def rows(self, rank):
        return rank * self.rows_per_rank, self.rows_per_rank

    def columns(self, rank):
        c0 = min(self.W, rank * self.cols_per_rank)
        return c0, min(self.W, c0 + self.cols_per_rank)

    def cosets(self, rank):
        """natural coset indices j (points g * w_N^(B*q + j)) whose rows this rank owns"""
        b0 = rank * self.rows_per_rank // self.n
        nb = self.rows_per_rank // self.n
        rb = self.rate_bits
        return [int(format(b, "0%db" % rb)[::-1], 2) if rb else 0 for b in range(b0, b0 + nb)]


class ShardedCommit:
    """from_values / from_coeffs over `world` ranks.  Buffers are allocated once and reused."""

    def __init__(self, engine, W, log_n, rate_bits, cap_height, is_values=True, rank=0, world=1, dist=None,
                 want_leaves=False, pipeline_chunks=None, gather_digests=True):
        self.eng, self.dist, self.rank, self.world = engine, dist, rank, world
        if pipeline_chunks is None:
            pipeline_chunks = 1 if os.environ.get("P2HOT_SYNC_COLLECTIVES") == "1" else 8
        self.pipeline_chunks = pipeline_chunks
        self.gather_digests = gather_digests
        self.plan = p = ShardPlan(W, log_n, rate_bits, cap_height, world)
        self.is_values = is_values
        mem = engine.mem
        self.column_range = p.columns(rank)
        self.row_begin, self.row_count = p.rows(rank)
        # coefficient buffer, padded to world * cols_per_rank columns so the all-gather is uniform
        self.coeffs_all = mem.empty(max(1, world * p.cols_per_rank), p.n)
        self.lde = mem.empty(W, self.row_count)
        self.leaves = mem.empty(self.row_count, W) if want_leaves else None
        self.digests = mem.zeros(max(1, p.num_digests), 4)
        self.cap = mem.zeros(1 << cap_height, 4)

    def run(self, cols_local):
        """cols_local: device [c1 - c0][n] -- this rank's columns (values on H_n, or coefficients).
        Returns dict(coeffs [W][n], lde [W][rows of this rank], digests, cap), all device buffers;
        digests / cap are the FULL tree's arrays on every rank (with gather_digests=False only this rank's slice
        [rank * digests_per_rank, (rank + 1) * digests_per_rank) of `digests` is filled; cap is always complete)."""
        eng, p, lib = self.eng, self.plan, self.eng.lib
        c0, c1 = self.column_range
        W = p.W
        if self.world == 1:
            eng.check(lib.p2hot_commit_dev(
                eng.ctx, eng.ptr(cols_local), cols_local.shape[1] if W else p.n, W, p.log_n, p.rate_bits, p.cap_height,
                1 if self.is_values else 0, 0, p.N, eng.ptr(self.coeffs_all), p.n, eng.ptr(self.lde), self.row_count,
                eng.ptr(self.leaves), eng.ptr(self.digests), eng.ptr(self.cap)))
        elif self.pipeline_chunks > 1 and p.cols_per_rank > 1:
            self._run_pipelined(cols_local)
        else:
            # 1. this rank's columns -> coefficient form, in its slot of the padded buffer
            slot = self.coeffs_all[self.rank * p.cols_per_rank:(self.rank + 1) * p.cols_per_rank]
            if c1 > c0:
                slot[:c1 - c0] = cols_local
                if self.is_values:
                    eng.ifft(slot[:c1 - c0], p.log_n)
            # 2. all-gather of coefficients (W*n*8 bytes over xGMI)
            self._all_gather(self.coeffs_all, slot)
            # 3. LDE + leaf sponge + Merkle levels of this rank's rows (whole cosets, whole cap subtrees)
            eng.check(lib.p2hot_commit_dev(
                eng.ctx, eng.ptr(self.coeffs_all), p.n, W, p.log_n, p.rate_bits, p.cap_height, 0, self.row_begin,
                self.row_count, None, 0, eng.ptr(self.lde), self.row_count, eng.ptr(self.leaves), eng.ptr(self.digests),
                eng.ptr(self.cap)))
        if self.world > 1:
            # 4. all-gather of this rank's contiguous digest slice and cap entries
            if p.digests_per_rank and self.gather_digests:
                d = self.digests[self.rank * p.digests_per_rank:(self.rank + 1) * p.digests_per_rank]
                self._all_gather(self.digests[:p.num_digests], d)
            k = self.cap[self.rank * p.cap_per_rank:(self.rank + 1) * p.cap_per_rank]
            self._all_gather(self.cap, k)
        return {"coeffs": self.coeffs_all[:W], "lde": self.lde, "leaves": self.leaves,
                "digests": self.digests[:p.num_digests], "cap": self.cap}

    def owner(self, leaf_index):
        """rank whose rows (and therefore whose digest slice) contain leaf `leaf_index` of the committed order"""
        return int(leaf_index) // self.plan.rows_per_rank

    def prove_local(self, leaf_indices):
        """(rows [m][W], Merkle paths [m][log2(N) - cap_height][4]) for leaves THIS rank owns, from its LDE block
        and its digest slice (MerkleTree::get + merkle_tree_prove, merkle_tree.rs:227, :151-190)."""
        eng, p = self.eng, self.plan
        idx = np.asarray(leaf_indices, dtype=np.uint64).reshape(-1)
        if any(self.owner(i) != self.rank for i in idx):
            raise ValueError("prove_local: a requested leaf belongs to another rank")
        layers = p.log_N - p.cap_height
        paths = eng.mem.zeros(max(1, len(idx)), max(layers, 1), 4)
        d_idx = eng.dev(idx if len(idx) else np.zeros(1, dtype=np.uint64))
        eng.check(eng.lib.p2hot_merkle_paths_dev(eng.ctx, eng.ptr(self.digests), p.log_N, p.cap_height, eng.ptr(d_idx),
                                                 len(idx), eng.ptr(paths)))
        rows = eng.gather_rows(self.lde, idx - np.uint64(self.row_begin)) if len(idx) and p.W else \
            np.zeros((len(idx), p.W), dtype=np.uint64)
        return eng.host(rows), eng.host(paths)[:len(idx), :layers]

    def _run_pipelined(self, cols_local):
        """steps 1-3 with the coefficient exchange hidden behind the NTTs (see the module docstring)"""
        eng, p, lib, mem = self.eng, self.plan, self.eng.lib, self.eng.mem
        c0, c1 = self.column_range
        cpr, W = p.cols_per_rank, p.W
        K = min(self.pipeline_chunks, cpr)
        cpk = -(-cpr // K)
        mine = c1 - c0
        mem.collective_fence()
        works, spans = [], []
        for k in range(K):
            lo, hi = k * cpk, min((k + 1) * cpk, cpr)            # rows of every rank's slot in this chunk
            if hi <= lo:
                continue
            my = self.coeffs_all[self.rank * cpr + lo:self.rank * cpr + hi]
            valid = max(0, min(hi, mine) - lo)                   # my real (non-padding) columns in the chunk
            if valid:
                my[:valid] = cols_local[lo:lo + valid]
                if self.is_values:
                    eng.ifft(my[:valid], p.log_n)
            outs = [mem.as_torch(self.coeffs_all[r * cpr + lo:r * cpr + hi]) for r in range(self.world)]
            works.append(self.dist.all_gather(outs, mem.as_torch(my).clone(), async_op=True))
            spans.append((lo, hi))
        for w, (lo, hi) in zip(works, spans):
            w.wait()                                             # NCCL: the compute stream waits, the host does not
            for r in range(self.world):
                cb = r * cpr + lo
                cnt = max(0, min(r * cpr + hi, W, (r + 1) * cpr) - cb)
                if cnt:                                          # LDE of these columns for this rank's coset rows
                    eng.check(lib.p2hot_coset_lde_dev(
                        eng.ctx, eng.ptr(self.coeffs_all[cb:cb + cnt]), cnt, p.n, p.log_n, p.rate_bits, COSET_SHIFT,
                        self.row_begin, self.row_count, eng.ptr(self.lde[cb:cb + cnt]), self.row_count))
        # leaf sponge + Merkle levels of this rank's rows, straight from the column-major LDE
        eng.check(lib.p2hot_merkle_dev(eng.ctx, eng.ptr(self.lde), 0, self.row_count, W, p.log_N, p.cap_height,
                                       self.row_begin, self.row_count, eng.ptr(self.digests), eng.ptr(self.cap)))
        if self.leaves is not None:
            eng.check(lib.p2hot_transpose_dev(eng.ctx, eng.ptr(self.lde), self.row_count, W, self.row_count,
                                              eng.ptr(self.leaves)))

    def _all_gather(self, full, mine):
        mem = self.eng.mem
        out = mem.as_torch(full).reshape(-1)
        inp = mem.as_torch(mine).reshape(-1).clone()  # not in place: the slice aliases `full`
        mem.collective_fence()
        self.dist.all_gather_into_tensor(out, inp)
