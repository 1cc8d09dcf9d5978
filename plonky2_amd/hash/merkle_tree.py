"""MerkleTree -- mirror of plonky2/src/hash/merkle_tree.rs (struct :46-62, new :193-224, get :227,
prove :231-237 / merkle_tree_prove :151-190) with the tree built by libp2hot on the GPU.
"""
import numpy as np

from ..engine import default_engine


class MerkleCap:  # merkle_tree.rs:19
    def __init__(self, entries):
        self.entries = np.asarray(entries, dtype=np.uint64).reshape(-1, 4)

    def height(self):
        return int(self.entries.shape[0]).bit_length() - 1

    def flatten(self):
        return self.entries.reshape(-1)


class MerkleTree:
    """leaves [n][w]; digests [2*(n - 2^cap_height)][4] in the reference layout (:50-57); cap [2^cap_height][4].

    `leaves` may be given lazily (a callable rows(idx_array) -> [m][w]) when the leaf matrix stays on the GPU.
    """

    def __init__(self, leaves, digests, cap, cap_height, n_leaves=None, leaf_getter=None, engine=None, path_getter=None,
                 digests_getter=None):
        """digests: host array, or a DEVICE buffer (then `engine` must be given): it stays on the GPU, proofs are
        gathered there (p2hot_merkle_paths_dev) and `.digests` copies it to the host only when somebody asks; or None
        with `path_getter` / `digests_getter` callables when the tree lives inside a p2hot_batch handle."""
        self._leaves = leaves
        self._getter = leaf_getter
        self._path_getter = path_getter
        self._digests_getter = digests_getter
        self._engine = engine
        self._digests_dev = digests if engine is not None and engine.mem.is_buffer(digests) else None
        self._digests = None if self._digests_dev is not None else digests
        self.cap = MerkleCap(cap)
        self.cap_height = cap_height
        self.n_leaves = n_leaves if n_leaves is not None else len(leaves)

    @property
    def digests(self):
        if self._digests is None:
            self._digests = self._digests_getter() if self._digests_dev is None else self._engine.host(self._digests_dev)
        return self._digests

    @classmethod
    def new(cls, leaves, cap_height, engine=None):  # merkle_tree.rs:193-224
        eng = engine or default_engine()
        leaves_h = np.ascontiguousarray(np.asarray(leaves, dtype=np.uint64))
        if leaves_h.ndim != 2:
            raise ValueError("leaves must be [n][w]")
        n, w = leaves_h.shape
        log_n = n.bit_length() - 1
        if n != 1 << log_n:
            raise ValueError("number of leaves must be a power of two")  # log2_strict, :194
        d_leaves = eng.dev(leaves_h)
        digests, cap = eng.merkle(d_leaves, 1, w, log_n, cap_height)
        return cls(leaves_h, digests, eng.host(cap), cap_height, engine=eng)

    @property
    def leaves(self):
        if self._leaves is None:
            self._leaves = self._getter(np.arange(self.n_leaves, dtype=np.uint64))
        return self._leaves

    def get(self, i):  # :227
        if self._leaves is not None:
            return self._leaves[i]
        return self._getter(np.asarray([i], dtype=np.uint64))[0]

    def prove(self, leaf_index):
        """merkle_tree_prove (:151-190): siblings bottom-up, [log2(n) - cap_height][4]"""
        return self.prove_many([leaf_index])[0]

    def prove_many(self, leaf_indices):
        """[m][log2(n) - cap_height][4]; gathered on the GPU when the digests live there"""
        n = self.n_leaves
        log_n = n.bit_length() - 1
        num_layers = log_n - self.cap_height
        idx = np.asarray(leaf_indices, dtype=np.uint64).reshape(-1)
        if idx.size and int(idx.max()) >= n:
            raise IndexError("leaf index %d out of range (%d leaves)" % (int(idx.max()), n))  # merkle_tree.rs:231 panics
        if self._path_getter is not None and self._digests is None:
            return self._path_getter(idx)
        if self._digests_dev is not None and self._digests is None:
            eng = self._engine
            out = eng.mem.zeros(len(idx), max(num_layers, 1), 4)
            d_idx = eng.dev(idx)  # named: the buffer must outlive the (asynchronous) call that reads it
            eng.check(eng.lib.p2hot_merkle_paths_dev(eng.ctx, eng.ptr(self._digests_dev), log_n, self.cap_height,
                                                     eng.ptr(d_idx), len(idx), eng.ptr(out)))
            return eng.host(out)[:, :num_layers]
        digests = np.asarray(self.digests, dtype=np.uint64).reshape(-1, 4)
        tree_len = digests.shape[0] >> self.cap_height
        out = np.zeros((len(idx), num_layers, 4), dtype=np.uint64)
        for q, leaf_index in enumerate(int(x) for x in idx):
            tree = digests[tree_len * (leaf_index >> num_layers):]
            pair_index = leaf_index & ((1 << num_layers) - 1)
            for i in range(num_layers):
                parity = pair_index & 1
                pair_index >>= 1
                siblings_index = (pair_index << (i + 1)) + (1 << i) - 1
                out[q, i] = tree[2 * siblings_index + (1 - parity)]
        return out
