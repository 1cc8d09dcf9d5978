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

    def __init__(self, leaves, digests, cap, cap_height, n_leaves=None, leaf_getter=None):
        self._leaves = leaves
        self._getter = leaf_getter
        self.digests = digests
        self.cap = MerkleCap(cap)
        self.cap_height = cap_height
        self.n_leaves = n_leaves if n_leaves is not None else len(leaves)

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
        return cls(leaves_h, eng.host(digests), eng.host(cap), cap_height)

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
        n = self.n_leaves
        num_layers = (n.bit_length() - 1) - self.cap_height
        digests = np.asarray(self.digests, dtype=np.uint64).reshape(-1, 4)
        tree_len = digests.shape[0] >> self.cap_height
        tree = digests[tree_len * (leaf_index >> num_layers):]
        pair_index = leaf_index & ((1 << num_layers) - 1)
        out = np.zeros((num_layers, 4), dtype=np.uint64)
        for i in range(num_layers):
            parity = pair_index & 1
            pair_index >>= 1
            siblings_index = (pair_index << (i + 1)) + (1 << i) - 1
            out[i] = tree[2 * siblings_index + (1 - parity)]
        return out
