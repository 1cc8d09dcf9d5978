#!/usr/bin/env python3
"""Threshold sweep for the three Poseidon mappings (tooling): times a batch of m permutations, a small Merkle tree, the C3
FRI commit phase and the 2^12-gate per-proof path with the word-per-lane kernels enabled up to `row` permutations per launch.
usage: row_sweep.py [row thresholds ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from plonky2_amd import Engine  # noqa: E402
from plonky2_amd.fri.oracle import FriBatchInfo, PolynomialBatch, eval_openings, prove_openings  # noqa: E402
from plonky2_amd.fri.prover import fri_committed_trees_device  # noqa: E402
from plonky2_amd.iop.challenger import Challenger  # noqa: E402
from plonky2_amd.util.synthetic import splitmix_columns_torch  # noqa: E402

P = 2**64 - 2**32 + 1
eng = Engine(0)
dev = torch.device("cuda:0")
thrs = [int(a) for a in sys.argv[1:]] or [0, 64, 512, 2048, 8192, 32768]


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def k12_path():
    n, rb, cap = 1 << 12, 3, 4
    wires = splitmix_columns_torch(torch, dev, 0, 135, n)
    zs = splitmix_columns_torch(torch, dev, 1000, 20, n)
    quo = splitmix_columns_torch(torch, dev, 2000, 16, n)

    def path():
        oracles = [PolynomialBatch.from_values(wires, rb, False, cap, engine=eng),
                   PolynomialBatch.from_values(zs, rb, False, cap, engine=eng),
                   PolynomialBatch.from_coeffs(quo, rb, False, cap, engine=eng)]
        ch = Challenger(eng)
        ch.observe_elements(np.arange(8, dtype=np.uint64))
        zeta = ch.get_extension_challenge()
        gz = [(zeta[0] * 7) % P, zeta[1]]
        eval_openings(oracles, [zeta, gz], eng)
        allp = [(oi, pi) for oi, W in enumerate((135, 20, 16)) for pi in range(W)]
        nxt = [(oi, pi) for oi, W in enumerate((135, 20)) for pi in range(W)]
        prove_openings([FriBatchInfo(zeta, allp), FriBatchInfo(gz, nxt)], oracles, ch, rb, cap, [4, 4], 16, 28, engine=eng)
    return path


path = k12_path()
planes = splitmix_columns_torch(torch, dev, 500, 2, 1 << 20)
rng = np.random.default_rng(5)
states = {m: eng.dev(rng.integers(0, P, size=(m, 12), dtype=np.uint64)) for m in (1, 16, 256, 1024, 4096, 16384)}
leaves = {(n, w): eng.dev(rng.integers(0, P, size=(n, w), dtype=np.uint64)) for (n, w) in ((256, 135), (2048, 20), (1024, 4))}
for thr in thrs:
    eng.check(eng.lib.p2hot_tune_row(eng.ctx, thr))
    line = ["row<=%-6d" % thr]
    for m, st in states.items():
        line.append("perm[%d] %.1f us" % (m, timed(lambda: eng.poseidon_permute(st), 50) * 1e3))
    for (n, w), lv in leaves.items():
        line.append("tree[%dx%d] %.1f us" % (n, w, timed(lambda: eng.merkle(lv, 1, w, n.bit_length() - 1, 2)) * 1e3))
    ch = Challenger(eng)
    line.append("c3_fri_commit %.3f ms" % timed(lambda: fri_committed_trees_device(planes, 20, ch, 3, 4, [4, 4, 4, 4], eng), 5))
    line.append("k12_path %.3f ms" % timed(path, 10))
    print("  ".join(line), flush=True)
