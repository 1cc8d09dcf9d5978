#!/usr/bin/env python3
"""Per-rank COMPUTE of a sharded commit on one GPU (tooling): rank 0 of a `world`-rank job with a transport hook that
returns at once (nothing is exchanged, so the results are not a commitment), timing exactly the kernels one rank of an
N-GPU run executes -- its share of the iNTT, the LDE of its cosets over all columns, its leaves and cap subtrees.  The
exchange itself is not in this number.   usage: rank_work.py [world ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from plonky2_amd import Engine, _lib  # noqa: E402
from plonky2_amd.distributed import ShardedCommit  # noqa: E402
from plonky2_amd.util.synthetic import splitmix_columns_torch  # noqa: E402

eng = Engine(0)
W, rb, cap = 135, 3, 4
for world in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    log_n = 20 + (world - 1).bit_length()
    job = ShardedCommit(eng, W, log_n, rb, cap, True, rank=0, world=world, dist=None, gather_digests=False,
                        transport="none" if world == 1 else "null")
    c0, c1 = job.column_range
    cols = splitmix_columns_torch(torch, eng.mem.device, c0, c1 - c0, 1 << log_n)
    job.run(cols)
    eng.profile(True)
    eng.profile_results(reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        job.run(cols)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    prof = eng.profile_results(reset=True)
    eng.profile(False)
    print("world %d (2^%d rows): rank 0 computes for %.1f ms/step  %s" % (
        world, log_n, ms, {k: round(v["ms"] / 3, 2) for k, v in prof.items()}), flush=True)
    job.comm.close()
    del job, cols
    torch.cuda.empty_cache()
    eng.check(eng.lib.p2hot_ctx_trim(eng.ctx))
