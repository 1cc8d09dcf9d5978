#!/usr/bin/env python3
"""Tooling: how much of the C3 FRI commit phase is kernel time and how much is gaps between kernels.
Run under `rocprofv3 --kernel-trace --output-format csv`; this script only issues the calls (3 warm-ups + 5 timed)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C

import numpy as np
import torch

from plonky2_amd import Engine
from plonky2_amd.iop.challenger import Challenger

eng = Engine(0)
log_n, rb, cap = 20, 3, 4
n = 1 << log_n
rng = np.random.default_rng(1)
co = rng.integers(0, 0xFFFFFFFF00000001, size=(n, 2), dtype=np.uint64)
ab = (C.c_uint * 4)(4, 4, 4, 4)
betas = np.zeros(8, dtype=np.uint64)
final = np.zeros(64, dtype=np.uint64)
for it in range(8):
    ch = Challenger(eng)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.check(eng.lib.p2hot_fri_commit(eng.ctx, co.ctypes.data, log_n, rb, cap, ab, 4, 0, 0, ch._h, None, None, None, betas.ctypes.data,
                                       final.ctypes.data))
    dt = (time.perf_counter() - t0) * 1e3
    if it >= 3:
        print("MARK call %d wall %.3f ms" % (it, dt))
