import sys, os, json, time
sys.path.insert(0, "/root/repo")
import torch
from bench import splitmix_columns_torch
from plonky2_amd import Engine
eng = Engine(0)
W, log_n = 135, 20
cols = splitmix_columns_torch(torch, eng.mem.device, 0, W, 1 << log_n)
for mode in (0, 3, 4):
    eng.check(eng.lib.p2hot_tune_ntt(eng.ctx, mode))
    for _ in range(2):
        eng.commit(cols, log_n, 3, 4, True)
    eng.profile(True); eng.profile_results(reset=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        eng.commit(cols, log_n, 3, 4, True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5 * 1e3
    pr = eng.profile_results(reset=True); eng.profile(False)
    print("tune_ntt", mode, "%.2f ms" % dt, {k: round(v["ms"] / v["launches"], 3) for k, v in pr.items() if "ntt" in k})
