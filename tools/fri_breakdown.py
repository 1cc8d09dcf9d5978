"""Times p2hot_fri_commit at the C3 shape with pre-touched host buffers: all outputs / without the leaf matrices /
without any output copy (tooling).  The first call also pays the page faults of the fresh host buffers."""
import sys, time, os, ctypes as C
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from plonky2_amd import Engine
from plonky2_amd.iop.challenger import Challenger
eng = Engine(0)
log_n, rb, cap = 20, 3, 4
n = 1 << log_n; N = n << rb
rng = np.random.default_rng(1)
co = rng.integers(0, 0xFFFFFFFF00000001, size=(n, 2), dtype=np.uint64)
arity = [4,4,4,4]
ab = (C.c_uint*4)(*arity)
leaves = np.zeros(2*(N + N//16 + N//256 + N//4096), dtype=np.uint64)
digests = np.zeros(4*2*(N//16 + N//256 + N//4096 + N//65536), dtype=np.uint64)
caps = np.zeros(4*16*4, dtype=np.uint64); betas=np.zeros(8,dtype=np.uint64); final=np.zeros((n>>16)*2+4,dtype=np.uint64)
import torch
for variant in ("all","no_leaves","no_outputs","all"):
    ch = Challenger(eng)
    torch.cuda.synchronize(); t0=time.perf_counter()
    eng.check(eng.lib.p2hot_fri_commit(eng.ctx, co.ctypes.data, log_n, rb, cap, ab, 4, 0, 0, ch._h,
        leaves.ctypes.data if variant=="all" else None,
        digests.ctypes.data if variant!="no_outputs" else None,
        caps.ctypes.data if variant!="no_outputs" else None, betas.ctypes.data, final.ctypes.data))
    t1=time.perf_counter()
    print(variant, "%.2f ms"%((t1-t0)*1e3))

# per-kernel HIP-event times of one device-only call
eng.profile(True)
eng.profile_results(reset=True)
ch = Challenger(eng)
eng.check(eng.lib.p2hot_fri_commit(eng.ctx, co.ctypes.data, log_n, rb, cap, ab, 4, 0, 0, ch._h, None, None, None, betas.ctypes.data,
                                   final.ctypes.data))
prof = eng.profile_results(reset=True)
eng.profile(False)
tot = 0.0
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
    print("  %-28s %8.3f ms  %4d launches" % (k, v["ms"], v["launches"]))
    tot += v["ms"]
print("  kernel time total %.3f ms" % tot)
