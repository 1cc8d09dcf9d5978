"""CPU-side probe (tooling): how many host cores the job may really use, and how the oracle's Merkle build scales."""
import os, sys, time, subprocess
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
print("nproc:", os.cpu_count(), "affinity:", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(f, open(f).read().strip())
    except Exception as e:
        pass
code = r'''
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from oracle import p2oracle as ora
rng = np.random.default_rng(0)
leaves = rng.integers(0, ora.P, size=(1 << 15, 135), dtype=np.uint64)
ora.merkle_tree(leaves, 4)
t0 = time.perf_counter(); ora.merkle_tree(leaves, 4); dt = time.perf_counter() - t0
print("threads %4d: merkle 2^15 x 135 %.3f s = %.2f Mperm/s" % (ora.num_threads(), dt, (1 << 15) * 18 / dt / 1e6))
'''
for th in (1, 8, 16, 32, 64, 128, 256):
    env = dict(os.environ, OMP_NUM_THREADS=str(th))
    print(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip())
