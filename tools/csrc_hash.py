#!/usr/bin/env python3
"""sha256 over the kernel sources (plonky2_amd/csrc/*, sorted by name): stamps profiles/pmc_traffic.json so that bench.py can
tell whether the committed counters were collected on the kernels it is timing."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_hash():
    d = os.path.join(ROOT, "plonky2_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        h.update(f.encode() + b"\0")
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(csrc_hash())
