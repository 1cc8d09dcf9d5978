#!/usr/bin/env python3
"""The reference's TimingTree output (`bench_recursion -vv`: lines "<| | ...><seconds>s to <scope>", plonky2/src/util/timing.rs:162-175)
of one or two runs, summed per scope name and tabulated side by side as a Markdown table -- the CPU-vs-GPU breakdown SURVEY section 5
asks for, in the reference's own scope names.

    python tools/timing_tree_table.py cpu.log [gpu.log] [--scopes "IFFT,FFT + blinding,..."]
"""
import argparse
import collections
import re

LINE = re.compile(r"((?:\| )*)(\d+\.\d+)s to (.+?)\s*$")
DEFAULT = ["prove", "IFFT", "FFT + blinding", "transpose LDEs", "build Merkle tree", "p2hot commit", "compute quotient polys", "compute partial products",
           "construct the opening set", "p2hot prove_openings", "reduce batch of", "perform final FFT", "fold codewords in the commitment phase",
           "find proof-of-work witness"]


def parse(path):
    """{scope name: (total seconds, occurrences)}; names with numbers in them ("reduce batch of 255 polynomials") are kept whole"""
    acc = collections.OrderedDict()
    for line in open(path, errors="replace"):
        m = LINE.search(line)
        if not m:
            continue
        name = m.group(3)
        t, k = acc.get(name, (0.0, 0))
        acc[name] = (t + float(m.group(2)), k + 1)
    return acc


def table(runs, labels, scopes):
    rows = []
    names = list(collections.OrderedDict((n, 1) for r in runs for n in r))
    for s in scopes:
        hits = [n for n in names if n == s or n.startswith(s + " ")]
        if not hits:
            continue
        cells = []
        for r in runs:
            t = sum(r[n][0] for n in hits if n in r)
            k = sum(r[n][1] for n in hits if n in r)
            cells.append("%.4f s (x%d)" % (t, k) if k else "--")
        rows.append("| `%s` | %s |" % (s, " | ".join(cells)))
    head = "| TimingTree scope | %s |\n|---|%s" % (" | ".join(labels), "---|" * len(labels))
    return head + "\n" + "\n".join(rows)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("logs", nargs="+")
    ap.add_argument("--labels", default=None)
    ap.add_argument("--scopes", default=None)
    a = ap.parse_args()
    runs = [parse(p) for p in a.logs]
    labels = a.labels.split(",") if a.labels else a.logs
    print(table(runs, labels, a.scopes.split(",") if a.scopes else DEFAULT))


if __name__ == "__main__":
    main()
