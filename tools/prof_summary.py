#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd .db or *_kernel_stats.csv) as CSV text.
usage: tools/prof_summary.py <dir-or-file> > profiles/<name>_kernel_stats.csv"""
import csv
import glob
import os
import sqlite3
import sys


def main(path):
    dbs = [path] if path.endswith(".db") else glob.glob(os.path.join(path, "**", "*.db"), recursive=True)
    csvs = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True)
    w = csv.writer(sys.stdout)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
    if csvs:
        for r in csv.DictReader(open(csvs[0])):
            w.writerow([r.get("Name"), r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage")])
        return
    db = sqlite3.connect(dbs[0])
    for name, calls, total, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        w.writerow([name, calls, "%.0f" % total, "%.0f" % avg, "%.3f" % pct])


if __name__ == "__main__":
    main(sys.argv[1])
