#!/usr/bin/env python
"""Shorten a rocprofv3 *_kernel_stats.csv: kernel name cut to 48 characters, calls, total ms, average us, max us, share.
usage: python profiles/kstats_short.py <kernel_stats.csv> [rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
print(f"{'kernel':48s} {'calls':>6s} {'total ms':>9s} {'avg us':>9s} {'max us':>9s} {'%':>6s}")
for r in rows[:n]:
    name = r["Name"].replace("void ", "").split("(")[0][:48]
    print(f"{name:48s} {int(r['Calls']):6d} {float(r['TotalDurationNs']) / 1e6:9.3f} {float(r['AverageNs']) / 1e3:9.1f} "
          f"{float(r['MaxNs']) / 1e3:9.1f} {float(r['Percentage']):6.2f}")
