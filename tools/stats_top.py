#!/usr/bin/env python3
"""The kernels of a rocprofv3 --stats run by total time.  usage: stats_top.py <kernel_stats.csv> [rows=16]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# {tot / 1e6:.1f} ms of kernels")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:n]:
    print(f"{float(r['TotalDurationNs']) / 1e6:9.1f} ms {100 * float(r['TotalDurationNs']) / tot:5.1f} %  calls {r['Calls']:>6}  avg {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:90]}")
