#!/usr/bin/env python3
"""Per-kernel totals of one step in a rocprofv3 kernel trace; a step starts at each launch of the delimiter kernel.
By default the second to last step (the last may be cut short by the end of the trace); `which` picks another (-1 = last).
usage: trace_breakdown.py <kernel_trace.csv> [delimiter=embedding_kernel] [min_kernels=50] [which=-2]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
delim = sys.argv[2] if len(sys.argv) > 2 else "embedding_kernel"
min_k = int(sys.argv[3]) if len(sys.argv) > 3 else 50
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if delim in r["Kernel_Name"]] + [len(rows)]
spans = [(a, b) for a, b in zip(idx, idx[1:]) if b - a >= min_k]
which = int(sys.argv[4]) if len(sys.argv) > 4 else -2
a, b = spans[which] if len(spans) > 1 else spans[-1]
step = rows[a:b]
t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
agg = collections.OrderedDict()
busy = 0
for r in step:
    n = re.sub(r"\(.*", "", r["Kernel_Name"])
    n = re.sub(r"^void ", "", n)[:90]
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    busy += d
    g = agg.setdefault(n, [0, 0])
    g[0] += 1
    g[1] += d
print(f"# step span {(t1 - t0) / 1e3:.1f} us, {len(step)} kernels, busy {busy / 1e3:.1f} us, idle {(t1 - t0 - busy) / 1e3:.1f} us")
print("kernel,launches,total_us,avg_us")
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n},{c},{d / 1e3:.1f},{d / c / 1e3:.2f}")
