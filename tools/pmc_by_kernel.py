#!/usr/bin/env python3
"""Per-kernel averages of every counter in a rocprofv3 --pmc output directory (kernels told apart by name and grid).
usage: pmc_by_kernel.py <dir>"""
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    per, names = collections.defaultdict(float), {}
    for r in csv.DictReader(open(f)):
        per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
        names[r["Dispatch_Id"]] = r["Kernel_Name"][:120] + " |grid=" + r.get("Grid_Size", "?")
    for (d, c), v in per.items():
        acc[names[d]][c].append(v)
for k, cs in sorted(acc.items()):
    print(k, {c: (round(sum(v) / len(v)), len(v)) for c, v in cs.items()})
