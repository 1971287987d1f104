#!/usr/bin/env python3
"""Average per-dispatch counter values of one kernel from rocprofv3 --pmc CSV output.
usage: pmc_summary.py <dir> <kernel substring>"""
import collections, csv, glob, sys
d, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (disp, name), v in per.items():
        acc[name].append(v)
for name, vals in sorted(acc.items()):
    vals = vals[1:] if len(vals) > 1 else vals          # drop the first (cold) dispatch
    print(f"{name:36s} {sum(vals) / len(vals):16.0f}   (n={len(vals)})")
