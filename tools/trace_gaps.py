#!/usr/bin/env python3
"""Idle time of the GPU in a rocprofv3 kernel trace, by the kernel that ran before each gap: gaps above `min_us` between
the end of the latest kernel so far and the start of the next one.  usage: trace_gaps.py <kernel_trace.csv> [min_us=20] [skip_over_ms=50]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
skip_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 50.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
end = int(rows[0]["End_Timestamp"])
last = rows[0]["Kernel_Name"]
by = collections.defaultdict(lambda: [0, 0.0])
hist = collections.Counter()
total = skipped = 0.0
for r in rows[1:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - end) / 1e3
    if gap > min_us:
        if gap > skip_ms * 1e3:
            skipped += gap
        else:
            total += gap
            key = re.sub(r"\(.*", "", last)[:60] + "  ->  " + re.sub(r"\(.*", "", r["Kernel_Name"])[:50]
            by[key][0] += 1
            by[key][1] += gap
            hist[min(int(gap // 250), 20)] += 1
    if e > end:
        end, last = e, r["Kernel_Name"]
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
print(f"span {span / 1e3:.1f} ms; gaps > {min_us:.0f} us: {total / 1e3:.1f} ms ({100 * total / span:.1f} %), {skipped / 1e3:.1f} ms more in gaps over {skip_ms:.0f} ms (start-up)")
print("gap length histogram (250-us bins):", dict(sorted(hist.items())))
for k, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"{t / 1e3:8.1f} ms in {c:5d} gaps, mean {t / c:7.0f} us: {k}")
# context of a few typical gaps (the kernels before and after, with their queue / stream ids)
shown = 0
end = int(rows[0]["End_Timestamp"])
for i, r in enumerate(rows[1:], 1):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - end) / 1e3
    if 300 < gap < 800 and "feed_token_ids" in r["Kernel_Name"] and shown < 3 and i > len(rows) // 2:
        shown += 1
        print(f"--- gap of {gap:.0f} us before row {i}")
        t0 = int(rows[i - 6]["Start_Timestamp"])
        for q in rows[i - 6:i + 5]:
            print(f"   +{(int(q['Start_Timestamp']) - t0) / 1e3:9.1f} us  dur {(int(q['End_Timestamp']) - int(q['Start_Timestamp'])) / 1e3:8.1f} us  queue {q.get('Queue_Id', '?'):>3} stream {q.get('Stream_Id', '?'):>3}  {re.sub(r'[(].*', '', q['Kernel_Name'])[:70]}")
    if e > end:
        end = e
