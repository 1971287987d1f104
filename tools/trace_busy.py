#!/usr/bin/env python3
"""GPU busy fraction of the measured phase of a run from a rocprofv3 kernel trace (CSV): the trace is cut at its
longest idle gap (the pause between a warm-up phase and the measured one) and the part after it is summed.
usage: trace_busy.py <kernel_trace.csv>"""
import csv
import sys

rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(sys.argv[1]))))
gaps = [(rows[i + 1][0] - max(e for _, e in rows[max(0, i - 4):i + 1]), i) for i in range(len(rows) - 1)]
cut = max(gaps)[1] + 1 if gaps else 0
part = rows[cut:]
busy, end = 0, part[0][0]
for s, e in part:                       # union of the intervals (kernels of one stream do not overlap; be safe anyway)
    if e > end:
        busy += e - max(s, end)
        end = e
span = part[-1][1] - part[0][0]
idle = sorted((part[i + 1][0] - part[i][1] for i in range(len(part) - 1)), reverse=True)
print(f"kernels {len(part)} (of {len(rows)}), span {span / 1e6:.1f} ms, busy {busy / 1e6:.1f} ms = {100 * busy / span:.1f} %, "
      f"idle {100 * (1 - busy / span):.1f} %; largest gaps (us): {[round(g / 1e3) for g in idle[:5]]}")
