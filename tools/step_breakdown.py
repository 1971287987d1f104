#!/usr/bin/env python3
"""Cut the last decode step out of a rocprofv3 kernel trace (CSV) and print per-kernel totals.
usage: step_breakdown.py <kernel_trace.csv> [kernel name part the step must contain]"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if ('sample_final_kernel' in r['Kernel_Name'] or 'sample_kernel' in r['Kernel_Name'])]
want = sys.argv[2] if len(sys.argv) > 2 else None
pairs = [(a, b) for a, b in zip(idx, idx[1:]) if b - a > 50 and (want is None or any(want in r['Kernel_Name'] for r in rows[a + 1:b + 1]))]
a, b = pairs[-1]
step = rows[a + 1:b + 1]
t0, t1 = int(step[0]['Start_Timestamp']), int(step[-1]['End_Timestamp'])
prev_end = int(rows[a]['End_Timestamp'])
print(f"# step span {(t1 - t0) / 1e3:.1f} us, {len(step)} kernels; gap after previous sample kernel {(t0 - prev_end) / 1e3:.1f} us")
agg = collections.OrderedDict()
busy = 0
for r in step:
    n = re.sub(r"\(.*", "", r['Kernel_Name'])[:64]
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    busy += d
    g = agg.setdefault(n, [0, 0])
    g[0] += 1
    g[1] += d
print("kernel,launches,total_us,avg_us")
for n, (c, d) in agg.items():
    print(f"{n},{c},{d / 1e3:.1f},{d / c / 1e3:.2f}")
print(f"# busy {busy / 1e3:.1f} us, idle inside step {(t1 - t0 - busy) / 1e3:.1f} us")
