#!/usr/bin/env python3
"""Counter passes -> the small JSON records bench.py quotes.

  evidence_json.py traffic <fetch pmc dir> <write pmc dir> <kernel substring> <algorithmic bytes per launch> <out.json>
      HBM bytes per launch of one kernel from FETCH_SIZE / WRITE_SIZE (rocprofv3, one counter per pass).  FETCH_SIZE is
      reported in KB and, on gfx950, tallies the 128-B requests of 16-B-per-lane streaming reads at 64 B: doubled here.
      Launches whose fetch is below a quarter of the largest (graph-capture launches over empty rows) are left out.
  evidence_json.py mfma <pmc dir> <out.json>
      Per kernel of a prefill chunk: matrix-pipe busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs divided by the
      kernel's active cycles (GRBM_GUI_ACTIVE summed over the 8 XCDs / 8)."""
import collections
import csv
import glob
import json
import re
import sys


def per_dispatch(d, counter, pat=None):
    vals = collections.defaultdict(float)
    names = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and (pat is None or pat in r["Kernel_Name"]):
                vals[r["Dispatch_Id"]] += float(r["Counter_Value"])
                names[r["Dispatch_Id"]] = r["Kernel_Name"]
    return vals, names


def traffic(fetch_dir, write_dir, pat, algorithmic, out):
    fetch, _ = per_dispatch(fetch_dir, "FETCH_SIZE", pat)
    write, _ = per_dispatch(write_dir, "WRITE_SIZE", pat)
    big = max(fetch.values())
    kept = [v for v in fetch.values() if v >= big / 4]
    wkept = sorted(write.values())[-len(kept):] if write else []
    fetch_kb = sum(kept) / len(kept)
    rec = {"kernel": pat, "launches": len(kept), "fetch_size_kb_avg": round(fetch_kb, 1),
           "correction": "x2 (gfx950 counts 64 B per 128-B request for 16-B-per-lane streaming reads)",
           "traffic_bytes_per_launch": int(fetch_kb * 1024 * 2), "algorithmic_bytes_per_launch": int(algorithmic),
           "write_bytes_per_launch": int(sum(wkept) / len(wkept) * 1024) if wkept else None}
    rec["measured_over_algorithmic"] = round(rec["traffic_bytes_per_launch"] / rec["algorithmic_bytes_per_launch"], 4)
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec))


def mfma(d, out):
    busy, names = per_dispatch(d, "SQ_VALU_MFMA_BUSY_CYCLES")
    active, _ = per_dispatch(d, "GRBM_GUI_ACTIVE")
    acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for disp, b in busy.items():
        if b <= 0 or active.get(disp, 0) <= 0:
            continue
        n = re.sub(r"\(.*", "", names[disp])
        n = re.sub(r"^void ", "", n)
        a = acc[n]
        a[0] += b / 1024.0
        a[1] += active[disp] / 8.0
        a[2] += 1
    rec = {n: {"mfma_busy_frac": round(a[0] / a[1], 4), "launches": a[2], "active_cycles_avg": round(a[1] / a[2])} for n, a in sorted(acc.items())}
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    if sys.argv[1] == "traffic":
        traffic(sys.argv[2], sys.argv[3], sys.argv[4], float(sys.argv[5]), sys.argv[6])
    else:
        mfma(sys.argv[2], sys.argv[3])
