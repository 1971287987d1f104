#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy summary from hipcc's resource-usage remarks.
usage: tools/kres.py file.hip [...]"""
import re, subprocess, sys
for f in sys.argv[1:]:
    p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                        "-c", f, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    cur = {}
    for line in p.stderr.splitlines():
        m = re.search(r"remark: [^:]+:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r":\d+:\d+: remark:\s+(.*?) \[-Rpass", line)
        if not m:
            m = re.search(r"\d+:\d+:\s+(.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
        elif ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
            if k.strip().startswith("LDS Size"):
                name = subprocess.run(["/usr/bin/c++filt", cur.get("name", "?")], capture_output=True, text=True).stdout.strip()
                name = re.sub(r"\(.*", "", name)[:70]
                print(f"{name:70s} vgpr {cur.get('VGPRs','?'):>4} agpr {cur.get('AGPRs','?'):>4} spill {cur.get('VGPRs Spill','?'):>3} "
                      f"scratch {cur.get('ScratchSize [bytes/lane]','?'):>4} occ {cur.get('Occupancy [waves/SIMD]','?'):>2} lds {v.strip():>6}")
