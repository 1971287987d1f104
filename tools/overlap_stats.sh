#!/usr/bin/env bash
# Per-kernel statistics of bench.py's prefill + decode with one and with two steps in flight (NANOVLLM_STEP_OVERLAP): how much
# longer each kernel of a chunk runs when it shares the chip with the other chunk's kernels.
set -euo pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
O="$R/gpurun_out/ov"
mkdir -p "$O"
cd /tmp
export TMPDIR=/tmp
for e in 0 1; do
  NANOVLLM_STEP_OVERLAP=$e timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/s$e" -o b -- python3 "$R/bench.py" --steps 8 --warmup 2 --no-extras > "$O/bench_$e.log" 2>&1
  cp "$(find "$O/s$e" -name '*kernel_stats.csv' | head -1)" "$O/kernel_stats_overlap$e.csv"
  rm -rf "$O/s$e"
done
ls "$O"
