#!/usr/bin/env bash
# One decode step of bench.py at batch $1 (context $2, default 1024) cut out of a rocprofv3 kernel trace -> gpurun_out/step_b$1.txt
set -euo pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"; B="$1"; C="${2:-1024}"
cd /tmp; export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/trace_b$B" -o t -- python3 "$R/bench.py" --batch "$B" --context "$C" --steps 8 --warmup 2 --no-extras > "$R/gpurun_out/trace_b$B.log" 2>&1
python3 "$R/tools/step_breakdown.py" "$(find "$R/gpurun_out/trace_b$B" -name '*kernel_trace.csv' | head -1)" feed_token_ids > "$R/gpurun_out/step_b$B.txt"
rm -rf "$R/gpurun_out/trace_b$B"
cat "$R/gpurun_out/step_b$B.txt"
