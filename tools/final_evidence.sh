#!/usr/bin/env bash
export NVH_TUNING=1   # the library reads its NVH_* routing switches from the environment only under this
# Bench line, profiles (tools/collect_evidence.sh), the Qwen3-14B prefill chunk by kernel and the var-len attention
# sweep in both grid orders, for the build in the tree; everything under gpurun_out/ev/.  Each GPU step runs under its own
# time limit; the script stops at the first failure.  Two parts, one GPU call each (a call is limited to 20 minutes):
#   final_evidence.sh 1   bench line + tools/collect_evidence.sh
#   final_evidence.sh 2   14B chunk, attention sweeps, serving, the reference's offline workload, steps-in-flight A/B
set -euo pipefail
part="${1:-1}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
O="$R/gpurun_out/ev"
mkdir -p "$O"
cd "$R"
if [ "$part" = 1 ]; then
    timeout -k 10 500 python3 -u bench.py > "$O/bench.json" 2> "$O/bench.err"
    bash tools/collect_evidence.sh > "$O/collect.log" 2>&1
    ls "$O"
    exit 0
fi
cd /tmp
export TMPDIR=/tmp
NANOVLLM_PROBE_MODEL=qwen3-14b timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$O/pf14" -o p -- python3 "$R/tools/prefill_probe.py" 1024 0 4 > "$O/prefill_probe_14b.log" 2>&1
python3 "$R/tools/trace_breakdown.py" "$(find "$O/pf14" -name '*kernel_trace.csv' | head -1)" > "$O/prefill_chunk_1024_14b.csv"
rm -rf "$O/pf14"
# the last chunk of an 8192-token prompt (1024 new tokens over 7168 cached) by kernel
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$O/pf8k" -o p -- python3 "$R/tools/prefill_probe.py" 1024 7168 4 > "$O/prefill_probe_at_7168.log" 2>&1
python3 "$R/tools/trace_breakdown.py" "$(find "$O/pf8k" -name '*kernel_trace.csv' | head -1)" embedding_kernel 50 -1 > "$O/prefill_chunk_1024_at_7168.csv"
rm -rf "$O/pf8k"
cd "$R"
{
    echo "# NVH_ATTN_HEAD_MAJOR=0 (query-block-major)"
    NVH_ATTN_HEAD_MAJOR=0 timeout -k 10 200 python3 -u tools/attn_varlen_sweep.py 2> /dev/null
    echo "# NVH_ATTN_HEAD_MAJOR=1 (kv-head-major)"
    NVH_ATTN_HEAD_MAJOR=1 timeout -k 10 200 python3 -u tools/attn_varlen_sweep.py 2> /dev/null
    echo "# default rule"
    timeout -k 10 200 python3 -u tools/attn_varlen_sweep.py 2> /dev/null
} > "$O/attn_varlen_grid_order.txt"
{
    NANOVLLM_PROBE_MODEL=qwen3-14b timeout -k 10 200 python3 -u tools/prefill_probe.py 1024 0 5 2> /dev/null | tail -1
    echo "# NVH_GEMM_NO_XCD=1"
    NVH_GEMM_NO_XCD=1 NANOVLLM_PROBE_MODEL=qwen3-14b timeout -k 10 200 python3 -u tools/prefill_probe.py 1024 0 5 2> /dev/null | tail -1
} > "$O/prefill_14b_xcd_order.txt"
timeout -k 10 200 python3 -u tools/prefill_two_streams_probe.py 1024 8 2 2> /dev/null | tail -1 > "$O/prefill_two_streams.txt"
timeout -k 10 200 python3 -u tools/prefill_two_streams_probe.py 1024 8 4 2> /dev/null | tail -1 >> "$O/prefill_two_streams.txt"
timeout -k 10 300 python3 -u serving_bench.py --num-requests 2048 --request-rate 2000 --random-input-len 128 --random-output-len 100 --max-num-batched-tokens 1024 --chunked-prefill --max-model-len 4096 > "$O/serving_rate2000_async.txt" 2> /dev/null
timeout -k 10 300 python3 -u serving_bench.py --num-requests 512 --request-rate 100 --random-input-len 128 --random-output-len 100 --max-num-batched-tokens 1024 --chunked-prefill --max-model-len 4096 > "$O/serving_rate100_async.txt" 2> /dev/null
# the reference's offline workload (its bench.py) through LLM.generate, both model shapes, and the steps-in-flight A/B
timeout -k 10 400 python3 -u tools/offline_bench.py --model qwen3-0.6b 2> /dev/null | tail -2 > "$O/offline_0.6b.txt"
timeout -k 10 400 python3 -u tools/offline_bench.py --model qwen3-0.6b --sync-scheduling 2> /dev/null | tail -2 > "$O/offline_0.6b_sync.txt"
timeout -k 10 900 python3 -u tools/offline_bench.py --model qwen3-14b 2> /dev/null | tail -2 > "$O/offline_14b.txt"
bash tools/ab_overlap.sh > "$O/steps_in_flight_ab.txt" 2>&1
ls "$O"
