#!/usr/bin/env bash
# Profiles of the current build on the GPU box, written under gpurun_out/ev/: per-kernel statistics and one decode step
# of bench.py, one 1024-token prefill chunk, the MFMA counters of that chunk (own pass), GPU busy share of a saturated serving run.  Each GPU step runs under its
# own time limit; the script stops at the first failure.
set -euo pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
O="$R/gpurun_out/ev"
mkdir -p "$O"
cd /tmp
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -o b -- python3 "$R/bench.py" --steps 16 --warmup 4 --no-extras > "$O/bench_stats.log" 2>&1
python3 "$R/tools/step_breakdown.py" "$(find "$O/stats" -name '*kernel_trace.csv' | head -1)" feed_token_ids > "$O/decode_step_B256_ahead.csv"
python3 "$R/tools/step_breakdown.py" "$(find "$O/stats" -name '*kernel_trace.csv' | head -1)" > "$O/decode_step_B256_sync.csv"
cp "$(find "$O/stats" -name '*kernel_stats.csv' | head -1)" "$O/bench_kernel_stats.csv"
rm -rf "$O/stats"
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$O/pf" -o p -- python3 "$R/tools/prefill_probe.py" 1024 0 6 > "$O/prefill_probe.log" 2>&1
python3 "$R/tools/trace_breakdown.py" "$(find "$O/pf" -name '*kernel_trace.csv' | head -1)" > "$O/prefill_chunk_1024.csv"
rm -rf "$O/pf"
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$O/pmc" -o m -- python3 "$R/tools/prefill_probe.py" 1024 0 4 > "$O/pmc.log" 2>&1
python3 "$R/tools/pmc_by_kernel.py" "$O/pmc" > "$O/prefill_pmc_by_kernel.txt"
python3 "$R/tools/evidence_json.py" mfma "$O/pmc" "$O/prefill_mfma_busy.json" > /dev/null
rm -rf "$O/pmc"
# HBM bytes of the headline decode attention launch (256 sequences x 1040 keys, the fused form the graph replays): FETCH_SIZE
# and WRITE_SIZE, one pass each
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/fetch" -o f -- python3 "$R/tools/attn_decode_probe.py" 256 1040 > "$O/fetch.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/write" -o w -- python3 "$R/tools/attn_decode_probe.py" 256 1040 > "$O/write.log" 2>&1
python3 "$R/tools/evidence_json.py" traffic "$O/fetch" "$O/write" "attn_decode_kernel<2, true>" $((256 * 1040 * 8 * 128 * 4)) "$O/attn_decode_traffic.json" > /dev/null
rm -rf "$O/fetch" "$O/write"
A="--num-requests 2048 --request-rate 2000 --random-input-len 128 --random-output-len 100 --max-num-batched-tokens 1024 --chunked-prefill --max-model-len 4096"
for mode in sync ahead; do
    flag="--sync-scheduling"; [ "$mode" = ahead ] && flag="--async-scheduling"
    timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$O/sv_$mode" -o s -- python3 "$R/serving_bench.py" $A $flag > "$O/serving_trace_$mode.log" 2>&1
    python3 "$R/tools/trace_busy.py" "$(find "$O/sv_$mode" -name '*kernel_trace.csv' | head -1)" > "$O/serving_busy_$mode.txt"
    rm -rf "$O/sv_$mode"
done
ls "$O"
