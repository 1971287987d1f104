#!/usr/bin/env bash
export NVH_TUNING=1   # the library reads its NVH_* routing switches from the environment only under this
# Saturated serving run (2048 requests at 2000 req/s, in 128 / out 100, budget 1024) under environment switches given as
# arguments ("VAR=value" each; "-" = none), same box, two rounds: total tok/s, TTFT, TPOT.
for e in "$@" "$@"; do
  cmd="python3 -u serving_bench.py --num-requests 2048 --request-rate 2000 --random-input-len 128 --random-output-len 100 --max-num-batched-tokens 1024 --chunked-prefill --max-model-len 4096"
  if [ "$e" = "-" ]; then out=$(timeout -k 10 300 $cmd 2>/dev/null) || exit 1; else out=$(env $e timeout -k 10 300 $cmd 2>/dev/null) || exit 1; fi
  echo "env=$e $(echo "$out" | grep -i -E "throughput|ttft|tpot" | tr '\n' ' ' | cut -c1-300)"
done
