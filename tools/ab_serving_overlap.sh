#!/usr/bin/env bash
# Saturated and arrival-bound serving runs (Qwen3-0.6B shape, in 128 / out 100) with one and two steps in flight.
A="--num-requests 2048 --random-input-len 128 --random-output-len 100 --max-num-batched-tokens 1024 --chunked-prefill --max-model-len 4096"
for rate in 2000 100; do
  for e in 0 1; do
    echo "rate=$rate overlap=$e $(NANOVLLM_STEP_OVERLAP=$e timeout -k 10 300 python serving_bench.py $A --request-rate $rate 2>/dev/null | tail -1 | cut -c1-400)"
  done
done
