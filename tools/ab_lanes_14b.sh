#!/usr/bin/env bash
# Qwen3-14B shape on one GPU: 1024-token prefill chunks with 2, 3, 4 steps in flight (same box).
for e in 4 3 2 4 3 2; do
  echo "lanes=$e starting"
  line=$(NANOVLLM_STEP_LANES=$e timeout -k 10 400 python bench.py --model qwen3-14b --batch 8 --context 2048 --steps 8 --warmup 2 --no-extras 2>/dev/null | tail -1)
  echo "lanes=$e $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["prefill"]["ms_per_chunk"], d["prefill"]["roofline"]["frac"])')"
done
