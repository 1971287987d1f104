#!/usr/bin/env bash
# B=256 decode step, 1024-token prefill chunk and the chunks of an 8192-token prompt with and without two steps in flight
# (NANOVLLM_STEP_OVERLAP), same box.
for e in 0 1 0 1; do
  line=$(NANOVLLM_STEP_OVERLAP=$e timeout -k 10 300 python bench.py --steps 32 --warmup 4 2>/dev/null | tail -1)
  echo "overlap=$e $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["prefill"]["ms_per_chunk"], d["prefill"]["roofline"]["frac"], d.get("prefill_8k"), d.get("prefill_8x128", {}).get("ms_per_step"), d.get("extras_error"))')"
done
