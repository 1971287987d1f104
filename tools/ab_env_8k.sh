#!/usr/bin/env bash
export NVH_TUNING=1   # the library reads its NVH_* routing switches from the environment only under this
# 8192-token prompts in 1024-token chunks (bench.py's prefill_8k) under environment switches given as arguments
# ("VAR=value" each; "-" = none), same box.
for e in "$@" "$@"; do
  if [ "$e" = "-" ]; then line=$(timeout -k 10 300 python bench.py --steps 16 --warmup 4 2>/dev/null); else line=$(env $e timeout -k 10 300 python bench.py --steps 16 --warmup 4 2>/dev/null); fi
  echo "env=$e $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["prefill_8k"]["mfma_frac"], d["prefill_8k"]["ms_per_chunk"])')"
done
