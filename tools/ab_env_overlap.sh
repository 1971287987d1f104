#!/usr/bin/env bash
export NVH_TUNING=1   # the library reads its NVH_* routing switches from the environment only under this
# Prefill chunk (two steps in flight) and B=256 decode step under environment switches given as arguments ("VAR=value,VAR=value" each; "-" = none), same box.
for e in "$@" "$@"; do
  if [ "$e" = "-" ]; then line=$(timeout -k 10 200 python bench.py --no-extras --steps 16 --warmup 4 2>/dev/null | tail -1); else line=$(env ${e//,/ } timeout -k 10 200 python bench.py --no-extras --steps 16 --warmup 4 2>/dev/null | tail -1); fi
  echo "env=$e $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["prefill"]["ms_per_chunk"])')"
done
