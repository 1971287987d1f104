#!/usr/bin/env bash
export NVH_TUNING=1   # the library reads its NVH_* routing switches from the environment only under this
# Decode step time at the batches given in BATCHES under key-range settings of the decode attention: each argument is
# "FULL:TARGET" (NVH_DECODE_FULL_WGS, NVH_DECODE_TARGET_WGS) or "-" for the defaults.  Same box for all of them.
set -e
for b in ${BATCHES:-64 128}; do
  for e in "$@"; do
    if [ "$e" = "-" ]; then unset NVH_DECODE_FULL_WGS NVH_DECODE_TARGET_WGS; else export NVH_DECODE_FULL_WGS=${e%%:*} NVH_DECODE_TARGET_WGS=${e##*:}; fi
    line=$(timeout -k 10 120 python bench.py --batch $b --no-extras --steps 200 --warmup 20)
    echo "B=$b full:target=$e $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')"
  done
done
