#!/usr/bin/env bash
export NVH_TUNING=1   # the library reads its NVH_* routing switches from the environment only under this
# Decode step time at small batches: add + RMSNorm in the next projection's prologue with K split between waves (fused),
# the same with split-K slabs (slabs), and the separate launches (unfused).
set -e
out=gpurun_out/small_batch_ab.txt
: > $out
for b in ${BATCHES:-1 2 4 8}; do
  for mode in ${MODES:-fused slabs unfused}; do
    unset NVH_NO_NORM_GEMM NVH_DECODE_SLABS
    if [ $mode = unfused ]; then export NVH_NO_NORM_GEMM=1; fi
    if [ $mode = slabs ]; then export NVH_DECODE_SLABS=1; fi
    line=$(timeout -k 10 120 python bench.py --batch $b --no-extras --steps 200 --warmup 20)
    echo "B=$b $mode $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d.get("synchronous_loop",{}).get("ms_per_step"))')" | tee -a $out
  done
done
