#!/usr/bin/env bash
# Var-len attention over long key ranges: the pair kernel (a wave = both heads of a GQA pair) against the one-head-per-wave
# kernel (NVH_ATTN_NO_PAIR), same box.  splits 0 = the library's suggestion.
export NVH_TUNING=1
C="1024:2048:1:0,1024:2048:1:2,1024:4096:1:0,1024:4096:1:4,1024:8192:1:0,1024:8192:1:4,1024:8192:1:8,512:4096:2:0,1024:16384:1:0"
echo "# pair kernel"; timeout -k 10 200 python3 -u tools/attn_varlen_sweep.py 16 8 $C 2>/dev/null
echo "# pair kernel, 4 stages"; NVH_ATTN_PAIR_STAGES=4 timeout -k 10 200 python3 -u tools/attn_varlen_sweep.py 16 8 $C 2>/dev/null
echo "# one head per wave"; NVH_ATTN_NO_PAIR=1 timeout -k 10 200 python3 -u tools/attn_varlen_sweep.py 16 8 $C 2>/dev/null
