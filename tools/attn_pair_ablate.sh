#!/usr/bin/env bash
# Timing-only builds of the pair attention kernel with parts of its tile loop compiled out (results are then wrong):
# bits 1 = no exp / convert, 2 = no score MFMAs, 4 = no P.V MFMAs, 8 = no tile staging inside the loop, 16 = no K fragment
# reads after the first, 32 = no V^T fragment reads.  `build` makes lib/ab/libnv_pair_<bits>.so here (no GPU needed);
# `run` times Lq 1024 over 8192 keys in 4 ranges with each of them on the GPU.
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
cs="$root/nano-vllm-v1_amd/csrc"
variants="${VARIANTS:-1 6 8 16 32 48 56 63}"
if [ "${1:-run}" = build ]; then
  mkdir -p "$root/nano-vllm-v1_amd/lib/ab"
  objs=$(ls "$cs"/.obj/*.o | grep -v attention_prefill_pair)
  for v in $variants; do
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -mllvm -amdgpu-spill-vgpr-to-agpr=0 \
        -DPAIR_ABLATE=$v -c "$cs/attention_prefill_pair.hip" -o /tmp/pair_ab_$v.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/nano-vllm-v1_amd/lib/ab/libnv_pair_$v.so" $objs /tmp/pair_ab_$v.o ) &
  done
  wait
  ls "$root/nano-vllm-v1_amd/lib/ab"
  exit 0
fi
C="1024:8192:1:4"
echo "# complete"; timeout -k 10 100 python3 -u tools/attn_varlen_sweep.py 16 8 $C 2>/dev/null | tail -1
for v in $variants; do
  echo "# ablate $v"; NANOVLLM_HIP_LIB=$root/nano-vllm-v1_amd/lib/ab/libnv_pair_$v.so timeout -k 10 100 python3 -u tools/attn_varlen_sweep.py 16 8 $C 2>/dev/null | tail -1
done
