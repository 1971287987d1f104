#!/usr/bin/env bash
# Diagnostic build of the pair attention kernel with cycle stamps around the phases of its tile loop (printed by one wave):
# builds nano-vllm-v1_amd/lib/ab/libnv_pair_stamps.so from the objects of the last build.sh run.  Run on the GPU with
#   NANOVLLM_HIP_LIB=$PWD/nano-vllm-v1_amd/lib/ab/libnv_pair_stamps.so python3 tools/attn_varlen_sweep.py 16 8 1024:8192:1:4
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
cs="$root/nano-vllm-v1_amd/csrc"
mkdir -p "$root/nano-vllm-v1_amd/lib/ab"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -mllvm -amdgpu-spill-vgpr-to-agpr=0 \
    -DPAIR_STAMPS=1 -c "$cs/attention_prefill_pair.hip" -o /tmp/pair_stamps.o
objs=$(ls "$cs"/.obj/*.o | grep -v attention_prefill_pair)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/nano-vllm-v1_amd/lib/ab/libnv_pair_stamps.so" $objs /tmp/pair_stamps.o
echo "built lib/ab/libnv_pair_stamps.so"
