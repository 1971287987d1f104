#!/usr/bin/env bash
# Build libnanovllm_hip.so for gfx950 (cross-compiles without a GPU).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../lib"
mkdir -p "$out" "$here/.obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function"
pids=()
for f in api elementwise attention_decode attention_prefill gemm sampler comm model_forward; do
    if [ ! -f "$here/.obj/$f.o" ] || [ "$here/$f.hip" -nt "$here/.obj/$f.o" ] || [ "$here/common.h" -nt "$here/.obj/$f.o" ] || [ "$here/../../include/nanovllm_hip.h" -nt "$here/.obj/$f.o" ]; then
        $HIPCC $FLAGS -c "$here/$f.hip" -o "$here/.obj/$f.o" &
        pids+=($!)
    fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$out/libnanovllm_hip.so" "$here"/.obj/*.o
echo "built $out/libnanovllm_hip.so"
# host-path helpers (CPython extension; optional at run time: the Python implementation is used without it)
PYINC="$(python3 -c 'import sysconfig, numpy; print("-I" + sysconfig.get_paths()["include"], "-I" + numpy.get_include())')"
gcc -O2 -shared -fPIC -Wall $PYINC "$here/host/hostpath.c" -o "$here/../nanovllm/_hostpath.so"
echo "built $here/../nanovllm/_hostpath.so"
