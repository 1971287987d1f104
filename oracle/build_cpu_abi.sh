#!/usr/bin/env bash
# CPU build of include/nanovllm_hip.h's operator entry points (test infrastructure): oracle/cpu_abi.cpp ->
# oracle/_build/libnanovllm_cpu.so.  Plain g++, no GPU toolchain.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
mkdir -p "$here/_build"
g++ -O2 -std=c++17 -fPIC -shared -fopenmp -ffp-contract=off -Wall -Wextra -Wno-unused-parameter \
    "$here/cpu_abi.cpp" -o "$here/_build/libnanovllm_cpu.so"
echo "built $here/_build/libnanovllm_cpu.so"
