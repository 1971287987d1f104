#!/usr/bin/env python3
"""One GEMM shape under the dispatcher's choice and under forced tiles / K-slab counts: time per call (weights rotating
over 6 copies), TFLOP/s.  Tile forcing needs NVH_TUNING=1: NVH_GEMM_FORCE_TILE=<index of enum Tile in csrc/gemm.hip>.
usage: gemm_shape_probe.py M N K [split_k ...]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd"))
sys.path.insert(0, str(ROOT / "tools"))
import torch
from nanovllm import ops
from bench_kernels import timeit
m, n, k = (int(a) for a in sys.argv[1:4])
splits = [int(a) for a in sys.argv[4:]] or [1]
BF = torch.bfloat16
x = torch.randn(m, k, device="cuda").to(BF)
ws = [(torch.randn(n, k, device="cuda") * 0.02).to(BF) for _ in range(6)]
for s in splits:
    if s == 1:
        us = timeit(lambda i: ops.linear(x, ws[i % 6]), iters=30)
    else:
        us = timeit(lambda i: ops.splitk_reduce(ops.linear_splitk(x, ws[i % 6], s)), iters=30)
    print(f"M={m} N={n} K={k} split_k={s}: {us:8.2f} us {2 * m * n * k / us / 1e6:7.1f} TFLOP/s", flush=True)
