#!/usr/bin/env python3
"""LM-head GEMM (vocabulary 151936, hidden 1024) at decode batch sizes, tile chosen by the library or forced with
NVH_GEMM_FORCE_TILE.  usage: lmhead_probe.py"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd")); sys.path.insert(0, str(ROOT / "tools"))
import torch
from nanovllm import ops
from bench_kernels import timeit
BF = torch.bfloat16
n, k = 151936, 1024
ws = [(torch.randn(n, k, device="cuda") * 0.02).to(BF) for _ in range(3)]
for m in (64, 128, 256, 512, 1024):
    x = torch.randn(m, k, device="cuda").to(BF)
    us = timeit(lambda i: ops.linear(x, ws[i % 3]), iters=9)
    print(f"M={m:5d}: {us:8.1f} us  {2 * m * n * k / us / 1e6:6.0f} TFLOP/s  weights at {n * k * 2 / us / 1e3:6.0f} GB/s", flush=True)
