#!/usr/bin/env python3
"""Weight-streaming rate of the M<=16 GEMM vs output width (how much is lost to a partly filled last round of
workgroups).  usage: gemv_probe.py [M=1] [K=5120]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd")); sys.path.insert(0, str(ROOT / "tools"))
import torch
from nanovllm import ops
from bench_kernels import timeit
BF = torch.bfloat16
m = int(sys.argv[1]) if len(sys.argv) > 1 else 1
k = int(sys.argv[2]) if len(sys.argv) > 2 else 5120
for n in (4096, 8192, 12288, 24576, 28672, 34816, 49152, 73728):
    copies = max(3, (900 << 20) // (n * k * 2) + 1)
    x = torch.randn(m, k, device="cuda").to(BF)
    ws = [(torch.randn(n, k, device="cuda") * 0.02).to(BF) for _ in range(copies)]
    us = timeit(lambda i: ops.linear(x, ws[i % copies]), iters=2 * copies)
    print(f"M={m} N={n:6d} K={k}: {n // 32:5d} workgroups {us:7.1f} us {n * k * 2 / us / 1e3:6.0f} GB/s", flush=True)
    del ws
