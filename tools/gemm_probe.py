#!/usr/bin/env python3
"""GEMM timing probes: hot (one weight copy, cache resident) vs cold (rotating copies), and time vs K."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd"))
sys.path.insert(0, str(ROOT / "tools"))
import torch
from nanovllm import ops
from bench_kernels import timeit
BF = torch.bfloat16
print(f"{'M':>6} {'N':>6} {'K':>6} {'hot us':>8} {'cold us':>8} {'TF hot':>7} {'TF cold':>7} {'us/ktile(cold)':>14}")
for m, n, k in [(1024, 4096, 512), (1024, 4096, 1024), (1024, 4096, 2048), (1024, 4096, 4096), (1024, 1024, 1024), (1024, 1024, 2048), (1024, 1024, 4096),
                (2048, 4096, 1024), (4096, 4096, 1024)]:
    copies = max(2, min(128, (600 << 20) // (n * k * 2) + 1))
    xs = [torch.randn(m, k, device="cuda").to(BF) for _ in range(8)]
    ws = [(torch.randn(n, k, device="cuda") * 0.05).to(BF) for _ in range(copies)]
    hot = timeit(lambda i: ops.linear(xs[0], ws[0]), iters=50)
    cold = timeit(lambda i: ops.linear(xs[i % 8], ws[i % copies]), iters=max(50, copies))
    f = 2 * m * n * k / 1e6
    print(f"{m:6d} {n:6d} {k:6d} {hot:8.2f} {cold:8.2f} {f / hot:7.1f} {f / cold:7.1f} {cold / (k / 64):14.3f}")
