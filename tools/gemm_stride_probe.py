#!/usr/bin/env python3
"""Does a power-of-two row stride cost the GEMM anything?  Same tile grid at K = 1024 (2 KB rows) and at neighbouring K
(rows of 2176 / 1920 B): time per K tile.  usage: gemm_stride_probe.py [M=1024] [N=4096]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "nano-vllm-v1_amd")); sys.path.insert(0, str(Path(__file__).resolve().parent))
import torch
from nanovllm import ops
from bench_kernels import timeit
m = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
for k in (896, 960, 1024, 1088, 1152, 2048, 2112, 3072, 3136):
    x = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    ws = [(torch.randn(n, k, device="cuda") * 0.05).to(torch.bfloat16) for _ in range(4)]
    us = timeit(lambda i: ops.linear(x, ws[i % 4]), iters=40)
    print(f"M={m} N={n} K={k:5d} (row stride {2 * k:5d} B): {us:7.2f} us  {us / (k // 64):6.3f} us per K tile  {2 * m * n * k / us / 1e6:7.1f} TFLOP/s", flush=True)
