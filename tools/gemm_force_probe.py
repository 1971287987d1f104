#!/usr/bin/env python3
"""One GEMM shape under the tile the environment forces (NVH_GEMM_FORCE_TILE): plain and SwiGLU epilogue.
usage: gemm_force_probe.py M N K"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd")); sys.path.insert(0, str(ROOT / "tools"))
import os
import torch
from nanovllm import ops
from bench_kernels import timeit
m, n, k = (int(v) for v in sys.argv[1:4])
x = (torch.rand(m, k, device="cuda") * 2 - 1).to(torch.bfloat16)
ws = [(torch.rand(n, k, device="cuda") * 2 - 1).to(torch.bfloat16) for _ in range(8)]
a = timeit(lambda i: ops.linear(x, ws[i % 8]), iters=24)
b = timeit(lambda i: ops.linear_swiglu(x, ws[i % 8]), iters=24)
print(f"tile {os.environ.get('NVH_GEMM_FORCE_TILE', 'auto'):>4} M={m} N={n} K={k}: plain {a:7.1f} us  swiglu {b:7.1f} us", flush=True)
