#!/usr/bin/env python3
"""Round quantisation of big-tile GEMMs: a grid a little over a whole number of rounds of workgroups against the same
product cut into a whole-round part and a remainder on smaller tiles.  usage: gemm_rounds_probe.py"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd")); sys.path.insert(0, str(ROOT / "tools"))
import torch
from nanovllm import ops
from bench_kernels import timeit
BF = torch.bfloat16
m = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for n, k in ((34816, 5120), (7168, 5120), (5120, 5120), (5120, 17408), (151936, 1024)):
    copies = max(2, min(8, (600 << 20) // (n * k * 2) + 1))
    x = (torch.rand(m, k, device="cuda") * 2 - 1).to(BF)
    ws = [(torch.rand(n, k, device="cuda") * 2 - 1).to(BF) for _ in range(copies)]
    us = timeit(lambda i: ops.linear(x, ws[i % copies]), iters=3 * copies)
    line = f"M={m} N={n:6d} K={k:6d}: plain {us:8.1f} us {2 * m * n * k / us / 1e6:6.0f} TF/s"
    if n % 64 == 0:
        us2 = timeit(lambda i: ops.linear_swiglu(x, ws[i % copies]), iters=3 * copies)
        line += f" | swiglu {us2:8.1f} us"
    for s in (2, 3, 4, 5, 6):
        if (k // 64) % s == 0 or True:
            us3 = timeit(lambda i: ops.linear_splitk(x, ws[i % copies], s), iters=3 * copies)
            line += f" | split{s} {us3:7.1f}"
    print(line, flush=True)
    del ws
