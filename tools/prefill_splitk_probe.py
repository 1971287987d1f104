#!/usr/bin/env python3
"""Row-parallel projections of a prefill chunk (M=1024, N=1024): GEMM + add_rmsnorm in one piece vs split-K
slabs summed inside add_rmsnorm.  usage: prefill_splitk_probe.py"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd"))
sys.path.insert(0, str(ROOT / "tools"))
import torch
from nanovllm import ops
from bench_kernels import timeit
BF = torch.bfloat16
for m, n, k in [(1024, 1024, 2048), (1024, 1024, 3072), (512, 1024, 2048), (512, 1024, 3072)]:
    copies = 24
    xs = [torch.randn(m, k, device="cuda").to(BF) for _ in range(8)]
    ws = [(torch.randn(n, k, device="cuda") * 0.05).to(BF) for _ in range(copies)]
    res = torch.randn(m, n, device="cuda").to(BF)
    nw = torch.ones(n, device="cuda", dtype=BF)
    def whole(i):
        y = ops.linear(xs[i % 8], ws[i % copies])
        ops.add_rmsnorm(y, res, nw, 1e-6)
    line = f"M={m} N={n} K={k}: gemm+norm {timeit(whole):6.2f} us (gemm alone {timeit(lambda i: ops.linear(xs[i % 8], ws[i % copies])):6.2f})"
    for s in (2, 3, 4):
        def split(i, s=s):
            p = ops.linear_splitk(xs[i % 8], ws[i % copies], s)
            ops.add_rmsnorm(None, res, nw, 1e-6, partials=p)
        line += f" | split {s}: {timeit(split):6.2f}"
    print(line)
