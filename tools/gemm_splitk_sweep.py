#!/usr/bin/env python3
"""Split-K sweep of one GEMM shape with rotating cold weights (28 copies, as the layers of a step):
time per launch for slab counts 1..8 - the slab write included, the consumer's extra read not.
usage: gemm_splitk_sweep.py M N K [M N K ...]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd"))
sys.path.insert(0, str(ROOT / "tools"))
import torch
from nanovllm import ops
from bench_kernels import timeit

args = [int(v) for v in sys.argv[1:]]
for m, n, k in zip(args[0::3], args[1::3], args[2::3]):
    x = (torch.rand(m, k, device="cuda") * 2 - 1).to(torch.bfloat16)
    ws = [(torch.rand(n, k, device="cuda") * 2 - 1).to(torch.bfloat16) for _ in range(28)]
    line = [f"M={m} N={n} K={k} suggested {ops.suggest_split_k(m, n, k) if hasattr(ops, 'suggest_split_k') else '?'}:"]
    line.append(f"s=1 {timeit(lambda i: ops.linear(x, ws[i % 28]), iters=56):.2f}")
    for s in (2, 3, 4, 6, 8):
        if k // s < 256:
            continue
        line.append(f"s={s} {timeit(lambda i: ops.linear_splitk(x, ws[i % 28], s), iters=56):.2f}")
    print(" ".join(line), "us", flush=True)
