#!/usr/bin/env python3
"""Narrow-output decode GEMMs (the O / down projections and their tensor-parallel shards) with K split between the waves of a
workgroup against the generic tiles: run once as is and once with NVH_NO_KWAVES=1.  usage: kwaves_probe.py [N]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd")); sys.path.insert(0, str(ROOT / "tools"))
import torch
from nanovllm import ops
from bench_kernels import timeit
BF = torch.bfloat16
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for k in (256, 384, 512, 768, 1024, 1536, 2048, 3072, 4096):
    line = []
    for m in (1, 8, 32):
        copies = max(3, min(64, (700 << 20) // (n * k * 2) + 1))
        x = torch.randn(m, k, device="cuda").to(BF)
        ws = [(torch.randn(n, k, device="cuda") * 0.02).to(BF) for _ in range(copies)]
        ref = x.float() @ ws[0].float().t()
        y = ops.linear(x, ws[0])
        err = float((y.float() - ref).abs().max() / ref.abs().max())
        assert err < 2e-2, (m, k, err)
        us = timeit(lambda i: ops.linear(x, ws[i % copies]), iters=4 * copies)
        line.append(f"M={m} {us:.2f}us")
        del ws
    print(f"N={n} K={k}: " + "  ".join(line))
