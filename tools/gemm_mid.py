#!/usr/bin/env python3
"""Layer projections of Qwen3-0.6B at decode batch sizes 64..512: time per call (cold weights), with the split-K
the library suggests and with forced values.  Tile forcing: NVH_GEMM_FORCE_TILE=<enum index> in the environment.
usage: gemm_mid.py [M ...]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd"))
sys.path.insert(0, str(ROOT / "tools"))
import torch
from nanovllm import ops
from bench_kernels import timeit
BF = torch.bfloat16
ms = [int(a) for a in sys.argv[1:]] or [256]
for m in ms:
    for name, n, k in (("qkv", 4096, 1024), ("o", 1024, 2048), ("gate_up", 6144, 1024), ("down", 1024, 3072)):
        copies = 28
        xs = [torch.randn(m, k, device="cuda").to(BF) for _ in range(4)]
        ws = [(torch.randn(n, k, device="cuda") * 0.05).to(BF) for _ in range(copies)]
        line = f"M={m:4d} {name:8s} N={n:5d} K={k:5d} suggest={ops.suggest_split_k(m, n, k)}"
        for s in (1, 2, 3, 4, 6, 8):
            if k // 64 < s:
                continue
            if s == 1:
                t = timeit(lambda i: ops.linear(xs[i % 4], ws[i % copies]))
            else:
                t = timeit(lambda i, s=s: ops.linear_splitk(xs[i % 4], ws[i % copies], s))
            line += f" | s{s}: {t:5.2f}"
        print(line, flush=True)
