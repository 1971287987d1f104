#!/usr/bin/env python3
"""Weight-streaming rate of the decode GEMMs at the Qwen3-14B layer shapes (cold weights, rotating copies)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd")); sys.path.insert(0, str(ROOT / "tools"))
import torch
from nanovllm import ops
from bench_kernels import timeit
BF = torch.bfloat16
shapes = [("qkv", 7168, 5120), ("o", 5120, 5120), ("gate_up", 34816, 5120), ("down", 5120, 17408)]
print(f"{'op':>8} {'M':>4} {'split':>5} {'us':>8} {'W GB/s':>8}")
for m in ([int(a) for a in sys.argv[1:]] or [1, 8, 32, 64, 256]):
    tot = 0.0
    for name, n, k in shapes:
        copies = max(3, (700 << 20) // (n * k * 2) + 1)
        x = torch.randn(m, k, device="cuda").to(BF)
        ws = [(torch.randn(n, k, device="cuda") * 0.02).to(BF) for _ in range(copies)]
        sp = ops.suggest_split_k(m, n, k) if name in ("o", "down") else 1
        if name == "gate_up":
            wi = [ops.interleave_gate_up(w) for w in ws]
            us = timeit(lambda i: ops.linear_swiglu(x, wi[i % copies]), iters=2 * copies)
        elif sp > 1:
            us = timeit(lambda i: ops.linear_splitk(x, ws[i % copies], sp), iters=2 * copies)
        else:
            us = timeit(lambda i: ops.linear(x, ws[i % copies]), iters=2 * copies)
        tot += us
        print(f"{name:>8} {m:4d} {sp:5d} {us:8.1f} {n * k * 2 / us / 1e3:8.0f}")
        del ws
    print(f"   layer GEMMs at M={m}: {tot:.1f} us ({sum(n * k * 2 for _, n, k in shapes) / tot / 1e3:.0f} GB/s)")
