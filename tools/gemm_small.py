#!/usr/bin/env python3
"""Decode GEMMs at M = 1..32 for the 0.6B and 14B layer shapes: weight-streaming rate and correctness."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd")); sys.path.insert(0, str(ROOT / "tools"))
import torch
from nanovllm import ops
from bench_kernels import timeit
BF = torch.bfloat16
models = {"0.6b": [("qkv", 4096, 1024), ("o", 1024, 2048), ("gate_up", 6144, 1024), ("down", 1024, 3072)],
          "14b": [("qkv", 7168, 5120), ("o", 5120, 5120), ("gate_up", 34816, 5120), ("down", 5120, 17408)]}
for tag, shapes in models.items():
    for m in (1, 8, 32):
        tot, line = 0.0, []
        for name, n, k in shapes:
            copies = max(3, min(64, (700 << 20) // (n * k * 2) + 1))
            x = torch.randn(m, k, device="cuda").to(BF)
            ws = [(torch.randn(n, k, device="cuda") * 0.02).to(BF) for _ in range(copies)]
            sp = ops.suggest_split_k(m, n, k) if name in ("o", "down") else 1
            ref = x.float() @ ws[0].float().t()
            if name == "gate_up":
                wi = [ops.interleave_gate_up(w) for w in ws]
                g, u = ref.bfloat16().float().chunk(2, -1)
                ref = (torch.nn.functional.silu(g).bfloat16().float() * u)
                y = ops.linear_swiglu(x, wi[0])
                us = timeit(lambda i: ops.linear_swiglu(x, wi[i % copies]), iters=2 * copies)
            elif sp > 1:
                y = ops.splitk_reduce(ops.linear_splitk(x, ws[0], sp))
                us = timeit(lambda i: ops.linear_splitk(x, ws[i % copies], sp), iters=2 * copies)
            else:
                y = ops.linear(x, ws[0])
                us = timeit(lambda i: ops.linear(x, ws[i % copies]), iters=2 * copies)
            err = float((y.float() - ref).abs().max() / ref.abs().max())
            assert err < 2e-2, (name, m, err)
            tot += us
            line.append(f"{name} {us:.1f}us/{n * k * 2 / us / 1e3:.0f}GB/s(s{sp})")
            del ws
        print(f"{tag} M={m}: " + "  ".join(line) + f"  | total {tot:.1f} us {sum(n * k * 2 for _, n, k in shapes) / tot / 1e3:.0f} GB/s")
