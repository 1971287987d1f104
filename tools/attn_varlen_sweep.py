#!/usr/bin/env python3
"""Var-len attention timing over the prefill shapes of a 1024-token chunk (16 q heads / 8 kv heads):
time per launch and TFLOP/s for a list of (Lq, Lk, batch, splits) cases.  Every launch reads another
layer's cache (28 copies), as consecutive layers of a step do.
usage: attn_varlen_sweep.py [hq=16] [hkv=8] [Lq:Lk:B:splits,...]      env NVH_ATTN_KS=1|2 forces the key-group count"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd"))
sys.path.insert(0, str(ROOT / "tools"))
import torch
from nanovllm import ops
from bench_kernels import timeit
hq = int(sys.argv[1]) if len(sys.argv) > 1 else 16
hkv = int(sys.argv[2]) if len(sys.argv) > 2 else 8
d, BF, L = 128, torch.bfloat16, 8
print(f"{'Lq':>6} {'Lk':>6} {'B':>3} {'splits':>6} {'us':>9} {'TFLOP/s':>8}")
CASES = [tuple(int(v) for v in c.split(":")) for c in sys.argv[3].split(",")] if len(sys.argv) > 3 else None
for lq, lk, b, splits in CASES or [(1024, 1024, 1, 1), (1024, 1024, 1, 2), (1024, 1024, 1, 4), (1024, 2048, 1, 1), (1024, 2048, 1, 2),
                          (1024, 4096, 1, 1), (1024, 4096, 1, 2), (1024, 8192, 1, 1), (1024, 8192, 1, 2), (1024, 8192, 1, 4),
                          (128, 128, 8, 1), (512, 512, 2, 1), (64, 8192, 1, 0), (16, 4096, 4, 0)]:
    pages = (lk + 255) // 256
    kcs = [torch.randn(b * pages + 1, 256, hkv, d, device="cuda").to(BF) for _ in range(L)]
    vcs = [torch.randn(b * pages + 1, 256, hkv, d, device="cuda").to(BF) for _ in range(L)]
    q = torch.randn(b * lq, hq, d, device="cuda").to(BF)
    bt = torch.arange(b * pages, dtype=torch.int32, device="cuda").view(b, pages)
    cu_q = torch.arange(b + 1, dtype=torch.int32, device="cuda") * lq
    cu_k = torch.arange(b + 1, dtype=torch.int32, device="cuda") * lk
    us = timeit(lambda i: ops.attn_varlen(q, kcs[i % L], vcs[i % L], cu_q, cu_k, lq, d ** -0.5, bt, max_seqlen_k=lk, num_splits=splits), iters=40)
    visible = b * (lq * (lk - lq) + lq * (lq + 1) // 2)
    print(f"{lq:6d} {lk:6d} {b:3d} {splits:6d} {us:9.2f} {4 * hq * d * visible / us / 1e6:8.1f}")
