#!/usr/bin/env python3
"""Decode attention bandwidth vs context length at serving-sized batches (paged cache, every launch on other pages).
usage: attn_decode_ctx_sweep.py [batch=512]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd")); sys.path.insert(0, str(ROOT / "tools"))
import torch
from nanovllm import ops
from bench_kernels import timeit
b = int(sys.argv[1]) if len(sys.argv) > 1 else 512
hq, hkv, d = 16, 8, 128
BF = torch.bfloat16
for ctx in (32, 64, 128, 192, 256, 512, 1024):
    pages_per = (ctx + 255) // 256
    sets = 6
    npages = sets * b * pages_per
    kc = torch.randn(npages, 256, hkv, d, device="cuda", dtype=BF)
    vc = torch.randn(npages, 256, hkv, d, device="cuda", dtype=BF)
    q = torch.randn(b, hq, d, device="cuda", dtype=BF)
    lens = torch.full((b,), ctx, dtype=torch.int32, device="cuda")
    bts = [(torch.arange(b * pages_per, dtype=torch.int32, device="cuda") + s * b * pages_per).view(b, pages_per) for s in range(sets)]
    sp = ops.suggest_decode_splits(b, hkv, ctx)
    t = timeit(lambda i: ops.attn_decode(q, kc, vc, lens, bts[i % sets], d ** -0.5, num_splits=sp), iters=24)
    byts = b * ctx * hkv * d * 2 * 2
    print(f"B={b} ctx={ctx:5d} splits={sp}: {t:7.2f} us  {byts / t / 1e3:7.0f} GB/s", flush=True)
    del kc, vc
