#!/usr/bin/env python3
"""Var-len attention time (kernel + merge) over prefill shapes, with the key split the library suggests.
usage: attn_shape_sweep.py"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd"))
sys.path.insert(0, str(ROOT / "tools"))
import torch
from nanovllm import ops, _hip
from bench_kernels import timeit
hq, hkv, d = 16, 8, 128
BF = torch.bfloat16
for nseq, lq, lk in [(1, 1024, 1024), (1, 1024, 2048), (1, 1024, 4096), (1, 1024, 8192), (8, 128, 128), (4, 256, 256), (2, 512, 512), (1, 512, 512), (1, 256, 1280)]:
    pages_per = (lk + 255) // 256
    kc = torch.randn(nseq * pages_per + 1, 256, hkv, d, device="cuda").to(BF)
    vc = torch.randn(nseq * pages_per + 1, 256, hkv, d, device="cuda").to(BF)
    q = torch.randn(nseq * lq, hq, d, device="cuda").to(BF)
    bt = torch.arange(nseq * pages_per, dtype=torch.int32, device="cuda").view(nseq, pages_per)
    cu_q = torch.arange(nseq + 1, dtype=torch.int32, device="cuda") * lq
    cu_k = torch.arange(nseq + 1, dtype=torch.int32, device="cuda") * lk
    sp = _hip.lib().nvh_attn_varlen_suggest_splits(nseq * lq, nseq, hq, hkv, lk)
    t = timeit(lambda i: ops.attn_varlen(q, kc, vc, cu_q, cu_k, lq, d ** -0.5, bt, max_seqlen_k=lk, num_splits=sp), iters=20)
    flops = 4 * hq * d * nseq * (lq * (lk - lq) + lq * (lq + 1) // 2)
    print(f"{nseq} x Lq={lq:5d} Lk={lk:5d} splits={sp:2d}: {t:7.2f} us  {flops / t / 1e6:6.1f} TFLOP/s", flush=True)
