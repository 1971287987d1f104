#!/usr/bin/env python3
"""The headline decode attention launch alone: B sequences x ctx keys, 16 / 8 heads, paged cache, every launch on other
pages (6 cache sets).  usage: attn_decode_probe.py [batch=256] [ctx=1040]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd")); sys.path.insert(0, str(ROOT / "tools"))
import torch
from nanovllm import ops
from bench_kernels import timeit
b = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 1040
hq, hkv, d = 16, 8, 128
BF = torch.bfloat16
pages_per = (ctx + 255) // 256
sets = 6
npages = sets * b * pages_per
kc = torch.randn(npages, 256, hkv, d, device="cuda", dtype=BF)
vc = torch.randn(npages, 256, hkv, d, device="cuda", dtype=BF)
q = torch.randn(b, hq, d, device="cuda", dtype=BF)
lens = torch.full((b,), ctx, dtype=torch.int32, device="cuda")
g = torch.Generator().manual_seed(0)
bts = [(torch.randperm(b * pages_per, generator=g).to(torch.int32).cuda() + s * b * pages_per).view(b, pages_per) for s in range(sets)]
sp = ops.suggest_decode_splits(b, hkv, 8256)
ref = ops.attn_decode(q, kc, vc, lens, bts[0], d ** -0.5, num_splits=sp).float()
for rep in range(3):
    t = timeit(lambda i: ops.attn_decode(q, kc, vc, lens, bts[i % sets], d ** -0.5, num_splits=sp), iters=48)
    byts = b * ctx * hkv * d * 2 * 2
    print(f"B={b} ctx={ctx} splits={sp}: {t:7.2f} us  {byts / t / 1e3:7.0f} GB/s  checksum {float(ref.abs().sum()):.3f}", flush=True)
# the fused form the decode graph replays: raw qkv rows in, q/k norm + RoPE + KV store of the new token inside the launch
qkv = torch.randn(b, (hq + 2 * hkv) * d, device="cuda", dtype=BF)
qw = torch.ones(d, device="cuda", dtype=BF)
kw = torch.ones(d, device="cuda", dtype=BF)
pos = torch.full((b,), ctx - 1, dtype=torch.int64, device="cuda")
half = torch.arange(64, dtype=torch.float32) / 64
ang = torch.arange(ctx + 8, dtype=torch.float32)[:, None] * (1e6 ** -half)[None, :]
cs = torch.cat([ang.cos(), ang.sin()], -1).contiguous().cuda()
for rep in range(3):
    t = timeit(lambda i: ops.attn_decode_fused_qkv(qkv, hq, hkv, qw, kw, 1e-6, pos, cs, kc, vc, lens, bts[i % sets], d ** -0.5, num_splits=sp), iters=48)
    print(f"fused  B={b} ctx={ctx} splits={sp}: {t:7.2f} us  {byts / t / 1e3:7.0f} GB/s", flush=True)
