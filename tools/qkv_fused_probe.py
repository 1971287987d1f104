#!/usr/bin/env python3
"""QKV projection + q/k epilogue + KV store: one call (GEMM epilogue) against GEMM followed by the epilogue kernel.
usage: qkv_fused_probe.py [rows=1024]"""
import sys
sys.path.insert(0, "nano-vllm-v1_amd"); sys.path.insert(0, "tools")
import torch
from nanovllm import ops
from bench_kernels import timeit
t = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
hq, hkv, d, k = 16, 8, 128, 1024
n = (hq + 2 * hkv) * d
BF = torch.bfloat16
x = torch.randn(t, k, device="cuda").to(BF)
ws = [(torch.randn(n, k, device="cuda") * 0.05).to(BF) for _ in range(8)]
qw = torch.ones(d, device="cuda", dtype=BF); kw = torch.ones(d, device="cuda", dtype=BF)
pos = torch.arange(t, device="cuda", dtype=torch.int64)
table = torch.randn(4096, 128, device="cuda", dtype=torch.float32)
kc = torch.zeros(8, 256, hkv, d, device="cuda", dtype=BF); vc = torch.zeros_like(kc)
slots = torch.arange(t, device="cuda", dtype=torch.int32)
def sep(i):
    y = ops.linear(x, ws[i % 8])
    ops.qknorm_rope_store_(y, hq, hkv, qw, kw, 1e-6, pos, table, kc, vc, slots)
def one(i):
    ops.linear_qkv_rope_store(x, ws[i % 8], hq, hkv, qw, kw, 1e-6, pos, table, kc, vc, slots)
print(f"rows {t}: GEMM alone {timeit(lambda i: ops.linear(x, ws[i % 8]), iters=40):.2f} us, GEMM + epilogue kernel {timeit(sep, iters=40):.2f} us, "
      f"one call {timeit(one, iters=40):.2f} us")
