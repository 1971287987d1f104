#!/usr/bin/env python3
"""The projections of a prefill chunk with the RMSNorm as a row scale against their plain forms, alone: HIP-event time per
launch over 12 launches replayed from a hipGraph (rotating weight copies).  usage: rowscale_probe.py [tokens=1024] [weight copies=12]"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd"))
from nanovllm import ops  # noqa: E402

t = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
h, hq, hkv, d, inter = 1024, 16, 8, 128, 3072
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 12          # weight copies the launches rotate over (1: weights stay in cache)
bf = torch.bfloat16
dev = "cuda"


def w(n, k):
    return [(torch.randn(n, k, device=dev) * 0.02).to(bf) for _ in range(copies)]


x = torch.randn(t, h, device=dev).to(bf)
xa = torch.randn(t, hq * d, device=dev).to(bf)
res = torch.randn(t, h, device=dev).to(bf)
wo, wg = w(h, hq * d), [ops.interleave_gate_up(v) for v in w(2 * inter, h)]
wq = w((hq + 2 * hkv) * d, h)
qw = kw = torch.ones(d, device=dev, dtype=bf)
pos = torch.arange(t, device=dev)
table = torch.randn(4096, 128, device=dev)
rowsq = torch.rand(t, 32, device=dev) * 30
gamma = torch.ones(h, device=dev, dtype=bf)
sk = ops.suggest_split_k(t, h, hq * d)
print(f"tokens {t}: suggested K-slabs for the O projection {sk}")


def o_slabs(i):
    if sk == 1:
        return ops.add_rmsnorm(ops.linear(xa, wo[i]), res, gamma, 1e-6)
    return ops.add_rmsnorm(None, res, gamma, 1e-6, partials=ops.linear_splitk(xa, wo[i], sk))


cases = {
    "o_proj": lambda i: ops.linear(xa, wo[i]),
    "o_proj (K-slabs as suggested) + add + RMSNorm": lambda i: o_slabs(i),
    "o_proj + add + RMSNorm": lambda i: ops.add_rmsnorm(ops.linear(xa, wo[i]), res, gamma, 1e-6),
    "o_proj + residual + row sums": lambda i: ops.linear_add_residual_(xa, wo[i], res),
    "gate_up + SwiGLU": lambda i: ops.linear_swiglu(x, wg[i]),
    "gate_up + SwiGLU, row-scaled": lambda i: ops.linear_swiglu_rowscaled(x, wg[i], rowsq, 1e-6),
    "qkv + epilogue": lambda i: ops.linear_qkv_rope_store(x, wq[i], hq, hkv, qw, kw, 1e-6, pos, table),
    "qkv + epilogue, row-scaled": lambda i: ops.linear_qkv_rope_store_rowscaled(x, wq[i], hq, hkv, qw, kw, 1e-6, pos, table, rowsq, 1e-6),
}
if ops.residual_rowsq_parts(t, h, hq * d) == 0:
    cases.pop("o_proj + residual + row sums")
from nanovllm import _hip
if not _hip.lib().nvh_gemm_rowscaled_supported(t, 2 * inter, h, 1):
    cases.pop("gate_up + SwiGLU, row-scaled")
if not _hip.lib().nvh_gemm_rowscaled_supported(t, (hq + 2 * hkv) * d, h, 2):
    cases.pop("qkv + epilogue, row-scaled")
for name, fn in cases.items():
    for i in range(12):
        fn(i % copies)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for i in range(12):
            fn(i % copies)
    graph.replay()
    best = 1e30
    for _ in range(5):
        s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        graph.replay()
        e0.record()
        torch.cuda.synchronize()
        best = min(best, s0.elapsed_time(e0) * 1e3 / 12)
    print(f"{name:34s} {best:7.2f} us")
