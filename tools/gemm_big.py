#!/usr/bin/env python3
"""Large GEMM shapes (4096^3 and the Qwen3-14B projections at 1024 rows): TFLOP/s of this library's GEMM, with the
tile chosen by the library or forced (env NVH_GEMM_FORCE_TILE), next to torch's library GEMM.
usage: gemm_big.py [lib]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd")); sys.path.insert(0, str(ROOT / "tools"))
import torch
from nanovllm import ops
from bench_kernels import timeit
BF = torch.bfloat16
with_lib = len(sys.argv) > 1
for m, n, k in ((4096, 4096, 4096), (8192, 8192, 4096), (1024, 7168, 5120), (1024, 5120, 5120), (1024, 34816, 5120), (1024, 5120, 17408),
                (2048, 4096, 4096), (1024, 4096, 1024), (1024, 6144, 1024)):
    copies = max(2, min(16, (600 << 20) // (n * k * 2) + 1))
    x = (torch.rand(m, k, device="cuda") * 2 - 1).to(BF)
    ws = [(torch.rand(n, k, device="cuda") * 2 - 1).to(BF) for _ in range(copies)]
    y = ops.linear(x, ws[0])
    ref = x.float() @ ws[0].float().t()
    err = float((y.float() - ref).abs().max() / ref.abs().max())
    us = timeit(lambda i: ops.linear(x, ws[i % copies]), iters=2 * copies)
    line = f"M={m:5d} N={n:6d} K={k:6d}: {us:9.1f} us {2 * m * n * k / us / 1e6:7.0f} TFLOP/s  max rel err {err:.2e}"
    if with_lib:
        ul = timeit(lambda i: torch.nn.functional.linear(x, ws[i % copies]), iters=2 * copies)
        line += f" | library {ul:9.1f} us {2 * m * n * k / ul / 1e6:7.0f} TFLOP/s"
    print(line, flush=True)
    del ws
