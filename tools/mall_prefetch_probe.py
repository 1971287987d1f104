#!/usr/bin/env python3
"""Does reading a weight matrix shortly before the M<=16 GEMM that uses it (so that it sits in the memory-side cache)
shorten that GEMM?  Times graphs of [gemm], [read] and [read(i+d); gemm(i)] over a 1 GB rotation of weights."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd")); sys.path.insert(0, str(ROOT / "tools"))
import torch
from nanovllm import ops
from bench_kernels import timeit
BF = torch.bfloat16
for name, n, k in (("qkv", 4096, 1024), ("o", 1024, 2048), ("gate_up", 6144, 1024)):
    copies = (1 << 30) // (n * k * 2)
    ws = [(torch.randn(n, k, device="cuda") * 0.02).to(BF) for _ in range(copies)]
    wi = [w.view(torch.int32) for w in ws]
    for m in (1, 8, 32):
        x = torch.randn(m, k, device="cuda").to(BF)
        gemm = timeit(lambda i: ops.linear(x, ws[i % copies]), iters=copies)
        read = timeit(lambda i: wi[i % copies].sum(), iters=copies)
        line = [f"{name} M={m}: gemm {gemm:.2f} us, read {read:.2f} us"]
        for d in (0, 1, 4):
            def both(i):
                wi[(i + d) % copies].sum()
                ops.linear(x, ws[i % copies])
            t = timeit(both, iters=copies)
            line.append(f"read(i+{d})+gemm(i) {t:.2f} (gemm part {t - read:.2f})")
        print("  ".join(line), flush=True)
    del ws, wi
