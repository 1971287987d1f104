#!/usr/bin/env python3
"""128 x 128 tiles on grids of at most one workgroup per CU: four-wave against eight-wave workgroups (run once as is and once
with NVH_QKV_NO_8WAVE=1).  us per launch, weights rotated through 6 copies."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd")); sys.path.insert(0, str(ROOT / "tools"))
import torch
from nanovllm import ops
from bench_kernels import timeit
BF = torch.bfloat16
for m, n, k in [(1024, 4096, 1024), (896, 4096, 1024), (640, 4096, 1024), (512, 4096, 1024), (1024, 2048, 1024), (1024, 4096, 2048), (768, 5120, 1024), (1024, 1024, 1024)]:
    x = torch.randn(m, k, device="cuda").to(BF)
    ws = [(torch.randn(n, k, device="cuda") * 0.05).to(BF) for _ in range(6)]
    t = timeit(lambda i: ops.linear(x, ws[i % 6]), iters=60)
    print(f"{m}x{n}x{k}: {t:6.2f} us  {2 * m * n * k / t / 1e6:6.0f} TFLOP/s", flush=True)
