#!/usr/bin/env python3
"""gate/up projection with the SwiGLU epilogue at prefill size.  usage: swiglu_probe.py [rows=1024]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "nano-vllm-v1_amd")); sys.path.insert(0, str(Path(__file__).resolve().parent))
import torch
from nanovllm import ops
from bench_kernels import timeit
t = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
k, inter = 1024, 3072
x = torch.randn(t, k, device="cuda").to(torch.bfloat16)
ws = [ops.interleave_gate_up((torch.randn(2 * inter, k, device="cuda") * 0.05).to(torch.bfloat16)) for _ in range(4)]
print(f"rows {t}: gate_up + SwiGLU {timeit(lambda i: ops.linear_swiglu(x, ws[i % 4]), iters=40):.2f} us")
