#!/usr/bin/env python3
"""One GEMM shape launched repeatedly (for counter passes).  usage: gemm_one.py M N K [launches=50]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "nano-vllm-v1_amd"))
import torch
from nanovllm import ops
m, n, k = (int(v) for v in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 50
x = torch.randn(m, k, device="cuda").to(torch.bfloat16)
ws = [(torch.randn(n, k, device="cuda") * 0.05).to(torch.bfloat16) for _ in range(4)]
for i in range(reps):
    ops.linear(x, ws[i % 4])
torch.cuda.synchronize()
