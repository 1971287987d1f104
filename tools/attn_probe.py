#!/usr/bin/env python3
"""Launch the var-len attention kernel a few times on one shape (for rocprofv3 --pmc / --kernel-trace runs).
usage: attn_probe.py Lq Lk splits [iters]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd"))
import torch
from nanovllm import ops
lq, lk, sp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
hq, hkv, d = 16, 8, 128
BF = torch.bfloat16
pages = (lk + 255) // 256
kc = torch.randn(pages + 1, 256, hkv, d, device="cuda").to(BF)
vc = torch.randn(pages + 1, 256, hkv, d, device="cuda").to(BF)
q = torch.randn(lq, hq, d, device="cuda").to(BF)
bt = torch.arange(pages, dtype=torch.int32, device="cuda").view(1, -1)
cu_q = torch.tensor([0, lq], dtype=torch.int32, device="cuda")
cu_k = torch.tensor([0, lk], dtype=torch.int32, device="cuda")
for _ in range(iters):
    ops.attn_varlen(q, kc, vc, cu_q, cu_k, lq, d ** -0.5, bt, max_seqlen_k=lk, num_splits=sp)
torch.cuda.synchronize()
print("done")
