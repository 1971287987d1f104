#!/usr/bin/env python3
"""Per-phase cycle totals inside the var-len attention kernel (needs the -DNVH_ATTN_PROF build of the library,
NANOVLLM_HIP_LIB=.../libnanovllm_hip_prof.so).  usage: attn_phase_probe.py Lq Lk splits"""
import ctypes, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd"))
import torch
from nanovllm import ops, _hip
lq, lk, sp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
hq, hkv, d = 16, 8, 128
BF = torch.bfloat16
pages = (lk + 255) // 256
kc = torch.randn(pages + 1, 256, hkv, d, device="cuda").to(BF)
vc = torch.randn(pages + 1, 256, hkv, d, device="cuda").to(BF)
q = torch.randn(lq, hq, d, device="cuda").to(BF)
bt = torch.arange(pages, dtype=torch.int32, device="cuda").view(1, -1)
cu_q = torch.tensor([0, lq], dtype=torch.int32, device="cuda")
cu_k = torch.tensor([0, lk], dtype=torch.int32, device="cuda")
lib = ctypes.CDLL(str(_hip._LIB_PATH))
buf = (ctypes.c_ulonglong * 8)()
for _ in range(3):
    ops.attn_varlen(q, kc, vc, cu_q, cu_k, lq, d ** -0.5, bt, max_seqlen_k=lk, num_splits=sp)
torch.cuda.synchronize()
lib.nvh_debug_attn_prof(buf, 1)
n = 5
for _ in range(n):
    ops.attn_varlen(q, kc, vc, cu_q, cu_k, lq, d ** -0.5, bt, max_seqlen_k=lk, num_splits=sp)
torch.cuda.synchronize()
lib.nvh_debug_attn_prof(buf, 0)
tiles = buf[4] / n
names = ["wait dma", "barrier", "issue dma", "tile compute"]
print(f"Lq={lq} Lk={lk} splits={sp}: {tiles:.0f} workgroup-tiles per launch; cycles per tile (wave 0, 100 MHz-independent shader clock):")
for i, nm in enumerate(names):
    print(f"  {nm:14s} {buf[i] / n / tiles:9.1f}")
