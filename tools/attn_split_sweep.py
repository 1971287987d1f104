import sys
sys.path.insert(0, "nano-vllm-v1_amd"); sys.path.insert(0, "tools")
import torch
from nanovllm import ops
from bench_kernels import timeit
hq, hkv, d = 16, 8, 128
BF = torch.bfloat16
for lq, lk in [(1024, 1024), (1024, 2048), (1024, 4096), (1024, 8192)]:
    pages = (lk + 255) // 256
    kc = torch.randn(pages + 1, 256, hkv, d, device="cuda").to(BF); vc = torch.randn(pages + 1, 256, hkv, d, device="cuda").to(BF)
    q = torch.randn(lq, hq, d, device="cuda").to(BF)
    bt = torch.arange(pages, dtype=torch.int32, device="cuda").view(1, -1)
    cu_q = torch.tensor([0, lq], dtype=torch.int32, device="cuda"); cu_k = torch.tensor([0, lk], dtype=torch.int32, device="cuda")
    line = f"Lq={lq} Lk={lk}:"
    for sp in (1, 2, 3, 4, 6, 8):
        t = timeit(lambda i: ops.attn_varlen(q, kc, vc, cu_q, cu_k, lq, d ** -0.5, bt, max_seqlen_k=lk, num_splits=sp), iters=20)
        line += f" s{sp} {t:6.1f}"
    print(line, flush=True)
