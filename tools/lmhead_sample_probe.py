#!/usr/bin/env python3
"""LM head with the sampler in its epilogue (vocabulary 151936, hidden 1024) at decode batch sizes: the grid cut into whole
rounds of 256 x 256 tiles + a launch of 128 x 128 tiles for the left-over columns, against the uncut grid
(nvh_set_switch NVH_LM_HEAD_NO_CUT), and the ids of both.  usage: lmhead_sample_probe.py"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd")); sys.path.insert(0, str(ROOT / "tools"))
import torch
from nanovllm import _hip, ops
from bench_kernels import timeit
BF = torch.bfloat16
n, k = 151936, 1024
ws = [(torch.randn(n, k, device="cuda") * 0.02).to(BF) for _ in range(3)]
for m in (230, 256, 384, 512):
    x = torch.randn(m, k, device="cuda").to(BF)
    t = torch.full((m,), 0.6, device="cuda")
    res = {}
    for cut in (True, False):
        with _hip.switch("NVH_LM_HEAD_NO_CUT", -1 if cut else 1):
            ids = ops.lm_head_sample(x, ws[0], t, 7, 3).cpu()
            us = timeit(lambda i: ops.lm_head_sample(x, ws[i % 3], t, 7, i), iters=9)
        res[cut] = (us, ids)
    same = torch.equal(res[True][1], res[False][1])
    print(f"M={m:4d}: cut {res[True][0]:7.1f} us | one grid {res[False][0]:7.1f} us | same ids: {same}", flush=True)
