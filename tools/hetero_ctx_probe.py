#!/usr/bin/env python3
"""Decode step time of a batch whose contexts are short except for a few long ones (a long-context request next to
many short ones).  usage: hetero_ctx_probe.py [batch=256] [short=128] [long=8192] [n_long=1]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd")); sys.path.insert(0, str(ROOT))
import torch
import bench as B
from nanovllm.config import Config
from nanovllm.engine.model_runner import ModelRunner
b = int(sys.argv[1]) if len(sys.argv) > 1 else 256
short = int(sys.argv[2]) if len(sys.argv) > 2 else 128
long_ = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
n_long = int(sys.argv[4]) if len(sys.argv) > 4 else 1
pages = b * (short // 256 + 2) + n_long * (long_ // 256 + 2) + 16
cfg = Config(str(B.MODEL_DIR), max_num_batched_tokens=1024, max_num_seqs=max(b, 8), max_model_len=long_ + 256, enforce_eager=False,
             chunked_prefill=True, num_kvcache_blocks=pages)
runner = ModelRunner(cfg, 0, [])
drv = B.Driver(runner)
g = torch.Generator().manual_seed(0)
for nl in (0, n_long):
    drv.reset()
    seqs = []
    for i in range(b):
        n = long_ if i < nl else short
        s = drv.new_seq(torch.randint(0, 10000, (n,), generator=g).tolist())
        s.num_cached_tokens = n - 1                      # stale cache contents: timing only
        drv.grow(s, n)
        seqs.append(s)
    dt = B.time_decode(drv, seqs, 32, 4, False)
    print(f"B={b}, {nl} sequences at context {long_}, the rest at {short}: {dt / 32 * 1e3:.3f} ms per step", flush=True)
runner.exit()
