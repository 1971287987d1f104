#!/usr/bin/env python3
"""Host cost of the pieces of one graph-replayed decode step.  usage: replay_probe.py [batch=256] [ctx=1024]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd")); sys.path.insert(0, str(ROOT))
import numpy as np
import torch
import bench as B
from nanovllm.config import Config
from nanovllm.engine import model_runner as MR

b = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
pages = b * ((ctx0 + 200) // 256 + 2) + 16
cfg = Config(str(B.MODEL_DIR), max_num_batched_tokens=1024, max_num_seqs=max(b, 8), max_model_len=8256, enforce_eager=False,
             chunked_prefill=True, num_kvcache_blocks=pages)
runner = MR.ModelRunner(cfg, 0, [])
drv = B.Driver(runner)
g = torch.Generator().manual_seed(0)
seqs = [drv.new_seq(torch.randint(0, 10000, (ctx0,), generator=g).tolist()) for _ in range(b)]
for s in seqs:
    s.num_cached_tokens = ctx0 - 1
    s.num_new_tokens = 1
    drv.grow(s, ctx0)
fb = MR.flatten_batch(seqs, 256)
G = runner._graph_for(fb)
n = len(seqs)
T = dict(fill=0.0, temps=0.0, copy=0.0, launch=0.0, sync=0.0)
steps = 50
with torch.inference_mode():
    for it in range(steps + 5):
        if it == 5:
            T = {k: 0.0 for k in T}
        t0 = time.perf_counter()
        h = G.h
        h["input_ids"][:n] = fb["input_ids"]; h["positions"][:n] = fb["positions"]; h["slot_mapping"][:n] = fb["slot_mapping"]
        h["slot_mapping"][n:] = -1; h["context_lens"][:n] = fb["context_lens"]; h["context_lens"][n:] = 0
        tb = fb["block_tables"]; h["block_tables"][:n, :tb.shape[1]] = tb
        t1 = time.perf_counter()
        h["temperatures"][:n] = np.fromiter((s.temperature for s in seqs), dtype=np.float32, count=n)
        t2 = time.perf_counter()
        G.dev.copy_(G.host, non_blocking=True)
        t3 = time.perf_counter()
        G.graph.replay()
        t4 = time.perf_counter()
        torch.cuda.synchronize()
        t5 = time.perf_counter()
        for k, v in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            T[k] += v
print(f"B={b}: per step, us: " + ", ".join(f"{k} {v / steps * 1e6:.0f}" for k, v in T.items()))
runner.exit()
