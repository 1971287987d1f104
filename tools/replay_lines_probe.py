#!/usr/bin/env python3
"""Where the host time of one graph-replay call goes (B=256 decode).  usage: replay_lines_probe.py"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd")); sys.path.insert(0, str(ROOT))
import numpy as np
import torch
import bench as B
from nanovllm.config import Config
from nanovllm.engine import model_runner as MR
b, ctx0 = 256, 1024
cfg = Config(str(B.MODEL_DIR), max_num_batched_tokens=1024, max_num_seqs=b, max_model_len=8256, enforce_eager=False,
             chunked_prefill=True, num_kvcache_blocks=b * 6 + 16)
runner = MR.ModelRunner(cfg, 0, [])
drv = B.Driver(runner)
g = torch.Generator().manual_seed(0)
seqs = [drv.new_seq(torch.randint(0, 10000, (ctx0,), generator=g).tolist()) for _ in range(b)]
for s in seqs:
    s.num_cached_tokens = ctx0 - 1
    s.num_new_tokens = 1
    drv.grow(s, ctx0)
fb = MR.flatten_batch(seqs, 256)
gr = runner._graph_for(fb)
T = {}
def lap(name, t0):
    t1 = time.perf_counter(); T[name] = T.get(name, 0.0) + t1 - t0; return t1
n = len(seqs)
with torch.inference_mode():
    for it in range(60):
        if it == 10: T = {}
        torch.cuda.synchronize()
        t = time.perf_counter()
        if not gr.event.query(): gr.event.synchronize()
        t = lap("event", t)
        h = gr.h
        h["input_ids"][:n] = fb["input_ids"]; h["positions"][:n] = fb["positions"]
        h["slot_mapping"][:n] = fb["slot_mapping"]; h["slot_mapping"][n:] = -1
        h["context_lens"][:n] = fb["context_lens"]; h["context_lens"][n:] = 0
        t = lap("small fills", t)
        tb = fb["block_tables"]; h["block_tables"][:n, :tb.shape[1]] = tb
        t = lap("block tables", t)
        h["temperatures"][:n] = np.fromiter((s.temperature for s in seqs), dtype=np.float32, count=n)
        t = lap("temperatures", t)
        gr.dev.copy_(gr.host, non_blocking=True)
        t = lap("h2d", t)
        gr.event.record()
        t = lap("event record", t)
        gr.graph.replay()
        t = lap("graph launch", t)
print("per call, us: " + ", ".join(f"{k} {v / 50 * 1e6:.1f}" for k, v in T.items()), f"| staging buffer {gr.host.numel()} B")
runner.exit()
