#!/usr/bin/env python3
"""Host-side time per decode step (the GPU idles through it): where the microseconds between the sampled ids
arriving and the next graph launch go.  usage: host_probe.py [batch=256] [ctx=1024]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd")); sys.path.insert(0, str(ROOT))
import torch
import bench as B
from nanovllm.config import Config
from nanovllm.engine import model_runner as MR
from nanovllm.utils.context import reset_context

b = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
pages = b * ((ctx0 + 200) // 256 + 2) + 16
cfg = Config(str(B.MODEL_DIR), max_num_batched_tokens=1024, max_num_seqs=max(b, 8), max_model_len=8256, enforce_eager=False,
             chunked_prefill=True, num_kvcache_blocks=pages)
runner = MR.ModelRunner(cfg, 0, [])
drv = B.Driver(runner)
g = torch.Generator().manual_seed(0)
seqs = [drv.new_seq(torch.randint(0, 10000, (ctx0,), generator=g).tolist()) for _ in range(b)]
for s in seqs:                                   # pretend the prompts are cached: decode timing does not need real K/V
    s.num_cached_tokens = ctx0 - 1
    drv.grow(s, ctx0)
T = {k: 0.0 for k in ("driver", "flatten", "replay_call", "logits_call", "sampler_call", "tolist_wait")}
steps = 40
_mode = torch.inference_mode()
_mode.__enter__()
for it in range(steps + 5):
    if it == 5:
        T = {k: 0.0 for k in T}
    t0 = time.perf_counter()
    for s in seqs:
        s.num_new_tokens = 1
        drv.grow(s, s.num_cached_tokens + 1)
    t1 = time.perf_counter()
    fb = MR.flatten_batch(seqs, 256)
    graph = runner._graph_for(fb)
    t2 = time.perf_counter()
    hidden, temps = graph.replay(fb, seqs, True)
    t3 = time.perf_counter()
    logits = runner.native.logits(hidden[:len(seqs)])
    t4 = time.perf_counter()
    ids = runner.sampler(logits, temps)
    t5 = time.perf_counter()
    toks = ids.tolist()
    t6 = time.perf_counter()
    reset_context()
    for s, t in zip(seqs, toks):
        s.num_cached_tokens += 1
        s.num_new_tokens = 0
        s.append_token(int(t))
    t7 = time.perf_counter()
    for k, v in zip(T, (t1 - t0 + t7 - t6, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)):
        T[k] += v
print(f"B={b}: per step, us: " + ", ".join(f"{k} {v / steps * 1e6:.0f}" for k, v in T.items()))
runner.exit()
