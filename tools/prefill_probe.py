#!/usr/bin/env python3
"""Run a few prefill steps of Qwen3-0.6B (one sequence, T new tokens over `prefix` cached tokens) so that a
rocprofv3 kernel trace of this process can be cut per step (tools/trace_breakdown.py).
usage: prefill_probe.py [T=1024] [prefix=0] [steps=6]"""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd"))
from nanovllm.config import Config  # noqa: E402
from nanovllm.engine.model_runner import ModelRunner  # noqa: E402
from nanovllm.engine.sequence import Sequence  # noqa: E402

t = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
prefix = int(sys.argv[2]) if len(sys.argv) > 2 else 0
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
import os
cfg = Config(str(ROOT / "nano-vllm-v1_amd" / "model_configs" / os.environ.get("NANOVLLM_PROBE_MODEL", "qwen3-0.6b")), max_num_batched_tokens=max(t, 1024), max_num_seqs=8,
             max_model_len=16384, enforce_eager=True, chunked_prefill=True, num_kvcache_blocks=64 * steps + 64)
runner = ModelRunner(cfg, 0, [])
g = torch.Generator().manual_seed(0)
page = 0
times = []
for i in range(steps):
    toks = torch.randint(0, 10000, (prefix + t + 1,), generator=g).tolist()
    s = Sequence(toks)
    n_pages = (prefix + t + 255) // 256
    s.block_table = list(range(page, page + n_pages))
    page += n_pages
    if prefix:
        s.num_cached_tokens, s.num_new_tokens = 0, prefix          # fill the cache (not the measured step)
        runner.run([s])
    s.num_cached_tokens, s.num_new_tokens = prefix, t
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    runner.run([s])
    torch.cuda.synchronize()
    times.append((time.perf_counter() - t0) * 1e3)
print("ms per step:", " ".join(f"{x:.3f}" for x in times))
runner.exit()
