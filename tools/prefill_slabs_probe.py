#!/usr/bin/env python3
"""A prefill step as one chain of launches vs as row slabs on side streams (nvh_step.num_slabs): GPU time per step
(HIP events), host time to enqueue it, and the largest difference of the final-norm hidden states between the two.
usage: prefill_slabs_probe.py [T=1024] [prefix=0] [batch=1] [lanes=1,2,4] [reps=6]
(`batch` sequences of T/batch new tokens each over `prefix` cached tokens each)"""
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd"))
from nanovllm.config import Config  # noqa: E402
from nanovllm.engine.model_runner import ModelRunner, flatten_batch  # noqa: E402
from nanovllm.engine.sequence import Sequence  # noqa: E402
from nanovllm.utils.context import reset_context  # noqa: E402

_im = torch.inference_mode()
_im.__enter__()
t = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
prefix = int(sys.argv[2]) if len(sys.argv) > 2 else 0
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
lanes_list = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "1,2,4").split(",")]
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 6
model = os.environ.get("NANOVLLM_PROBE_MODEL", "qwen3-0.6b")
per = t // batch
pages_per = (prefix + per + 255) // 256
cfg = Config(str(ROOT / "nano-vllm-v1_amd" / "model_configs" / model), max_num_batched_tokens=max(t, 1024), max_num_seqs=max(8, batch),
             max_model_len=16384, enforce_eager=True, chunked_prefill=True, num_kvcache_blocks=batch * pages_per + 8)
runner = ModelRunner(cfg, 0, [])
g = torch.Generator().manual_seed(0)
seqs = []
for i in range(batch):
    s = Sequence(torch.randint(0, 10000, (prefix + per + 1,), generator=g).tolist())
    s.block_table = list(range(i * pages_per, (i + 1) * pages_per))
    seqs.append(s)
if prefix:                                   # fill the cache with the prefix keys (not measured)
    for s in seqs:
        s.num_cached_tokens, s.num_new_tokens = 0, prefix
    for i in range(0, batch, 4):
        runner.run(seqs[i:i + 4])
for s in seqs:
    s.num_cached_tokens, s.num_new_tokens = prefix, per


def one_step():
    fb = flatten_batch(seqs, runner.block_size)
    fb["fed"] = None
    ids, pos, _, _ = runner.prepare_model_input(seqs, fb)
    h0 = time.perf_counter()
    hidden = runner.forward_hidden(ids, pos, fb)
    host_ms = (time.perf_counter() - h0) * 1e3
    reset_context()
    return hidden, host_ms, fb.get("slabs")


ref = None
for lanes in lanes_list:
    runner.prefill_lanes = lanes
    hidden, _, slabs = one_step()
    torch.cuda.synchronize()
    nslabs = len(slabs["rows"]) if slabs else 1
    if ref is None:
        ref = hidden.float().clone()
        diff = 0.0
    else:
        diff = (hidden.float() - ref).abs().max().item()
    best, host = 1e9, 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        _, h_ms, _ = one_step()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e))
        host = min(host, h_ms)
    # the same step replayed from a hipGraph (the launches of all lanes captured through their fork / join events)
    fb = flatten_batch(seqs, runner.block_size)
    fb["fed"] = None
    ids, pos, _, _ = runner.prepare_model_input(seqs, fb)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ghidden = runner.forward_hidden(ids, pos, fb)
    reset_context()
    graph.replay()
    torch.cuda.synchronize()
    gdiff = (ghidden.float() - hidden.float()).abs().max().item()
    gbest, ghost = 1e9, 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        h0 = time.perf_counter()
        graph.replay()
        ghost = min(ghost, (time.perf_counter() - h0) * 1e3)
        e.record()
        torch.cuda.synchronize()
        gbest = min(gbest, s.elapsed_time(e))
    # ... and eight replays back to back (the launch of one overlaps the run of the one before)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(8):
        graph.replay()
    e.record()
    torch.cuda.synchronize()
    g8 = s.elapsed_time(e) / 8
    print(f"{model} T={t} prefix={prefix} batch={batch} lanes={lanes} slabs={nslabs}: {best:.3f} ms per step on the GPU, "
          f"{host:.3f} ms to enqueue, max |hidden - one chain| = {diff:.4f} (max |hidden| {ref.abs().max().item():.2f}) | "
          f"as a graph: {gbest:.3f} ms, {ghost:.3f} ms to launch, {g8:.3f} ms per replay over 8, |graph - eager| = {gdiff:.4f}", flush=True)
    del graph
runner.exit()
