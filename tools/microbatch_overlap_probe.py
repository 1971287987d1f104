#!/usr/bin/env python3
"""Decode step of B sequences as ONE native forward vs TWO half-batch forwards on two streams inside one captured
graph (attention of one half - HBM-bound - under the projections of the other - latency-bound).
usage: microbatch_overlap_probe.py [B=256] [ctx=1024]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd"))
import numpy as np
import torch
from nanovllm.config import Config
from nanovllm.engine.model_runner import ModelRunner
from nanovllm.native import MODE_DECODE

_im = torch.inference_mode()
_im.__enter__()
b = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
PAGE = 256
pages_per = ctx // PAGE + 2
cfg = Config(str(ROOT / "nano-vllm-v1_amd" / "model_configs" / "qwen3-0.6b"), max_num_batched_tokens=1024, max_num_seqs=b,
             max_model_len=ctx + 512, chunked_prefill=True, num_kvcache_blocks=b * pages_per + 8)
runner = ModelRunner(cfg, 0, [])
g = runner.graphs[b]
h, dv = g.h, g.d
h["input_ids"][:] = np.arange(b) + 5
h["positions"][:] = ctx
h["context_lens"][:] = ctx + 1
for i in range(b):
    h["block_tables"][i, :pages_per] = np.arange(pages_per) + i * pages_per
    h["slot_mapping"][i] = (i * pages_per + ctx // PAGE) * PAGE + ctx % PAGE
g.dev.copy_(g.host)
torch.cuda.synchronize()

def timed(graph, n=30):
    graph.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            graph.replay()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n)
    return best

one = timed(g.graph)
ref = g.out.clone()
nat = runner.native
for parts in (2, 4):
    hb = b // parts
    splits = runner.decode_splits(hb)
    out = torch.empty(b, nat.desc.hidden, dtype=torch.bfloat16, device="cuda")
    wss = [torch.empty(nat.workspace_bytes(hb, hb, splits), dtype=torch.uint8, device="cuda") for _ in range(parts)]
    side = [torch.cuda.Stream() for _ in range(parts - 1)]
    def fwd(i):
        sl = slice(i * hb, (i + 1) * hb)
        nat.forward(dv["input_ids"][sl], dv["positions"][sl], mode=MODE_DECODE, batch=hb, slot_mapping=dv["slot_mapping"][sl],
                    context_lens=dv["context_lens"][sl], block_tables=dv["block_tables"][sl], decode_splits=splits,
                    out=out[sl], workspace=wss[i])
    for i in range(parts):
        fwd(i)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        cur = torch.cuda.current_stream()
        for i, s in enumerate(side):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                fwd(i + 1)
        fwd(0)
        for s in side:
            cur.wait_stream(s)
    t = timed(gr)
    err = float((out.float() - ref.float()).abs().max())
    # the same halves one after the other on one stream
    gs = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gs):
        for i in range(parts):
            fwd(i)
    ts = timed(gs)
    print(f"B={b} ctx={ctx}: one forward {one:.3f} ms | {parts} parts side by side {t:.3f} ms, in sequence {ts:.3f} ms (splits {splits}, max diff vs one forward {err:.3g})", flush=True)
