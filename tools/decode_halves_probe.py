#!/usr/bin/env python3
"""A decode step of B sequences as ONE chain of launches against the same step cut into `pieces` row ranges that run side
by side on their own streams (the attention of one range streams K/V while the other ranges run their launch-bound
projections).  Each form is captured into a hipGraph and timed over replays (final-norm hidden states only, no LM head).
usage: decode_halves_probe.py [B=256] [ctx=1040] [pieces=2,4]"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd"))
from nanovllm import native as N  # noqa: E402
from nanovllm.config import Config  # noqa: E402
from nanovllm.engine.model_runner import ModelRunner  # noqa: E402

_im = torch.inference_mode()
_im.__enter__()
b = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 1040
pieces_list = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "2,4").split(",")]
pages_per = (ctx + 255) // 256
cfg = Config(str(ROOT / "nano-vllm-v1_amd" / "model_configs" / "qwen3-0.6b"), max_num_batched_tokens=1024, max_num_seqs=b, max_model_len=4096,
             enforce_eager=True, chunked_prefill=True, num_kvcache_blocks=b * pages_per + 8)
runner = ModelRunner(cfg, 0, [])
nat = runner.native
runner.kv_cache.normal_(0, 0.5)
dev = "cuda"
ids = torch.randint(0, 10000, (b,), device=dev)
pos = torch.full((b,), ctx - 1, dtype=torch.int64, device=dev)
tables = torch.arange(b * pages_per, dtype=torch.int32, device=dev).view(b, pages_per)
ctx_lens = torch.full((b,), ctx, dtype=torch.int32, device=dev)
slots = (tables[:, (ctx - 1) // 256].to(torch.int64) * 256 + (ctx - 1) % 256).to(torch.int32)


def step(rows, out, ws):
    r = slice(*rows)
    n = rows[1] - rows[0]
    return nat.forward(ids[r], pos[r], mode=N.MODE_DECODE, batch=n, slot_mapping=slots[r], context_lens=ctx_lens[r], block_tables=tables[r],
                       decode_splits=1, out=out[r], workspace=ws)


def run(pieces):
    out = torch.empty(b, nat.desc.hidden, dtype=torch.bfloat16, device=dev)
    bounds = [(i * b // pieces, (i + 1) * b // pieces) for i in range(pieces)]
    wss = [torch.empty(nat.workspace_bytes(r1 - r0, r1 - r0, 1), dtype=torch.uint8, device=dev) for r0, r1 in bounds]
    streams = [torch.cuda.Stream() for _ in range(pieces - 1)]

    def body():
        cur = torch.cuda.current_stream()
        fork = torch.cuda.Event()
        fork.record(cur)
        for i, rows in enumerate(bounds):
            if i == 0:
                step(rows, out, wss[0])
            else:
                s = streams[i - 1]
                s.wait_event(fork)
                with torch.cuda.stream(s):
                    step(rows, out, wss[i])
        for s in streams:
            e = torch.cuda.Event()
            e.record(s)
            cur.wait_event(e)

    body()
    torch.cuda.synchronize()
    eager = 1e9
    for _ in range(3):
        s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        for _ in range(6):
            body()
        e0.record()
        torch.cuda.synchronize()
        eager = min(eager, s0.elapsed_time(e0) / 6)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        for _ in range(8):
            g.replay()
        e0.record()
        torch.cuda.synchronize()
        best = min(best, s0.elapsed_time(e0) / 8)
    return best, out.float().clone(), eager


base, ref, base_e = run(1)
print(f"B={b} ctx={ctx}: one chain {base:.3f} ms per step as a graph, {base_e:.3f} ms as eager launches (hidden states only)", flush=True)
for p in pieces_list:
    ms, got, ms_e = run(p)
    print(f"  {p} row ranges on {p} streams: {ms:.3f} ms per step ({base / ms:.3f}x), eager {ms_e:.3f} ms, max |diff| to the one chain {float((got - ref).abs().max()):.4f} (max |hidden| {float(ref.abs().max()):.2f})", flush=True)
runner.exit()
