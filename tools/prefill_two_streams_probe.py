#!/usr/bin/env python3
"""Independent prefill steps (one T-token prompt each) one after the other on one stream vs spread over S streams
(each with a workspace of its own): how much of a step's launch-latency-bound time another step can fill.
usage: prefill_two_streams_probe.py [T=1024] [steps=8] [streams=2]"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd"))
from nanovllm.config import Config  # noqa: E402
from nanovllm.engine.model_runner import ModelRunner, flatten_batch  # noqa: E402
from nanovllm.engine.sequence import Sequence  # noqa: E402
from nanovllm.native import MODE_PACKED, MODE_PAGED  # noqa: E402
from nanovllm.utils.context import get_context, reset_context  # noqa: E402

_im = torch.inference_mode()
_im.__enter__()
t = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nstreams = int(sys.argv[3]) if len(sys.argv) > 3 else 2
cfg = Config(str(ROOT / "nano-vllm-v1_amd" / "model_configs" / "qwen3-0.6b"), max_num_batched_tokens=max(t, 1024), max_num_seqs=8,
             max_model_len=16384, enforce_eager=True, chunked_prefill=True, num_kvcache_blocks=8 * steps + 64)
runner = ModelRunner(cfg, 0, [])
nat = runner.native
g = torch.Generator().manual_seed(0)
work, page = [], 0
for i in range(steps):
    s = Sequence(torch.randint(0, 10000, (t + 1,), generator=g).tolist())
    n_pages = (t + 255) // 256
    s.block_table = list(range(page, page + n_pages))
    page += n_pages
    s.num_cached_tokens, s.num_new_tokens = 0, t
    fb = flatten_batch([s], runner.block_size)
    fb["fed"] = None
    ids, pos, _, _ = runner.prepare_model_input([s], fb)
    c = get_context()
    work.append(dict(ids=ids, pos=pos, mode=MODE_PACKED if c.block_tables is None else MODE_PAGED, cu_q=c.cu_seqlens_q,
                     cu_k=c.cu_seqlens_k, slot=c.slot_mapping, cl=c.context_lens, bt=c.block_tables, mq=fb["max_seqlen_q"],
                     mk=fb["max_seqlen_k"]))
    reset_context()
torch.cuda.synchronize()
nbytes = nat.workspace_bytes(t, 1, 1)
wss = [torch.empty(nbytes, dtype=torch.uint8, device="cuda") for _ in range(nstreams)]
outs = [torch.empty(t, nat.desc.hidden, dtype=torch.bfloat16, device="cuda") for _ in range(steps)]


def fwd(i, lane):
    w = work[i]
    nat.forward(w["ids"], w["pos"], mode=w["mode"], batch=1, cu_seqlens_q=w["cu_q"], cu_seqlens_k=w["cu_k"], max_seqlen_q=w["mq"],
                max_seqlen_k=w["mk"], slot_mapping=w["slot"], context_lens=w["cl"], block_tables=w["bt"], out=outs[i], workspace=wss[lane])


def timed(fn):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / steps)
    return best


def in_sequence():
    for i in range(steps):
        fwd(i, 0)


side = [torch.cuda.Stream() for _ in range(nstreams - 1)]


def side_by_side():
    cur = torch.cuda.current_stream()
    for s in side:
        s.wait_stream(cur)
    for i in range(steps):
        lane = i % nstreams
        if lane == 0:
            fwd(i, 0)
        else:
            with torch.cuda.stream(side[lane - 1]):
                fwd(i, lane)
    for s in side:
        cur.wait_stream(s)


one = timed(in_sequence)
ref = [o.clone() for o in outs]
two = timed(side_by_side)
same = all(torch.equal(a, b) for a, b in zip(ref, outs))
print(f"T={t}: {steps} steps in sequence {one:.3f} ms per step | over {nstreams} streams {two:.3f} ms per step (bit-identical: {same})", flush=True)
runner.exit()
