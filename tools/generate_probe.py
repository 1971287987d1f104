#!/usr/bin/env python3
"""Offline throughput through the front-end (LLM.generate), the shape of the reference's bench.py: N prompts of
`in_len` random ids, `out_len` new tokens each, ignore_eos.  usage: generate_probe.py [n=256] [in_len=1024] [out_len=128]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd"))
import torch
from nanovllm import LLM, SamplingParams
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
in_len = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
out_len = int(sys.argv[3]) if len(sys.argv) > 3 else 128
model = str(ROOT / "nano-vllm-v1_amd" / "model_configs" / "qwen3-0.6b")
llm = LLM(model, max_num_batched_tokens=1024, max_num_seqs=max(n, 8), max_model_len=4096, chunked_prefill=True)
g = torch.Generator().manual_seed(0)
prompts = [torch.randint(0, 10000, (in_len,), generator=g).tolist() for _ in range(n)]
sp = SamplingParams(temperature=0.6, max_tokens=out_len, ignore_eos=True)
llm.generate(prompts[:2], SamplingParams(temperature=0.6, max_tokens=4, ignore_eos=True), use_tqdm=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
steps = [0, 0.0, 0.0]
orig_step = llm.step
def step():
    a = time.perf_counter(); r = orig_step(); steps[0] += 1; steps[1] += time.perf_counter() - a; return r
llm.step = step
outs = llm.generate(prompts, sp, use_tqdm=False)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{n} x ({in_len} in, {out_len} out): {dt:.2f} s, {n * out_len / dt:.0f} generated tok/s, {n * (in_len + out_len) / dt:.0f} total tok/s, "
      f"{steps[0]} steps, {steps[1] / steps[0] * 1e3:.2f} ms per step")
llm.exit()
