#!/usr/bin/env python3
"""Long-context chunked prefill with prefix reuse through the front-end: a first wave of N prompts of 8192 tokens,
then a second wave whose prompts share their first 4096 tokens (16 pages) with the first wave's.  The second wave
only prefills what is not cached.  usage: prefix_wave_probe.py [model=qwen3-0.6b] [n=8] [out=64]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd"))
import torch
from nanovllm import LLM, SamplingParams
name = sys.argv[1] if len(sys.argv) > 1 else "qwen3-0.6b"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
out = int(sys.argv[3]) if len(sys.argv) > 3 else 64
llm = LLM(str(ROOT / "nano-vllm-v1_amd" / "model_configs" / name), max_num_batched_tokens=1024, max_num_seqs=64,
          max_model_len=8192 + out + 8, chunked_prefill=True, num_kvcache_blocks=n * 2 * 34 + 8)
g = torch.Generator().manual_seed(0)
wave1 = [torch.randint(0, 10000, (8192,), generator=g).tolist() for _ in range(n)]
wave2 = [p[:4096] + torch.randint(0, 10000, (4096,), generator=g).tolist() for p in wave1]
sp = SamplingParams(temperature=0.6, max_tokens=out, ignore_eos=True)
llm.generate([wave1[0][:64]], SamplingParams(temperature=0.6, max_tokens=2, ignore_eos=True), use_tqdm=False)
for tag, wave in (("first wave", wave1), ("second wave (4096-token prefix cached)", wave2)):
    steps = [0]
    orig = llm.step
    def step():
        steps[0] += 1
        return orig()
    llm.step = step
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = llm.generate(wave, sp, use_tqdm=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    llm.step = orig
    assert all(len(r["token_ids"]) == out for r in res)
    print(f"{tag}: {n} x (8192 in, {out} out) in {dt:.3f} s, {steps[0]} steps, {n * (8192 + out) / dt:.0f} tok/s in+out", flush=True)
llm.exit()
