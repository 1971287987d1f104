#!/usr/bin/env python3
"""The reference's offline throughput workload (its bench.py:9-28) through this package's `LLM.generate`:
256 requests, prompt length ~U[100, 1024] of token ids ~U[0, 10000], output length ~U[100, 1024], `random.seed(0)`,
temperature 0.6, ignore_eos, max_model_len 4096, decode graphs on, every other engine setting at its default (16384
batched tokens, 512 sequences, no chunked prefill, KV pool sized from free memory).  Synthetic weights of the named
architecture (there are no checkpoints offline).  Prints the reference script's line (output tokens / wall time) and a
JSON line with input + output tokens per second as well.

usage: offline_bench.py [--model qwen3-0.6b] [--num-seqs 256] [--sync-scheduling] [--num-kvcache-blocks N]"""
import argparse
import json
import sys
import time
from pathlib import Path
from random import randint, seed

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd"))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen3-0.6b", help="directory name under nano-vllm-v1_amd/model_configs")
    ap.add_argument("--num-seqs", type=int, default=256)
    ap.add_argument("--max-input-len", type=int, default=1024)
    ap.add_argument("--max-output-len", type=int, default=1024)
    ap.add_argument("--sync-scheduling", action="store_true", help="the reference's loop instead of the engine's default (one step ahead)")
    ap.add_argument("--num-kvcache-blocks", type=int, default=-1, help="-1: sized from free memory, as the reference does")
    args = ap.parse_args(argv)
    from nanovllm import LLM, SamplingParams

    seed(0)
    model_dir = ROOT / "nano-vllm-v1_amd" / "model_configs" / args.model
    t0 = time.time()
    llm = LLM(str(model_dir), enforce_eager=False, max_model_len=4096, async_scheduling=not args.sync_scheduling,
              num_kvcache_blocks=args.num_kvcache_blocks)
    init_s = time.time() - t0
    prompt_token_ids = [[randint(0, 10000) for _ in range(randint(100, args.max_input_len))] for _ in range(args.num_seqs)]
    sampling_params = [SamplingParams(temperature=0.6, ignore_eos=True, max_tokens=randint(100, args.max_output_len))
                       for _ in range(args.num_seqs)]
    llm.generate([[1, 2, 3, 4]], SamplingParams(), use_tqdm=False)          # the reference warms up with one short request
    t = time.time()
    outs = llm.generate(prompt_token_ids, sampling_params, use_tqdm=False)
    t = time.time() - t
    assert [len(o["token_ids"]) for o in outs] == [sp.max_tokens for sp in sampling_params]
    total_tokens = sum(sp.max_tokens for sp in sampling_params)
    input_tokens = sum(len(p) for p in prompt_token_ids)
    print(f"Total: {total_tokens}tok, Time: {t:.2f}s, Throughput: {total_tokens / t:.2f}tok/s")
    print(json.dumps(dict(workload="reference bench.py shape: 256 requests, in/out ~U[100,1024], seed 0, LLM.generate", model=args.model,
                          requests=args.num_seqs, input_tokens=input_tokens, output_tokens=total_tokens, time_s=round(t, 3),
                          output_tok_s=round(total_tokens / t, 1), total_tok_s=round((total_tokens + input_tokens) / t, 1),
                          engine_loop="synchronous" if args.sync_scheduling else "one step ahead (default)",
                          kv_pages=llm.config.num_kvcache_blocks, init_s=round(init_s, 1))))
    llm.exit()


if __name__ == "__main__":
    main()
