"""Online serving benchmark in the shape of the reference's serving_bench.py:16-167: requests with random
token-id prompts arrive as a Poisson process and are fed to `LLM.add_request` while the engine steps;
reports throughput (input + output tokens over wall time), mean TTFT, mean TPOT and mean latency.

Same knobs and seeds as the reference script (random.seed(100), numpy.random.seed(100), prompts
randint(0, 10000), temperature 0.6, ignore_eos).  `--model` is a local directory with a config.json
(synthetic random-init weights when it holds no checkpoint).  Prints the reference's report block and then
one JSON line.

    python serving_bench.py --model nano-vllm-v1_amd/model_configs/qwen3-0.6b --num-requests 1024 \
        --request-rate 10 --max-num-batched-tokens 1024 --max-num-seqs 1024 --random-input-len 128 \
        --random-output-len 100 --chunked-prefill
"""
from __future__ import annotations

import argparse
import json
import math
import sys
import time
from pathlib import Path
from random import randint, seed

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd"))


class RequestClock:
    """Timestamps of one request: scheduled arrival, first generated token, completion."""
    __slots__ = ("input_len", "arrival", "first_token", "done", "output_len")

    def __init__(self, input_len: int, arrival: float):
        self.input_len, self.arrival = input_len, arrival
        self.first_token = self.done = math.nan
        self.output_len = 0

    def saw_token(self, now: float):
        if math.isnan(self.first_token):
            self.first_token = now

    def finished(self, now: float, output_len: int):
        self.saw_token(now)
        self.done, self.output_len = now, output_len

    @property
    def ttft(self):
        return self.first_token - self.arrival

    @property
    def tpot(self):
        return (self.done - self.first_token) / (self.output_len - 1) if self.output_len > 1 else math.nan

    @property
    def latency(self):
        return self.done - self.arrival


def serve(engine, prompts, sampling_params, arrival_offsets, progress=None):
    """Drive the engine until every request has arrived and finished.  Returns ({seq_id: RequestClock}, wall s)."""
    clocks: dict[int, RequestClock] = {}
    sent, n = 0, len(prompts)
    t0 = time.perf_counter()
    while sent < n or not engine.is_finished():
        now = time.perf_counter()
        while sent < n and now - t0 >= arrival_offsets[sent]:
            engine.add_request(prompts[sent], sampling_params[sent])
            seq = engine.scheduler.waiting[-1]
            clocks[seq.seq_id] = RequestClock(len(prompts[sent]), t0 + arrival_offsets[sent])
            sent += 1
        if engine.is_finished():
            time.sleep(min(0.01, max(0.0, t0 + arrival_offsets[sent] - time.perf_counter())))
            continue
        finished, _ = engine.step()
        now = time.perf_counter()
        for seq in engine.scheduler.running:           # its first generated id has reached the host
            c = clocks.get(seq.seq_id)
            if c is not None and math.isnan(c.first_token) and len(seq) > seq.num_prompt_tokens and seq[seq.num_prompt_tokens] >= 0:
                c.saw_token(now)
        for seq_id, ids in finished:
            if seq_id in clocks:
                clocks[seq_id].finished(now, len(ids))
                if progress is not None:
                    progress.update(1)
    return clocks, time.perf_counter() - t0


def summarize(clocks, wall):
    done = [c for c in clocks.values() if not math.isnan(c.done)]
    tok_in = sum(c.input_len for c in clocks.values())
    tok_out = sum(c.output_len for c in done)
    tpots = [c.tpot for c in done if not math.isnan(c.tpot)]
    return dict(total_time_s=wall, requests=len(clocks), completed=len(done), input_tokens=tok_in, output_tokens=tok_out,
                throughput_tok_s=(tok_in + tok_out) / wall,
                mean_ttft_ms=1e3 * float(np.mean([c.ttft for c in clocks.values() if not math.isnan(c.first_token)])),
                mean_tpot_ms=1e3 * float(np.mean(tpots)) if tpots else math.nan,
                mean_latency_s=float(np.mean([c.latency for c in done])))


def main(argv=None):
    ap = argparse.ArgumentParser(description="Serving benchmark (Poisson arrivals) for the MI355X nano-vllm path.")
    ap.add_argument("--model", type=str, default=str(ROOT / "nano-vllm-v1_amd" / "model_configs" / "qwen3-0.6b"))
    ap.add_argument("--num-requests", type=int, default=256)
    ap.add_argument("--request-rate", type=float, default=8)
    ap.add_argument("--max-num-batched-tokens", type=int, default=2048)
    ap.add_argument("--max-num-seqs", type=int, default=2048)
    ap.add_argument("--max-model-len", type=int, default=40960)
    ap.add_argument("--tensor-parallel-size", type=int, default=1)
    ap.add_argument("--random-input-len", type=int, default=128)
    ap.add_argument("--random-output-len", type=int, default=128)
    ap.add_argument("--chunked-prefill", action="store_true")
    ap.add_argument("--enforce-eager", action="store_true")
    ap.add_argument("--async-scheduling", action="store_true", default=True,
                    help="the engine launches step n+1 before it reads step n's ids (the engine's default)")
    ap.add_argument("--sync-scheduling", dest="async_scheduling", action="store_false",
                    help="the reference's loop: schedule, run, read ids back, postprocess")
    ap.add_argument("--num-kvcache-blocks", type=int, default=-1, help="KV pages (default: from free memory)")
    ap.add_argument("--warmup-requests", type=int, default=50)
    args = ap.parse_args(argv)

    seed(100)
    np.random.seed(100)
    from nanovllm import LLM, SamplingParams

    print(f"\n--- Running benchmark with --num-requests {args.num_requests} --request-rate {args.request_rate} ---")
    llm = LLM(args.model, enforce_eager=args.enforce_eager, max_model_len=args.max_model_len,
              max_num_batched_tokens=args.max_num_batched_tokens, max_num_seqs=args.max_num_seqs,
              tensor_parallel_size=args.tensor_parallel_size, chunked_prefill=args.chunked_prefill,
              num_kvcache_blocks=args.num_kvcache_blocks, async_scheduling=args.async_scheduling)

    def batch(n):
        return ([[randint(0, 10000) for _ in range(args.random_input_len)] for _ in range(n)],
                [SamplingParams(temperature=0.6, ignore_eos=True, max_tokens=args.random_output_len) for _ in range(n)])

    if args.warmup_requests:
        llm.generate(*batch(args.warmup_requests), use_tqdm=False)
    prompts, sps = batch(args.num_requests)
    arrivals = np.cumsum(np.random.exponential(1.0 / args.request_rate, args.num_requests))
    before = llm.scheduler.num_preemptions
    clocks, wall = serve(llm, prompts, sps, arrivals)
    r = summarize(clocks, wall)
    r["preemptions"] = llm.scheduler.num_preemptions - before
    print("--- Benchmark Results ---")
    print(f"Total time: {r['total_time_s']:.2f}s")
    print(f"Requests sent: {r['requests']}")
    print(f"Throughput: {r['throughput_tok_s']:.2f} tokens/s")
    print(f"Average TTFT: {r['mean_ttft_ms']:.2f} ms")
    print(f"Average TPOT: {r['mean_tpot_ms']:.2f} ms")
    print(f"Average latency: {r['mean_latency_s']:.2f} s")
    print("-------------------------\n")
    r["config"] = {k: getattr(args, k) for k in ("model", "num_requests", "request_rate", "max_num_batched_tokens", "max_num_seqs",
                                                  "tensor_parallel_size", "random_input_len", "random_output_len",
                                                  "chunked_prefill", "enforce_eager", "async_scheduling")}
    print(json.dumps(r))
    llm.exit()
    return r


if __name__ == "__main__":
    main()
