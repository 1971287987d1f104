#!/usr/bin/env python3
"""Benchmark of the nano-vllm operator path on MI355X (gfx950).

Metric (BASELINE.json): decode tok/s + prefill tok/s, Qwen3-0.6B, bf16, tensor parallel over
the launched ranks; % of the HBM roofline (decode) and of the bf16 MFMA peak (prefill).

A "step" is one pass of the hot path - ModelRunner.run(): batch flattening, one H2D copy,
the forward (hipGraph replay for decode), LM head, sampler, token ids back on the host - over
B sequences that each decode one token at ~1024 tokens of context (synthetic token ids,
random-init weights of the Qwen3-0.6B architecture).  `value` = B * steps / wall time.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--context C]
  N > 1: either as typed (the script starts its N ranks itself through torch.distributed.run) or under a launcher:
         python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  Extra keys: `roofline` (dominant kernel, HIP-event timed),
`step_roofline` (all algorithmic bytes of a step / step time), `prefill` (1024-token chunks),
`decode_sweep` (other batch sizes), `cpu_baseline` (the CPU oracle on the host cores).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd"))
sys.path.insert(0, str(ROOT))

# HIP IPC between the rank processes (the tensor-parallel exchange) needs the runtime's legacy IPC mode off on
# dmabuf-only hosts; it must be in the environment before the first HIP call and is inherited by child ranks.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def launch_ranks_if_needed(argv):
    """`python bench.py --gpus N` typed as is (no launcher, WORLD_SIZE unset): start the N ranks as child processes
    through torch.distributed.run - the same command line a launcher would use - BEFORE this process touches the
    GPU, relay their output (rank 0 prints the JSON line) and exit with the launcher's status.  The reference starts
    its own ranks as well (engine/llm_engine.py:101-111)."""
    if "WORLD_SIZE" in os.environ:
        return
    n = 1
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    if n <= 1:
        return
    import socket
    import subprocess
    with socket.socket() as sock:                      # a free rendezvous port
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + list(argv)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "4"))
    sys.exit(subprocess.run(cmd, env=env).returncode)


if __name__ == "__main__":
    launch_ranks_if_needed(sys.argv[1:])

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak
MFMA_PEAK_TFLOPS = 2500.0    # dense bf16 spec peak
MODEL_DIR = ROOT / "nano-vllm-v1_amd" / "model_configs" / "qwen3-0.6b"
PAGE = 256


def algorithmic_bytes_per_decode_step(hf, contexts, tp=1):
    """Weights (every matrix once, LM head included) + KV rows of every context token, bf16."""
    d = hf.head_dim
    p_layer = hf.hidden_size * (hf.num_attention_heads + 2 * hf.num_key_value_heads) * d \
        + hf.num_attention_heads * d * hf.hidden_size + 3 * hf.hidden_size * hf.intermediate_size
    weights = 2 * (hf.num_hidden_layers * p_layer + hf.vocab_size * hf.hidden_size) // tp
    kv_per_token = 2 * hf.num_hidden_layers * hf.num_key_value_heads * d * 2 // tp
    return weights + kv_per_token * int(sum(contexts)), weights, kv_per_token


def algorithmic_flops_per_chunk(hf, t, cached, tp=1):
    d = hf.head_dim
    p_layer = hf.hidden_size * (hf.num_attention_heads + 2 * hf.num_key_value_heads) * d \
        + hf.num_attention_heads * d * hf.hidden_size + 3 * hf.hidden_size * hf.intermediate_size
    linear = t * 2 * hf.num_hidden_layers * p_layer
    visible = t * cached + t * (t + 1) // 2
    attn = 4 * hf.num_hidden_layers * hf.num_attention_heads * d * visible
    return (linear + attn) / tp


class Driver:
    """Minimal stand-in for the scheduler: sequential page allocation, one list of live sequences."""

    def __init__(self, runner):
        from nanovllm.engine.sequence import Sequence
        self.Sequence = Sequence
        self.runner = runner
        self.next_page = 0
        self.num_pages = runner.config.num_kvcache_blocks

    def reset(self):
        self.next_page = 0

    def new_seq(self, tokens, temperature=0.6):
        from nanovllm.sampling_params import SamplingParams
        return self.Sequence(tokens, SamplingParams(temperature=temperature, max_tokens=1 << 30, ignore_eos=True))

    def grow(self, s, ctx):
        while len(s.block_table) * PAGE < ctx:
            assert self.next_page < self.num_pages, "KV pool exhausted"
            s.block_table.append(self.next_page)
            self.next_page += 1

    def prefill(self, seqs, chunk, ahead=False):
        """Chunked prefill, one sequence at a time, `chunk` tokens per step; returns (seconds, tokens, cached) per chunk.
        `ahead`: the engine loop that runs one step ahead (Config.async_scheduling) - a chunk is launched before the
        previous one has finished, its time is the interval between the completions of consecutive chunks."""
        from nanovllm.engine.sequence import PLACEHOLDER
        times = []
        if ahead:
            inflight = None
            t_prev = None

            def finish(entry):
                nonlocal t_prev
                tokens, s, pos, n, cached = entry
                toks = tokens.result()                               # waits for that chunk
                now = time.perf_counter()
                if t_prev is not None:
                    times.append((now - t_prev, n, cached))
                t_prev = now
                if toks:
                    s.token_ids[pos] = toks[0]
                    s.last_token = s.token_ids[-1]

            for s in seqs:
                prompt_len = len(s)
                while s.num_cached_tokens < prompt_len:
                    n = min(chunk, prompt_len - s.num_cached_tokens)
                    cached = s.num_cached_tokens
                    s.num_new_tokens = n
                    self.grow(s, cached + n)
                    tokens = self.runner.run_async([s])
                    s.num_cached_tokens += n
                    s.num_new_tokens = 0
                    pos = -1
                    if tokens.need:
                        s.append_token(PLACEHOLDER)
                        s.pending_slot = 0
                        pos = len(s) - 1
                    if inflight is not None:
                        finish(inflight)
                    inflight = (tokens, s, pos, n, cached)
            if inflight is not None:
                finish(inflight)
            return times
        for s in seqs:
            prompt_len = len(s)
            while s.num_cached_tokens < prompt_len:
                n = min(chunk, prompt_len - s.num_cached_tokens)
                s.num_new_tokens = n
                self.grow(s, s.num_cached_tokens + n)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                toks, need = self.runner.run([s])
                torch.cuda.synchronize()
                times.append((time.perf_counter() - t0, n, s.num_cached_tokens))
                toks = self.share_tokens(toks, int(need.numel()))
                s.num_cached_tokens += n
                s.num_new_tokens = 0
                if toks:
                    s.append_token(toks[0])
        return times

    @staticmethod
    def share_tokens(toks, n):
        """Only rank 0 samples; every rank must feed the same ids to the next step (the engine's control
        plane does this with the scheduler's sequences; here one small broadcast stands in for it)."""
        if n and dist.is_initialized() and dist.get_world_size() > 1:
            t = torch.tensor(toks if toks is not None else [0] * n, dtype=torch.int64, device=collective_device())
            dist.broadcast(t, 0)
            toks = t.tolist()
        return toks

    def schedule_decode(self, seqs):
        for s in seqs:
            s.num_new_tokens = 1
            if len(s.block_table) * PAGE <= s.num_cached_tokens:     # the new token starts a page
                self.grow(s, s.num_cached_tokens + 1)

    def decode_steps_ahead(self, seqs, steps):
        """`steps` decode steps the way the engine runs them with Config.async_scheduling (LLMEngine._step_ahead):
        step n+1 is scheduled and launched - its input ids fed from step n's sampler output on the device - before
        step n's ids are read back.  Same launches, same ids as `decode_step`; the host work hides behind the GPU."""
        from nanovllm.engine.sequence import PLACEHOLDER
        inflight = None

        def fill(entry):
            tokens, positions = entry
            for s, pos, t in zip(seqs, positions, tokens.result()):
                s.token_ids[pos] = t
                s.last_token = s.token_ids[-1]

        for _ in range(steps):
            self.schedule_decode(seqs)
            tokens = self.runner.run_async(seqs)
            positions = []
            for row, s in enumerate(seqs):
                s.num_cached_tokens += 1
                s.num_new_tokens = 0
                s.append_token(PLACEHOLDER)
                s.pending_slot = row
                positions.append(len(s) - 1)
            if inflight is not None:
                fill(inflight)
            inflight = (tokens, positions)
        if inflight is not None:
            fill(inflight)

    def decode_step(self, seqs):
        self.schedule_decode(seqs)
        toks, need = self.runner.run(seqs)
        toks = self.share_tokens(toks, len(seqs))
        for s, t in zip(seqs, toks):                                  # ids arrive as Python ints
            s.num_cached_tokens += 1
            s.num_new_tokens = 0
            s.append_token(t)


def collective_device():
    """RCCL moves device tensors; the gloo rehearsal (ranks sharing one GPU) gets host tensors."""
    return "cuda" if dist.get_backend() == "nccl" else "cpu"


def note(msg):
    """Progress marker on stderr (NANOVLLM_BENCH_VERBOSE=1): tells where a multi-rank run stopped."""
    if os.environ.get("NANOVLLM_BENCH_VERBOSE") == "1":
        print(f"[bench rank {os.environ.get('RANK', '0')}] {msg}", file=sys.stderr, flush=True)


def sync_all(distributed, runner=None):
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    if runner is not None and getattr(runner, "comm", None) is not None:
        runner.comm.check()                     # raises if an exchange timed out instead of reporting a bogus time


def time_decode(driver, seqs, steps, warmup, distributed, ahead=False):
    """Seconds for `steps` decode steps (max over the ranks).  `ahead`: the engine loop that runs one step ahead of
    its sampled ids; its last step's ids are read back inside the timed region."""
    for _ in range(warmup):
        driver.decode_step(seqs)
    sync_all(distributed, driver.runner)
    t0 = time.perf_counter()
    if ahead:
        driver.decode_steps_ahead(seqs, steps)
    else:
        for _ in range(steps):
            driver.decode_step(seqs)
    sync_all(distributed, driver.runner)
    dt = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([dt], device=collective_device())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def kernel_roofline(runner, seqs, iters):
    """Average duration of the dominant decode kernel exactly as the step runs it: attn_decode_kernel<G, fused> -
    q/k norm + RoPE + KV store of the new token + paged attention straight from the raw qkv rows
    (nvh_attn_decode_fused_qkv, the instance the captured graph replays) - timed with HIP events on the launch
    stream; every launch reads a different layer's cache, as a real step does.  Run after everything else: the
    launches store synthetic K/V rows for the newest token of every sequence."""
    from nanovllm import ops
    from nanovllm.engine.model_runner import flatten_batch
    hf = runner.config.hf_config
    tp = runner.world_size
    fb = flatten_batch(seqs)
    ctx = torch.from_numpy(fb["context_lens"]).cuda()
    tables = torch.from_numpy(fb["block_tables"]).cuda()
    positions = (ctx.long() - 1).clamp(min=0)
    b = len(seqs)
    hq, hkv, d = hf.num_attention_heads // tp, hf.num_key_value_heads // tp, hf.head_dim
    qkv = torch.randn(b, (hq + 2 * hkv) * d, device="cuda").to(torch.bfloat16)
    attn0 = runner.model.model.layers[0].self_attn
    cos_sin = attn0.rotary_emb.table2d()
    splits = runner.decode_splits(b)
    layers = hf.num_hidden_layers
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(layers)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(layers)]
    total_ms, n = 0.0, 0
    for it in range(iters + 1):
        for li in range(layers):
            starts[li].record()
            ops.attn_decode_fused_qkv(qkv, hq, hkv, attn0.q_norm.weight, attn0.k_norm.weight, attn0.q_norm.eps, positions, cos_sin,
                                      runner.kv_cache[0, li], runner.kv_cache[1, li], ctx, tables, d ** -0.5, num_splits=splits)
            ends[li].record()
        torch.cuda.synchronize()
        if it:                              # first sweep is warm-up
            total_ms += sum(s.elapsed_time(e) for s, e in zip(starts, ends))
            n += layers
    avg_ms = total_ms / n
    bytes_per_launch = int(fb["context_lens"].sum()) * hkv * d * 2 * 2      # K and V rows, bf16
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
    # HBM bytes per launch from the PMC pass committed under profiles/ (FETCH_SIZE, own rocprofv3 run,
    # doubled as gfx950 requires), scaled from that run's context length to this one's.
    traffic, traffic_note = None, None
    pmc = ROOT / "profiles" / "r03_attn_decode_traffic.json"
    if not pmc.exists():
        pmc = ROOT / "profiles" / "r01_attn_decode_traffic.json"
    if pmc.exists() and tp == 1 and b == 256:
        rec = json.loads(pmc.read_text())
        ratio = rec["traffic_bytes_per_launch"] / rec.get("algorithmic_bytes_per_launch", rec.get("algorithmic_bytes_per_launch_at_ctx1027"))
        traffic = int(bytes_per_launch * ratio)
        traffic_note = f"profiles/{pmc.name}: measured/algorithmic = {ratio:.4f}"
    g = hq // hkv
    return dict(bound="hbm", kernel=f"attn_decode_kernel<{g}, true>" + (" + merge" if splits > 1 else ""),
                entry_point="nvh_attn_decode_fused_qkv (the launch the decode graph replays)", achieved=round(achieved, 1),
                peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic,
                traffic_source=traffic_note, avg_launch_us=round(avg_ms * 1e3, 2), bytes_per_launch=bytes_per_launch,
                launches_timed=n)


@torch.inference_mode()
def prefill_kernel_roofline(hf, tokens=1024, iters=3):
    """The kernels of one 1024-token prefill chunk (no cached prefix), each timed with HIP events on the launch stream
    over rotating weight copies (a layer's weights are cold when its kernels run): flops, time and fraction of the bf16
    MFMA peak per kernel, and which of them takes the largest share of the chunk."""
    from nanovllm import ops
    h, d = hf.hidden_size, hf.head_dim
    hq, hkv, inter = hf.num_attention_heads, hf.num_key_value_heads, hf.intermediate_size
    copies = 12
    bf = torch.bfloat16

    def w(n, k):
        return [(torch.randn(n, k, device="cuda") * 0.02).to(bf) for _ in range(copies)]

    x = torch.randn(tokens, h, device="cuda").to(bf)
    xa = torch.randn(tokens, hq * d, device="cuda").to(bf)
    xi = torch.randn(tokens, inter, device="cuda").to(bf)
    wq, wo, wd = w((hq + 2 * hkv) * d, h), w(h, hq * d), w(h, inter)
    wg = [ops.interleave_gate_up(t) for t in w(2 * inter, h)]
    q = torch.randn(tokens, hq, d, device="cuda").to(bf)
    kv = [torch.randn(2, tokens, hkv, d, device="cuda").to(bf) for _ in range(copies)]
    cu = torch.tensor([0, tokens], dtype=torch.int32, device="cuda")
    cases = {
        "qkv_proj gemm": (lambda i: ops.linear(x, wq[i]), 2 * tokens * h * (hq + 2 * hkv) * d),
        "o_proj gemm": (lambda i: ops.linear(xa, wo[i]), 2 * tokens * h * hq * d),
        "gate_up gemm + SwiGLU": (lambda i: ops.linear_swiglu(x, wg[i]), 2 * tokens * h * 2 * inter),
        "down_proj gemm": (lambda i: ops.linear(xi, wd[i]), 2 * tokens * h * inter),
        "attn_varlen_kernel": (lambda i: ops.attn_varlen(q, kv[i][0], kv[i][1], cu, cu, tokens, d ** -0.5), 4 * hq * d * tokens * (tokens + 1) // 2),
    }
    out, total_us = {}, 0.0
    for name, (fn, flops) in cases.items():
        # `copies` launches captured into one hipGraph (no host launch cost between them), replayed between two events
        for i in range(copies):
            fn(i)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for i in range(copies):
                fn(i)
        graph.replay()
        best = 1e30
        for _ in range(iters):
            s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0.record()
            graph.replay()
            e0.record()
            torch.cuda.synchronize()
            best = min(best, s0.elapsed_time(e0) * 1e3 / copies)
        us = best
        total_us += us
        out[name] = dict(us=round(us, 2), tflops=round(flops / us / 1e6, 1), frac=round(flops / us / 1e6 / MFMA_PEAK_TFLOPS, 4))
    dominant = max(out, key=lambda k: out[k]["us"])
    # matrix-pipe busy fraction of the kernels of a chunk as the step launches them (counter pass, tools/evidence_json.py)
    pmc = ROOT / "profiles" / "r03_prefill_mfma_busy.json"
    busy = {k: v["mfma_busy_frac"] for k, v in json.loads(pmc.read_text()).items()} if pmc.exists() else None
    return dict(bound="mfma", kernel=dominant, achieved=out[dominant]["tflops"], peak=MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                frac=out[dominant]["frac"], per_kernel=out, per_layer_us=round(total_us, 1),
                mfma_busy_by_step_kernel=busy, mfma_busy_source=(f"profiles/{pmc.name}" if pmc.exists() else None),
                note="HIP-event time per launch over 12 back-to-back launches replayed from a hipGraph (rotating cold weights); one layer = these five + two add/RMSNorm launches")


def cpu_baseline(hf):
    """The CPU oracle (oracle/ref_model.py, torch-CPU eager) on the same architecture and weight recipe, driven through
    the scheduler and page manager like the engine drives the GPU runner: decode at B = 1 and B = 8 (128-token contexts)
    and one 1024-token prefill chunk; wall clock, median of 5 after one warm-up each."""
    from types import SimpleNamespace
    from oracle import ref_model as M
    from nanovllm.engine.scheduler import Scheduler
    from nanovllm.engine.sequence import Sequence
    from nanovllm.sampling_params import SamplingParams
    torch.manual_seed(0)
    shape = M.ModelShape(hidden_size=hf.hidden_size, num_hidden_layers=hf.num_hidden_layers,
                         num_attention_heads=hf.num_attention_heads, num_key_value_heads=hf.num_key_value_heads,
                         head_dim=hf.head_dim, intermediate_size=hf.intermediate_size, vocab_size=hf.vocab_size,
                         max_position_embeddings=4096, tie_word_embeddings=True)
    w = M.random_weights(shape, seed=0)
    pages = 16
    runner = M.RefRunner(shape, w, num_blocks=pages)
    g = torch.Generator().manual_seed(0)

    def engine():
        return Scheduler(SimpleNamespace(chunked_prefill=True, max_model_len=4096, max_num_seqs=8, max_num_batched_tokens=1024,
                                         eos=-1, num_kvcache_blocks=pages, kvcache_block_size=PAGE))

    def step(sched):
        seqs = sched.schedule()
        t0 = time.perf_counter()
        logits, need = runner.run(seqs)
        dt = time.perf_counter() - t0
        sched.postprocess(seqs, logits.float().argmax(-1).tolist() if need.numel() else [], need)
        return dt, len(seqs)

    def median(xs):
        return sorted(xs)[len(xs) // 2]

    def decode(b, ctx=128):
        sched = engine()
        for _ in range(b):
            sched.add(Sequence(torch.randint(0, 10000, (ctx,), generator=g).tolist(),
                               SamplingParams(temperature=0.6, max_tokens=64, ignore_eos=True)))
        step(sched)                                           # prefill of all b prompts (one step: b * 128 <= 1024)
        step(sched)                                           # warm-up decode step
        return b / median([step(sched)[0] for _ in range(5)])

    def prefill(tokens=1024):
        times = []
        for _ in range(6):
            sched = engine()
            sched.add(Sequence(torch.randint(0, 10000, (tokens + 8,), generator=g).tolist(),
                               SamplingParams(temperature=0.6, max_tokens=8, ignore_eos=True)))
            times.append(step(sched)[0])                      # the first chunk: exactly `tokens` new tokens, no prefix
        return tokens / median(times[1:])

    t0 = time.perf_counter()
    b8, b1, pf = decode(8), decode(1), prefill()
    return dict(value=round(b8, 2), unit="tok/s", cores=torch.get_num_threads(), kind="port",
                sample=f"Qwen3-0.6B shape through the scheduler (oracle/ref_model.py, torch-CPU eager bf16), median of 5 steps after "
                       f"a warm-up: decode B=8 ctx=128 (= value), decode B=1 ctx=128, one 1024-token prefill chunk; {time.perf_counter() - t0:.0f}s in all",
                decode_b8_tok_s=round(b8, 2), decode_b1_tok_s=round(b1, 2), prefill_chunk_tok_s=round(pf, 1),
                host_cpus=os.cpu_count())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--context", type=int, default=1024)
    ap.add_argument("--no-extras", action="store_true", help="skip sweeps, long prefill and the CPU baseline")
    ap.add_argument("--model", default="qwen3-0.6b", help="directory name under nano-vllm-v1_amd/model_configs (default: the headline model)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if args.gpus != world:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started {world} ranks")
    # Rehearsal on a one-GPU box: NANOVLLM_BENCH_SHARE_DEVICE=1 puts every rank on cuda:0 with gloo as the
    # control plane (RCCL refuses two ranks on one device); the data path is the xGMI exchange either way.
    share_device = os.environ.get("NANOVLLM_BENCH_SHARE_DEVICE", "0") == "1"
    if share_device:
        local_rank = 0
        os.environ["LOCAL_RANK"] = "0"
    torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from nanovllm.config import Config
    from nanovllm.engine.model_runner import ModelRunner

    b, ctx0 = args.batch, args.context
    total_steps = args.steps + args.warmup
    max_len = 8192 + 64
    pages_decode = b * ((ctx0 + total_steps + 64 + PAGE) // PAGE + 3)          # room for the comparison runs and sweeps too
    pages = max(pages_decode, 8 * (max_len // PAGE + 1)) + 16
    model_dir = MODEL_DIR.parent / args.model
    model_name = args.model.replace("qwen3", "Qwen3").replace("b", "B")
    cfg = Config(str(model_dir), max_num_batched_tokens=1024, max_num_seqs=max(b, 8), max_model_len=max_len,
                 tensor_parallel_size=world, enforce_eager=os.environ.get("NANOVLLM_BENCH_EAGER", "0") == "1", chunked_prefill=True, num_kvcache_blocks=pages)
    hf = cfg.hf_config
    t_init = time.perf_counter()
    note("building the runner")
    runner = ModelRunner(cfg, rank, [])
    init_s = time.perf_counter() - t_init
    note("runner ready")
    driver = Driver(runner)
    g = torch.Generator().manual_seed(0)

    def prompts(n, length):
        return [torch.randint(0, 10000, (length,), generator=g).tolist() for _ in range(n)]

    # ---- prefill of the decode batch: B prompts x context tokens, one 1024-token chunk per step
    # The loop that runs one step ahead of its sampled ids (Config.async_scheduling, the engine's default): what a plain
    # LLM(model) does; the synchronous loop of the reference is timed next to it for the decode steps.  With
    # several ranks every rank runs the same schedule and the ids reach all of them on the device (comm.broadcast_ids).
    ahead = os.environ.get("NANOVLLM_BENCH_SYNC", "0") != "1"
    seqs = [driver.new_seq(p) for p in prompts(b, ctx0)]
    ptimes = driver.prefill(seqs, 1024, ahead)
    note("prefill done")
    steady = [x for x in ptimes[len(ptimes) // 4:]] or ptimes
    if steady:
        chunk_s = sum(x[0] for x in steady) / len(steady)
        chunk_tokens = sum(x[1] for x in steady) / len(steady)
        chunk_flops = sum(algorithmic_flops_per_chunk(hf, x[1], x[2], world) for x in steady) / len(steady)
        prefill = dict(tok_s=round(chunk_tokens / chunk_s, 1), ms_per_chunk=round(chunk_s * 1e3, 3),
                       chunk_tokens=int(chunk_tokens), chunks_timed=len(steady),
                       roofline=dict(bound="mfma", achieved=round(chunk_flops / chunk_s / 1e12, 2), peak=MFMA_PEAK_TFLOPS,
                                     unit="TFLOP/s", frac=round(chunk_flops / chunk_s / 1e12 / MFMA_PEAK_TFLOPS, 4)),
                       steps_in_flight=runner.LANES if (ahead and runner.step_overlap) else 1,
                       note="one sequence per step, T=chunk_tokens new tokens, no cached prefix; eager launches, host included"
                            + (" (engine one step ahead: a chunk's time is the interval between consecutive completions"
                               + ("; chunks are launched up to steps_in_flight ahead, each on its own stream, ordered against its predecessor per layer" if runner.step_overlap else "")
                               + ")" if ahead else ""))
    else:
        prefill = dict(note="fewer than two chunks in the prefill of this batch: not timed")

    # ---- the timed region: K decode steps of B sequences
    dt = time_decode(driver, seqs, args.steps, args.warmup, distributed, ahead=ahead)
    note("decode timed")
    value = b * args.steps / dt
    contexts = [s.num_cached_tokens - args.steps / 2 for s in seqs]          # mean context over the timed steps
    step_bytes, w_bytes, kv_tok = algorithmic_bytes_per_decode_step(hf, contexts, world)
    step_gbs = step_bytes / (dt / args.steps) / 1e9
    out = dict(metric=f"decode tok/s ({model_name}, bf16; prefill tok/s and roofline fractions in `prefill`/`roofline`)",
               value=round(value, 1), unit="tok/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
               ms_per_step=round(dt / args.steps * 1e3, 4), higher_is_better=True, scaling="strong", vs_baseline=None,
               dtype="bf16", data=f"synthetic token ids, random-init weights ({model_name} architecture)",
               config=dict(workload=f"{model_name} TP={world} bf16 decode, B={b} sequences x ~{ctx0} context tokens, "
                                    "chunked_prefill on, max_num_batched_tokens=1024, hipGraph decode",
                           batch=b, context=ctx0, parallelism=f"tp{world}", kv_page=PAGE,
                           engine_loop=("one step ahead (ModelRunner.run_async; Config.async_scheduling, the engine's default): step n+1 is launched before step n's ids are read back"
                                        if ahead else "synchronous (schedule, run, read ids back, postprocess)")),
               step_roofline=dict(bound="hbm", achieved=round(step_gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                                  frac=round(step_gbs / HBM_PEAK_GBS, 4), bytes_per_step=int(step_bytes),
                                  weight_bytes=int(w_bytes), kv_bytes_per_context_token=int(kv_tok)),
               prefill=prefill, init_s=round(init_s, 1))
    if ahead:
        d_sync = time_decode(driver, seqs, args.steps, 2, distributed)
        out["synchronous_loop"] = dict(tok_s=round(b * args.steps / d_sync, 1), ms_per_step=round(d_sync / args.steps * 1e3, 4),
                                       note="same steps with the ids read back before the next step is scheduled (contexts one timed region longer)")
    if distributed:
        from nanovllm import comm as comm_mod
        out["tp_exchange"] = ("xgmi peer all-reduce (one kernel per projection, inside the native step and its hipGraph)"
                              if runner.comm is not None else
                              (f"torch.distributed ({dist.get_backend()}) sums between the parts of the native step"
                               if runner.native is not None else f"torch.distributed ({dist.get_backend()}) between operator modules"))
        out["comm_fallback_reason"] = comm_mod.fallback_reason      # None: the peer exchange passed its start-up check
        out["decode_graphs"] = bool(runner.graphs)
        out["native_executor"] = runner.native is not None
        if share_device:
            out["config"]["rehearsal"] = "all ranks share cuda:0; not a scaling measurement"
    def extras():
        # ---- decode at other batch sizes (same sequences, subsets)
        sweep = {}
        for bb in (1, 8, 32, 128):
            if bb >= b:
                continue
            sub = seqs[:bb]
            k = 32
            d2 = time_decode(driver, sub, k, 4, distributed, ahead=ahead)
            sb, _, _ = algorithmic_bytes_per_decode_step(hf, [s.num_cached_tokens - k / 2 for s in sub], world)
            note(f"sweep B={bb} done")
            sweep[f"B{bb}"] = dict(tok_s=round(bb * k / d2, 1), ms_per_step=round(d2 / k * 1e3, 4),
                                   hbm_frac=round(sb / (d2 / k) / 1e9 / HBM_PEAK_GBS, 4))
            if ahead:
                sweep[f"B{bb}"]["ms_per_step_synchronous_loop"] = round(time_decode(driver, sub, k, 2, distributed) / k * 1e3, 4)
        out["decode_sweep"] = sweep
        # ---- long-context chunked prefill: 8192-token prompts, 8 chunks of 1024 each
        driver_pages = driver.next_page
        long_seqs = [driver.new_seq(p) for p in prompts(2, 8192)]
        lt = driver.prefill(long_seqs, 1024, ahead)[-8:]                         # second prompt (first warms up)
        tot_s = sum(x[0] for x in lt)
        tot_f = sum(algorithmic_flops_per_chunk(hf, x[1], x[2], world) for x in lt)
        out["prefill_8k"] = dict(tok_s=round(sum(x[1] for x in lt) / tot_s, 1), tflops=round(tot_f / tot_s / 1e12, 2),
                                 mfma_frac=round(tot_f / tot_s / 1e12 / MFMA_PEAK_TFLOPS, 4),
                                 ms_per_chunk=[round(x[0] * 1e3, 3) for x in lt])
        # ---- the serving shape of a prefill step: 8 prompts x 128 tokens in one 1024-token step
        times = []
        for it in range(12):
            driver.reset()
            batch = [driver.new_seq(p) for p in prompts(8, 128)]
            for s in batch:
                s.num_new_tokens = len(s)
                driver.grow(s, len(s))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            toks, need = runner.run(batch)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
            driver.share_tokens(toks, int(need.numel()))
        t8 = sorted(times[2:])[len(times[2:]) // 2]
        f8 = 8 * algorithmic_flops_per_chunk(hf, 128, 0, world)
        out["prefill_8x128"] = dict(tok_s=round(1024 / t8, 1), ms_per_step=round(t8 * 1e3, 3), tflops=round(f8 / t8 / 1e12, 2),
                                    mfma_frac=round(f8 / t8 / 1e12 / MFMA_PEAK_TFLOPS, 4),
                                    note="8 prompts of 128 tokens in one step, logits + sampling for all 8 included; median of 10")
        # ---- decode at other context lengths.  The sequences are declared cached up to their last token, so the
        # attention streams pages left behind by the runs above: same bytes and launches as a real batch, no
        # meaningful ids (timing only).
        ctx_sweep = {}
        # The third case is uneven: one long context among short ones (the decode attention runs over a work list).
        for bb, cc, first in ((min(b, 256), 128, 0), (min(b, 32), 8192, 0), (min(b, 256), 128, 8192)):
            k = 32
            driver.reset()
            lens = [first or cc] + [cc] * (bb - 1)
            if sum((n + k + 8) // PAGE + 1 for n in lens) > driver.num_pages:
                continue
            fake = [driver.new_seq(p) for n in lens for p in prompts(1, n)]
            for s, n in zip(fake, lens):
                s.num_cached_tokens = n - 1
                driver.grow(s, n)
            d2 = time_decode(driver, fake, k, 4, distributed)
            sb, _, _ = algorithmic_bytes_per_decode_step(hf, [s.num_cached_tokens - k / 2 for s in fake], world)
            tag = f"B{bb}_ctx{cc}" + (f"_one{first}" if first else "")
            note(f"context sweep {tag} done")
            ctx_sweep[tag] = dict(tok_s=round(bb * k / d2, 1), ms_per_step=round(d2 / k * 1e3, 4),
                                  hbm_frac=round(sb / (d2 / k) / 1e9 / HBM_PEAK_GBS, 4))
        out["decode_context_sweep"] = ctx_sweep

    if not args.no_extras:
        # the headline above is complete; a failure in the secondary measurements must not lose it (every rank runs the
        # same code on the same shapes, so an error is raised on all of them alike)
        try:
            extras()
        except Exception as e:  # noqa: BLE001
            out["extras_error"] = f"{type(e).__name__}: {e}"
            note(f"extras failed: {out['extras_error']}")
        if world == 1 and args.model == "qwen3-0.6b":
            out["prefill"]["roofline"]["kernel"] = prefill_kernel_roofline(hf)
        if rank == 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(hf)
    out["roofline"] = kernel_roofline(runner, seqs, iters=4)     # last: it overwrites the newest K/V row of every sequence
    if rank == 0:
        print(json.dumps(out))
    runner.exit()
    if distributed and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
