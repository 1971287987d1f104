"""Worker of tests/test_tp_gpu.py: one tensor-parallel rank of the tiny model.  All ranks share cuda:0 (one-GPU
test box); gloo on 127.0.0.1 is the control plane, the xGMI exchange carries the data.  Rank 0 checks the
logits of a prefill + decode run against the single-rank oracle."""
import os
import sys
from pathlib import Path

import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd"))
sys.path.insert(0, str(ROOT / "tests"))


def rccl_capture_check(port):
    """One rank, RCCL process group of size one: the native step in parts with dist.all_reduce between them must run
    eagerly and inside a captured hipGraph with the same bits, and agree with the one-call step to rounding (the
    one-call step may hand split-K slabs to its norms, the parts never do)."""
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0, device_id=torch.device("cuda", 0))
    from conftest import TINY
    from test_model_gpu import TINY_DIR, hf_named, seq
    from oracle import ref_model as M
    from nanovllm.config import Config
    from nanovllm.engine.model_runner import ModelRunner, flatten_batch
    from nanovllm.native import MODE_DECODE, NativeQwen3
    cfg = Config(TINY_DIR, max_num_batched_tokens=1024, max_num_seqs=8, max_model_len=2048, enforce_eager=True,
                 num_kvcache_blocks=12, chunked_prefill=True)
    runner = ModelRunner(cfg, 0, [])
    shape = M.ModelShape(**TINY)
    runner.load_weights(hf_named(M.random_weights(shape, seed=0, std=0.05), shape))
    g = torch.Generator().manual_seed(31)
    prompts = [torch.randint(0, 1024, (n,), generator=g).tolist() for n in (5, 300, 257)]
    tables = [[0], [1, 2], [3, 4]]
    runner.forward_logits([seq(p, 0, len(p), t) for p, t in zip(prompts, tables)])       # fills the cache
    calls = []
    parts = NativeQwen3(runner.model, reduce_fn=lambda t: (calls.append(t.numel()), dist.all_reduce(t))[1])
    fb = flatten_batch([seq(p + [7], len(p), 1, t) for p, t in zip(prompts, tables)])
    dev = {k: torch.from_numpy(fb[k]).cuda() for k in ("input_ids", "positions", "slot_mapping", "context_lens", "block_tables")}

    def step(native):
        return native.forward(dev["input_ids"], dev["positions"], mode=MODE_DECODE, batch=3, slot_mapping=dev["slot_mapping"],
                              context_lens=dev["context_lens"], block_tables=dev["block_tables"], decode_splits=2)
    whole = step(runner.native)
    eager = step(parts)
    assert len(calls) == 2 * shape.num_hidden_layers + 1, calls       # one sum after the embedding, two per layer
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        captured = step(parts)
    captured.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(captured, eager), "captured parts + RCCL sums differ from the eager ones"
    err = (eager.float() - whole.float()).abs()
    assert bool((err <= 3e-2 + 2e-2 * whole.float().abs()).all()), float(err.max())
    runner.exit()
    dist.destroy_process_group()
    print("rank 0 OK")


def check_one_step_ahead(runner, seqs, tables, rank, world):
    """Two steps launched back to back with run_async on every rank: the second takes its input ids from the first
    one's sampler output on the device (rank 0 samples; the ids reach the other ranks through the exchange).  Every
    rank must read back the same ids, and the second step's ids must be what a synchronous step computes from the
    first step's ids (near-greedy temperature)."""
    from nanovllm.engine.sequence import PLACEHOLDER
    for s in seqs:
        s.temperature = 1e-9

    def advance(first):
        for i, s in enumerate(seqs):
            s.num_cached_tokens += s.num_new_tokens
            s.append_token(PLACEHOLDER)
            s.pending_slot = i
            s.num_new_tokens = 1
            s.block_table = tables[i][:(len(s) + 255) // 256]

    p1 = runner.run_async(seqs)
    advance(True)
    p2 = runner.run_async(seqs)                      # input ids: placeholders, fed from p1's ids on the device
    ids1, ids2 = p1.result(), p2.result()
    assert len(ids1) == len(ids2) == len(seqs)
    both = torch.tensor(ids1 + ids2, dtype=torch.int64)
    gathered = [torch.empty_like(both) for _ in range(world)]
    dist.all_gather(gathered, both)
    assert all(torch.equal(g, both) for g in gathered), "the ranks read back different ids"
    for s, t in zip(seqs, ids1):
        s.token_ids[-1] = t
        s.last_token = t
    logits, _, _ = runner.forward_logits(seqs)       # the second step again, synchronously, from the real ids
    if rank == 0:
        assert logits.float().argmax(-1).tolist() == ids2, "the step launched ahead did not see the previous step's ids"


def main():
    rank, world, port, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    if mode == "rccl1":
        return rccl_capture_check(port)
    os.environ["LOCAL_RANK"] = "0"
    os.environ["NANOVLLM_NATIVE"] = "0" if mode == "modules" else "1"
    if mode == "fallback":                       # no peer exchange: native step in parts + torch.distributed sums between them
        os.environ["NANOVLLM_XGMI_AR"] = "0"
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", world_size=world, rank=rank)
    from conftest import TINY
    from test_model_gpu import TINY_DIR, hf_named, seq
    from oracle import ref_model as M
    from nanovllm import comm
    from nanovllm.config import Config
    from nanovllm.engine.model_runner import ModelRunner

    # "long": one 2200-token prompt in a single step: 2200 x 256 x 2 B = 1.07 MiB per row-parallel sum -> two-shot exchange
    budget = 4096 if mode == "long" else 1024
    cfg = Config(TINY_DIR, max_num_batched_tokens=budget, max_num_seqs=8, max_model_len=4096 if mode == "long" else 2048,
                 tensor_parallel_size=world, enforce_eager=(mode != "graph"), num_kvcache_blocks=12, chunked_prefill=True)
    runner = ModelRunner(cfg, rank, [])
    if mode != "fallback" and comm.active() is None and comm.fallback_reason == "timeout":
        raise RuntimeError("start-up self-test: exchange timed out")
    assert (comm.active() is not None) == (mode != "fallback"), "xGMI exchange state"
    assert (runner.native is not None) == (mode != "modules")       # "fallback": native parts + torch.distributed sums
    shape = M.ModelShape(**TINY)
    weights = M.random_weights(shape, seed=0, std=0.05)
    runner.load_weights(hf_named(weights, shape))
    oracle = M.RefRunner(shape, weights, 12) if rank == 0 else None

    g = torch.Generator().manual_seed(31)
    lengths = (2200, 40, 257) if mode == "long" else (5, 300, 257)
    prompts = [torch.randint(0, 1024, (n,), generator=g).tolist() for n in lengths]
    tables = [[0, 1, 2, 3, 4, 5, 6, 7, 8], [9], [10, 11]] if mode == "long" else [[0], [1, 2], [3, 4]]
    seqs = [seq(p, 0, len(p), t) for p, t in zip(prompts, tables)]
    states = [M.SeqState(list(p), 0, len(p), list(t)) for p, t in zip(prompts, tables)]
    worst = 0.0
    for step in range(3 if mode == "long" else 5):
        logits, _, need = runner.forward_logits(seqs)
        assert need.tolist() == [0, 1, 2]
        if rank == 0:
            ref, _ = oracle.run(states)
            got = logits.float().cpu()
            err = (got - ref.float()).abs()
            worst = max(worst, float(err.max()))
            assert bool((err <= 3e-2 + 2e-2 * ref.float().abs()).all()), f"step {step}: max err {float(err.max()):.4g}"
            assert torch.equal(got.argmax(-1), ref.float().argmax(-1)), f"step {step}: greedy ids differ"
            toks = got.argmax(-1)
        else:
            assert logits is None
            toks = torch.zeros(3, dtype=torch.int64)
        dist.broadcast(toks, 0)
        for i, (s, st, t) in enumerate(zip(seqs, states, toks.tolist())):
            s.num_cached_tokens += s.num_new_tokens
            s.append_token(t)
            s.num_new_tokens = 1
            s.block_table = tables[i][:(len(s) + 255) // 256]
            st.token_ids.append(t)
            st.num_cached_tokens, st.num_new_tokens = s.num_cached_tokens, 1
            st.block_table = list(s.block_table)
    if mode == "graph":
        assert runner.graphs
    if runner.comm is not None:
        runner.comm.check()
    calls = runner.sampler.calls
    tok, need = runner.run(seqs)                     # ids for rank 0's caller only
    if rank == 0:
        assert len(tok) == 3
        print(f"max |logit err| vs oracle over the steps: {worst:.4g}")
    else:
        assert tok is None
    assert runner.shard_sampling == (mode != "modules")
    if runner.shard_sampling:
        # the ids came from the ranks' vocabulary shards; the sampler over the gathered logits draws the same ones
        from nanovllm import ops
        assert runner.sampler.calls == calls + 1
        logits, temps, _ = runner.forward_logits(seqs)
        if rank == 0:
            want = ops.sample(logits, temps, runner.sampler.seed, calls + 1).tolist()
            assert tok == want, (tok, want)
    if mode in ("native", "graph", "fallback"):
        check_one_step_ahead(runner, seqs, tables, rank, world)
    runner.exit()
    dist.destroy_process_group()
    print(f"rank {rank} OK")


if __name__ == "__main__":
    import faulthandler
    faulthandler.dump_traceback_later(75, exit=False)      # a hung rank shows where before the test's deadline kills it
    try:
        main()
    except Exception as e:  # noqa: BLE001
        if "exchange timed out" in str(e):       # a peer was not scheduled on the shared device: see run_ranks()
            print(f"rank not co-scheduled: {e}", flush=True)
            os._exit(77)
        raise
