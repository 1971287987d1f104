"""Qwen3-0.6B at its real shape (28 layers, hidden 1024, vocabulary 151936; synthetic weights) is too large for the
CPU oracle to follow in test time, so these tests check properties that do not depend on size: a prompt gives the
same logits whether it is prefilled at once or in chunks, a captured decode graph equals the eager step bit for
bit, the order of the sequences in a batch does not change a single bit of any row, and a sequence that borrows
the pages of a shared prefix sees the logits of one that computed it itself."""
import os
from pathlib import Path

import pytest
import torch

from test_model_gpu import seq

pytestmark = pytest.mark.gpu
MODEL = str(Path(__file__).resolve().parent.parent / "nano-vllm-v1_amd" / "model_configs" / "qwen3-0.6b")


def make(enforce_eager, budget=1024, pages=40, max_seqs=16):
    os.environ["NANOVLLM_NATIVE"] = "1"
    from nanovllm.config import Config
    from nanovllm.engine.model_runner import ModelRunner
    cfg = Config(MODEL, max_num_batched_tokens=budget, max_num_seqs=max_seqs, max_model_len=4096, enforce_eager=enforce_eager,
                 num_kvcache_blocks=pages, chunked_prefill=True)
    return ModelRunner(cfg, 0, [])


def close(a, b, atol=6e-2, rtol=3e-2):
    a, b = a.float().cpu(), b.float().cpu()
    err = (a - b).abs()
    assert bool((err <= atol + rtol * b.abs()).all()), f"max err {float(err.max()):.4g}"


def greedy_agrees(a, b, atol=6e-2, rtol=3e-2):
    """b's greedy id scores within tolerance of a's best in a (exact bf16 ties may resolve either way)."""
    a, b = a.float().cpu(), b.float().cpu()
    top = a.max(-1).values
    picked = a.gather(1, b.argmax(-1, keepdim=True)).squeeze(1)
    assert bool((picked >= top - (atol + rtol * top.abs())).all())


@pytest.fixture(scope="module")
def eager():
    r = make(True, budget=2048)
    yield r
    r.exit()


def prompts(lengths, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randint(0, 10000, (n,), generator=g).tolist() for n in lengths]


def test_chunked_prefill_equals_one_pass(eager):
    p = prompts([1500], 1)[0]
    table = [0, 1, 2, 3, 4, 5]
    whole, _, need = eager.forward_logits([seq(p, 0, len(p), table)])
    assert need.tolist() == [0]
    s = seq(p, 0, 0, table)
    logits = None
    for n in (700, 300, 500):                                   # page-crossing chunks
        s.num_new_tokens = n
        logits, _, _ = eager.forward_logits([s])
        s.num_cached_tokens += n
    close(logits, whole)
    greedy_agrees(logits, whole)


def test_batch_order_does_not_change_any_row(eager):
    ps = prompts([300, 17, 600, 129, 256], 2)
    tables = [[6, 7], [8], [9, 10, 11], [12], [13, 14]]
    first = [seq(p, 0, len(p), t) for p, t in zip(ps, tables)]
    logits, _, _ = eager.forward_logits(first)
    toks = logits.float().argmax(-1).tolist()

    def decode_batch(order):
        batch = []
        for i in order:
            batch.append(seq(ps[i] + [toks[i]], len(ps[i]), 1, tables[i]))
        out, _, _ = eager.forward_logits(batch)
        return out.cpu()

    for t, p in zip(tables, ps):                               # every table must hold position len(p)
        assert len(t) * 256 > len(p)
    a = decode_batch([0, 1, 2, 3, 4])
    order = [3, 0, 4, 2, 1]
    b = decode_batch(order)
    for row, i in enumerate(order):
        assert torch.equal(b[row], a[i]), f"sequence {i} changed with its position in the batch"


def test_graph_replay_equals_eager_at_full_size(eager):
    graphed = make(False, budget=1024, pages=24)
    try:
        ps = prompts([200, 513, 40], 3)
        tables = [[0], [1, 2, 3], [4]]
        outs = []
        for r in (eager, graphed):
            seqs = [seq(p, 0, len(p), t) for p, t in zip(ps, tables)]
            logits, _, _ = r.forward_logits(seqs)
            steps = []
            for _ in range(3):
                for s, t in zip(seqs, logits.float().argmax(-1).tolist()):
                    s.num_cached_tokens += s.num_new_tokens
                    s.append_token(t)
                    s.num_new_tokens = 1
                logits, _, _ = r.forward_logits(seqs)
                steps.append(logits.cpu())
            outs.append(torch.stack(steps))
        assert graphed.graphs
        assert torch.equal(outs[0], outs[1])
    finally:
        graphed.exit()


def test_shared_prefix_pages_give_the_same_logits(eager):
    shared = prompts([512], 4)[0]
    tail_a, tail_b = prompts([40, 90], 5)
    a = seq(shared + tail_a, 0, 552, [20, 21, 22])
    eager.forward_logits([a])                                   # fills pages 20, 21 with the shared prefix
    own = seq(shared + tail_b, 0, 602, [23, 24, 25])            # computes the prefix itself
    borrowed = seq(shared + tail_b, 512, 90, [20, 21, 26])      # prefix hit: only the tail is scheduled
    lo, _, _ = eager.forward_logits([own])
    lb, _, need = eager.forward_logits([borrowed])
    assert need.tolist() == [0]
    close(lb, lo)
    greedy_agrees(lb, lo)


def test_decode_attention_properties_at_benchmark_size():
    """The benchmark's dominant kernel at its benchmark shape (256 sequences x ~1000 keys, 16/8 heads): values that are
    constant per kv head must come back unchanged (the softmax weights sum to one), and moving the pages of the cache
    to other physical pages must not change a bit."""
    from nanovllm import ops
    b, hq, hkv, d = 256, 16, 8, 128
    g = torch.Generator().manual_seed(7)
    lens = torch.randint(900, 1100, (b,), generator=g, dtype=torch.int32)
    pages_per = 5
    npages = b * pages_per
    kc = (torch.randn(npages, 256, hkv, d, generator=g) * 0.5).to(torch.bfloat16).cuda()
    vc = torch.randn(npages, 256, hkv, d, generator=g).to(torch.bfloat16).cuda()
    q = torch.randn(b, hq, d, generator=g).to(torch.bfloat16).cuda()
    perm = torch.randperm(npages, generator=g).to(torch.int32)
    bt = perm.view(b, pages_per).cuda()
    scale = d ** -0.5
    ctx = lens.cuda()
    out = ops.attn_decode(q, kc, vc, ctx, bt, scale, num_splits=1)
    # constant values per kv head
    const = torch.arange(1, hkv + 1, dtype=torch.float32).view(1, 1, hkv, 1).expand(npages, 256, hkv, d).to(torch.bfloat16).cuda().contiguous()
    same = ops.attn_decode(q, kc, const, ctx, bt, scale, num_splits=1).float()
    want = torch.arange(1, hkv + 1, dtype=torch.float32).repeat_interleave(hq // hkv).view(1, hq, 1).cuda()
    assert float((same - want).abs().max()) <= 2e-2
    # relocate every page
    perm2 = torch.randperm(npages, generator=g)
    inv = torch.empty_like(perm2)
    inv[perm2] = torch.arange(npages)
    kc2, vc2 = kc[perm2.cuda()], vc[perm2.cuda()]               # new page j holds old page perm2[j]
    bt2 = inv.to(torch.int32).cuda()[bt.long()]                 # old page p now lives at inv[p]
    moved = ops.attn_decode(q, kc2, vc2, ctx, bt2, scale, num_splits=1)
    assert torch.equal(out, moved)
    # key-range splits agree with the unsplit result to rounding
    split = ops.attn_decode(q, kc, vc, ctx, bt, scale, num_splits=4).float()
    assert float((split - out.float()).abs().max()) <= 2e-2


def test_lm_head_gemm_at_full_vocabulary_against_fp32():
    """256 rows x 151936 outputs x 1024: every element against an fp32 matmul of the same bf16 operands."""
    from nanovllm import ops
    g = torch.Generator().manual_seed(8)
    x = torch.randn(256, 1024, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(151936, 1024, generator=g) * 0.03).to(torch.bfloat16).cuda()
    y = ops.linear(x, w).float()
    ref = x.float() @ w.float().t()
    err = (y - ref).abs()
    assert bool((err <= 1e-2 + 8e-3 * ref.abs()).all()), f"max err {float(err.max()):.4g}"     # one bf16 rounding of the fp32 sum


@pytest.mark.parametrize("tokens", [1024, 800])
def test_fused_qkv_epilogue_route_equals_two_launch_route(eager, tokens):
    """A chunk of 769..1024 tokens takes the QKV projection whose epilogue applies the q/k RMSNorm + RoPE and stores K/V
    (one launch inside the native step); smaller chunks project first and run the epilogue kernel afterwards.  The K/V
    rows of the FIRST layer depend on nothing but the embedding, the norm, that projection and its epilogue, row by row,
    so both routes must leave the same bits in the cache; the logits after the last chunk agree to rounding (attention
    sums over the chunks in a different order)."""
    p = prompts([tokens], 40 + tokens)[0]
    half = tokens // 2
    one_table, two_table = [28, 29, 30, 31], [32, 33, 34, 35]
    a, _, _ = eager.forward_logits([seq(p, 0, tokens, one_table)])          # one chunk: fused epilogue
    s = seq(p, 0, 0, two_table)
    b = None
    for n in (half, tokens - half):                                         # 512 / 400 rows: two launches
        s.num_new_tokens = n
        b, _, _ = eager.forward_logits([s])
        s.num_cached_tokens += n
    kv = eager.kv_cache
    pages = (tokens + 255) // 256
    for which in (0, 1):
        fused = kv[which, 0, one_table[0]:one_table[0] + pages].reshape(-1, kv.shape[-2], kv.shape[-1])[:tokens]
        split = kv[which, 0, two_table[0]:two_table[0] + pages].reshape(-1, kv.shape[-2], kv.shape[-1])[:tokens]
        assert torch.equal(fused, split), f"layer-0 {'KV'[which]} rows differ between the routes"
    close(b, a)
    greedy_agrees(b, a)


@pytest.mark.parametrize("tokens,running", [(1024, 0), (896, 0), (1024, 9)])
def test_rmsnorm_as_a_row_scale_route_agrees_with_the_norm_launches(eager, switches, tokens, running):
    """Steps of 832..1024 tokens at hidden size 1024 apply every RMSNorm but the first layer's input norm as a row scale
    inside the projection that consumes it (norm weight folded into the projection weight, row sums left by the O / down
    projection's epilogue): two launches less per layer, other rounding points.  Against the same step with the norm
    launches (NVH_NO_ROWSCALE): logits within the tolerance of the other route comparisons of this file, greedy ids equal up
    to ties, and the first layer's K/V rows - which depend on nothing the route changes - bit for bit.  `running`: that
    many one-token sequences lead the step (they take the decode attention kernel; their raw qkv rows are scaled too)."""
    from nanovllm import _hip
    lib = _hip.lib()
    assert eager.native.rowscale and lib.nvh_gemm_residual_rowsq_parts(tokens, 1024, 2048) == 32 and lib.nvh_gemm_residual_rowsq_parts(tokens, 1024, 3072) == 32
    olds = prompts([300 + 11 * i for i in range(running)], 90)
    new = prompts([tokens - running], 91 + tokens)[0]
    outs = {}
    for mode in ("rowscale", "launches"):
        switches("NVH_NO_ROWSCALE", 1 if mode == "launches" else -1)
        eager.kv_cache.zero_()
        seqs = []
        for i, p in enumerate(olds):                                           # contexts of the running sequences (small steps: same route both times)
            s_ = seq(p, 0, len(p), [2 * i, 2 * i + 1])
            eager.forward_logits([s_])
            s_.num_cached_tokens, s_.num_new_tokens = len(p), 1
            s_.append_token(7 + i)
            seqs.append(s_)
        seqs.append(seq(new, 0, len(new), [30, 31, 32, 33]))
        logits, _, _ = eager.forward_logits(seqs)
        torch.cuda.synchronize()
        outs[mode] = (logits.cpu(), eager.kv_cache[:, 0, 30:34].cpu().clone())
    switches("NVH_NO_ROWSCALE", -1)
    assert torch.equal(outs["rowscale"][1], outs["launches"][1])
    assert not torch.equal(outs["rowscale"][0], outs["launches"][0])          # (the route really was another one)
    close(outs["rowscale"][0], outs["launches"][0])
    greedy_agrees(outs["launches"][0], outs["rowscale"][0])


@pytest.mark.parametrize("batch", [1, 5, 16, 27])
def test_decode_prologue_fusion_at_full_size(eager, switches, batch):
    """Decode steps of 1..32 sequences at the real shape (rows of 1024: sixteen k-tiles resident per workgroup, 128 and
    96..192 column tiles, two row tiles from 17 rows): the residual add + RMSNorm inside the QKV and gate/up projections
    with K-slabs kept gives the bits of the separate launches; the default route (O / down projections over the waves
    of a workgroup, no slabs) stays within the logits tolerance of them and picks the same greedy ids up to ties."""
    ps = prompts([40 + 7 * i for i in range(batch)], 50 + batch)
    g = torch.Generator().manual_seed(7)
    forced = torch.randint(0, 10000, (2, batch), generator=g).tolist()
    outs = {}
    for mode in ("default", "slabs", "separate"):
        switches("NVH_DECODE_SLABS", 1 if mode == "slabs" else -1)
        switches("NVH_NO_NORM_GEMM", 1 if mode == "separate" else -1)
        eager.kv_cache.zero_()
        seqs = [seq(p, 0, len(p), [i]) for i, p in enumerate(ps)]
        eager.forward_logits(seqs)
        steps = []
        for k in range(2):
            for s_, t in zip(seqs, forced[k]):
                s_.num_cached_tokens += s_.num_new_tokens
                s_.append_token(t)
                s_.num_new_tokens = 1
            logits, _, _ = eager.forward_logits(seqs)
            steps.append(logits.cpu())
        outs[mode] = torch.stack(steps)
    if batch <= 2:                                   # beyond two rows the executor does not take the slab form by itself
        assert torch.equal(outs["slabs"], outs["separate"])
    else:
        close(outs["slabs"].flatten(0, 1), outs["separate"].flatten(0, 1))
    close(outs["default"].flatten(0, 1), outs["separate"].flatten(0, 1))
    greedy_agrees(outs["separate"].flatten(0, 1), outs["default"].flatten(0, 1))


def test_kv_pool_sized_from_free_memory():
    """num_kvcache_blocks <= 0 (the reference's only mode, engine/model_runner.py:319-375): the pool takes what
    gpu_memory_utilization leaves after the weights and the peak activations of the warm-up step, and the count is
    written back into the config for the scheduler.  The utilisation is set so the pool stays near 3 GB."""
    from nanovllm.config import Config
    from nanovllm.engine.model_runner import ModelRunner
    import gc
    gc.collect()                                                 # tensors of earlier tests that only a cycle keeps alive
    torch.cuda.empty_cache()                                     # as the runner's warm-up does: only live tensors count
    free, total = torch.cuda.mem_get_info()
    util = (total - free + (5 << 30)) / total                    # current use + weights/activations (~2 GB) + ~3 GB of pool
    cfg = Config(MODEL, max_num_batched_tokens=1024, max_num_seqs=8, max_model_len=2048, enforce_eager=True,
                 num_kvcache_blocks=-1, gpu_memory_utilization=util, chunked_prefill=True)
    runner = ModelRunner(cfg, 0, [])
    try:
        n = cfg.num_kvcache_blocks
        assert n > 0 and runner.kv_cache.shape[2] == n
        block = runner.kv_block_bytes()
        assert block == 2 * 28 * 256 * 8 * 128 * 2                # 28 MiB per page over all layers
        assert (1 << 30) < n * block < (8 << 30), f"{n} pages"     # ~3 GB; more if memory of earlier tests was released meanwhile
        free_after, _ = torch.cuda.mem_get_info()
        assert total - free_after <= total * util + (256 << 20), "the pool overshoots gpu_memory_utilization"
        p = prompts([300], 9)[0]
        logits, _, need = runner.forward_logits([seq(p, 0, 300, [n - 2, n - 1])])      # the last pages are usable
        assert need.tolist() == [0] and bool(torch.isfinite(logits.float()).all())
    finally:
        runner.exit()


def test_sampler_inside_the_lm_head_draws_the_ids_of_the_two_launches():
    """256 sequences: the LM head takes its 256 x 256 tile and samples in its epilogue (the logits are never written).
    Same ids as the projection followed by the sampler at temperature 0.8 - a prefill step that selects the last rows
    of 256 prompts, then decode steps through the captured graph - and 255 sequences take the two launches."""
    r = make(False, budget=1024, pages=260, max_seqs=256)
    try:
        ps = prompts([3] * 256, 9)
        w = r.model.lm_head.weight
        from nanovllm import ops
        assert ops.lm_head_sample_fused(256, w.shape[0], w.shape[1]) and not ops.lm_head_sample_fused(200, w.shape[0], w.shape[1])

        def three_steps(fused, count):
            r.fused_head, r.sampler.calls = fused, 0
            seqs = [seq(p, 0, len(p), [i]) for i, p in enumerate(ps[:count])]
            for s in seqs:
                s.temperature = 0.8
            drawn = []
            for _ in range(3):
                ids, need = r.run(seqs)
                assert len(ids) == count
                drawn.append(ids)
                for s, t in zip(seqs, ids):
                    s.num_cached_tokens += s.num_new_tokens
                    s.append_token(t)
                    s.num_new_tokens = 1
            return drawn

        calls = []
        real = ops.lm_head_sample
        ops.lm_head_sample = lambda *a: (calls.append(a[0].shape[0]), real(*a))[1]
        try:
            a = three_steps(True, 256)
            assert calls == [256, 256, 256]
            assert three_steps(False, 256) == a and len(calls) == 3
            assert len(set(a[2])) > 100                         # sampled, not a constant
            three_steps(True, 200)
            assert len(calls) == 3                              # too few rows for the tile: projection, then sampler
        finally:
            ops.lm_head_sample = real
    finally:
        r.exit()
