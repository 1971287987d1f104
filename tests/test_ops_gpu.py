"""Parity of every C-ABI operator (HIP, gfx950) against the CPU oracle on the same seeded
inputs and against the golden vectors produced by the reference modules.

Stated tolerances
  * copies / gathers / argmax / slot arithmetic: bit-exact;
  * RoPE: bit-exact (same fp32 op order, table built on the host);
  * RMSNorm family, SiLU*mul: <= 2 bf16 ulp (two roundings: a 1-ulp flip of the first can move the
    second) on <= 1 % of elements (reduction order / expf ulp);
  * GEMM: |err| <= 2^-7*|ref| + 2e-3*max|ref| (one bf16 rounding of an fp32 sum whose accumulation
    order differs; near-zero outputs are bounded absolutely), and <= 3 % of elements differ at all;
  * attention: |err| <= 1.5e-2 + 2e-2*|ref| (bf16 P in the PV product, exp2 vs exp).
"""
import math

import pytest
import torch

from conftest import bf16_ulp_diff
from oracle import ref_ops as R

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    from nanovllm import _hip, ops as o
    _hip.check(_hip.lib().nvh_device_check(0), "nvh_device_check")
    return o


def dev(t):
    return t.cuda()


def rnd(*s, seed=0, scale=1.0, dtype=BF):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*s, generator=g) * scale).to(dtype)


def assert_ulp(got, ref, max_ulp=2, max_frac=0.01):
    got = got.cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    d = bf16_ulp_diff(got, ref)
    frac = float((d > 0).float().mean())
    assert int(d.max()) <= max_ulp and frac <= max_frac, f"max ulp {int(d.max())}, frac differing {frac:.4f}"


def assert_gemm(got, ref, max_frac=0.03):
    got = got.cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    g, r = got.float(), ref.float()
    err = (g - r).abs()
    tol = 2.0 ** -7 * r.abs() + 2e-3 * float(r.abs().max())
    frac = float((got != ref).float().mean())
    assert bool((err <= tol).all()) and frac <= max_frac, f"max err {float(err.max()):.4g}, frac differing {frac:.4f}"


def assert_exact(got, ref):
    got, ref = got.cpu(), ref.cpu()
    assert got.shape == ref.shape and torch.equal(got, ref), f"mismatch: {int((got != ref).sum())} elements"


def assert_close(got, ref, atol, rtol):
    got = got.cpu().float()
    ref = ref.float()
    err = (got - ref).abs()
    bad = err > atol + rtol * ref.abs()
    assert not bool(bad.any()), f"max err {float(err.max()):.4g}, {int(bad.sum())} / {bad.numel()} outside tolerance"


# ------------------------------------------------------------------ golden vectors (reference modules)
def test_golden_rmsnorm(ops, f1):
    for tag in ("rms_2d", "rms_heads"):
        y = ops.rmsnorm(dev(f1.bf(tag + "_x")), dev(f1.bf(tag + "_w")), 1e-6)
        assert_ulp(y, f1.bf(tag + "_y"))
    res = dev(f1.bf("addrms_r").clone())
    y, res2 = ops.add_rmsnorm(dev(f1.bf("addrms_x")), res, dev(f1.bf("addrms_w")), 1e-6)
    assert_ulp(y, f1.bf("addrms_y"))
    assert_exact(res2, f1.bf("addrms_res"))


def test_golden_rope_bit_exact(ops, f1):
    table = R.rope_table(128, 40960, 1e6).squeeze(1).contiguous()
    q, k = dev(f1.bf("rope_q").clone()), dev(f1.bf("rope_k").clone())
    ops.rope_(dev(f1.t("rope_pos")), q, k, dev(table))
    assert_exact(q, f1.bf("rope_qo"))
    assert_exact(k, f1.bf("rope_ko"))


def test_golden_silu_linear_embedding(ops, f1):
    assert_ulp(ops.silu_mul(dev(f1.bf("silu_x"))), f1.bf("silu_y"))
    for tag in ("lin", "lin_bias"):
        b = dev(f1.bf(tag + "_b")) if tag + "_b" in f1 else None
        y = ops.linear(dev(f1.bf(tag + "_x")), dev(f1.bf(tag + "_w")), b)
        assert_gemm(y, f1.bf(tag + "_y"))
    assert_exact(ops.embedding(dev(f1.t("emb_ids")), dev(f1.bf("emb_w"))), f1.bf("emb_y"))
    hidden, cu = dev(f1.bf("head_hidden")), dev(f1.t("head_cu"))
    need = dev(f1.t("head_subset_need"))
    rows = ops.gather_last_rows(hidden, cu, need, 2)
    assert_exact(rows, R.last_token_rows(f1.bf("head_hidden"), f1.t("head_cu"), f1.t("head_subset_need")))
    rows_all = ops.gather_last_rows(hidden, cu, None, 3)
    assert_exact(rows_all, R.last_token_rows(f1.bf("head_hidden"), f1.t("head_cu"), f1.t("head_empty_need")))


def test_golden_sampler(ops, f1):
    tok = ops.sample_with_noise(dev(f1.t("samp_logits")), dev(f1.t("samp_temps")), dev(f1.t("samp_noise")))
    assert_exact(tok, f1.t("samp_tok"))


# ------------------------------------------------------------------ seeded inputs vs the oracle
@pytest.mark.parametrize("rows,dim", [(1, 1024), (7, 1024), (1024, 1024), (33, 5120), (5, 256), (64, 128), (3, 8192)])
def test_rmsnorm_shapes(ops, rows, dim):
    x, w = rnd(rows, dim, seed=1, scale=2.0), (rnd(dim, seed=2).float() * 0.1 + 1).to(BF)
    assert_ulp(ops.rmsnorm(dev(x), dev(w), 1e-6), R.rms_norm(x, w, 1e-6))
    r = rnd(rows, dim, seed=3, scale=2.0)
    yr, nr = R.add_rms_norm(x, r, w, 1e-6)
    rd = dev(r.clone())
    y, res = ops.add_rmsnorm(dev(x), rd, dev(w), 1e-6)
    assert_ulp(y, yr)
    assert_exact(res, nr)
    assert res.data_ptr() == rd.data_ptr()        # residual updated in place


def test_add_rmsnorm_from_splitk_slabs(ops):
    m, n, k = 5, 1024, 2048
    x, w = rnd(m, k, seed=4), rnd(n, k, seed=5, scale=0.05)
    gamma = (rnd(n, seed=6).float() * 0.1 + 1).to(BF)
    res = rnd(m, n, seed=7)
    slabs = ops.linear_splitk(dev(x), dev(w), 4)
    assert slabs.shape == (4, m, n)
    y_lin = R.linear(x, w)
    assert_gemm(ops.splitk_reduce(slabs), y_lin)
    y, nres = ops.add_rmsnorm(None, dev(res.clone()), dev(gamma), 1e-6, partials=slabs)
    yr, rr = R.add_rms_norm(y_lin, res, gamma, 1e-6)
    assert_close(y, yr, 2e-2, 2e-2)
    assert_close(nres, rr, 2e-2, 2e-2)


@pytest.mark.parametrize("t,raw,store", [(1024, 0, True), (900, 37, True), (1024, 1024, True), (770, 5, False), (100, 3, True)])
def test_fused_qkv_projection_equals_separate_launches(ops, t, raw, store):
    """QKV projection with the q/k epilogue and the KV store in the GEMM's epilogue (128x128 tiles, from 769 rows on)
    against GEMM + qknorm_rope_store on the rows past `raw`: same bits in qkv and in both caches; the oracle's
    per-head norm + RoPE on one of the rows."""
    hq, hkv, d, k = 16, 8, 128, 1024
    n = (hq + 2 * hkv) * d
    x, w = rnd(t, k, seed=51), rnd(n, k, seed=52, scale=0.05)
    qw, kw = (rnd(d, seed=53).float() * 0.2 + 1).to(BF), (rnd(d, seed=54).float() * 0.2 + 1).to(BF)
    g = torch.Generator().manual_seed(55)
    pos = torch.randint(0, 4000, (t,), generator=g)
    slots = torch.randperm(5 * 256, generator=g)[:t].to(torch.int32)
    slots[t // 2] = -1
    table = dev(R.rope_table(128, 4096, 1e6).squeeze(1).contiguous())
    kc0, vc0 = rnd(5, 256, hkv, d, seed=56), rnd(5, 256, hkv, d, seed=57)
    # separate launches
    qkv_a = ops.linear(dev(x), dev(w))
    kca, vca = dev(kc0.clone()), dev(vc0.clone())
    if raw < t:
        ops.qknorm_rope_store_(qkv_a[raw:], hq, hkv, dev(qw), dev(kw), 1e-6, dev(pos[raw:]), table, kca if store else None,
                               vca if store else None, dev(slots[raw:]) if store else None)
    # one call
    kcb, vcb = dev(kc0.clone()), dev(vc0.clone())
    qkv_b = ops.linear_qkv_rope_store(dev(x), dev(w), hq, hkv, dev(qw), dev(kw), 1e-6, dev(pos), table, kcb if store else None,
                                      vcb if store else None, dev(slots) if store else None, raw_rows=raw)
    assert_exact(qkv_b, qkv_a)
    assert_exact(kcb, kca)
    assert_exact(vcb, vca)
    if raw < t:
        r = t - 1
        y = R.linear(x[r:r + 1], w)
        q = R.rms_norm(y[:, :hq * d].reshape(1, hq, d), qw, 1e-6)
        kk = R.rms_norm(y[:, hq * d:(hq + hkv) * d].reshape(1, hkv, d), kw, 1e-6)
        qr, kr = R.rope(pos[r:r + 1], q, kk, R.rope_table(128, 4096, 1e6))
        assert_close(qkv_b[r, :hq * d].view(hq, d), qr[0], 3e-2, 3e-2)
        assert_close(qkv_b[r, hq * d:(hq + hkv) * d].view(hkv, d), kr[0], 3e-2, 3e-2)


@pytest.mark.parametrize("t", [1024, 896])
def test_rmsnorm_as_a_row_scale_against_the_separate_launches(ops, t):
    """The route of prefill-sized steps that drops the add + RMSNorm launches: the O / down projection adds its product to
    the residual in place and leaves per-row sums of squares; the QKV and gate/up projections run over the un-normalised
    residual with the norm weight folded into their weights and scale each output row.  The producer is checked bit for
    bit against projection + add (same kernel, same rounding points) and its sums against fp32; the consumers against the
    separate launches AND against the oracle's fp32-accumulated result - their error to the oracle must not exceed the
    separate launches' by more than a fraction (the rounding points move, the tolerance does not)."""
    h, hq, hkv, d, inter, eps = 1024, 16, 8, 128, 3072, 1e-6
    qsz, n_qkv = hq * d, (hq + 2 * hkv) * d
    attn, w_o = rnd(t, qsz, seed=61), rnd(h, qsz, seed=62, scale=0.03)
    res0 = rnd(t, h, seed=63)
    gamma_post = (rnd(h, seed=64).float() * 0.3 + 1).to(BF)
    gamma_in = (rnd(h, seed=65).float() * 0.3 + 1).to(BF)
    parts = ops.residual_rowsq_parts(t, h, qsz)
    assert parts == h // 32
    # ---- producer
    y = ops.linear(dev(attn), dev(w_o))
    want_sum = y.float() + dev(res0).float()
    res = dev(res0.clone())
    rowsq = ops.linear_add_residual_(dev(attn), dev(w_o), res)
    assert_exact(res, want_sum.to(BF))
    torch.testing.assert_close(rowsq.sum(-1).cpu(), (want_sum * want_sum).sum(-1).cpu(), rtol=2e-5, atol=0)
    # ---- gate/up + SwiGLU consumer
    w_gu = rnd(2 * inter, h, seed=66, scale=0.03)
    w_int = ops.interleave_gate_up(dev(w_gu))
    xn, _ = ops.add_rmsnorm(y, dev(res0.clone()), dev(gamma_post), eps)
    sep = ops.linear_swiglu(xn, w_int)
    w_fold = ops.interleave_gate_up(ops.fold_norm_weight(dev(w_gu), dev(gamma_post)))
    got = ops.linear_swiglu_rowscaled(res, w_fold, rowsq, eps)
    xs = want_sum.cpu()
    xn32 = xs * torch.rsqrt((xs * xs).mean(-1, keepdim=True) + eps) * gamma_post.float()
    gu = xn32 @ w_gu.float().T
    want = torch.nn.functional.silu(gu[:, :inter]) * gu[:, inter:]
    e_sep, e_got = (sep.cpu().float() - want).abs(), (got.cpu().float() - want).abs()
    assert_close(got, want, 3e-2, 3e-2)
    assert float(e_got.mean()) <= 1.25 * float(e_sep.mean()) + 1e-6, (float(e_got.mean()), float(e_sep.mean()))
    # ---- QKV consumer (q/k norm + RoPE + KV store in its epilogue)
    w_qkv = rnd(n_qkv, h, seed=67, scale=0.03)
    qw, kw = (rnd(d, seed=68).float() * 0.2 + 1).to(BF), (rnd(d, seed=69).float() * 0.2 + 1).to(BF)
    g = torch.Generator().manual_seed(70)
    pos = torch.randint(0, 4000, (t,), generator=g)
    slots = torch.randperm(5 * 256, generator=g)[:t].to(torch.int32)
    table = dev(R.rope_table(128, 4096, 1e6).squeeze(1).contiguous())
    kc0, vc0 = rnd(5, 256, hkv, d, seed=71), rnd(5, 256, hkv, d, seed=72)
    xn_in = ops.rmsnorm(res, dev(gamma_in), eps)
    kca, vca = dev(kc0.clone()), dev(vc0.clone())
    qkv_a = ops.linear_qkv_rope_store(xn_in, dev(w_qkv), hq, hkv, dev(qw), dev(kw), eps, dev(pos), table, kca, vca, dev(slots), raw_rows=3)
    kcb, vcb = dev(kc0.clone()), dev(vc0.clone())
    rowsq_res = (res.float() * res.float()).view(t, parts, 32).sum(-1).contiguous()     # sums of the bf16 residual itself: what a plain RMSNorm of it sees
    qkv_b = ops.linear_qkv_rope_store_rowscaled(res, ops.fold_norm_weight(dev(w_qkv), dev(gamma_in)), hq, hkv, dev(qw), dev(kw), eps, dev(pos),
                                                table, rowsq_res, eps, kcb, vcb, dev(slots), raw_rows=3)
    assert_close(qkv_b, qkv_a.cpu(), 4e-2, 4e-2)
    assert_close(kcb, kca.cpu(), 4e-2, 4e-2)
    assert_close(vcb, vca.cpu(), 4e-2, 4e-2)
    assert float((qkv_b.float() - qkv_a.float()).abs().mean()) < 4e-3


def test_store_kvcache_strided_and_skips(ops):
    t, hq, hkv, d = 37, 4, 2, 128
    qkv = rnd(t, (hq + 2 * hkv) * d, seed=8)
    q, k, v = qkv.split([hq * d, hkv * d, hkv * d], dim=-1)
    k, v = k.view(t, hkv, d), v.view(t, hkv, d)
    g = torch.Generator().manual_seed(9)
    slots = torch.randperm(3 * 256, generator=g)[:t].to(torch.int32)
    slots[5] = -1
    slots[36] = -1
    kc, vc = rnd(3, 256, hkv, d, seed=10), rnd(3, 256, hkv, d, seed=11)
    kr, vr = kc.clone(), vc.clone()
    R.store_kvcache(k, v, kr, vr, slots)
    qkv_d = dev(qkv)
    _, kd, vd = qkv_d.split([hq * d, hkv * d, hkv * d], dim=-1)
    kcd, vcd = dev(kc), dev(vc)
    ops.store_kvcache(kd.view(t, hkv, d), vd.view(t, hkv, d), kcd, vcd, dev(slots))
    assert_exact(kcd, kr)
    assert_exact(vcd, vr)


def test_fused_qknorm_rope_store_equals_sequence(ops):
    t, hq, hkv, d = 19, 16, 8, 128
    qkv = rnd(t, (hq + 2 * hkv) * d, seed=12, scale=1.5)
    qw, kw = (rnd(d, seed=13).float() * 0.2 + 1).to(BF), (rnd(d, seed=14).float() * 0.2 + 1).to(BF)
    pos = torch.tensor([0, 1, 2, 255, 256, 257, 4095, 40959] + list(range(100, 111)), dtype=torch.int64)
    table = R.rope_table(128, 40960, 1e6)
    slots = torch.arange(t, dtype=torch.int32) * 3 + 1
    slots[4] = -1
    q, k, v = qkv.split([hq * d, hkv * d, hkv * d], dim=-1)
    qn = R.rms_norm(q.reshape(t, hq, d), qw, 1e-6)
    kn = R.rms_norm(k.reshape(t, hkv, d), kw, 1e-6)
    qr, kr = R.rope(pos, qn, kn, table)
    kc, vc = torch.zeros(1, 256, hkv, d, dtype=BF), torch.zeros(1, 256, hkv, d, dtype=BF)
    R.store_kvcache(kr, v.reshape(t, hkv, d), kc, vc, slots)

    qkv_d = dev(qkv.clone())
    kcd, vcd = dev(torch.zeros_like(kc)), dev(torch.zeros_like(vc))
    ops.qknorm_rope_store_(qkv_d, hq, hkv, dev(qw), dev(kw), 1e-6, dev(pos), dev(table.squeeze(1).contiguous()), kcd, vcd, dev(slots))
    qd, kd, vd = qkv_d.cpu().split([hq * d, hkv * d, hkv * d], dim=-1)
    assert_ulp(qd.reshape(t, hq, d), qr, max_ulp=1, max_frac=0.01)
    assert_ulp(kd.reshape(t, hkv, d), kr, max_ulp=1, max_frac=0.01)
    assert_exact(vd, v)                                   # v untouched
    assert_exact(vcd, vc)                                 # v scatter is a pure copy
    assert_exact(kcd[0][slots[slots >= 0].long()], kd.reshape(t, hkv, d)[slots >= 0])   # cache rows == rotated k rows
    # the separate entry points sum the squares in a different lane order: at most the last bit moves
    qkv_s = dev(qkv.clone())
    qs, ks, vs = qkv_s.split([hq * d, hkv * d, hkv * d], dim=-1)
    qn_d = ops.rmsnorm(qs.reshape(t, hq, d), dev(qw), 1e-6)
    kn_d = ops.rmsnorm(ks.reshape(t, hkv, d), dev(kw), 1e-6)
    ops.rope_(dev(pos), qn_d, kn_d, dev(table.squeeze(1).contiguous()))
    assert_ulp(qn_d.cpu(), qd.reshape(t, hq, d), max_ulp=1, max_frac=0.01)
    assert_ulp(kn_d.cpu(), kd.reshape(t, hkv, d), max_ulp=1, max_frac=0.01)


def test_silu_mul_shapes(ops):
    for rows, inter in ((1, 3072), (300, 3072), (17, 512)):
        x = rnd(rows, 2 * inter, seed=15, scale=3.0)
        assert_ulp(ops.silu_mul(dev(x)), R.silu_mul(x))


def test_embedding_vocab_parallel(ops):
    table = rnd(64, 256, seed=16)
    ids = torch.tensor([0, 63, 17, 40, 31, 32], dtype=torch.int64)
    for ws in (1, 2, 4):
        n = 64 // ws
        for rk in range(ws):
            shard = table[rk * n:(rk + 1) * n].contiguous()
            got = ops.embedding(dev(ids), dev(shard), vocab_start=rk * n)
            assert_exact(got, R.embedding(ids, shard, vocab_start=rk * n, tp_size=ws))


def test_feed_token_ids_copies_sampled_rows_into_the_next_input(ops):
    """input_ids[dst[i]] = sampled[src[i]] (the ids an engine running one step ahead hands to the next step on the
    device); untouched positions keep their value."""
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(0, 1000, (300,), generator=g, dtype=torch.int64)
    sampled = torch.randint(1000, 2000, (40,), generator=g, dtype=torch.int64)
    dst = torch.randperm(300, generator=g)[:33].to(torch.int32)
    src = torch.randint(0, 40, (33,), generator=g, dtype=torch.int32)
    want = ids.clone()
    want[dst.long()] = sampled[src.long()]
    got = dev(ids)
    ops.feed_token_ids(got, dev(dst), dev(sampled), dev(src))
    assert torch.equal(got.cpu(), want)
    ops.feed_token_ids(got, dev(dst[:0]), dev(sampled), dev(src[:0]))          # nothing to feed: no launch
    assert torch.equal(got.cpu(), want)


@pytest.mark.parametrize("world,rows,vocab", [(2, 3, 1024), (4, 7, 151936), (8, 256, 151936), (8, 1, 4096)])
def test_sampling_over_vocabulary_shards_draws_the_gathered_rows_ids(ops, world, rows, vocab):
    """Every rank offers the best candidate of its shard (noise of the full row's columns), the candidates are summed
    as bf16 bytes, every rank picks the winner: the ids of the sampler over the whole rows - also when the best score
    occurs twice (lowest index wins) and at a temperature that makes the noise irrelevant."""
    g = torch.Generator().manual_seed(world * 1000 + rows)
    logits = (torch.randn(rows, vocab, generator=g) * 3).to(torch.bfloat16)
    logits[0, vocab - 5] = logits[0].max() + 1                    # rows whose maximum sits in the last / first shard
    if rows > 1:
        logits[rows - 1, 3] = logits[rows - 1].max() + 1
    temps = torch.full((rows,), 0.8)
    temps[0] = 1e-6                                               # near-greedy: the planted maximum must win
    ld, td = dev(logits), dev(temps)
    want = ops.sample(ld, td, 11, 5)
    shard = vocab // world
    enc = None
    for r in range(world):
        e = ops.sample_shard(ld[:, r * shard:(r + 1) * shard], td, 11, 5, r * shard, r, world)
        assert float(e.float().abs().sum() - e[r].float().abs().sum()) == 0.0        # only its own segment
        enc = e if enc is None else enc + e
    got = ops.sample_pick(enc.contiguous())
    assert torch.equal(got.cpu(), want.cpu())
    assert int(got[0]) == vocab - 5
    # an exact tie between two shards: the lower index wins, as in the scan over the whole row
    tie = torch.zeros(1, vocab, dtype=torch.bfloat16)
    tie[0, shard - 1] = tie[0, shard + 1] = 50.0
    t1 = dev(torch.full((1,), 1e-9))
    enc = sum(ops.sample_shard(dev(tie)[:, r * shard:(r + 1) * shard], t1, 3, 1, r * shard, r, world) for r in range(world))
    assert int(ops.sample_pick(enc.contiguous())[0]) == int(ops.sample(dev(tie), t1, 3, 1)[0]) == shard - 1


@pytest.mark.parametrize("m,n,k", [(256, 151936, 256), (230, 151936, 128), (460, 151936, 64), (1024, 16384, 192), (700, 30000, 64),
                                   (300, 151936, 64), (850, 151936, 64)])           # row tiles with 41 % / 17 % padding
def test_lm_head_with_sampler_in_its_epilogue_draws_the_separate_calls_ids(ops, m, n, k):
    """The projection whose workgroups score their own 256 columns (logits never written) against the projection
    followed by the sampler (reference: layers/embed_head.py:204 then layers/sampler.py:76-80): the same ids, bit for
    bit - ragged rows, a half-empty last column tile, near-greedy rows, and a row whose best logit occurs twice."""
    assert ops.lm_head_sample_fused(m, n, k)
    assert not ops.lm_head_sample_fused(64, n, k) and not ops.lm_head_sample_fused(m, 1024, k)
    x, w = rnd(m, k, seed=m + 3), rnd(n, k, seed=n + k, scale=0.2)
    w[n - 1] = w[7]                                               # columns 7 and n-1 give the same logit in every row
    x[1] = 40 * w[7] / w[7].norm()                                # ... and row 1 makes it the largest by far
    temps = torch.rand(m, generator=torch.Generator().manual_seed(k)) * 1.2 + 0.3
    temps[0] = temps[1] = 1e-6                                    # near-greedy
    xd, wd, td = dev(x), dev(w), dev(temps)
    logits = ops.linear(xd, wd)
    for seed, offset in [(0, 0), (1234567, 99)]:
        want = ops.sample(logits, td, seed, offset)
        got = ops.lm_head_sample(xd, wd, td, seed, offset)
        assert torch.equal(got.cpu(), want.cpu())
    assert int(got[1]) == 7 and int(got[0]) == int(logits[0].float().argmax())
    with pytest.raises(RuntimeError):
        ops.lm_head_sample(xd[:64], wd, td[:64], 0, 0)            # does not take the 256 x 256 tile: refused, not computed otherwise


GEMM_SHAPES = [(1, 4096, 1024), (9, 160, 192), (16, 1024, 2048), (17, 6144, 1024), (33, 1024, 3072), (64, 1024, 256),
               (100, 4096, 1024), (128, 1024, 1024), (300, 6144, 1024), (1024, 1024, 3072), (5, 151936, 256),
               (256, 6144, 1024), (256, 151936, 64), (400, 151936, 128), (400, 50176, 128),   # LM head: 256x256 tile, padded rows; 256x128 tile (two row tiles)
               (5, 26624, 192), (40, 4096, 1024), (64, 7168, 512),         # 16x64 two-wave tile; 64x64 at 33..64 rows
               (1024, 16384, 256), (1000, 15400, 192), (2048, 8192, 64),   # 256x256 eight-wave tile (full, ragged, one k-tile)
               (1024, 17408, 128), (512, 35000, 64),                       # ... whole rounds + a remainder launch on smaller tiles
               (1, 1024, 2048), (7, 1024, 3072), (17, 1024, 3072), (32, 1000, 2048), (3, 2048, 1536), (16, 1024, 4096),   # K split between the waves of a workgroup (decode O / down projections)
               (8, 5120, 1600), (16, 5120, 4352), (32, 2048, 3200), (64, 1024, 3072), (47, 1024, 2048), (3, 1000, 2048), (20, 520, 1536),   # ... wider outputs, K up to 4608, up to 64 rows
               (1000, 1024, 2048), (256, 4096, 1024), (200, 4000, 1152),   # one 64 x 64 tile per CU over a long K: two wave groups over alternate k-tiles (ragged rows / columns)
               (520, 1536, 128), (576, 768, 64), (600, 768, 128), (1100, 1280, 192)]   # XCD-aware tile order: strips of 8 / 4 column tiles, odd and ragged row-tile counts, a tile count that 8 does not divide


@pytest.mark.parametrize("m,n,k", GEMM_SHAPES)
def test_gemm_vs_oracle(ops, m, n, k):
    x, w = rnd(m, k, seed=m + n), rnd(n, k, seed=k + 1, scale=0.05)
    ref = R.linear(x, w)
    y = ops.linear(dev(x), dev(w))
    assert_gemm(y, ref)


def test_gemm_bias_strided_input_and_splitk(ops):
    m, n, k = 45, 1024, 2048
    x_wide = rnd(m, k + 64, seed=20)
    x = x_wide[:, :k]                                   # row stride k+64
    w, b = rnd(n, k, seed=21, scale=0.05), rnd(n, seed=22)
    ref = R.linear(x, w, b)
    assert_gemm(ops.linear(dev(x_wide)[:, :k], dev(w), dev(b)), ref)
    for s in (2, 3, 8):
        slabs = ops.linear_splitk(dev(x_wide)[:, :k], dev(w), s)
        assert_gemm(ops.splitk_reduce(slabs, dev(b)), ref)


def test_gemm_k_over_waves_bias_and_strided_rows(ops):
    """Decode-sized batch, narrow output, long K: K is split between the waves of a workgroup; bias and strided input rows."""
    m, n, k = 9, 1024, 3072
    x_wide = rnd(m, k + 64, seed=23)
    w, b = rnd(n, k, seed=24, scale=0.05), rnd(n, seed=25)
    assert_gemm(ops.linear(dev(x_wide)[:, :k], dev(w), dev(b)), R.linear(x_wide[:, :k], w, b))


def test_gemm_two_wave_groups_bias_and_strided_rows(ops):
    """One 64 x 64 tile per CU over K >= 1024 (eight-wave workgroups, two groups over alternate k-tiles): bias and strided
    input rows."""
    m, n, k = 960, 1024, 1280
    x_wide = rnd(m, k + 128, seed=26)
    w, b = rnd(n, k, seed=27, scale=0.05), rnd(n, seed=28)
    assert_gemm(ops.linear(dev(x_wide)[:, :k], dev(w), dev(b)), R.linear(x_wide[:, :k], w, b))


def test_gemm_bias_across_the_round_cut(ops):
    """A 256 x 256 grid cut into a whole-round launch and a remainder launch: the bias and the output columns of the
    remainder start where the first launch ended."""
    m, n, k = 1024, 17408, 64
    x, w, b = rnd(m, k, seed=31), rnd(n, k, seed=32, scale=0.05), rnd(n, seed=33)
    assert_gemm(ops.linear(dev(x), dev(w), dev(b)), R.linear(x, w, b))


@pytest.mark.parametrize("m,n,k,s", [(1024, 5120, 384, 3), (1000, 5000, 512, 3), (512, 5120, 384, 6), (2048, 2560, 256, 3)])
def test_gemm_k_slabs_on_the_256_tile(ops, m, n, k, s):
    """Tiles x slabs of one nearly full round take the eight-wave 256 x 256 tile (the Qwen3-14B o / down projections at
    1024 rows): slabs summed by the reduce kernel and by add + RMSNorm against the oracle's linear."""
    x, w = rnd(m, k, seed=m + 1), rnd(n, k, seed=n + 2, scale=0.05)
    ref = R.linear(x, w)
    slabs = ops.linear_splitk(dev(x), dev(w), s)
    assert slabs.shape == (s, m, n)
    assert_gemm(ops.splitk_reduce(slabs), ref)
    assert_gemm(ops.linear(dev(x), dev(w)), ref)


@pytest.mark.parametrize("m,inter,k", [(1, 3072, 1024), (20, 512, 256), (256, 3072, 1024), (1024, 3072, 1024), (70, 96, 192),
                                       (3, 13312, 128), (16, 17408, 64), (48, 3072, 256),    # 16x64 tile (832 / 1088 narrow tiles)
                                       (1024, 8192, 256), (1024, 8704, 128),                 # 256x256 eight-wave tile; with a remainder launch
                                       (900, 3072, 256), (1024, 3072, 1600)])                # eight-wave 128x192 tiles with ragged rows; 128x96 tiles (K beyond the 192-wide rule)
def test_gemm_swiglu_epilogue_bit_identical_to_separate_kernels(ops, m, inter, k):
    x, w = rnd(m, k, seed=m + 1), rnd(2 * inter, k, seed=k + 2, scale=0.05)
    xd, wd = dev(x), dev(w)
    sep = ops.silu_mul(ops.linear(xd, wd))
    wi = ops.interleave_gate_up(wd)
    # interleaved rows: block b = 16 gate rows then the 16 up rows of the same outputs
    assert torch.equal(wi[:16].cpu(), w[:16]) and torch.equal(wi[16:32].cpu(), w[inter:inter + 16])
    fused = ops.linear_swiglu(xd, wi)
    assert_exact(fused, sep.cpu())
    assert_gemm(fused, R.silu_mul(R.linear(x, w)), max_frac=0.05)


@pytest.mark.parametrize("m,n,k,slabs,swiglu", [(1, 4096, 1024, 12, False), (1, 6144, 1024, 8, True), (2, 4096, 1024, 0, False),
                                                (5, 1000, 512, 3, False), (16, 6144, 1024, 0, True), (17, 4096, 1024, 0, False),
                                                (32, 2048, 1024, 2, True), (9, 96, 64, 5, False), (3, 4096, 960, 2, False),
                                                (32, 6144, 1024, 0, True), (21, 6144, 1024, 3, True), (29, 4096, 1024, 0, False),
                                                (64, 4096, 1024, 0, False), (45, 2048, 512, 0, True), (50, 4096, 1024, 0, False),
                                                (20, 6176, 1024, 0, True), (2, 96, 128, 0, True), (12, 1000, 1024, 0, False)])   # ragged last tiles   # 17..32 rows: row tiles of 16; 64-column tiles with two loader waves
def test_add_rmsnorm_in_the_projection_prologue_bit_identical(ops, m, n, k, slabs, swiglu):
    """Residual add + RMSNorm + projection in one launch (decode batches) against the two launches: the projection
    output and the new residual carry the same bits, from a bf16 input or from split-K slabs of the previous projection,
    plain or with the SwiGLU epilogue; the input residual is left alone; against the oracle within the GEMM tolerance."""
    prev_in, prev_w = rnd(m, 1024, seed=61), rnd(k, 1024, seed=62, scale=0.05)
    res, gamma = rnd(m, k, seed=63), (rnd(k, seed=64).float() * 0.1 + 1).to(BF)
    w = rnd(n, k, seed=65, scale=0.05)
    wd = dev(w)
    if swiglu:
        wd = ops.interleave_gate_up(wd)
    partials = ops.linear_splitk(dev(prev_in), dev(prev_w), slabs) if slabs else None
    prev = None if slabs else ops.linear(dev(prev_in), dev(prev_w))
    res_a = dev(res.clone())
    xa, res_a = ops.add_rmsnorm(prev, res_a, dev(gamma), 1e-6, partials=partials)
    ya = ops.linear_swiglu(xa, wd) if swiglu else ops.linear(xa, wd)
    res_b = dev(res.clone())
    yb, new_res = ops.add_rmsnorm_linear(prev, res_b, dev(gamma), 1e-6, wd, partials=partials, swiglu=swiglu)
    assert_exact(yb, ya.cpu())
    assert_exact(new_res, res_a.cpu())
    assert_exact(res_b, res)
    prev_ref = R.linear(prev_in, prev_w)
    x_ref, _ = R.add_rms_norm(prev_ref, res, gamma, 1e-6)
    y_ref = R.linear(x_ref, w)
    assert_gemm(yb, R.silu_mul(y_ref) if swiglu else y_ref, max_frac=0.05)


def test_rmsnorm_without_an_add_in_the_projection_prologue(ops):
    """The first layer's form of the fused launch: nothing to add, the rows themselves are the residual stream.  Bits of
    nvh_rmsnorm + nvh_gemm_bf16_nt, rows copied to the second buffer."""
    from nanovllm import _hip
    m, n, k = 7, 4096, 1024
    x, gamma, w = rnd(m, k, seed=71), (rnd(k, seed=72).float() * 0.1 + 1).to(BF), rnd(n, k, seed=73, scale=0.05)
    xd, wd, gd = dev(x), dev(w), dev(gamma)
    want = ops.linear(ops.rmsnorm(xd, gd, 1e-6), wd)
    y, res_out = torch.empty(m, n, dtype=BF, device="cuda"), torch.empty(m, k, dtype=BF, device="cuda")
    _hip.call("nvh_add_rmsnorm_gemm_bf16_nt", 0, 0, 0, xd.data_ptr(), res_out.data_ptr(), gd.data_ptr(), 1e-6, wd.data_ptr(), y.data_ptr(), n,
              m, n, k, 0, torch.cuda.current_stream().cuda_stream)
    assert_exact(y, want.cpu())
    assert_exact(res_out, x)


def test_add_rmsnorm_projection_argument_checks(ops):
    from nanovllm import _hip
    lib = _hip.lib()
    a = torch.zeros(4, 1024, dtype=BF, device="cuda")
    w = torch.zeros(64, 1024, dtype=BF, device="cuda")
    args = lambda res_out, m, k: (a.data_ptr(), 0, 0, a.data_ptr(), res_out, a.data_ptr(), 1e-6, w.data_ptr(), a.data_ptr(), 64, m, 64, k, 0, 0)
    assert lib.nvh_add_rmsnorm_gemm_bf16_nt(*args(a.data_ptr(), 4, 1024)) == -22           # residual_out aliases residual_in
    b = torch.zeros(4, 1024, dtype=BF, device="cuda")
    assert lib.nvh_add_rmsnorm_gemm_bf16_nt(*args(b.data_ptr(), 65, 1024)) == -38           # too many rows: NVH_ENOSYS
    assert lib.nvh_add_rmsnorm_gemm_bf16_nt(*args(b.data_ptr(), 4, 2048)) == -38            # rows too wide
    assert not ops.fused_add_rmsnorm_linear(65, 4096, 1024) and not ops.fused_add_rmsnorm_linear(4, 4096, 5120)
    assert not ops.fused_add_rmsnorm_linear(48, 4096, 1024)      # the kernels take up to 64 rows, callers are advised to stop at 32
    assert ops.fused_add_rmsnorm_linear(1, 4096, 1024, 12) and ops.fused_add_rmsnorm_linear(1, 6144, 1024, 8, True)


def _paged_case(seed, lens_q, lens_k, hq, hkv, num_pages=40):
    """Random q for the new tokens and a cache holding every sequence's lk keys in shuffled pages."""
    g = torch.Generator().manual_seed(seed)
    d = 128
    b = len(lens_q)
    kc, vc = rnd(num_pages, 256, hkv, d, seed=seed + 1), rnd(num_pages, 256, hkv, d, seed=seed + 2)
    perm = torch.randperm(num_pages, generator=g).tolist()
    tables, used = [], 0
    for lk in lens_k:
        nb = (lk + 255) // 256
        tables.append(perm[used:used + nb])
        used += nb
    assert used <= num_pages
    width = max(1, max(len(t) for t in tables))
    bt = torch.tensor([t + [-1] * (width - len(t)) for t in tables], dtype=torch.int32)
    q = rnd(sum(lens_q), hq, d, seed=seed + 3)
    cu_q = torch.tensor([0] + list(torch.tensor(lens_q).cumsum(0)), dtype=torch.int32)
    cu_k = torch.tensor([0] + list(torch.tensor(lens_k).cumsum(0)), dtype=torch.int32)
    return q, kc, vc, bt, cu_q, cu_k


@pytest.mark.parametrize("hq,hkv", [(16, 8), (8, 8), (8, 2), (10, 2), (8, 1)])
@pytest.mark.parametrize("splits", [1, 0, 5])
def test_attn_decode_vs_oracle(ops, hq, hkv, splits):
    lens_k = [1, 255, 256, 257, 1000, 64, 0, 700]
    b = len(lens_k)
    q, kc, vc, bt, _, _ = _paged_case(30 + hq, [1] * b, lens_k, hq, hkv)
    ctx = torch.tensor(lens_k, dtype=torch.int32)
    scale = 128 ** -0.5
    ref = R.attention_decode(q, kc, vc, ctx, bt, scale)
    got = ops.attn_decode(dev(q), dev(kc), dev(vc), dev(ctx), dev(bt), scale, num_splits=splits)
    assert_close(got, ref, 1.5e-2, 2e-2)
    assert float(got[6].abs().max()) == 0.0             # padded row (length 0) -> zeros


@pytest.mark.parametrize("hq,hkv", [(16, 8), (4, 4), (8, 2), (10, 2), (8, 1)])
@pytest.mark.parametrize("splits", [1, 5])
def test_attn_decode_fused_qkv_equals_sequence(ops, hq, hkv, splits):
    """One launch (norm + RoPE + store + attention from raw qkv) == qknorm_rope_store then attn_decode, bit for bit."""
    d = 128
    lens_k = [1, 255, 256, 257, 1000, 64, 0, 700]
    b = len(lens_k)
    _, kc, vc, bt, _, _ = _paged_case(70 + hq, [1] * b, lens_k, hq, hkv)
    qkv = rnd(b, (hq + 2 * hkv) * d, seed=71, scale=1.5)
    qw, kw = (rnd(d, seed=72).float() * 0.2 + 1).to(BF), (rnd(d, seed=73).float() * 0.2 + 1).to(BF)
    ctx = torch.tensor(lens_k, dtype=torch.int32)
    pos = (ctx.long() - 1).clamp(min=0)
    slots = torch.tensor([int(bt[i, (n - 1) // 256]) * 256 + (n - 1) % 256 if n > 0 else -1 for i, n in enumerate(lens_k)],
                         dtype=torch.int32)
    table = dev(R.rope_table(128, 4096, 1e6).squeeze(1).contiguous())
    scale = d ** -0.5
    # sequence of two launches
    qkv_a, kca, vca = dev(qkv.clone()), dev(kc.clone()), dev(vc.clone())
    ops.qknorm_rope_store_(qkv_a, hq, hkv, dev(qw), dev(kw), 1e-6, dev(pos), table, kca, vca, dev(slots))
    ref = ops.attn_decode(qkv_a[:, :hq * d].view(b, hq, d), kca, vca, dev(ctx), dev(bt), scale, num_splits=splits)
    # fused launch
    qkv_b, kcb, vcb = dev(qkv.clone()), dev(kc.clone()), dev(vc.clone())
    got = ops.attn_decode_fused_qkv(qkv_b, hq, hkv, dev(qw), dev(kw), 1e-6, dev(pos), table, kcb, vcb, dev(ctx), dev(bt),
                                    scale, num_splits=splits)
    assert_exact(got, ref)
    assert_exact(kcb, kca)
    assert_exact(vcb, vca)
    assert_exact(qkv_b.cpu(), qkv)                       # raw projection left untouched
    assert float(got[6].abs().max()) == 0.0


def test_attn_decode_merge_kernels_agree(ops):
    """The range merge has a few-rows and a many-rows kernel (more than 2048 output rows); they must give the same
    bits for the same sequence: 8 sequences alone vs the same 8 at the front of a batch of 160."""
    hq, hkv = 16, 8
    lens8 = [1, 255, 256, 257, 1000, 64, 0, 700]
    lens_k = lens8 + [300] * 152
    b = len(lens_k)
    q, kc, vc, bt, _, _ = _paged_case(91, [1] * b, lens_k, hq, hkv, num_pages=2 * b + 8)
    ctx = torch.tensor(lens_k, dtype=torch.int32)
    scale = 128 ** -0.5
    for splits in (5, 16):
        big = ops.attn_decode(dev(q), dev(kc), dev(vc), dev(ctx), dev(bt), scale, num_splits=splits)
        small = ops.attn_decode(dev(q[:8]), dev(kc), dev(vc), dev(ctx[:8]), dev(bt[:8]), scale, num_splits=splits)
        assert_exact(big[:8], small.cpu())
        ref = R.attention_decode(q[:16], kc, vc, ctx[:16], bt[:16], scale)
        assert_close(big[:16], ref, 1.5e-2, 2e-2)


def test_attn_decode_adaptive_ranges(ops):
    """Long and short sequences in one launch: 16 ranges offered, at least 512 keys each.  Sequences of up to 512 keys
    take one range and must equal the unsplit launch bit for bit; longer ones are merged from partial results."""
    hq, hkv, d = 16, 8, 128
    lens_k = [100, 8192, 513, 0, 2049, 512, 1, 3000]
    b = len(lens_k)
    q, kc, vc, bt, _, _ = _paged_case(93, [1] * b, lens_k, hq, hkv, num_pages=64)
    ctx = torch.tensor(lens_k, dtype=torch.int32)
    scale = d ** -0.5
    ref = R.attention_decode(q, kc, vc, ctx, bt, scale)
    one = ops.attn_decode(dev(q), dev(kc), dev(vc), dev(ctx), dev(bt), scale, num_splits=1)
    got = ops.attn_decode(dev(q), dev(kc), dev(vc), dev(ctx), dev(bt), scale, num_splits=16, min_range=512)
    assert_close(got, ref, 1.5e-2, 2e-2)
    for i, n in enumerate(lens_k):
        if n <= 512:
            assert torch.equal(got[i], one[i]), f"sequence of {n} keys changed under the long-context plan"
    assert float(got[3].abs().max()) == 0.0
    # fused variant: same bits as norm/rope/store followed by the plain kernel, K/V rows stored by the right range
    qkv = rnd(b, (hq + 2 * hkv) * d, seed=94, scale=1.5)
    qw, kw = (rnd(d, seed=95).float() * 0.2 + 1).to(BF), (rnd(d, seed=96).float() * 0.2 + 1).to(BF)
    pos = (ctx.long() - 1).clamp(min=0)
    slots = torch.tensor([int(bt[i, (n - 1) // 256]) * 256 + (n - 1) % 256 if n > 0 else -1 for i, n in enumerate(lens_k)], dtype=torch.int32)
    table = dev(R.rope_table(128, 16384, 1e6).squeeze(1).contiguous())
    qkv_a, kca, vca = dev(qkv.clone()), dev(kc.clone()), dev(vc.clone())
    ops.qknorm_rope_store_(qkv_a, hq, hkv, dev(qw), dev(kw), 1e-6, dev(pos), table, kca, vca, dev(slots))
    sep = ops.attn_decode(qkv_a[:, :hq * d].view(b, hq, d), kca, vca, dev(ctx), dev(bt), scale, num_splits=16, min_range=512)
    kcb, vcb = dev(kc.clone()), dev(vc.clone())
    fused = ops.attn_decode_fused_qkv(dev(qkv.clone()), hq, hkv, dev(qw), dev(kw), 1e-6, dev(pos), table, kcb, vcb, dev(ctx), dev(bt),
                                      scale, num_splits=16, min_range=512)
    assert_exact(fused, sep)
    assert_exact(kcb, kca)
    assert_exact(vcb, vca)


@pytest.mark.parametrize("hq,hkv", [(40, 8), (4, 4), (32, 4)])
def test_attn_decode_work_list_group_sizes(ops, hq, hkv):
    """The work list with 5, 1 and 8 query heads per kv head, 40 sequences of which three are long."""
    d = 128
    g = torch.Generator().manual_seed(hq)
    lens_k = torch.randint(1, 300, (40,), generator=g).tolist()
    lens_k[3], lens_k[17], lens_k[39] = 4000, 1531, 2048
    b = len(lens_k)
    q, kc, vc, bt, _, _ = _paged_case(7, [1] * b, lens_k, hq, hkv, num_pages=sum((n + 255) // 256 for n in lens_k) + 1)
    ctx = torch.tensor(lens_k, dtype=torch.int32)
    scale = d ** -0.5
    plan = ops.plan_decode_items(lens_k, max_ranges=16, range_keys=256, capacity=b + 64)
    assert int(plan["seq_active"].sum()) > b + 20
    items = ops.upload_decode_items(plan, "cuda")
    ref = R.attention_decode(q, kc, vc, ctx, bt, scale)
    one = ops.attn_decode(dev(q), dev(kc), dev(vc), dev(ctx), dev(bt), scale, num_splits=1)
    got = ops.attn_decode(dev(q), dev(kc), dev(vc), dev(ctx), dev(bt), scale, items=items)
    assert_close(got, ref, 1.5e-2, 2e-2)
    single = torch.from_numpy(plan["seq_active"] == 1)
    assert torch.equal(got.cpu()[single], one.cpu()[single])


@pytest.mark.parametrize("capacity", [0, 24, 11])
def test_attn_decode_work_list(ops, capacity):
    """Key ranges named one by one (plan_decode_items): only the long sequences are split.  One-range sequences equal
    the unsplit launch bit for bit, the rest are merged from partial results; a list squeezed into a fixed capacity
    (wider ranges, unused slots) gives the same answers; the fused variant equals norm/rope/store + the plain one."""
    hq, hkv, d = 16, 8, 128
    lens_k = [100, 8192, 513, 0, 2049, 512, 1, 3000]
    b = len(lens_k)
    q, kc, vc, bt, _, _ = _paged_case(93, [1] * b, lens_k, hq, hkv, num_pages=64)
    ctx = torch.tensor(lens_k, dtype=torch.int32)
    scale = d ** -0.5
    plan = ops.plan_decode_items(lens_k, max_ranges=8, range_keys=512, capacity=capacity)
    covered = {(int(s), int(r)) for s, r in zip(plan["item_seq"], plan["item_range"]) if s >= 0}
    assert covered == {(s, r) for s in range(b) for r in range(int(plan["seq_active"][s]))}
    assert all(int(a) * int(c) >= n and int(a) <= 8 for a, c, n in zip(plan["seq_active"], plan["seq_chunk"], lens_k))
    if capacity:
        assert plan["item_seq"].size == capacity
    else:
        assert int(plan["seq_active"][1]) == 8 and int(plan["seq_active"][2]) == 1
    items = ops.upload_decode_items(plan, "cuda")
    ref = R.attention_decode(q, kc, vc, ctx, bt, scale)
    one = ops.attn_decode(dev(q), dev(kc), dev(vc), dev(ctx), dev(bt), scale, num_splits=1)
    got = ops.attn_decode(dev(q), dev(kc), dev(vc), dev(ctx), dev(bt), scale, items=items)
    assert_close(got, ref, 1.5e-2, 2e-2)
    for i in range(b):
        if int(plan["seq_active"][i]) == 1:
            assert torch.equal(got[i], one[i]), f"sequence of {lens_k[i]} keys changed under the work list"
    assert float(got[3].abs().max()) == 0.0
    qkv = rnd(b, (hq + 2 * hkv) * d, seed=94, scale=1.5)
    qw, kw = (rnd(d, seed=95).float() * 0.2 + 1).to(BF), (rnd(d, seed=96).float() * 0.2 + 1).to(BF)
    pos = (ctx.long() - 1).clamp(min=0)
    slots = torch.tensor([int(bt[i, (n - 1) // 256]) * 256 + (n - 1) % 256 if n > 0 else -1 for i, n in enumerate(lens_k)], dtype=torch.int32)
    table = dev(R.rope_table(128, 16384, 1e6).squeeze(1).contiguous())
    qkv_a, kca, vca = dev(qkv.clone()), dev(kc.clone()), dev(vc.clone())
    ops.qknorm_rope_store_(qkv_a, hq, hkv, dev(qw), dev(kw), 1e-6, dev(pos), table, kca, vca, dev(slots))
    sep = ops.attn_decode(qkv_a[:, :hq * d].view(b, hq, d), kca, vca, dev(ctx), dev(bt), scale, items=items)
    kcb, vcb = dev(kc.clone()), dev(vc.clone())
    fused = ops.attn_decode_fused_qkv(dev(qkv.clone()), hq, hkv, dev(qw), dev(kw), 1e-6, dev(pos), table, kcb, vcb, dev(ctx), dev(bt),
                                      scale, items=items)
    assert_exact(fused, sep)
    assert_exact(kcb, kca)
    assert_exact(vcb, vca)


def test_attn_decode_strided_q_long_context(ops):
    hq, hkv, d = 16, 8, 128
    lens_k = [8192, 3000]
    q, kc, vc, bt, _, _ = _paged_case(41, [1, 1], lens_k, hq, hkv, num_pages=48)
    qkv = torch.zeros(2, (hq + 2 * hkv) * d, dtype=BF)
    qkv[:, :hq * d] = q.reshape(2, -1)
    ctx = torch.tensor(lens_k, dtype=torch.int32)
    scale = d ** -0.5
    ref = R.attention_decode(q, kc, vc, ctx, bt, scale)
    qv = dev(qkv)[:, :hq * d].view(2, hq, d)
    for splits in (1, 16, 64):
        got = ops.attn_decode(qv, dev(kc), dev(vc), dev(ctx), dev(bt), scale, num_splits=splits)
        assert_close(got, ref, 1.5e-2, 2e-2)


VARLEN_CASES = [
    # (lens_q, lens_k): fresh prefill, chunk crossing pages, prefix hit, decode rows in a mixed batch
    ([5, 1, 9], [5, 1, 9]),
    ([64, 65, 31, 32, 33], [64, 65, 31, 32, 33]),
    ([300], [300]),
    ([44, 1, 257, 1], [300, 700, 513, 1]),
    ([1024], [2048]),
    ([100, 3], [100 + 256, 259]),
]


@pytest.mark.parametrize("hq,hkv", [(16, 8), (4, 4), (8, 2), (10, 2), (8, 1)])
@pytest.mark.parametrize("case", range(len(VARLEN_CASES)))
def test_attn_varlen_paged_vs_oracle(ops, hq, hkv, case):
    lens_q, lens_k = VARLEN_CASES[case]
    q, kc, vc, bt, cu_q, cu_k = _paged_case(50 + case, lens_q, lens_k, hq, hkv)
    scale = 128 ** -0.5
    ref = R.attention_varlen(q, None, None, cu_q, cu_k, scale, kc, vc, bt)
    # key-range splits: heuristic (0), none (1), and counts that leave some ranges empty for short sequences
    for splits in (0, 1, 3, 8):
        got = ops.attn_varlen(dev(q), dev(kc), dev(vc), dev(cu_q), dev(cu_k), max(lens_q), scale, dev(bt),
                              max_seqlen_k=max(lens_k), num_splits=splits)
        assert_close(got, ref, 1.5e-2, 2e-2)


@pytest.mark.parametrize("hq,hkv", [(16, 8), (10, 2)])
def test_attn_varlen_long_cut_launch_vs_oracle(ops, hq, hkv):
    """Key ranges over 6144+ keys: such a launch walks its grid kv-head-major like an uncut one (shorter cut launches
    walk it query-block-major); both orders must cover every (query block, head, range) once."""
    lens_q, lens_k = [130, 40], [6500, 6210]
    q, kc, vc, bt, cu_q, cu_k = _paged_case(77, lens_q, lens_k, hq, hkv, num_pages=56)
    scale = 128 ** -0.5
    ref = R.attention_varlen(q, None, None, cu_q, cu_k, scale, kc, vc, bt)
    for splits in (2, 5):
        got = ops.attn_varlen(dev(q), dev(kc), dev(vc), dev(cu_q), dev(cu_k), max(lens_q), scale, dev(bt),
                              max_seqlen_k=max(lens_k), num_splits=splits)
        assert_close(got, ref, 1.5e-2, 2e-2)


@pytest.mark.parametrize("hq,hkv", [(16, 8), (8, 1)])
def test_attn_varlen_packed_strided_vs_oracle(ops, hq, hkv):
    """Mode without a block table: K/V are the packed new tokens, given as strided views of qkv."""
    d = 128
    lens = [70, 1, 200, 33]
    t = sum(lens)
    qkv = rnd(t, (hq + 2 * hkv) * d, seed=60)
    q, k, v = qkv.split([hq * d, hkv * d, hkv * d], dim=-1)
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    scale = d ** -0.5
    ref = R.attention_varlen(q.reshape(t, hq, d), k.reshape(t, hkv, d), v.reshape(t, hkv, d), cu, cu, scale)
    qd, kd, vd = dev(qkv).split([hq * d, hkv * d, hkv * d], dim=-1)
    got = ops.attn_varlen(qd.view(t, hq, d), kd.view(t, hkv, d), vd.view(t, hkv, d), dev(cu), dev(cu), max(lens), scale)
    assert_close(got, ref, 1.5e-2, 2e-2)


def test_attn_integer_exact_layout_check(ops):
    """Small-integer Q/K/V with an ASYMMETRIC V: a swapped fragment map cannot pass.
    One key per query is made dominant so softmax is one-hot up to e^-30 and the
    output must equal that V row exactly."""
    hq, hkv, d = 2, 1, 128
    lq = lk = 96
    k = torch.zeros(lk, hkv, d)
    q = torch.zeros(lq, hq, d)
    for j in range(lk):
        k[j, 0, j % d] = 8.0
    target = [(i * 7) % (i + 1) for i in range(lq)]          # any key <= i
    for i in range(lq):
        for h in range(hq):
            tj = target[i] if h == 0 else (i - target[i])
            q[i, h, tj % d] = 8.0 * math.sqrt(d) / 2
    v = (torch.arange(lk * d).reshape(lk, 1, d) % 251).float() - 125
    cu = torch.tensor([0, lq], dtype=torch.int32)
    out = ops.attn_varlen(dev(q.to(BF)), dev(k.to(BF)), dev(v.to(BF)), dev(cu), dev(cu), lq, d ** -0.5).cpu().float()
    for i in range(lq):
        # the other keys leak e^-32 each: far below half a bf16 ulp of any non-zero integer
        assert torch.allclose(out[i, 0], v[target[i], 0], rtol=0, atol=1e-6), f"query {i} head 0"
        assert torch.allclose(out[i, 1], v[i - target[i], 0], rtol=0, atol=1e-6), f"query {i} head 1"


def test_argmax_and_sampling(ops):
    g = torch.Generator().manual_seed(70)
    logits = torch.randn(6, 151936, generator=g).to(BF)
    logits[2, 777] = logits[2, 5000] = 9.0                   # tie -> lowest index
    assert_exact(ops.argmax(dev(logits)), R.greedy(logits))
    lf = torch.randn(3, 1000, generator=g) * 2
    assert_exact(ops.argmax(dev(lf)), R.greedy(lf))
    # explicit noise: identical to the oracle's deterministic form
    temps = torch.tensor([0.6, 1.0, 2.0])
    noise = torch.empty(3, 1000).exponential_(1, generator=g)
    assert_exact(ops.sample_with_noise(dev(lf), dev(temps), dev(noise)), R.sample_with_noise(lf, temps, noise))
    # temperature -> 0 collapses to greedy
    cold = torch.full((6,), 1e-9)
    untied = logits.clone()
    untied[2, 5000] = 8.0                                    # an exact tie is split by the noise, as in the reference
    assert_exact(ops.sample(dev(untied), dev(cold), seed=1, offset=0), R.greedy(untied))
    # distribution: chi-square of 20000 draws from a 6-way distribution, temperature 0.7
    base = torch.tensor([2.0, 1.0, 0.5, 0.0, -1.0, -3.0])
    n = 20000
    rows = base.repeat(n, 1).contiguous()
    t = torch.full((n,), 0.7)
    draws = ops.sample(dev(rows), dev(t), seed=1234, offset=5).cpu()
    p = torch.softmax(base / 0.7, 0)
    counts = torch.bincount(draws, minlength=6).float()
    chi2 = float(((counts - n * p) ** 2 / (n * p)).sum())
    assert chi2 < 25.0, chi2                                  # 5 dof: P(chi2 > 25) ~ 1e-4
    again = ops.sample(dev(rows), dev(t), seed=1234, offset=5).cpu()
    assert torch.equal(draws, again)                          # counter-based: reproducible
    other = ops.sample(dev(rows), dev(t), seed=1234, offset=6).cpu()
    assert not torch.equal(draws, other)


def test_sample_noise_is_never_zero(ops):
    """The generated Exp(1) noise must stay strictly positive: a zero would make -log(E) = +inf and hand the row to a
    random column whatever its logit (2^-24 per element before the fix: about one token in a hundred at this
    vocabulary).  19 million draws per call, four calls; column 7 leads by a margin (z/T = 200) that no finite noise
    term can overturn (-log E <= 17 for E >= 2^-24), so every sampled id must be 7."""
    rows, vocab = 128, 151936
    logits = torch.zeros(rows, vocab, dtype=torch.float32)
    logits[:, 7] = 10.0
    temps = torch.full((rows,), 0.05)
    dl, dt = dev(logits), dev(temps)
    for offset in range(4):
        ids = ops.sample(dl, dt, seed=1234, offset=offset)
        assert ids.cpu().tolist() == [7] * rows, f"offset {offset}: {(ids.cpu() != 7).sum().item()} rows left the leading column"


def test_attn_decode_in_launch_merge_equals_merge_launch(ops, switches):
    """With NVH_DECODE_MERGE_IN_LAUNCH=1 key ranges are merged by the last workgroup of a (sequence, kv head) to arrive,
    inside the attention launch; it must give the bits of the separate merge launch (the default).  Checked on every launch of a run of
    back-to-back launches with other kernels between them (the hand-off between workgroups has to hold under load and
    with warm caches), on uneven contexts, for three group sizes, the fused-qkv variant and the work list."""
    import os
    d = 128
    for hq, hkv, splits in ((16, 8, 8), (10, 2, 5), (8, 1, 16)):
        g = torch.Generator().manual_seed(hq)
        lens_k = torch.randint(1, 3000, (48,), generator=g).tolist()
        lens_k[5], lens_k[11], lens_k[40] = 0, 1, 4096
        b = len(lens_k)
        q, kc, vc, bt, _, _ = _paged_case(200 + hq, [1] * b, lens_k, hq, hkv, num_pages=sum((n + 255) // 256 for n in lens_k) + 1)
        dq, dkc, dvc, dbt = dev(q), dev(kc), dev(vc), dev(bt)
        ctx = dev(torch.tensor(lens_k, dtype=torch.int32))
        scale = d ** -0.5
        want = ops.attn_decode(dq, dkc, dvc, ctx, dbt, scale, num_splits=splits).clone()
        x, w = dev(rnd(256, 1024, seed=1)), dev(rnd(1024, 1024, seed=2, scale=0.05))
        outs = []
        switches("NVH_DECODE_MERGE_IN_LAUNCH", 1)
        for it in range(12):
            outs.append(ops.attn_decode(dq, dkc, dvc, ctx, dbt, scale, num_splits=splits))
            ops.linear(x, w)                                  # other work between the launches
        switches("NVH_DECODE_MERGE_IN_LAUNCH", -1)
        for it, got in enumerate(outs):
            assert torch.equal(got, want), f"launch {it}: {int((got != want).sum())} elements differ (hq {hq}, hkv {hkv})"
        ref = R.attention_decode(q, kc, vc, torch.tensor(lens_k, dtype=torch.int32), bt, scale)
        assert_close(want, ref, 1.5e-2, 2e-2)
        # work list: the same comparison
        plan = ops.plan_decode_items(lens_k, max_ranges=16, range_keys=256, capacity=b + 64)
        items = ops.upload_decode_items(plan, "cuda")
        want_items = ops.attn_decode(dq, dkc, dvc, ctx, dbt, scale, items=items).clone()
        switches("NVH_DECODE_MERGE_IN_LAUNCH", 1)
        for it in range(4):
            assert torch.equal(ops.attn_decode(dq, dkc, dvc, ctx, dbt, scale, items=items), want_items)
        switches("NVH_DECODE_MERGE_IN_LAUNCH", -1)
