"""The CPU build of include/nanovllm_hip.h (oracle/cpu_abi.cpp -> oracle/_build/libnanovllm_cpu.so): the operator
entry points of the C ABI on host memory, bound with the very ctypes signatures the product uses for the HIP
library.  Pinned by the golden vectors the reference's own modules produced (tests/golden) and checked against the
Python restatement (oracle/ref_ops.py) where the reference holds no fixture.  CPU only; test infrastructure - the
product never loads this library."""
import ctypes
import subprocess

import pytest
import torch

from conftest import ROOT, bf16_ulp_diff
from oracle import ref_ops as R

BF = torch.bfloat16
SRC = ROOT / "oracle" / "cpu_abi.cpp"
LIB = ROOT / "oracle" / "_build" / "libnanovllm_cpu.so"
EXPORTS = ["nvh_version", "nvh_last_error", "nvh_device_check", "nvh_store_kvcache", "nvh_rmsnorm", "nvh_add_rmsnorm", "nvh_rope",
           "nvh_qknorm_rope_store", "nvh_silu_mul", "nvh_embedding", "nvh_gather_last_rows", "nvh_gemm_bf16_nt",
           "nvh_gemm_workspace_bytes", "nvh_splitk_reduce", "nvh_add_rmsnorm_gemm_bf16_nt", "nvh_attn_varlen", "nvh_attn_varlen_workspace_bytes", "nvh_attn_decode",
           "nvh_attn_decode_workspace_bytes", "nvh_sample_workspace_bytes", "nvh_argmax", "nvh_sample_with_noise"]


@pytest.fixture(scope="module")
def cpu():
    if not LIB.exists() or LIB.stat().st_mtime < SRC.stat().st_mtime:
        subprocess.run(["bash", str(ROOT / "oracle" / "build_cpu_abi.sh")], check=True)
    from nanovllm import _hip
    lib = ctypes.CDLL(str(LIB))
    for name in EXPORTS:
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = _hip.SIGNATURES[name]        # the product's own binding of the same header
    return lib


def ok(lib, rc):
    assert rc == 0, lib.nvh_last_error().decode()


def rnd(*shape, seed=0, scale=1.0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).to(BF)


def close_ulp(got, want, max_ulp=1, max_frac=0.02):
    d = bf16_ulp_diff(got, want)
    assert got.shape == want.shape and int(d.max()) <= max_ulp and float((d > 0).float().mean()) <= max_frac, \
        f"max ulp {int(d.max())}, differing {float((d > 0).float().mean()):.4f}"


def close_abs(got, want, atol=4e-3, rtol=1e-2):
    err = (got.float() - want.float()).abs()
    assert got.shape == want.shape and bool((err <= atol + rtol * want.float().abs()).all()), f"max err {float(err.max()):.4g}"


def test_exports_and_error_convention(cpu):
    assert cpu.nvh_version() >= 100
    assert cpu.nvh_device_check(0) == -19 and b"CPU build" in cpu.nvh_last_error()
    assert cpu.nvh_gemm_bf16_nt(None, 0, None, None, None, 0, 1, 8, 64, 1, None, None) == -22
    assert b"null pointer" in cpu.nvh_last_error()
    p = ctypes.c_void_p(16)
    assert cpu.nvh_rope(p, p, 0, 1, p, 0, 1, p, 1, 1, 64, None) == -38 and b"head_dim" in cpu.nvh_last_error()
    assert cpu.nvh_gemm_workspace_bytes(5, 1024, 4) == 4 * 5 * 1024 * 4


def test_norms_against_the_reference_vectors(cpu, f1):
    x, w = f1.bf("rms_2d_x").clone(), f1.bf("rms_2d_w")
    y = torch.empty_like(x)
    ok(cpu, cpu.nvh_rmsnorm(x.data_ptr(), 256, w.data_ptr(), y.data_ptr(), 256, 5, 256, 1e-6, None))
    close_ulp(y, f1.bf("rms_2d_y"), 1, 0.005)
    # per-head norm: rows of 128 inside [5, 4, 128], in place
    xh, wh = f1.bf("rms_heads_x").clone(), f1.bf("rms_heads_w")
    ok(cpu, cpu.nvh_rmsnorm(xh.data_ptr(), 128, wh.data_ptr(), xh.data_ptr(), 128, 20, 128, 1e-6, None))
    close_ulp(xh, f1.bf("rms_heads_y"), 1, 0.005)
    x, res, w = f1.bf("addrms_x").clone(), f1.bf("addrms_r").clone(), f1.bf("addrms_w")
    y = torch.empty_like(x)
    ok(cpu, cpu.nvh_add_rmsnorm(x.data_ptr(), res.data_ptr(), w.data_ptr(), y.data_ptr(), 7, 256, 1e-6, None, 0, None))
    assert torch.equal(res, f1.bf("addrms_res"))
    close_ulp(y, f1.bf("addrms_y"), 1, 0.005)


def test_rope_silu_embedding_rows_bit_exact(cpu, f1):
    table = R.rope_table(128, 40960, 1e6).squeeze(1).contiguous()
    q, k, pos = f1.bf("rope_q").clone(), f1.bf("rope_k").clone(), f1.t("rope_pos")
    ok(cpu, cpu.nvh_rope(pos.data_ptr(), q.data_ptr(), 4 * 128, 4, k.data_ptr(), 2 * 128, 2, table.data_ptr(), 40960, 8, 128, None))
    assert torch.equal(q, f1.bf("rope_qo")) and torch.equal(k, f1.bf("rope_ko"))
    x = f1.bf("silu_x")
    y = torch.empty(6, 256, dtype=BF)
    ok(cpu, cpu.nvh_silu_mul(x.data_ptr(), y.data_ptr(), 6, 256, None))
    close_ulp(y, f1.bf("silu_y"), 1, 0.002)
    ids, w = f1.t("emb_ids"), f1.bf("emb_w")
    out = torch.empty(5, 64, dtype=BF)
    ok(cpu, cpu.nvh_embedding(ids.data_ptr(), w.data_ptr(), out.data_ptr(), 5, 64, 0, 300, None))
    assert torch.equal(out, f1.bf("emb_y"))
    # a vocabulary shard: ids outside [100, 200) give zero rows
    ok(cpu, cpu.nvh_embedding(ids.data_ptr(), w[100:200].contiguous().data_ptr(), out.data_ptr(), 5, 64, 100, 100, None))
    assert torch.equal(out, R.embedding(ids, w[100:200], 100, 2))
    hidden, cu = f1.bf("head_hidden"), f1.t("head_cu")
    for tag in ("head_subset", "head_empty"):
        need = f1.t(tag + "_need")
        n = need.numel() or cu.numel() - 1
        rows = torch.empty(n, 64, dtype=BF)
        ok(cpu, cpu.nvh_gather_last_rows(hidden.data_ptr(), cu.data_ptr(), need.data_ptr() if need.numel() else None, rows.data_ptr(), n, 64, None))
        assert torch.equal(rows, R.last_token_rows(hidden, cu, need))
        logits = torch.empty(n, 300, dtype=BF)
        ok(cpu, cpu.nvh_gemm_bf16_nt(rows.data_ptr(), 64, w.data_ptr(), None, logits.data_ptr(), 300, n, 300, 64, 1, None, None))
        close_ulp(logits, f1.bf(tag + "_logits"))


def test_linear_and_k_slabs(cpu, f1):
    for tag in ("lin", "lin_bias"):
        x, w = f1.bf(tag + "_x"), f1.bf(tag + "_w")
        b = f1.bf(tag + "_b") if tag + "_b" in f1 else None
        y = torch.empty(9, 160, dtype=BF)
        ok(cpu, cpu.nvh_gemm_bf16_nt(x.data_ptr(), 192, w.data_ptr(), b.data_ptr() if b is not None else None, y.data_ptr(), 160, 9, 160, 192,
                                     1, None, None))
        close_ulp(y, f1.bf(tag + "_y"))
    # strided input rows, three K-slabs, the reduce and the norm that consumes slabs
    m, n, k = 7, 256, 320
    xw, w, res, g = rnd(m, k + 64, seed=1), rnd(n, k, seed=2, scale=0.05), rnd(m, n, seed=3), (rnd(n, seed=4).float() * 0.1 + 1).to(BF)
    slabs = torch.empty(3, m, n)
    assert cpu.nvh_gemm_workspace_bytes(m, n, 3) == slabs.numel() * 4
    ok(cpu, cpu.nvh_gemm_bf16_nt(xw.data_ptr(), k + 64, w.data_ptr(), None, None, 0, m, n, k, 3, slabs.data_ptr(), None))
    want = R.linear(xw[:, :k], w)
    y = torch.empty(m, n, dtype=BF)
    ok(cpu, cpu.nvh_splitk_reduce(slabs.data_ptr(), 3, None, y.data_ptr(), n, m, n, None))
    close_ulp(y, want)
    out, r2 = torch.empty(m, n, dtype=BF), res.clone()
    ok(cpu, cpu.nvh_add_rmsnorm(None, r2.data_ptr(), g.data_ptr(), out.data_ptr(), m, n, 1e-6, slabs.data_ptr(), 3, None))
    want_y, want_r = R.add_rms_norm(y, res, g, 1e-6)
    assert torch.equal(r2, want_r)
    close_ulp(out, want_y, 1, 0.01)
    assert cpu.nvh_gemm_bf16_nt(xw.data_ptr(), k + 64, w.data_ptr(), None, None, 0, m, n, k, 6, slabs.data_ptr(), None) == -22   # 5 k-tiles


def test_add_rmsnorm_inside_the_projection(cpu, f1):
    """The one-call form of residual add + RMSNorm + projection: the reference's add_rms_forward vector goes in, its residual
    comes out bit-exact in the SECOND buffer, the projection of its normalised rows within one ulp; with the SwiGLU epilogue
    against the oracle's silu_mul over the de-interleaved product; slabs as input; aliasing and sizes refused."""
    x, res, g = f1.bf("addrms_x"), f1.bf("addrms_r"), f1.bf("addrms_w")
    m, k = x.shape
    kp = (k + 63) // 64 * 64
    if kp != k or m > 32:       # the vector's shape does not fit the entry point: use a seeded one of the same kind
        m, k = 5, 256
        x, res, g = rnd(m, k, seed=11), rnd(m, k, seed=12), (rnd(k, seed=13).float() * 0.1 + 1).to(BF)
    n = 96
    w = rnd(n, k, seed=14, scale=0.05)
    want_x, want_r = R.add_rms_norm(x, res, g, 1e-6)
    y, r_out, r_in = torch.empty(m, n, dtype=BF), torch.empty(m, k, dtype=BF), res.clone()
    ok(cpu, cpu.nvh_add_rmsnorm_gemm_bf16_nt(x.data_ptr(), None, 0, r_in.data_ptr(), r_out.data_ptr(), g.data_ptr(), 1e-6, w.data_ptr(),
                                            y.data_ptr(), n, m, n, k, 0, None))
    assert torch.equal(r_out, want_r) and torch.equal(r_in, res)
    close_abs(y, R.linear(want_x, w))
    # SwiGLU over interleaved rows: block b = gate rows 16b.. then up rows 16b..
    inter = n // 2
    wi = torch.cat([torch.cat([w[16 * b:16 * b + 16], w[inter + 16 * b:inter + 16 * b + 16]]) for b in range(inter // 16)])
    ys = torch.empty(m, inter, dtype=BF)
    ok(cpu, cpu.nvh_add_rmsnorm_gemm_bf16_nt(x.data_ptr(), None, 0, r_in.data_ptr(), r_out.data_ptr(), g.data_ptr(), 1e-6, wi.data_ptr(),
                                            ys.data_ptr(), inter, m, n, k, 1, None))
    close_abs(ys, R.silu_mul(R.linear(want_x, w)))
    # two slabs that add up to x exactly
    slabs = torch.stack([x.float() * 0.5, x.float() * 0.5]).contiguous()
    y2 = torch.empty(m, n, dtype=BF)
    ok(cpu, cpu.nvh_add_rmsnorm_gemm_bf16_nt(None, slabs.data_ptr(), 2, r_in.data_ptr(), r_out.data_ptr(), g.data_ptr(), 1e-6, w.data_ptr(),
                                            y2.data_ptr(), n, m, n, k, 0, None))
    assert torch.equal(y2, y) and torch.equal(r_out, want_r)
    assert cpu.nvh_add_rmsnorm_gemm_bf16_nt(x.data_ptr(), None, 0, r_in.data_ptr(), r_in.data_ptr(), g.data_ptr(), 1e-6, w.data_ptr(),
                                           y.data_ptr(), n, m, n, k, 0, None) == -22
    assert cpu.nvh_add_rmsnorm_gemm_bf16_nt(x.data_ptr(), None, 0, r_in.data_ptr(), r_out.data_ptr(), g.data_ptr(), 1e-6, w.data_ptr(),
                                           y.data_ptr(), n, 65, n, k, 0, None) == -38


def test_qk_epilogue_and_cache_store(cpu):
    t, hq, hkv, d = 6, 4, 2, 128
    table = R.rope_table(d, 4096, 1e6)
    qkv = rnd(t, (hq + 2 * hkv) * d, seed=5)
    qw, kw = (rnd(d, seed=6).float() * 0.1 + 1).to(BF), (rnd(d, seed=7).float() * 0.1 + 1).to(BF)
    pos = torch.tensor([0, 1, 255, 256, 1000, 4095])
    slots = torch.tensor([3, -1, 700, 0, 1023, 512], dtype=torch.int32)
    kc, vc = torch.zeros(4, 256, hkv, d, dtype=BF), torch.zeros(4, 256, hkv, d, dtype=BF)
    q, k, v = qkv.split([hq * d, hkv * d, hkv * d], dim=-1)
    qn = R.rms_norm(q.reshape(t, hq, d), qw, 1e-6)
    kn = R.rms_norm(k.reshape(t, hkv, d), kw, 1e-6)
    qr, kr = R.rope(pos, qn, kn, table)
    want_k, want_v = torch.zeros_like(kc), torch.zeros_like(vc)
    R.store_kvcache(kr, v.reshape(t, hkv, d), want_k, want_v, slots)
    buf = qkv.clone()
    ok(cpu, cpu.nvh_qknorm_rope_store(buf.data_ptr(), buf.shape[1], hq, hkv, qw.data_ptr(), kw.data_ptr(), 1e-6, pos.data_ptr(),
                                      table.squeeze(1).contiguous().data_ptr(), 4096, kc.data_ptr(), vc.data_ptr(), slots.data_ptr(), t, None))
    close_ulp(buf[:, :hq * d].reshape(t, hq, d), qr, 1, 0.01)
    close_ulp(buf[:, hq * d:(hq + hkv) * d].reshape(t, hkv, d), kr, 1, 0.01)
    assert torch.equal(buf[:, (hq + hkv) * d:], v)
    assert torch.equal(vc, want_v) and torch.equal(kc.view(-1, hkv * d)[1], want_k.view(-1, hkv * d)[1])
    close_ulp(kc, want_k, 1, 0.01)


@pytest.mark.parametrize("hq,hkv", [(4, 2), (8, 1), (5, 1)])
def test_attention_packed_paged_and_decode(cpu, hq, hkv):
    d, scale = 128, 128 ** -0.5
    lq, lk = [5, 70, 1, 300], [5, 300, 257, 300]                 # no prefix, a prefix across a page edge, one query, exactly the keys
    cu_q = torch.tensor([0] + list(torch.tensor(lq).cumsum(0)), dtype=torch.int32)
    cu_k = torch.tensor([0] + list(torch.tensor(lk).cumsum(0)), dtype=torch.int32)
    q = rnd(int(cu_q[-1]), hq, d, seed=8)
    kc, vc = rnd(9, 256, hkv, d, seed=9), rnd(9, 256, hkv, d, seed=10)
    tables = torch.tensor([[8, -1], [3, 1], [0, 2], [5, 4]], dtype=torch.int32)
    out = torch.empty_like(q)
    ok(cpu, cpu.nvh_attn_varlen(q.data_ptr(), hq * d, kc.data_ptr(), vc.data_ptr(), 0, cu_q.data_ptr(), cu_k.data_ptr(), tables.data_ptr(), 2,
                                4, max(lq), max(lk), int(cu_q[-1]), hq, hkv, d, scale, out.data_ptr(), 1, None, None))
    want = R.attention_varlen(q, None, None, cu_q, cu_k, scale, kc, vc, tables)
    close_abs(out, want)
    # packed: K/V are views of a fused [T, (Hq + 2 Hkv) D] buffer (token stride = its row length)
    t = int(cu_q[-1])
    qkv = rnd(t, (hq + 2 * hkv) * d, seed=11)
    qv, kv_, vv = qkv[:, :hq * d], qkv[:, hq * d:(hq + hkv) * d], qkv[:, (hq + hkv) * d:]
    out2 = torch.empty(t, hq, d, dtype=BF)
    ok(cpu, cpu.nvh_attn_varlen(qv.data_ptr(), qkv.shape[1], kv_.data_ptr(), vv.data_ptr(), qkv.shape[1], cu_q.data_ptr(), cu_q.data_ptr(), None, 0,
                                4, max(lq), max(lq), t, hq, hkv, d, scale, out2.data_ptr(), 1, None, None))
    want2 = R.attention_varlen(qv.reshape(t, hq, d), kv_.reshape(t, hkv, d), vv.reshape(t, hkv, d), cu_q, cu_q, scale)
    close_abs(out2, want2)
    # decode: one query per sequence, a padded row (context 0) gives zeros
    ctx = torch.tensor([257, 0, 1, 512], dtype=torch.int32)
    qd = rnd(4, hq, d, seed=12)
    outd = torch.full((4, hq, d), 7.0, dtype=BF)
    dtab = torch.tensor([[8, 6], [7, 7], [0, 2], [5, 4]], dtype=torch.int32)
    ok(cpu, cpu.nvh_attn_decode(qd.data_ptr(), hq * d, kc.data_ptr(), vc.data_ptr(), ctx.data_ptr(), dtab.data_ptr(), 2, 4, hq, hkv, d, scale,
                                outd.data_ptr(), 1, 0, None, None))
    close_abs(outd, R.attention_decode(qd, kc, vc, ctx, dtab, scale))
    assert float(outd[1].abs().max()) == 0.0


def test_sampler_vectors_and_greedy_ties(cpu, f1):
    logits, temps, noise = f1.t("samp_logits"), f1.t("samp_temps"), f1.t("samp_noise")
    out = torch.empty(4, dtype=torch.int64)
    ok(cpu, cpu.nvh_sample_with_noise(logits.data_ptr(), 1, 300, temps.data_ptr(), noise.data_ptr(), out.data_ptr(), 4, 300, None, None))
    assert torch.equal(out, f1.t("samp_tok"))
    z = rnd(3, 50, seed=13)
    z[1, 7] = z[1, 31] = 9.0                                      # the lowest index of the maximum wins
    g = torch.empty(3, dtype=torch.int64)
    ok(cpu, cpu.nvh_argmax(z.data_ptr(), 0, 50, g.data_ptr(), 3, 50, None, None))
    assert torch.equal(g, R.greedy(z)) and int(g[1]) == 7
