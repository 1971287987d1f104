"""nvh_attn_decode_noalloc: the decode attention entry point that allocates nothing (its per-range partial
results live in a device array inside the library) must give the bits of nvh_attn_decode with a caller
workspace, call after call on the same GPU - the second call smaller than the first, so stale partial
results of the first call sit in the array while the second runs.  Also compared with the CPU oracle, and
sizes past the built-in array must be refused.

Collected last on purpose (file name): the round-end run stops at the first failure.
"""
import pytest
import torch

from oracle import ref_ops as R

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _rnd(*s, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*s, generator=g).to(BF)


def _case(seed, lens_k, hq, hkv, num_pages):
    g = torch.Generator().manual_seed(seed)
    kc, vc = _rnd(num_pages, 256, hkv, 128, seed=seed + 1), _rnd(num_pages, 256, hkv, 128, seed=seed + 2)
    perm = torch.randperm(num_pages, generator=g).tolist()
    tables, used = [], 0
    for lk in lens_k:
        nb = (lk + 255) // 256
        tables.append(perm[used:used + nb])
        used += nb
    assert used <= num_pages
    width = max(1, max(len(t) for t in tables))
    bt = torch.tensor([t + [-1] * (width - len(t)) for t in tables], dtype=torch.int32)
    return _rnd(len(lens_k), hq, 128, seed=seed + 3), kc, vc, bt, torch.tensor(lens_k, dtype=torch.int32)


def _both(hip, q, kc, vc, ctx, bt, splits, min_range):
    """(workspace entry point, no-allocation entry point) on the same device tensors, through the C ABI."""
    b, hq, d = q.shape
    hkv = kc.shape[2]
    scale = d ** -0.5
    st = torch.cuda.current_stream().cuda_stream
    common = (q.data_ptr(), q.stride(0), kc.data_ptr(), vc.data_ptr(), ctx.data_ptr(), bt.data_ptr(), bt.stride(0),
              b, hq, hkv, d, scale)
    ref = torch.full((b, hq, d), float("nan"), dtype=BF, device="cuda")
    ws = torch.empty(max(hip.lib().nvh_attn_decode_workspace_bytes(b, hq, d, splits) // 4, 1), dtype=torch.float32, device="cuda")
    hip.call("nvh_attn_decode", *common, ref.data_ptr(), splits, min_range, ws.data_ptr(), st)
    got = torch.full((b, hq, d), float("nan"), dtype=BF, device="cuda")
    torch.cuda.synchronize()
    # Two witnesses that the call allocates nothing: the device's own count of free bytes (hipMemGetInfo sees hipMalloc,
    # hipMallocAsync and pool growth by anyone in the process, not just torch's caching allocator) does not move, and the
    # call is accepted under stream capture, where an allocation or a synchronisation fails the capture.
    free_before = torch.cuda.mem_get_info()[0]
    before = torch.cuda.memory_allocated()
    hip.call("nvh_attn_decode_noalloc", *common, got.data_ptr(), splits, min_range, st)
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
    assert torch.cuda.mem_get_info()[0] == free_before
    captured = torch.full((b, hq, d), float("nan"), dtype=BF, device="cuda")
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        hip.call("nvh_attn_decode_noalloc", *common, captured.data_ptr(), splits, min_range, torch.cuda.current_stream().cuda_stream)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(captured, got), "the captured call replays other bits than the direct one"
    return ref.cpu(), got.cpu()


def test_noalloc_decode_two_calls_second_smaller():
    from nanovllm import _hip as hip
    hip.check(hip.lib().nvh_device_check(0), "nvh_device_check")
    hq, hkv = 16, 8
    # call 1: 12 sequences, 8 key ranges offered; long, short, page-edge, empty
    lens1 = [1, 255, 256, 257, 3000, 64, 0, 700, 2049, 513, 1000, 4096]
    q, kc, vc, bt, ctx = _case(5, lens1, hq, hkv, num_pages=80)
    dq, dkc, dvc, dbt, dctx = q.cuda(), kc.cuda(), vc.cuda(), bt.cuda(), ctx.cuda()
    ref1, got1 = _both(hip, dq, dkc, dvc, dctx, dbt, 8, 0)
    assert torch.equal(got1, ref1), f"{int((got1 != ref1).sum())} elements differ from nvh_attn_decode"
    oracle1 = R.attention_decode(q, kc, vc, ctx, bt, 128 ** -0.5)
    err = (got1.float() - oracle1.float()).abs()
    assert bool((err <= 1.5e-2 + 2e-2 * oracle1.float().abs()).all()), float(err.max())
    assert float(got1[6].abs().max()) == 0.0                     # padded row
    # call 2, right after, smaller: 5 of those sequences (other rows, other lengths per row), 4 ranges of >= 512 keys
    pick = [4, 1, 8, 6, 11]
    q2, bt2, ctx2 = q[pick].contiguous(), bt[pick].contiguous(), ctx[pick].contiguous()
    ref2, got2 = _both(hip, q2.cuda(), dkc, dvc, ctx2.cuda(), bt2.cuda(), 4, 512)
    assert torch.equal(got2, ref2), f"{int((got2 != ref2).sum())} elements differ from nvh_attn_decode (second call)"
    oracle2 = R.attention_decode(q2, kc, vc, ctx2, bt2, 128 ** -0.5)
    err = (got2.float() - oracle2.float()).abs()
    assert bool((err <= 1.5e-2 + 2e-2 * oracle2.float().abs()).all()), float(err.max())
    assert float(got2[3].abs().max()) == 0.0
    # one range per sequence uses no partial results at all
    ref3, got3 = _both(hip, q2.cuda(), dkc, dvc, ctx2.cuda(), bt2.cuda(), 1, 0)
    assert torch.equal(got3, ref3)


def test_noalloc_decode_rejects_oversize():
    from nanovllm import _hip as hip
    cap = hip.lib().nvh_attn_decode_noalloc_max_bytes()
    assert cap == 8 << 20
    hq, hkv, b, splits = 16, 8, 64, 32                           # 64 x 16 x 32 x 130 x 4 B = 17 MB > 8 MiB
    assert hip.lib().nvh_attn_decode_workspace_bytes(b, hq, 128, splits) > cap
    q, kc, vc, bt, ctx = _case(9, [10] * b, hq, hkv, num_pages=b)
    out = torch.zeros(b, hq, 128, dtype=BF, device="cuda")
    dq, dkc, dvc, dbt, dctx = q.cuda(), kc.cuda(), vc.cuda(), bt.cuda(), ctx.cuda()
    rc = hip.lib().nvh_attn_decode_noalloc(dq.data_ptr(), dq.stride(0), dkc.data_ptr(), dvc.data_ptr(), dctx.data_ptr(),
                                           dbt.data_ptr(), dbt.stride(0), b, hq, hkv, 128, 128 ** -0.5, out.data_ptr(), splits, 0,
                                           torch.cuda.current_stream().cuda_stream)
    assert rc == -22 and b"built-in" in hip.lib().nvh_last_error()
    torch.cuda.synchronize()
    assert float(out.abs().max()) == 0.0                         # nothing was launched
