#!/usr/bin/env python3
"""Micro-benchmarks of single entry points on the GPU (HIP-event timed, many iterations over
rotating buffers so nothing is served from the Infinity Cache by accident)."""
import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd"))
import torch
from nanovllm import ops

BF = torch.bfloat16


def timeit(fn, iters=50, warmup=5):
    """Per-call GPU time: `iters` calls captured into one hipGraph (no host launch cost inside the
    timed region), replayed 3 times; the best replay counts."""
    for i in range(warmup):
        fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(iters):
            fn(i)
    g.replay()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e))
    return best / iters * 1e3   # us


def gemm(shapes):
    """Weights rotate through > 512 MB of copies, so every launch streams them from HBM (cold
    Infinity Cache), as consecutive layers of a model do."""
    print(f"{'M':>6} {'N':>7} {'K':>6} {'split':>5} {'us':>9} {'TFLOP/s':>9} {'W GB/s':>9}")
    for m, n, k in shapes:
        copies = max(2, min(128, (600 << 20) // (n * k * 2) + 1))
        xs = [torch.randn(m, k, device="cuda").to(BF) for _ in range(min(copies, 8))]
        ws = [(torch.randn(n, k, device="cuda") * 0.05).to(BF) for _ in range(copies)]
        for split in (1, ops.suggest_split_k(m, n, k)):
            if split == 1:
                us = timeit(lambda i: ops.linear(xs[i % len(xs)], ws[i % copies]), iters=max(50, copies))
            else:
                us = timeit(lambda i: ops.linear_splitk(xs[i % len(xs)], ws[i % copies], split), iters=max(50, copies))
            print(f"{m:6d} {n:7d} {k:6d} {split:5d} {us:9.2f} {2 * m * n * k / us / 1e6:9.1f} {2 * n * k / us / 1e3:9.1f}")
            if split == 1 and ops.suggest_split_k(m, n, k) == 1:
                break


def gemm_vs_library(shapes):
    """Calibration only: the same shapes through torch's library GEMM (hipBLASLt/rocBLAS), to see what the
    hardware allows at these sizes.  Not used by the product."""
    print(f"{'M':>6} {'N':>7} {'K':>6} {'ours us':>9} {'TF':>7} {'lib us':>9} {'TF':>7}")
    for m, n, k in shapes:
        copies = max(2, min(128, (600 << 20) // (n * k * 2) + 1))
        xs = [torch.randn(m, k, device="cuda").to(BF) for _ in range(min(copies, 8))]
        ws = [(torch.randn(n, k, device="cuda") * 0.05).to(BF) for _ in range(copies)]
        ours = timeit(lambda i: ops.linear(xs[i % len(xs)], ws[i % copies]), iters=max(50, copies))
        lib = timeit(lambda i: torch.nn.functional.linear(xs[i % len(xs)], ws[i % copies]), iters=max(50, copies))
        f = 2 * m * n * k / 1e6
        print(f"{m:6d} {n:7d} {k:6d} {ours:9.2f} {f / ours:7.1f} {lib:9.2f} {f / lib:7.1f}")


def attn_prefill(cases, hq=16, hkv=8):
    print(f"{'Lq':>6} {'Lk':>6} {'us':>9} {'TFLOP/s':>9}")
    d = 128
    for lq, lk in cases:
        pages = (lk + 255) // 256
        kc = torch.randn(pages + 1, 256, hkv, d, device="cuda").to(BF)
        vc = torch.randn(pages + 1, 256, hkv, d, device="cuda").to(BF)
        q = torch.randn(lq, hq, d, device="cuda").to(BF)
        bt = torch.arange(pages, dtype=torch.int32, device="cuda").view(1, -1)
        cu_q = torch.tensor([0, lq], dtype=torch.int32, device="cuda")
        cu_k = torch.tensor([0, lk], dtype=torch.int32, device="cuda")
        vis = lq * (lk - lq) + lq * (lq + 1) // 2
        for sp in (0, 1, 2, 4, 8):
            us = timeit(lambda i: ops.attn_varlen(q, kc, vc, cu_q, cu_k, lq, d ** -0.5, bt, max_seqlen_k=lk, num_splits=sp), iters=20)
            print(f"{lq:6d} {lk:6d} splits={sp} {us:9.2f} {4 * hq * d * vis / us / 1e6:9.1f}")


def attn_decode(cases, hq=16, hkv=8, layers=8):
    print(f"{'B':>5} {'ctx':>6} {'splits':>6} {'us':>9} {'GB/s':>9}")
    d = 128
    for b, ctx in cases:
        pages = b * ((ctx + 255) // 256)
        kcs = [torch.randn(pages, 256, hkv, d, device="cuda").to(BF) for _ in range(layers)]
        vcs = [torch.randn(pages, 256, hkv, d, device="cuda").to(BF) for _ in range(layers)]
        q = torch.randn(b, hq, d, device="cuda").to(BF)
        bt = torch.arange(pages, dtype=torch.int32, device="cuda").view(b, -1)
        cl = torch.full((b,), ctx, dtype=torch.int32, device="cuda")
        sp = ops.suggest_decode_splits(b, hkv, ctx)
        us = timeit(lambda i: ops.attn_decode(q, kcs[i % layers], vcs[i % layers], cl, bt, d ** -0.5, num_splits=sp), iters=40)
        print(f"{b:5d} {ctx:6d} {sp:6d} {us:9.2f} {b * ctx * hkv * d * 4 / us / 1e3:9.1f}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="*", default=["gemm", "prefill", "decode"])
    a = ap.parse_args()
    if "gemm" in a.what:
        gemm([(1024, 4096, 1024), (1024, 6144, 1024), (1024, 1024, 2048), (1024, 1024, 3072),
              (256, 4096, 1024), (256, 6144, 1024), (256, 1024, 2048), (256, 1024, 3072), (256, 151936, 1024),
              (1, 4096, 1024), (1, 6144, 1024), (1, 1024, 2048), (1, 1024, 3072), (1, 151936, 1024),
              (16, 4096, 1024), (16, 1024, 3072), (64, 6144, 1024), (4096, 4096, 4096)])
    if "gemmlib" in a.what:
        gemm_vs_library([(1024, 4096, 1024), (1024, 6144, 1024), (1024, 1024, 2048), (1024, 1024, 3072),
                         (256, 4096, 1024), (256, 6144, 1024), (256, 1024, 2048), (256, 1024, 3072),
                         (4096, 4096, 4096), (8192, 8192, 8192)])
    if "prefill" in a.what:
        attn_prefill([(1024, 1024), (1024, 2048), (1024, 4096), (1024, 8192), (128, 128)])
    if "decode" in a.what:
        attn_decode([(256, 1024), (32, 1024), (1, 1024), (1, 8192), (64, 4096)])


def attn_decode_layout_probe():
    """Same bytes and workgroup count as B=256/Hkv=8, but every workgroup's K/V rows are contiguous
    (Hkv=1): tells how much the 2-KiB token stride of the [page,256,Hkv,D] layout costs."""
    d = 128
    for (b, hq, hkv, ctx) in ((256, 16, 8, 1024), (2048, 2, 1, 1024)):
        layers = 8
        pages = b * ((ctx + 255) // 256)
        kcs = [torch.randn(pages, 256, hkv, d, device="cuda").to(BF) for _ in range(layers)]
        vcs = [torch.randn(pages, 256, hkv, d, device="cuda").to(BF) for _ in range(layers)]
        q = torch.randn(b, hq, d, device="cuda").to(BF)
        bt = torch.arange(pages, dtype=torch.int32, device="cuda").view(b, -1)
        cl = torch.full((b,), ctx, dtype=torch.int32, device="cuda")
        us = timeit(lambda i: ops.attn_decode(q, kcs[i % layers], vcs[i % layers], cl, bt, d ** -0.5, num_splits=1), iters=40)
        print(f"B={b} Hq={hq} Hkv={hkv} ctx={ctx}: {us:9.2f} us {b * ctx * hkv * d * 4 / us / 1e3:9.1f} GB/s")


if __name__ == "__main__" and "layout" in sys.argv:
    attn_decode_layout_probe()
