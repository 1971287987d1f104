"""Worker of tests/test_comm_gpu.py: one rank of a 2..4-rank job whose ranks all sit on cuda:0 (the test box has
one GPU; the peer mappings then cross process boundaries but not devices).  Control plane: gloo on 127.0.0.1."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-v1_amd"))

from nanovllm import comm  # noqa: E402


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", world_size=world, rank=rank)
    c = comm.XgmiComm(rank, world, 8 << 20)
    if not c.self_test():
        raise RuntimeError("self-test: exchange timed out" if c.self_test_timed_out else "self-test against gloo: wrong sums")

    def expect(n, seed):
        parts = [torch.randn(n, generator=torch.Generator().manual_seed(seed * 100 + r)).to(torch.bfloat16) for r in range(world)]
        acc = torch.zeros(n, dtype=torch.float32)
        for p in parts:                      # rank order, fp32 accumulation, one rounding
            acc += p.float()
        return parts[rank].cuda(), acc.to(torch.bfloat16)

    # one-shot sizes (incl. tails that leave blocks partly empty), the two-shot boundary and beyond
    for i, n in enumerate([8, 16, 1000 * 8, 1024, 256 * 1024, 512 * 1024, 512 * 1024 + 8, 1024 * 1024 + 24, 4 << 20]):
        x, ref = expect(n, i)
        got = c.all_reduce(x.clone())
        assert torch.equal(got.cpu(), ref), f"n={n}: {(got.cpu() != ref).sum()} mismatches"
        # producer wrote straight into the staging area
        buf = c.buffer(n)
        buf.copy_(x)
        out = torch.empty_like(x)
        c.all_reduce(buf, out)
        assert torch.equal(out.cpu(), ref), f"in-place n={n}"
    # back-to-back calls reuse the staging area: the end-of-call barrier must protect it
    xs = [expect(64 * 1024, 50 + k) for k in range(16)]
    outs = [c.all_reduce(x.clone()) for x, _ in xs]
    for (x, ref), got in zip(xs, outs):
        assert torch.equal(got.cpu(), ref)
    # graph capture + replays (epochs live in device memory)
    x, ref = expect(256 * 1024, 77)
    static = x.clone()
    c.all_reduce(static.clone())                       # warm
    torch.cuda.synchronize()
    dist.barrier()
    g = torch.cuda.CUDAGraph()
    res = torch.empty_like(static)
    with torch.cuda.graph(g):
        c.all_reduce(static, res)
    for k in range(3):
        x2, ref2 = expect(256 * 1024, 80 + k)
        static.copy_(x2)
        g.replay()
        assert torch.equal(res.cpu(), ref2), f"graph replay {k}"
    # all-reduce fused with residual add + RMSNorm == the two separate calls, bit for bit
    from nanovllm import ops
    for rows, dim in ((1, 1024), (37, 1024), (256, 1024), (5, 5120), (3, 8192)):
        x, _ = expect(rows * dim, 300 + rows)
        x = x.view(rows, dim)
        g = torch.Generator().manual_seed(7)
        res = torch.randn(rows, dim, generator=g).to(torch.bfloat16).cuda()
        w = (torch.randn(dim, generator=g) * 0.1 + 1).to(torch.bfloat16).cuda()
        summed = c.all_reduce(x.clone())
        res_a = res.clone()
        y_a, _ = ops.add_rmsnorm(summed, res_a, w, 1e-6)
        res_b = res.clone()
        y_b = c.all_reduce_add_rmsnorm(x.clone(), res_b, w, 1e-6)
        assert torch.equal(y_a, y_b) and torch.equal(res_a, res_b), f"fused exchange {rows}x{dim}"
        buf = c.buffer(rows * dim).view(rows, dim)          # producer wrote into the staging area
        buf.copy_(x)
        res_c = res.clone()
        y_c = c.all_reduce_add_rmsnorm(buf, res_c, w, 1e-6)
        assert torch.equal(y_a, y_c) and torch.equal(res_a, res_c), f"fused exchange in place {rows}x{dim}"
    # logits gather: column-wise concatenation on rank 0
    for rows, cols in ((1, 8), (3, 18992), (64, 1000 * 8)):
        g = torch.Generator().manual_seed(900 + rows)
        full = torch.randn(rows, world * cols, generator=g).to(torch.bfloat16)
        mine = full[:, rank * cols:(rank + 1) * cols].contiguous().cuda()
        got = c.gather_cols(mine, 0)
        if rank == 0:
            assert torch.equal(got.cpu(), full), f"gather {rows}x{cols}"
        else:
            assert got is None
    # argument checks
    try:
        c.all_reduce(torch.zeros(12, dtype=torch.bfloat16, device="cuda"))
        raise AssertionError("count % 8 != 0 accepted")
    except Exception as e:  # noqa: BLE001
        assert "multiple of 8" in str(e), e
    c.check()
    # timing (informational)
    x = torch.zeros(256 * 1024, dtype=torch.bfloat16, device="cuda")
    torch.cuda.synchronize()
    dist.barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(50):
        c.all_reduce(x)
    ev1.record()
    torch.cuda.synchronize()
    if rank == 0:
        print(f"allreduce 512 KiB x{world} ranks on one device: {ev0.elapsed_time(ev1) / 50 * 1000:.1f} us/call")
    c.check()
    dist.barrier()
    c.close()
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
