"""Tensor-parallel all-reduce over xGMI peer mappings (csrc/comm.hip), with torch.distributed as the
control plane and as the fallback.

The reference sums row-parallel partial outputs with `dist.all_reduce` (NCCL ring; layers/linear.py:407-409,
layers/embed_head.py:128-129).  Here every rank exports one staging region through HIP IPC and a single
kernel per rank sums all peers' regions in rank order; the process group is only used to swap the IPC
handles at start-up and to cross-check the first result.  `all_reduce()` below is what the layers call:
it uses the peer path when one is installed and the message fits, `dist.all_reduce` otherwise.
"""
from __future__ import annotations

import ctypes
import os
import sys

import torch
import torch.distributed as dist

from . import _hip

_ACTIVE: "XgmiComm | None" = None


def _on_host(group=None) -> bool:
    """True when the torch.distributed backend works on host memory (gloo: CPU rehearsals, several ranks sharing
    one test GPU).  Device tensors are then staged through the host explicitly; gloo's own device-tensor path was
    seen to hang with two processes on one GPU."""
    return dist.get_backend(group) != "nccl"


def _dist_sum(t: torch.Tensor, group=None) -> torch.Tensor:
    """In-place sum over the ranks through torch.distributed (RCCL in deployment)."""
    if t.is_cuda and _on_host(group):
        h = t.cpu()
        dist.all_reduce(h, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, group=group)
    return t


class XgmiComm:
    def __init__(self, rank: int, world: int, max_bytes: int, group=None):
        """Collective: every rank of `group` must call it.  Raises HipError on EVERY rank if any rank could not
        allocate, export or map the exchange regions (the ranks agree on the outcome before anything is used)."""
        lib = _hip.lib()
        self.rank, self.world, self.group = rank, world, group
        self.max_bytes = (int(max_bytes) + 15) // 16 * 16
        self._h = None
        problem, blob = None, None
        try:
            handle = ctypes.c_void_p()
            _hip.check(lib.nvh_comm_create(rank, world, self.max_bytes, ctypes.byref(handle)), "nvh_comm_create")
            self._h = handle
            mine = ctypes.create_string_buffer(lib.nvh_comm_handle_bytes())
            _hip.check(lib.nvh_comm_get_handle(self._h, mine), "nvh_comm_get_handle")
            blob = mine.raw
        except _hip.HipError as e:
            problem = str(e)
        gathered: list = [None] * world
        dist.all_gather_object(gathered, blob, group=group)
        if problem is None and all(b is not None for b in gathered):
            try:
                _hip.check(lib.nvh_comm_open_peers(self._h, b"".join(gathered)), "nvh_comm_open_peers")
            except _hip.HipError as e:
                problem = str(e)
        elif problem is None:
            problem = "a peer could not create its exchange region"
        verdicts: list = [None] * world
        dist.all_gather_object(verdicts, problem, group=group)      # doubles as the barrier before first use
        bad = [f"rank {r}: {p}" for r, p in enumerate(verdicts) if p is not None]
        if bad:
            self.close()
            raise _hip.HipError("xGMI exchange unavailable (" + "; ".join(bad) + ")")
        self.buffer_ptr = int(lib.nvh_comm_buffer(self._h))

    def buffer(self, numel: int) -> torch.Tensor:
        """bf16 view of the first `numel` elements of this rank's staging area: a producer that writes its
        partial result here saves the staging copy (then pass the view itself to all_reduce with a separate out)."""
        assert 0 < numel * 2 <= self.max_bytes

        class _Raw:
            __cuda_array_interface__ = {"shape": (numel,), "typestr": "<u2", "data": (self.buffer_ptr, False), "version": 3}

        return torch.as_tensor(_Raw(), device="cuda").view(torch.bfloat16)

    @property
    def handle(self) -> int:
        return self._h.value

    def all_reduce(self, x: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        """Sum of x over the ranks.  x: contiguous bf16, numel % 8 == 0.  Returns `out` (default: x itself)."""
        assert x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous()
        out = x if out is None else out
        _hip.call("nvh_allreduce_bf16", self._h, x.data_ptr(), out.data_ptr(), x.numel(),
                  torch.cuda.current_stream().cuda_stream)
        return out

    def all_reduce_add_rmsnorm(self, x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
        """residual += sum over ranks of x (in place); returns RMSNorm(residual) * weight.  x, residual: [rows, dim] bf16."""
        assert x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous() and residual.is_contiguous() and x.shape == residual.shape
        rows, dim = x.numel() // x.shape[-1], x.shape[-1]
        y = torch.empty_like(residual)
        _hip.call("nvh_allreduce_add_rmsnorm_bf16", self._h, x.data_ptr(), residual.data_ptr(), weight.data_ptr(), y.data_ptr(),
                  rows, dim, float(eps), torch.cuda.current_stream().cuda_stream)
        return y

    def gather_cols(self, x: torch.Tensor, root: int = 0) -> torch.Tensor | None:
        """x [rows, cols] per rank -> [rows, world*cols] on `root` (rank order along the columns), None elsewhere."""
        assert x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous() and x.dim() == 2
        rows, cols = x.shape
        out = torch.empty(rows, self.world * cols, dtype=x.dtype, device=x.device) if self.rank == root else None
        _hip.call("nvh_gather_cols_bf16", self._h, x.data_ptr(), out.data_ptr() if out is not None else 0, rows, cols, root,
                  torch.cuda.current_stream().cuda_stream)
        return out

    def fits(self, x: torch.Tensor) -> bool:
        return x.dtype == torch.bfloat16 and x.is_contiguous() and x.numel() % 8 == 0 and x.numel() * 2 <= self.max_bytes

    def check(self) -> None:
        """Synchronises; raises if an exchange timed out waiting for a peer."""
        rc = _hip.lib().nvh_comm_error(self._h)
        if rc:
            raise _hip.HipError(_hip.lib().nvh_last_error().decode())

    def self_test(self, sizes=(8, 1 << 12, 1 << 18, 3 << 18), soak: int = 64) -> bool:
        """Start-up verification against torch.distributed: a few message sizes (one- and two-shot), then a soak of
        back-to-back fused exchanges and gathers shaped like decode steps.  The verdict is agreed between the ranks."""
        ok, timed_out = True, False
        try:
            for i, n in enumerate(sizes):
                if n * 2 > self.max_bytes:
                    continue
                g = torch.Generator(device="cpu").manual_seed(1234 + 17 * i + self.rank)
                x = torch.randn(n, generator=g, dtype=torch.float32).to(torch.bfloat16).cuda()
                ref = x.float()
                _dist_sum(ref, self.group)
                got = self.all_reduce(x.clone())
                self.check()
                ok = ok and bool(torch.allclose(got.float(), ref, atol=0.0, rtol=2.0 ** -7))
            # soak: every call changes the data, the results of all calls are folded into one checksum row
            rows, dim = 8, 1024
            if ok and rows * dim * 2 <= self.max_bytes:
                g = torch.Generator(device="cpu").manual_seed(99 + self.rank)
                xs = torch.randn(soak, rows, dim, generator=g).to(torch.bfloat16).cuda()
                w = torch.ones(dim, dtype=torch.bfloat16, device="cuda")
                res = torch.zeros(rows, dim, dtype=torch.bfloat16, device="cuda")
                res_ref = torch.zeros(rows, dim, dtype=torch.float32, device="cuda")
                for k in range(soak):
                    self.all_reduce_add_rmsnorm(xs[k], res, w, 1e-6)
                    if k % 8 == 7 and rows * 8 * 2 <= self.max_bytes:
                        self.gather_cols(xs[k, :, :8].contiguous(), 0)
                total = xs.float().sum(0)
                _dist_sum(total, self.group)
                self.check()
                # res accumulated bf16-rounded sums; compare loosely (rounding per step), exact agreement is tested elsewhere
                ok = ok and bool((res.float() - total).abs().max() <= 0.05 * total.abs().max() + 0.5)
        except _hip.HipError as e:                       # a peer never answered: report, let every rank agree below
            print(f"[nanovllm] rank {self.rank}: {e}", file=sys.stderr)
            ok, timed_out = False, True
        flag = torch.tensor([1 if ok else 0, 0 if timed_out else 1], dtype=torch.int32,
                            device="cpu" if _on_host(self.group) else "cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        self.self_test_timed_out = flag[1].item() == 0     # some rank waited in vain for a peer (as opposed to wrong sums)
        return bool(flag[0].item())

    def close(self) -> None:
        if self._h is not None:
            _hip.lib().nvh_comm_destroy(self._h)
            self._h = None


fallback_reason: "str | None" = None      # why the last install() returned None: disabled | setup | timeout | mismatch


def install(rank: int, world: int, max_bytes: int, group=None) -> "XgmiComm | None":
    """Create the peer path for this process, verify it, make it the one `all_reduce` uses.  Returns None
    (and says so on stderr) when NANOVLLM_XGMI_AR=0 or the verification fails; callers then run on RCCL."""
    global _ACTIVE, fallback_reason
    fallback_reason = None
    if os.environ.get("NANOVLLM_XGMI_AR", "1") == "0":
        fallback_reason = "disabled"
        return None
    try:
        comm = XgmiComm(rank, world, max_bytes, group)
    except _hip.HipError as e:
        print(f"[nanovllm] rank {rank}: {e}; using torch.distributed", file=sys.stderr)
        fallback_reason = "setup"
        return None
    if not comm.self_test():
        fallback_reason = "timeout" if comm.self_test_timed_out else "mismatch"
        print(f"[nanovllm] rank {rank}: xGMI all-reduce self-test FAILED ({fallback_reason}); using torch.distributed", file=sys.stderr)
        comm.close()
        return None
    _ACTIVE = comm
    return comm


def uninstall() -> None:
    global _ACTIVE
    if _ACTIVE is not None:
        _ACTIVE.close()
        _ACTIVE = None


def active() -> "XgmiComm | None":
    return _ACTIVE


def gather_cols(x: torch.Tensor, root: int = 0) -> torch.Tensor | None:
    """Vocabulary shards [rows, cols] of every rank -> [rows, world*cols] on `root`, None on the other ranks."""
    if _ACTIVE is not None and _ACTIVE.fits(x) and x.shape[-1] % 8 == 0:
        return _ACTIVE.gather_cols(x, root)
    world, rank = dist.get_world_size(), dist.get_rank()
    src = x.cpu() if (x.is_cuda and _on_host()) else x
    parts = [torch.empty_like(src) for _ in range(world)] if rank == root else None
    dist.gather(src, parts, root)
    return torch.cat(parts, -1).to(x.device) if rank == root else None


def broadcast_ids(ids: "torch.Tensor | None", count: int, device) -> torch.Tensor:
    """Rank 0's `count` sampled ids (int64 on the device; None on the other ranks) -> the same ids on every rank, on
    the device, without the host.  They travel as three bf16 bytes each through `all_reduce` - the very sum over the
    ranks the activations take (peer exchange or torch.distributed), the other ranks adding zeros - so no second kind
    of collective is involved."""
    padded = (3 * count + 7) // 8 * 8
    enc = torch.empty(padded, dtype=torch.bfloat16, device=device)
    stream = torch.cuda.current_stream().cuda_stream
    _hip.call("nvh_ids_to_bytes_bf16", 0 if ids is None else ids.data_ptr(), enc.data_ptr(), count, padded, stream)
    all_reduce(enc)
    out = torch.empty(count, dtype=torch.int64, device=device)
    _hip.call("nvh_bytes_bf16_to_ids", enc.data_ptr(), out.data_ptr(), count, stream)
    return out


def all_reduce(y: torch.Tensor) -> torch.Tensor:
    """In-place sum over the tensor-parallel ranks."""
    if _ACTIVE is not None and _ACTIVE.fits(y):
        return _ACTIVE.all_reduce(y)
    return _dist_sum(y)
