#!/usr/bin/env python3
"""Writes tests/golden/gemm_dispatch.json: for every projection of the three Qwen3 shapes (and their tensor-parallel shards)
at a ladder of row counts, the kernel(s) the library's GEMM dispatcher picks (nvh_gemm_plan: nothing is launched, no GPU
needed), the split-K count the step executor would ask for and whether the add + RMSNorm goes into the projection.
tests/test_dispatch_cpu.py compares the built library against this file, so that an edit of one threshold cannot silently
move a tuned shape onto another kernel: regenerate it on purpose, and look at the diff."""
import ctypes as C
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd"))
from nanovllm import _hip  # noqa: E402

MODELS = {  # hidden, q heads, kv heads, intermediate, tensor-parallel sizes
    "qwen3-0.6b": (1024, 16, 8, 3072, (1,)),
    "qwen3-14b": (5120, 40, 8, 17408, (1, 4)),
    "qwen3-32b": (5120, 64, 8, 25600, (1, 8)),
}
ROWS = (1, 8, 32, 64, 128, 256, 512, 800, 1024)
VOCAB = 151936


def plan(lib, m, n, k, split, flavor):
    buf = C.create_string_buffer(512)
    rc = lib.nvh_gemm_plan(m, n, k, split, flavor, buf, len(buf))
    return buf.value.decode() if rc == 0 else f"error {rc}"


def table():
    lib = _hip.lib()
    out = {}
    for name, (h, hq, hkv, inter, tps) in MODELS.items():
        for tp in tps:
            qsz, kvsz, i = hq * 128 // tp, hkv * 128 // tp, inter // tp
            for m in ROWS:
                key = f"{name} tp{tp} M={m}"
                so, sd = lib.nvh_gemm_suggest_split_k(m, h, qsz), lib.nvh_gemm_suggest_split_k(m, h, i)
                out[key] = {
                    "qkv": plan(lib, m, qsz + 2 * kvsz, h, 1, 2),
                    "o": plan(lib, m, h, qsz, 1, 0), "o_split": so, "o_slabs": plan(lib, m, h, qsz, so, 0) if so > 1 else None,
                    "gate_up_swiglu": plan(lib, m, 2 * i, h, 1, 1),
                    "down": plan(lib, m, h, i, 1, 0), "down_split": sd, "down_slabs": plan(lib, m, h, i, sd, 0) if sd > 1 else None,
                    "norm_in_qkv": lib.nvh_add_rmsnorm_gemm_fused(m, qsz + 2 * kvsz, h, 0, 0),
                    "norm_in_gate_up": lib.nvh_add_rmsnorm_gemm_fused(m, 2 * i, h, 0, 1),
                    "lm_head": plan(lib, m, VOCAB // tp, h, 1, 0),
                    "lm_head_sampling": plan(lib, m, VOCAB // tp, h, 1, 3) if lib.nvh_lm_head_sample_fused(m, VOCAB // tp, h) else None,
                }
    return out


if __name__ == "__main__":
    path = ROOT / "tests" / "golden" / "gemm_dispatch.json"
    path.write_text(json.dumps(table(), indent=1) + "\n")
    print(f"wrote {path}")
