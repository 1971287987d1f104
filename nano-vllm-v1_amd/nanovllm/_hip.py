"""ctypes binding of libnanovllm_hip.so (the C ABI declared in include/nanovllm_hip.h).

There is no fallback: if the shared library is missing or a call fails, an
exception is raised.  Build the library with ``nano-vllm-v1_amd/csrc/build.sh``
(or ``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_LIB_PATH = Path(os.environ.get("NANOVLLM_HIP_LIB", Path(__file__).resolve().parent.parent / "lib" / "libnanovllm_hip.so"))

i32, i64, u64, f32, vp = C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_void_p

# name -> (restype, argtypes); mirrors include/nanovllm_hip.h one to one.
SIGNATURES = {
    "nvh_version": (i32, []),
    "nvh_last_error": (C.c_char_p, []),
    "nvh_device_check": (i32, [i32]),
    "nvh_set_switch": (i32, [C.c_char_p, i64]),
    "nvh_get_switch": (i64, [C.c_char_p]),
    "nvh_store_kvcache": (i32, [vp, i64, vp, i64, vp, vp, vp, i64, i64, vp]),
    "nvh_rmsnorm": (i32, [vp, i64, vp, vp, i64, i64, i64, f32, vp]),
    "nvh_add_rmsnorm": (i32, [vp, vp, vp, vp, i64, i64, f32, vp, i64, vp]),
    "nvh_rope": (i32, [vp, vp, i64, i64, vp, i64, i64, vp, i64, i64, i64, vp]),
    "nvh_qknorm_rope_store": (i32, [vp, i64, i64, i64, vp, vp, f32, vp, vp, i64, vp, vp, vp, i64, vp]),
    "nvh_gemm_qkv_rope_store": (i32, [vp, i64, vp, vp, i64, i64, i64, i64, i64, i64, vp, vp, f32, vp, vp, i64, vp, vp, vp, i64, vp]),
    "nvh_silu_mul": (i32, [vp, vp, i64, i64, vp]),
    "nvh_embedding": (i32, [vp, vp, vp, i64, i64, i64, i64, vp]),
    "nvh_gather_last_rows": (i32, [vp, vp, vp, vp, i64, i64, vp]),
    "nvh_feed_token_ids": (i32, [vp, vp, vp, vp, i64, vp]),
    "nvh_ids_to_bytes_bf16": (i32, [vp, vp, i64, i64, vp]),
    "nvh_sample_shard_workspace_bytes": (i64, [i64, i64]),
    "nvh_sample_shard": (i32, [vp, i32, i64, vp, vp, i64, i64, i64, i32, i32, u64, u64, vp, vp]),
    "nvh_sample_pick": (i32, [vp, vp, i64, i32, vp]),
    "nvh_lm_head_sample_fused": (i32, [i64, i64, i64]),
    "nvh_lm_head_sample_workspace_bytes": (i64, [i64, i64]),
    "nvh_lm_head_sample": (i32, [vp, i64, vp, vp, vp, i64, i64, i64, u64, u64, vp, vp]),
    "nvh_bytes_bf16_to_ids": (i32, [vp, vp, i64, vp]),
    "nvh_gemm_bf16_nt": (i32, [vp, i64, vp, vp, vp, i64, i64, i64, i64, i32, vp, vp]),
    "nvh_gemm_workspace_bytes": (i64, [i64, i64, i32]),
    "nvh_gemm_suggest_split_k": (i32, [i64, i64, i64]),
    "nvh_gemm_plan": (i32, [i64, i64, i64, i32, i32, C.c_char_p, i64]),
    "nvh_splitk_reduce": (i32, [vp, i64, vp, vp, i64, i64, i64, vp]),
    "nvh_gemm_swiglu_bf16_nt": (i32, [vp, i64, vp, vp, i64, i64, i64, i64, vp]),
    "nvh_interleave_gate_up": (i32, [vp, vp, i64, i64, vp]),
    "nvh_gemm_residual_rowsq_parts": (i32, [i64, i64, i64]),
    "nvh_gemm_residual_rowsq_bf16_nt": (i32, [vp, i64, vp, vp, vp, i64, i64, i64, vp]),
    "nvh_gemm_qkv_epilogue_in_gemm": (i32, [i64, i64, i64]),
    "nvh_gemm_rowscaled_supported": (i32, [i64, i64, i64, i32]),
    "nvh_gemm_qkv_rope_store_rowscaled": (i32, [vp, i64, vp, vp, i64, i64, i64, i64, i64, i64, vp, vp, f32, vp, vp, i64, vp, vp, vp, i64, vp, i32, f32, vp]),
    "nvh_gemm_swiglu_rowscaled_bf16_nt": (i32, [vp, i64, vp, vp, i64, i64, i64, i64, vp, i32, f32, vp]),
    "nvh_add_rmsnorm_gemm_bf16_nt": (i32, [vp, vp, i64, vp, vp, vp, f32, vp, vp, i64, i64, i64, i64, i32, vp]),
    "nvh_add_rmsnorm_gemm_fused": (i32, [i64, i64, i64, i64, i32]),
    "nvh_attn_varlen": (i32, [vp, i64, vp, vp, i64, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, i64, f32, vp, i32, vp, vp]),
    "nvh_attn_varlen_workspace_bytes": (i64, [i64, i64, i64, i32]),
    "nvh_attn_varlen_suggest_splits": (i32, [i64, i64, i64, i64, i64]),
    "nvh_attn_decode": (i32, [vp, i64, vp, vp, vp, vp, i64, i64, i64, i64, i64, f32, vp, i32, i32, vp, vp]),
    "nvh_attn_decode_noalloc": (i32, [vp, i64, vp, vp, vp, vp, i64, i64, i64, i64, i64, f32, vp, i32, i32, vp]),
    "nvh_attn_decode_noalloc_max_bytes": (i64, []),
    "nvh_comm_handle_bytes": (i32, []),
    "nvh_comm_create": (i32, [i32, i32, i64, vp]),
    "nvh_comm_get_handle": (i32, [vp, vp]),
    "nvh_comm_open_peers": (i32, [vp, C.c_char_p]),
    "nvh_comm_buffer": (vp, [vp]),
    "nvh_comm_max_bytes": (i64, [vp]),
    "nvh_allreduce_bf16": (i32, [vp, vp, vp, i64, vp]),
    "nvh_allreduce_add_rmsnorm_bf16": (i32, [vp, vp, vp, vp, vp, i64, i64, f32, vp]),
    "nvh_gather_cols_bf16": (i32, [vp, vp, vp, i64, i64, i32, vp]),
    "nvh_comm_error": (i32, [vp]),
    "nvh_comm_destroy": (i32, [vp]),
    "nvh_attn_decode_items": (i32, [vp, i64, vp, vp, vp, vp, i64, i64, i64, i64, i64, f32, vp, vp, vp, vp, vp, i64, i32, vp, vp]),
    "nvh_attn_decode_fused_qkv_items": (i32, [vp, i64, vp, vp, f32, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, f32, vp,
                                              vp, vp, vp, vp, i64, i32, vp, vp]),
    "nvh_attn_decode_fused_qkv": (i32, [vp, i64, vp, vp, f32, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, f32, vp, i32, i32, vp, vp]),
    "nvh_attn_decode_workspace_bytes": (i64, [i64, i64, i64, i32]),
    "nvh_attn_decode_suggest_splits": (i32, [i64, i64, i64]),
    "nvh_sample_workspace_bytes": (i64, [i64, i64]),
    "nvh_sample": (i32, [vp, i32, i64, vp, vp, i64, i64, u64, u64, vp, vp]),
    "nvh_sample_with_noise": (i32, [vp, i32, i64, vp, vp, vp, i64, i64, vp, vp]),
    "nvh_argmax": (i32, [vp, i32, i64, vp, i64, i64, vp, vp]),
    # struct pointers (nvh_qwen3_model*, nvh_step*) travel as void*; layouts are in nanovllm/native.py
    "nvh_qwen3_prepare_layer_events": (i32, [i32]),
    "nvh_qwen3_workspace_bytes": (i64, [vp, i64, i64, i32]),
    "nvh_qwen3_forward": (i32, [vp, vp, vp, vp, i64, vp]),
    "nvh_qwen3_forward_parts": (i32, [vp, vp, vp, vp, i64, i32, i32, vp]),
    "nvh_qwen3_exchange_buffers": (i32, [vp, i64, i64, i32, vp, vp, vp]),
    "nvh_qwen3_logits": (i32, [vp, vp, vp, vp, i64, vp, vp, vp]),
}

_lib = None


class HipError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load (once) and return the shared library; raises if it is not built."""
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            raise HipError(f"{_LIB_PATH} not found: build it with nano-vllm-v1_amd/csrc/build.sh; "
                           "there is no CPU or PyTorch fallback for the operator path")
        handle = C.CDLL(str(_LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)          # AttributeError if the library lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(code: int, what: str) -> None:
    if code != 0:
        raise HipError(f"{what} failed ({code}): {lib().nvh_last_error().decode()}")


def call(name: str, *args) -> None:
    check(getattr(lib(), name)(*args), name)


class switch:
    """``with _hip.switch("NVH_NO_NORM_GEMM"): ...`` - one routing switch of the library set for the block (value -1
    unsets it), restored afterwards.  For tests and measurements that compare two routes inside one process."""

    def __init__(self, name: str, value: int = 1):
        self.name, self.value = name.encode(), value

    def __enter__(self):
        self.prev = lib().nvh_get_switch(self.name)
        if self.prev == -2:
            raise HipError(f"unknown switch {self.name.decode()}")
        call("nvh_set_switch", self.name, self.value)
        return self

    def __exit__(self, *exc):
        call("nvh_set_switch", self.name, self.prev)
        return False
