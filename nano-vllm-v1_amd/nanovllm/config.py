"""Engine configuration: the twelve fields (names and defaults) of the reference's config.py:53-64, plus
``async_scheduling`` (not in the reference; ON by default): the engine schedules and launches step n+1 before it has
read step n's sampled ids, so the GPU never waits for the host between steps.  Same tokens as the synchronous loop up
to what depends on WHEN a value is seen (eos ends a sequence one step later: the token computed past it is dropped);
``async_scheduling=False`` runs the reference's loop (engine/llm_engine.py:117-126): schedule, run, read ids, postprocess.

``model`` is a local directory holding a Hugging Face ``config.json`` (and optionally
``*.safetensors`` weights; without them the runner initialises synthetic weights).
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass
from types import SimpleNamespace

import torch


def load_hf_config(path: str):
    """AutoConfig when transformers is importable, else the raw config.json as a namespace."""
    try:
        from transformers import AutoConfig
        return AutoConfig.from_pretrained(path)
    except ImportError:
        with open(os.path.join(path, "config.json")) as f:
            raw = json.load(f)
        raw["torch_dtype"] = getattr(torch, raw.get("torch_dtype", "bfloat16"))
        return SimpleNamespace(**raw)


def model_dtype(hf_config) -> torch.dtype:
    dt = getattr(hf_config, "dtype", None) or getattr(hf_config, "torch_dtype", None) or torch.bfloat16
    return getattr(torch, dt) if isinstance(dt, str) else dt


@dataclass
class Config:
    model: str
    max_num_batched_tokens: int = 16384
    max_num_seqs: int = 512
    max_model_len: int = 40960
    gpu_memory_utilization: float = 0.9
    tensor_parallel_size: int = 1
    enforce_eager: bool = False
    hf_config: object | None = None
    eos: int = -1
    kvcache_block_size: int = 256
    num_kvcache_blocks: int = -1
    chunked_prefill: bool = False
    async_scheduling: bool = True

    def __post_init__(self):
        # the reference asserts a multiple of 256 (config.py:82) but its Sequence.block_size is the constant 256
        # (engine/sequence.py:73) and every kernel here indexes 256-token pages: any other size would read wrong keys
        assert self.kvcache_block_size == 256, "the KV page is 256 tokens (kvcache_block_size must stay 256)"
        assert 1 <= self.tensor_parallel_size <= 8
        if self.hf_config is None:
            self.hf_config = load_hf_config(self.model)
        self.max_model_len = min(self.max_model_len, self.hf_config.max_position_embeddings)
