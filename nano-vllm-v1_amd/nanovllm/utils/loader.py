"""Weight loading: safetensors checkpoints (reference behaviour: utils/loader.py:44-94), any
iterable of (name, tensor) pairs in Hugging Face naming, or synthetic initialisation.

Checkpoint names are mapped onto packed parameters with the model's
``packed_modules_mapping`` ({"q_proj": ("qkv_proj", "q"), ...}); every parameter's own
``weight_loader`` does the tensor-parallel slicing.
"""
from __future__ import annotations

import os
import zlib
from glob import glob

import torch
from torch import nn


def default_weight_loader(param: nn.Parameter, loaded_weight: torch.Tensor):
    param.data.copy_(loaded_weight)


# Checkpoint entries that have no parameter of their own: the output projection of a tied model
# (models/qwen3.py:481-482 aliases it to the embedding) and rotary tables some exporters include.
_SKIPPABLE = ("lm_head.weight", "rotary_emb.inv_freq")


def load_tensors(model: nn.Module, named_tensors, loaded: set | None = None) -> int:
    """Feed (name, tensor) pairs through the per-parameter loaders; returns how many were consumed.
    A name that matches no parameter raises (as the reference's get_parameter does, utils/loader.py:81-88)
    unless it is one of the entries a tied / exported checkpoint may carry without a parameter behind it.
    ``loaded`` collects the parameter names that received data."""
    packed = getattr(model, "packed_modules_mapping", {})
    params = dict(model.named_parameters())
    n = 0
    for name, tensor in named_tensors:
        for key, (target, shard_id) in packed.items():
            if key in name:
                pname = name.replace(key, target)
                if pname not in params:
                    raise KeyError(f"checkpoint tensor {name!r} maps to {pname!r}, which is not a parameter of the model")
                param = params[pname]
                param.weight_loader(param, tensor, shard_id)
                if loaded is not None:
                    loaded.add((pname, shard_id))              # a packed parameter is complete only with every one of its shards
                break
        else:
            if name not in params:
                if name.endswith(_SKIPPABLE):
                    continue
                raise KeyError(f"checkpoint tensor {name!r} matches no parameter of the model")
            pname, param = name, params[name]
            getattr(param, "weight_loader", default_weight_loader)(param, tensor)
        if loaded is not None:
            loaded.add(pname)
        n += 1
    return n


def check_all_loaded(model: nn.Module, loaded: set) -> None:
    """Every parameter must have received data; a tied lm_head shares the embedding's storage."""
    packed = getattr(model, "packed_modules_mapping", {})
    shards_of = {}                                  # packed target (qkv_proj, gate_up_proj) -> the shard ids that fill it
    for target, shard_id in packed.values():
        shards_of.setdefault(target, set()).add(shard_id)
    incomplete = []
    for name in sorted(n for n in loaded if isinstance(n, str)):
        for target, ids in shards_of.items():
            if f".{target}." in f".{name}" or name.startswith(target + "."):
                lacking = sorted(str(i) for i in ids if (name, i) not in loaded)
                if lacking:
                    incomplete.append(f"{name} (shards {', '.join(lacking)} missing)")
    if incomplete:
        raise RuntimeError(f"{len(incomplete)} packed parameters were only partly loaded from the checkpoint: {incomplete[:8]}"
                           f"{' ...' if len(incomplete) > 8 else ''}")
    seen_ptrs = {p.data_ptr() for name, p in model.named_parameters() if name in loaded}
    missing = [name for name, p in model.named_parameters() if name not in loaded and p.data_ptr() not in seen_ptrs]
    if missing:
        raise RuntimeError(f"{len(missing)} parameters were not loaded from the checkpoint: {missing[:8]}"
                           f"{' ...' if len(missing) > 8 else ''}")


def load_model(model: nn.Module, path: str) -> int:
    from safetensors import safe_open
    n = 0
    loaded: set = set()
    for file in sorted(glob(os.path.join(path, "*.safetensors"))):
        with safe_open(file, "pt", "cpu") as f:
            n += load_tensors(model, ((name, f.get_tensor(name)) for name in f.keys()), loaded)
    check_all_loaded(model, loaded)
    return n


def has_checkpoint(path: str) -> bool:
    return bool(glob(os.path.join(path, "*.safetensors")))


def synthetic_tensors(hf_config, seed: int = 0, std: float = 0.02, device="cuda", dtype=torch.bfloat16):
    """Full (unsharded) random weights in Hugging Face naming, generated on ``device`` one tensor at
    a time; the stream for a tensor depends only on (seed, its name), so every tensor-parallel rank
    draws the same full matrix before its loader slices it."""
    c = hf_config
    d = getattr(c, "head_dim", None) or c.hidden_size // c.num_attention_heads
    hq, hkv, h, inter, v = c.num_attention_heads, c.num_key_value_heads, c.hidden_size, c.intermediate_size, c.vocab_size

    def mat(name, *shape):
        g = torch.Generator(device=device)
        g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
        return name, (torch.randn(*shape, generator=g, device=device, dtype=torch.float32) * std).to(dtype)

    def ones(name, n):
        return name, torch.ones(n, device=device, dtype=dtype)

    yield mat("model.embed_tokens.weight", v, h)
    for i in range(c.num_hidden_layers):
        p = f"model.layers.{i}."
        yield mat(p + "self_attn.q_proj.weight", hq * d, h)
        yield mat(p + "self_attn.k_proj.weight", hkv * d, h)
        yield mat(p + "self_attn.v_proj.weight", hkv * d, h)
        yield mat(p + "self_attn.o_proj.weight", h, hq * d)
        yield ones(p + "self_attn.q_norm.weight", d)
        yield ones(p + "self_attn.k_norm.weight", d)
        yield mat(p + "mlp.gate_proj.weight", inter, h)
        yield mat(p + "mlp.up_proj.weight", inter, h)
        yield mat(p + "mlp.down_proj.weight", h, inter)
        yield ones(p + "input_layernorm.weight", h)
        yield ones(p + "post_attention_layernorm.weight", h)
    yield ones("model.norm.weight", h)
    if not getattr(c, "tie_word_embeddings", False):
        yield mat("lm_head.weight", v, h)
