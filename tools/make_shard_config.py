#!/usr/bin/env python3
"""Write a model config that has the shapes ONE tensor-parallel rank computes (heads, kv heads, MLP width and
vocabulary divided by tp; hidden size kept), so the per-rank kernels of a TP run can be profiled on one GPU
(without the exchange).  usage: make_shard_config.py <model dir name> <tp>  ->  model_configs/<name>-tp<tp>shard"""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent / "nano-vllm-v1_amd" / "model_configs"
name, tp = sys.argv[1], int(sys.argv[2])
cfg = json.load(open(ROOT / name / "config.json"))
for k in ("num_attention_heads", "num_key_value_heads", "intermediate_size", "vocab_size"):
    assert cfg[k] % tp == 0
    cfg[k] //= tp
cfg["tie_word_embeddings"] = False          # the shard's embedding rows and LM-head rows are both V/tp
out = ROOT / f"{name}-tp{tp}shard"
out.mkdir(exist_ok=True)
json.dump(cfg, open(out / "config.json", "w"), indent=1)
print(out.name)
