import json
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"
for p in (ROOT, ROOT / "nano-vllm-v1_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture
def switches():
    """set(name, value) for the library's routing switches (value -1 unsets); everything touched is restored afterwards."""
    from nanovllm import _hip
    touched = {}

    def set_switch(name: str, value: int = 1):
        key = name.encode()
        if key not in touched:
            touched[key] = _hip.lib().nvh_get_switch(key)
            assert touched[key] != -2, f"unknown switch {name}"
        _hip.call("nvh_set_switch", key, value)

    yield set_switch
    for key, prev in touched.items():
        _hip.call("nvh_set_switch", key, prev)


def from_bits(a: np.ndarray) -> torch.Tensor:
    """uint16 bit patterns -> bf16 tensor."""
    return torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16)


class Golden:
    def __init__(self, name):
        self.z = np.load(GOLDEN / name)

    def bf(self, key):
        return from_bits(self.z[key])

    def t(self, key):
        return torch.from_numpy(self.z[key].copy())

    def __contains__(self, key):
        return key in self.z


@pytest.fixture(scope="session")
def f1():
    return Golden("f1_ops.npz")


@pytest.fixture(scope="session")
def f1b():
    return Golden("f1b_tp_shards.npz")


@pytest.fixture(scope="session")
def f2():
    return Golden("f2_tiny_qwen3.npz")


@pytest.fixture(scope="session")
def f4():
    return json.loads((GOLDEN / "f4_sched_traces.json").read_text())


@pytest.fixture(scope="session")
def f5():
    return json.loads((GOLDEN / "f5_hashes.json").read_text())


def bf16_ulp_diff(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Distance in bf16 units-in-the-last-place between two bf16 tensors (monotone
    integer mapping of the bit patterns)."""
    def key(x):
        i = x.contiguous().view(torch.int16).to(torch.int32) & 0xFFFF
        return torch.where(i >= 0x8000, 0x8000 - i, i)
    return (key(a) - key(b)).abs()


TINY = dict(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
            head_dim=128, intermediate_size=512, vocab_size=1024, max_position_embeddings=4096)
