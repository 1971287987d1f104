"""Temperature sampling by the exponential race (reference: nanovllm/layers/sampler.py:41-81):
token = argmax(softmax(logits / T) / E), E ~ Exp(1).  Kernel nvh_sample (csrc/sampler.hip) draws
E from a counter-based generator keyed by (seed, call counter, row, column)."""
import torch
from torch import nn

from .. import ops


class Sampler(nn.Module):
    def __init__(self, seed: int = 0):
        super().__init__()
        self.seed = seed
        self.calls = 0

    def forward(self, logits: torch.Tensor, temperatures: torch.Tensor) -> torch.Tensor:
        self.calls += 1
        return ops.sample(logits, temperatures, self.seed, self.calls)
