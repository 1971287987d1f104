"""Per-request sampling options (front-end contract of the reference's sampling_params.py:19-45)."""
from dataclasses import dataclass


@dataclass
class SamplingParams:
    temperature: float = 1.0
    max_tokens: int = 64
    ignore_eos: bool = False

    def __post_init__(self):
        # the sampler always races against exponential noise; temperature 0 is undefined for it
        assert self.temperature > 1e-10, "greedy sampling is not permitted"
