"""nano-vllm operator path for AMD Instinct MI355X (gfx950): public names of the reference package."""
from .sampling_params import SamplingParams

__all__ = ["LLM", "SamplingParams"]


def __getattr__(name):
    # LLM pulls in the engine (and torch.distributed); import it on first use
    if name == "LLM":
        from .llm import LLM
        return LLM
    raise AttributeError(name)
