"""nano-vllm operator path for AMD Instinct MI355X (gfx950): public names of the reference package."""
import os as _os

# The tensor-parallel exchange shares device memory between the rank processes through HIP IPC, which on hosts
# whose driver only offers dmabuf handles needs the runtime's legacy IPC mode switched off.  It has to be in the
# environment before the first HIP call of the process (and is inherited by the spawned rank processes).
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from .sampling_params import SamplingParams  # noqa: E402

__all__ = ["LLM", "SamplingParams"]


def __getattr__(name):
    # LLM pulls in the engine (and torch.distributed); import it on first use
    if name == "LLM":
        from .llm import LLM
        return LLM
    raise AttributeError(name)
