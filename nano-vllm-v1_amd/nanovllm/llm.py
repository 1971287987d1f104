"""`LLM` is the engine under its user-facing name (reference: nanovllm/llm.py)."""
from .engine.llm_engine import LLMEngine


class LLM(LLMEngine):
    pass
