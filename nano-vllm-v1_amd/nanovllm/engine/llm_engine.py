"""Front-end loop: requests in, scheduler -> runner -> scheduler per step, finished sequences out
(surface of the reference's engine/llm_engine.py:60-268: LLMEngine(model, **config fields), add_request,
step, is_finished, generate, exit).

Tensor parallelism keeps the reference's process model: rank 0 lives in the caller's process, ranks 1..n-1
are spawned, build their own ModelRunner and sit in its loop until "exit" arrives over the shared-memory
control channel.  A model directory without tokenizer files is served with token-id prompts only
(the text fields of the results are then empty); eos then comes from the model's config.json.
"""
from __future__ import annotations

import atexit
import os
from dataclasses import fields
from time import perf_counter

import torch.multiprocessing as mp

from ..config import Config
from ..sampling_params import SamplingParams
from .model_runner import ModelRunner
from .scheduler import Scheduler
from .sequence import Sequence

_TOKENIZER_FILES = ("tokenizer.json", "tokenizer.model", "vocab.json", "tokenizer_config.json")


def _load_tokenizer(path: str):
    if not any(os.path.exists(os.path.join(path, f)) for f in _TOKENIZER_FILES):
        return None
    from transformers import AutoTokenizer
    return AutoTokenizer.from_pretrained(path, use_fast=True)


class LLMEngine:
    def __init__(self, model: str, **kwargs):
        names = {f.name for f in fields(Config)}
        config = Config(model, **{k: v for k, v in kwargs.items() if k in names})
        self.config = config
        self.ps, self.events = [], []
        ctx = mp.get_context("spawn")
        for rank in range(1, config.tensor_parallel_size):
            event = ctx.Event()
            process = ctx.Process(target=ModelRunner, args=(config, rank, event))
            process.start()
            self.ps.append(process)
            self.events.append(event)
        self.model_runner = ModelRunner(config, 0, self.events)
        self.tokenizer = _load_tokenizer(config.model)
        if self.tokenizer is not None:
            config.eos = self.tokenizer.eos_token_id
        else:
            eos = getattr(config.hf_config, "eos_token_id", -1)
            config.eos = eos[0] if isinstance(eos, (list, tuple)) else (-1 if eos is None else eos)
        self.scheduler = Scheduler(config)
        self._inflight = None                   # async_scheduling: (placeholders, ids on their way) of the step last launched
        self._closed = False
        atexit.register(self.exit)

    def exit(self):
        if self._closed:
            return
        self._closed = True
        self.model_runner.call("exit")
        del self.model_runner
        for p in self.ps:
            p.join()

    def add_request(self, prompt: str | list[int], sampling_params: SamplingParams):
        if isinstance(prompt, str):
            if self.tokenizer is None:
                raise ValueError(f"{self.config.model} has no tokenizer files: pass token-id prompts")
            prompt = self.tokenizer.encode(prompt)
        self.scheduler.add(Sequence(prompt, sampling_params))

    def step(self):
        if self.config.async_scheduling:
            return self._step_ahead()
        seqs = self.scheduler.schedule()
        token_ids, need = self.model_runner.call("run", seqs)
        self.scheduler.postprocess(seqs, token_ids, need.tolist())
        outputs = [(seq.seq_id, seq.completion_token_ids) for seq in seqs if seq.is_finished]
        num_total_tokens = sum(len(seq) for seq in seqs if seq.is_finished)
        return outputs, num_total_tokens

    def _step_ahead(self):
        """One step of the engine running one step ahead of its results (Config.async_scheduling; the reference's loop,
        engine/llm_engine.py:117-126, is the synchronous branch above): schedule and launch the next step from what is
        known without the ids still on the GPU - lengths, pages, budgets; the ids themselves are fed to it on the
        device - and only then wait for the previous step's ids, while the GPU is busy.  What the reference decides
        from an id's VALUE (eos) is decided one step later; the one token computed past an eos is dropped."""
        launched = None
        if not self.scheduler.is_finished():
            seqs = self.scheduler.schedule()
            tokens = self.model_runner.call("run_async", seqs)     # every rank launches; rank 0's ids reach the others on the device
            launched = (self.scheduler.advance(seqs, tokens.need), tokens)
        done = []
        if self._inflight is not None:
            pending, tokens = self._inflight
            done = self.scheduler.commit(pending, tokens.result())
        self._inflight = launched
        outputs = [(seq.seq_id, seq.completion_token_ids) for seq in done]
        return outputs, sum(len(seq) for seq in done)

    def is_finished(self) -> bool:
        return self.scheduler.is_finished() and self._inflight is None

    def generate(self, prompts: list[str] | list[list[int]], sampling_params: SamplingParams | list[SamplingParams],
                 use_tqdm: bool = True) -> list[dict]:
        pbar = None
        if use_tqdm:
            from tqdm.auto import tqdm
            pbar = tqdm(total=len(prompts), desc="Generating", dynamic_ncols=True)
        if not isinstance(sampling_params, list):
            sampling_params = [sampling_params] * len(prompts)
        for prompt, sp in zip(prompts, sampling_params):
            self.add_request(prompt, sp)
        finished: dict[int, list[int]] = {}
        total_tokens, t0 = 0, perf_counter()
        while not self.is_finished():
            outputs, n = self.step()
            total_tokens += n
            for seq_id, token_ids in outputs:
                finished[seq_id] = token_ids
            if pbar is not None:
                pbar.set_postfix({"total_throughput": f"{int(total_tokens / (perf_counter() - t0))}tok/s"})
                pbar.update(len(outputs))
        if pbar is not None:
            pbar.close()
        decode = self.tokenizer.decode if self.tokenizer is not None else (lambda ids: "")
        return [{"text": decode(ids), "token_ids": ids} for _, ids in sorted(finished.items())]
