#!/usr/bin/env python3
"""Where the host spends a step of the saturated serving run (serving_bench.py's workload, engine one step ahead): per
engine step, the time to schedule + launch the next step, the time spent WAITING for the previous step's ids (0 = the GPU
had already finished: that step boundary left the GPU idle) and the time to commit.
usage: serving_host_probe.py [serving_bench.py arguments]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd"))
import serving_bench  # noqa: E402
from nanovllm.engine import llm_engine as E  # noqa: E402
from nanovllm.engine import model_runner as MR  # noqa: E402

rec = []
cur = {}
orig_result = MR.PendingTokens.result


def result(self):
    t0 = time.perf_counter()
    r = orig_result(self)
    cur["wait"] = cur.get("wait", 0.0) + time.perf_counter() - t0
    cur["ids"] = len(r)
    return r


MR.PendingTokens.result = result
orig_step = E.LLMEngine._step_ahead


def step(self):
    cur.clear()
    t0 = time.perf_counter()
    sched = self.scheduler
    o_sched, o_run, o_commit = sched.schedule, self.model_runner.call, sched.commit

    def timed(name, fn):
        def w(*a, **k):
            t = time.perf_counter()
            r = fn(*a, **k)
            cur[name] = cur.get(name, 0.0) + time.perf_counter() - t
            return r
        return w
    sched.schedule, self.model_runner.call, sched.commit = timed("schedule", o_sched), timed("launch", o_run), timed("commit", o_commit)
    try:
        out = orig_step(self)
    finally:
        sched.schedule, self.model_runner.call, sched.commit = o_sched, o_run, o_commit
    cur["total"] = time.perf_counter() - t0
    cur["t"] = t0
    rec.append(dict(cur))
    return out


E.LLMEngine._step_ahead = step
serving_bench.main(sys.argv[1:] or ["--num-requests", "2048", "--request-rate", "2000", "--random-input-len", "128", "--random-output-len", "100",
                                    "--max-num-batched-tokens", "1024", "--chunked-prefill", "--max-model-len", "4096"])
steps = rec[len(rec) // 6:]                      # (skips the warm-up requests' steps)
n = len(steps)
ms = lambda k: 1e3 * sum(s.get(k, 0.0) for s in steps) / n   # noqa: E731
commit_incl_wait = ms("commit")
gaps = [1e3 * (b["t"] - a["t"] - a["total"]) for a, b in zip(steps, steps[1:])]
nowait = sum(1 for s in steps if s.get("wait", 0.0) < 30e-6)
print(f"{n} steps: per step {ms('total'):.3f} ms inside engine.step + {sum(gaps) / len(gaps):.3f} ms in the harness between steps; "
      f"schedule {ms('schedule'):.3f}, launch (prepare + enqueue) {ms('launch'):.3f}, waiting for ids {ms('wait'):.3f}, commit {commit_incl_wait:.3f} ms; "
      f"{nowait} steps ({100 * nowait / n:.0f} %) did not have to wait (GPU already idle)")
