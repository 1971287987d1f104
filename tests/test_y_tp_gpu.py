"""Tensor-parallel model steps on real processes: 2 ranks (Hq 2 / Hkv 1 / vocab 512 each) of the tiny model,
sharing the one GPU of the test box, against the single-rank oracle.  Covers the shard loaders, the native
executor's exchange calls (embedding sum, O/down all-reduce, logits gather), graph capture with the exchange
inside, and the operator-module path that calls the same exchange from Python."""
import os
import subprocess
import sys

import pytest

from test_y_comm_gpu import run_ranks

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["native", "graph", "modules", "fallback"])
def test_tp2_logits_match_single_rank_oracle(mode):
    results = run_ranks("_tp_worker.py", 2, (mode,), deadline_s=110.0)
    for r, (rc, out) in enumerate(results):
        assert rc == 0 and f"rank {r} OK" in out, f"rank {r} rc={rc}\n{out[-3000:]}"
    print(results[0][1][-300:])
