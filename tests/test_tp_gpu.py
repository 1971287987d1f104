"""Tensor-parallel model steps on real processes: 2 ranks (Hq 2 / Hkv 1 / vocab 512 each) of the tiny model,
sharing the one GPU of the test box, against the single-rank oracle.  Covers the shard loaders, the native
executor's exchange calls (embedding sum, O/down all-reduce, logits gather), graph capture with the exchange
inside, and the operator-module path that calls the same exchange from Python."""
import os
import subprocess
import sys

import pytest

from test_comm_gpu import _free_port

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["native", "graph", "modules"])
def test_tp2_logits_match_single_rank_oracle(mode):
    world, port = 2, _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, "-u", os.path.join(HERE, "_tp_worker.py"), str(r), str(world), str(port), mode],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=110)
            outs.append(out)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} OK" in out, f"rank {r} rc={p.returncode}\n{out[-3000:]}"
    print(outs[0][-300:])
