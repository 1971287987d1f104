"""xGMI all-reduce (csrc/comm.hip) across real processes.  The test box has one GPU, so all ranks share
cuda:0: IPC export/import, the flag protocol, one-/two-shot paths, staging reuse and graph replay are
exercised; the cross-device memory path is not."""
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_allreduce_ranks_on_one_device(world):
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, "-u", os.path.join(HERE, "_comm_worker.py"), str(r), str(world), str(port)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=100)
            outs.append(out)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} OK" in out, f"rank {r} rc={p.returncode}\n{out[-3000:]}"
    print(outs[0])
