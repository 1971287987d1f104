"""xGMI all-reduce (csrc/comm.hip) across real processes.  The test box has one GPU, so all ranks share
cuda:0: IPC export/import, the flag protocol, one-/two-shot paths, staging reuse and graph replay are
exercised; the cross-device memory path is not."""
import os
import re
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


NOT_COSCHEDULED = 77      # worker exit code: an exchange timed out waiting for a peer

_RECORD = re.compile(r"exchange timed out: rank (\d+), block (\d+) waited for rank (\d+) to reach epoch (\d+) \(saw (\d+)\)")


def timeout_is_scheduling(results) -> bool:
    """Decide what an exchange timeout among ranks that SHARE one GPU means, from the record the kernel left
    (nvh_comm_error: waiting rank, block, silent rank, epoch expected, epoch seen).  Several processes on one device
    are time-sliced; a rank that is not scheduled for seconds makes its peers' bounded spins expire although the
    protocol is fine - that cannot happen with one process per GPU.  It is taken for that, and the test skipped, only
    when every record shows the silent peer exactly one exchange behind (an exchange is two flag rounds: the gap is 1
    or 2) and the silent peer's own process did not fail for another reason.  Anything else - a peer several exchanges
    behind, ahead of its waiter, crashed, or killed at the deadline - is a failure.  The records are printed either way."""
    records = []
    for r, (rc, out) in enumerate(results):
        for m in _RECORD.finditer(out or ""):
            records.append(tuple(int(x) for x in m.groups()))
    for rec in records:
        print("exchange timeout record: rank %d, block %d waited for rank %d: expected epoch %d, saw %d" % rec)
    if not records:
        return False
    for waiter, _block, silent, expected, seen in records:
        gap = expected - seen
        if not (1 <= gap <= 2) or silent >= len(results):
            return False
        rc_silent, out_silent = results[silent]
        if rc_silent not in (0, NOT_COSCHEDULED) or "[killed at the deadline]" in (out_silent or ""):
            return False
    return True


def run_ranks(script: str, world: int, extra_args=(), deadline_s: float = 90.0):
    """Start `world` ranks of a worker script on the one GPU of the box and collect (returncode, output) per rank.
    Workers exit with NOT_COSCHEDULED when (and only when) their failure is an exchange timeout; whether that is a
    skip or a failure is decided by timeout_is_scheduling() from the kernel's record.  Wrong results always fail."""
    import time
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NVH_COMM_TIMEOUT_MS="5000")
    procs = [subprocess.Popen([sys.executable, "-u", os.path.join(HERE, script), str(r), str(world), str(port), *extra_args],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    end = time.monotonic() + deadline_s
    results = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=max(1.0, end - time.monotonic()))
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
            out = (out or "") + "\n[killed at the deadline]"
        results.append((p.returncode, out))
    if any(rc == NOT_COSCHEDULED for rc, _ in results):
        if timeout_is_scheduling(results):
            pytest.skip("ranks sharing one device were not co-scheduled: a peer was exactly one exchange behind when the "
                        "bounded spin expired (record above); not a property of the exchange")
        pytest.fail("exchange timeout that is not explained by time-slicing of the shared device:\n" +
                    "\n".join(f"--- rank {r} rc={rc}\n{(out or '')[-1500:]}" for r, (rc, out) in enumerate(results)))
    return results


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_allreduce_ranks_on_one_device(world):
    results = run_ranks("_comm_worker.py", world)
    for r, (rc, out) in enumerate(results):
        assert rc == 0 and f"rank {r} OK" in out, f"rank {r} rc={rc}\n{out[-3000:]}"
    print(results[0][1][-300:])
