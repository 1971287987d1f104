"""The GEMM dispatcher's choices for every projection of the three model shapes, pinned (no GPU needed: nvh_gemm_plan
names the launches without making them).  The table is tests/golden/gemm_dispatch.json, written by
oracle/gen_dispatch_table.py from the library as it was when the shapes were tuned; editing a threshold in the dispatcher
shows up here as a named shape that moved to another kernel."""
import json
import os

import pytest

from conftest import GOLDEN, ROOT


@pytest.fixture(scope="module")
def current():
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_dispatch_table", ROOT / "oracle" / "gen_dispatch_table.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert os.environ.get("NVH_TUNING") != "1", "routing switches from the environment would change the table"
    return mod.table()


def test_every_tuned_shape_lands_on_its_kernel(current):
    want = json.loads((GOLDEN / "gemm_dispatch.json").read_text())
    assert sorted(current) == sorted(want)
    moved = [f"{shape} {proj}: {want[shape][proj]!r} -> {current[shape][proj]!r}"
             for shape in want for proj in want[shape] if current[shape][proj] != want[shape][proj]]
    assert not moved, "shapes that changed kernel (regenerate the table if intended):\n" + "\n".join(moved)


def test_the_headline_shapes_take_the_kernels_they_were_tuned_on(current):
    """Spelled out for the shapes the benchmark line is quoted on (Qwen3-0.6B, one GPU)."""
    chunk = current["qwen3-0.6b tp1 M=1024"]
    assert chunk["qkv"] == "tile 128x128 waves 2x4 stages 3 qkv grid 32x8x1"              # one round, epilogue inside
    assert chunk["o"] == chunk["down"] == "kgroups 64x64 stages 3 grid 16x16"
    assert chunk["gate_up_swiglu"] == "tile 128x192 waves 2x4 stages 3 swiglu grid 32x8x1"
    one = current["qwen3-0.6b tp1 M=1"]
    assert one["o"] == one["down"] == "kwaves 8 cols 8 grid 128x1" and one["norm_in_qkv"] == one["norm_in_gate_up"] == 1
    big = current["qwen3-0.6b tp1 M=256"]
    assert big["lm_head_sampling"].startswith("tile 256x256 waves 2x4 stages 2 sample")


def test_switches_are_ignored_without_the_tuning_gate():
    """A routing switch in the environment changes nothing unless NVH_TUNING=1 is set as well (read once per process, so
    checked in a child process)."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); from nanovllm import _hip; print(_hip.lib().nvh_get_switch(b'NVH_GEMM_NO_KGROUPS'))"
            % str(ROOT / "nano-vllm-v1_amd"))
    env = dict(os.environ, NVH_GEMM_NO_KGROUPS="1")
    env.pop("NVH_TUNING", None)
    assert subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip() == "-1"
    env["NVH_TUNING"] = "1"
    assert subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip() == "1"
