"""Worker of tests/test_y_tp_gpu.py: the front-end with tensor_parallel_size 2 as a user runs it (`LLM(...)`
spawns rank 1, the shared-memory control channel carries every step, weights come from a safetensors
checkpoint through the shard loaders).  Both ranks share cuda:0 (one-GPU test box, gloo control plane).
argv: model_dir out_json port [ahead]      (ahead: async_scheduling=True, the engine one step ahead of its sampled ids)"""
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd"))


def main():
    model_dir, out_json, port = sys.argv[1], sys.argv[2], sys.argv[3]
    ahead = len(sys.argv) > 4 and sys.argv[4] == "ahead"
    os.environ["NANOVLLM_SHARE_DEVICE"] = "1"
    os.environ["NANOVLLM_PORT"] = port
    import torch
    from nanovllm import LLM, SamplingParams, comm

    llm = LLM(model_dir, tensor_parallel_size=2, max_num_batched_tokens=200, max_num_seqs=8, max_model_len=2048,
              num_kvcache_blocks=24, chunked_prefill=True, enforce_eager=False, async_scheduling=ahead)
    if comm.active() is None and comm.fallback_reason == "timeout":
        raise RuntimeError("start-up self-test: exchange timed out")
    assert comm.active() is not None, "xGMI exchange not installed"
    g = torch.Generator().manual_seed(11)
    prompts = [torch.randint(0, 1024, (n,), generator=g).tolist() for n in (5, 300, 257, 64)]
    outs = llm.generate(prompts, SamplingParams(temperature=1e-9, max_tokens=6, ignore_eos=True), use_tqdm=False)
    llm.exit()
    json.dump({"prompts": prompts, "completions": [o["token_ids"] for o in outs]}, open(out_json, "w"))
    print("engine OK")


if __name__ == "__main__":
    import faulthandler
    faulthandler.dump_traceback_later(75, exit=False)
    try:
        main()
    except Exception as e:  # noqa: BLE001
        if "exchange timed out" in str(e):
            print(f"rank not co-scheduled: {e}", flush=True)
            os._exit(77)
        raise
