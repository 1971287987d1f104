import sys, json, os, torch, pathlib, tempfile
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/nano-vllm-v1_amd"); sys.path.insert(0, "/root/repo/tests")
from test_config_shapes_gpu import build, prompts, DIMS_14B
from test_model_gpu import seq
from oracle import ref_model as M
std = float(sys.argv[1]) if len(sys.argv) > 1 else 0.02
tmp = pathlib.Path(tempfile.mkdtemp())
runner, oracle, shape = build(tmp, "qwen3-14b", DIMS_14B, enforce_eager=True, pages=10, std=std)
ps = prompts([200, 130, 61, 257], 3, shape.vocab_size)
tables = [[0], [1], [2], [3, 4]]
seqs = [seq(p, 0, len(p), t) for p, t in zip(ps, tables)]
states = [M.SeqState(list(p), 0, len(p), list(t)) for p, t in zip(ps, tables)]
logits, _, _ = runner.forward_logits(seqs)
ref, _ = oracle.run(states)
g, r = logits.float().cpu(), ref.float()
err = (g - r).abs()
i = err.argmax()
print("std", std, "logit std", float(r.std()), "max|ref|", float(r.abs().max()))
print("max err", float(err.max()), "ref there", float(r.flatten()[i]), "got", float(g.flatten()[i]))
print("quantiles of err", [float(err.flatten().quantile(q)) for q in (0.5, 0.99, 0.999)] if err.numel() < 1e6 else float(err.mean()))
print("mean err", float(err.mean()), "frac > 0.03+0.02|ref|", float((err > 0.03 + 0.02 * r.abs()).float().mean()))
print("argmax equal", torch.equal(g.argmax(-1), r.argmax(-1)))
runner.exit()
