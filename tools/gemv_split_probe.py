import sys
sys.path.insert(0, "nano-vllm-v1_amd"); sys.path.insert(0, "tools")
import torch
from nanovllm import ops
from bench_kernels import timeit
BF = torch.bfloat16
for m in ([int(a) for a in sys.argv[1:]] or [1, 8]):
    for name, n, k in (("qkv", 7168, 5120), ("o", 5120, 5120), ("down", 5120, 17408)):
        copies = max(3, (700 << 20) // (n * k * 2) + 1)
        x = torch.randn(m, k, device="cuda").to(BF)
        ws = [(torch.randn(n, k, device="cuda") * 0.02).to(BF) for _ in range(copies)]
        line = f"M={m} {name:5s} suggest={ops.suggest_split_k(m, n, k)} s1 {timeit(lambda i: ops.linear(x, ws[i % copies]), iters=2 * copies):6.1f}"
        for s in (2, 3, 4, 5, 6, 8):
            t = timeit(lambda i: ops.linear_splitk(x, ws[i % copies], s), iters=2 * copies)
            t2 = timeit(lambda i: ops.splitk_reduce(ops.linear_splitk(x, ws[i % copies], s)), iters=2 * copies)
            line += f" | s{s} {t:5.1f} (+reduce {t2:5.1f})"
        print(line, flush=True)
        del ws
