for v in "" "NVH_GEMM_KSUB=22" "NVH_GEMM_KSUB=23"; do echo "== $v"; env $v timeout -k 10 120 python - <<'PY' 2>&1 | grep -v amdgpu.ids || exit 1
import sys; sys.path.insert(0,'nano-vllm-v1_amd'); sys.path.insert(0,'tools')
import torch
from nanovllm import ops
from bench_kernels import timeit
BF=torch.bfloat16
for m in (256, 128, 1024):
  for n,k,split in ((4096,1024,1),(6144,1024,1),(1024,2048,0),(1024,3072,0)):
    copies=40
    x=torch.randn(m,k,device='cuda').to(BF); ws=[(torch.randn(n,k,device='cuda')*0.05).to(BF) for _ in range(copies)]
    sp = ops.suggest_split_k(m,n,k) if split==0 else 1
    ref=(x.float()@ws[0].float().t())
    if sp>1:
        y=ops.splitk_reduce(ops.linear_splitk(x, ws[0], sp)); us=timeit(lambda i: ops.linear_splitk(x, ws[i%copies], sp), iters=80)
    else:
        y=ops.linear(x, ws[0]); us=timeit(lambda i: ops.linear(x, ws[i%copies]), iters=80)
    err=float((y.float()-ref).abs().max()/ref.abs().max())
    print(f"M={m} N={n} K={k} split={sp}: {us:.2f} us  err {err:.1e}")
PY
done
