for v in "" "NVH_GEMM_NO_256=1"; do echo "== $v"; env $v timeout -k 10 120 python - <<'PY' 2>&1 | grep -v amdgpu.ids || exit 1
import sys; sys.path.insert(0,'nano-vllm-v1_amd'); sys.path.insert(0,'tools')
import torch
from nanovllm import ops
from bench_kernels import timeit
BF=torch.bfloat16
for m,n,k in ((256,151936,1024),(512,151936,1024),(4096,4096,4096),(8192,8192,8192),(1024,6144*4,1024)):
    x=torch.randn(m,k,device='cuda').to(BF); ws=[(torch.randn(n,k,device='cuda')*0.05).to(BF) for _ in range(3)]
    y=ops.linear(x, ws[0]); ref=(x.float()@ws[0].float().t())
    err=float((y.float()-ref).abs().max()/ref.abs().max())
    us=timeit(lambda i: ops.linear(x, ws[i%3]), iters=20)
    print(f"M={m} N={n} K={k}: {us:.1f} us  {2*m*n*k/us/1e6:.0f} TF  rel err {err:.2e}")
PY
done
