import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from frankenstein_amd import kernels as K
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/n
M=196608
for (N,Kd) in [(3072,384),(384,1536),(1152,384),(384,384)]:
    a=torch.randn(M,Kd,device='cuda',dtype=torch.bfloat16); w=torch.randn(N,Kd,device='cuda',dtype=torch.bfloat16)
    out=torch.empty(M,N,device='cuda',dtype=torch.bfloat16)
    ms=t(lambda: K.gemm_nt(a,w,out=out))
    print(f"dbg={os.environ.get('FK_GEMM_DBG','0')} NT M={M} N={N} K={Kd}: {ms:.3f} ms  {2*M*N*Kd/ms/1e9:.0f} TF/s", flush=True)
