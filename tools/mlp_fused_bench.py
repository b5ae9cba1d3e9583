"""Timing of fk_mlp_bwd_fused at the cfg2 shape (M = 196 608, H = 1536, d = 384) against the two launches it replaces."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frankenstein_amd import kernels as K
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
M, d, H = 32 * 6144, 384, 1536
def rnd(*s): return (torch.randn(*s, device=dev, generator=g) * 0.5).to(torch.bfloat16)
dy, h13, w2t, w13t = rnd(M, d), rnd(M, 2 * H), rnd(H, d), rnd(d, 2 * H)
def t(f, n=5):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[n // 2] * 1e3
two = lambda: K.gemm_nt(K.gemm_nt_dswiglu(dy, w2t, h13), w13t)
print(f"{os.environ.get('FRANKEN_HIP_LIB', 'in-tree').split('/')[-1]:24s} fused {t(lambda: K.mlp_bwd_fused(dy, w2t, h13, w13t)):8.1f} us   two launches {t(two):8.1f} us")
