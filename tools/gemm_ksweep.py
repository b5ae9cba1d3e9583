"""Per-tile fixed cost vs per-k cost of the NT GEMMs: time(K) for K = 64..1536 at M = 196608, N in {384, 1152, 3072}; the intercept is
the epilogue + store + tile turnaround, the slope the main loop."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frankenstein_amd import kernels as K

M = 32 * 6144
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
def rnd(*s): return (torch.randn(*s, device=dev, generator=g) * 0.5).to(torch.bfloat16)
def t(f, n=5):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 2)
    return sorted(ts)[n // 2] * 1e3
for N in (384, 1152, 3072):
    for Kd in (64, 128, 384, 768, 1536):
        x, w = rnd(M, Kd), rnd(N, Kd)
        line = f"N={N:5d} K={Kd:5d}  plain {t(lambda: K.gemm_nt(x, w, None)):8.1f} us"
        if N == 3072:
            line += f"   swiglu {t(lambda: K.gemm_nt_swiglu(x, w)):8.1f} us"
        if N == 384:
            r = rnd(M, N)
            line += f"   +res {t(lambda: K.gemm_nt(x, w, None, residual=r)):8.1f} us"
        print(line, flush=True)
        del x, w
