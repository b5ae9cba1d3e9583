"""Isolated timing of the cfg2 LayerNorm forward / backward ([196608, 384] bf16)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frankenstein_amd import kernels as K
M, d = 32 * 6144, 384
x = torch.randn(M, d, device="cuda").bfloat16(); dy = torch.randn(M, d, device="cuda").bfloat16(); dres = torch.randn(M, d, device="cuda").bfloat16()
g = torch.ones(d, device="cuda"); b = torch.zeros(d, device="cuda")
y, mean, rstd = K.norm_fwd(x, g, b, 1e-5)
def t(f, n=20):
    f(); torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3
tf = t(lambda: K.norm_fwd(x, g, b, 1e-5)); tb = t(lambda: K.norm_bwd(dy, x, g, mean, rstd, dres=dres))
print(f"norm_fwd {tf:.1f} us  {2 * M * d * 2 / tf / 1e6:.2f} TB/s   norm_bwd {tb:.1f} us  {4 * M * d * 2 / tb / 1e6:.2f} TB/s")
