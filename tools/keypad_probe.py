"""What does the key-padding mask cost the small-sequence attention (SimpleMAE decoder: N = 600; encoder: N = 150) when nothing is padded?
Same call with MASK_NONE and with Mask.from_padding(all valid): forward + backward, pre-scaled kernels, B = 32 and 256."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frankenstein_amd import kernels as K
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
def t(f, n=7):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[n // 2] * 1e3
for B in (32, 256):
    for N in (150, 600):
        H, D = 6, 64
        qkv = (torch.randn(B, N, 3 * H * D, device=dev, generator=g) * 0.5).bfloat16()
        q, k, v = (qkv[..., i * H * D:(i + 1) * H * D].unflatten(-1, (H, D)) for i in range(3))
        do = (torch.randn(B, N, H, D, device=dev, generator=g) * 0.5).bfloat16()
        valid = torch.ones(B, N, dtype=torch.bool, device=dev)
        for name, mask in (("none", K.NO_MASK), ("keypad(all valid)", K.Mask.from_padding(valid, valid))):
            o, lse = K.attn_fwd(q, k, v, mask, q_prescaled=True)
            dqkv = torch.empty_like(qkv)
            dq, dk, dv = (dqkv[..., i * H * D:(i + 1) * H * D].unflatten(-1, (H, D)) for i in range(3))
            tf = t(lambda: K.attn_fwd(q, k, v, mask, q_prescaled=True))
            tb = t(lambda: K.attn_bwd(q, k, v, o, do, lse, dq, dk, dv, mask, q_prescaled=True))
            print(f"B={B:3d} N={N:3d} {name:18s} fwd {tf:7.1f} us  bwd {tb:7.1f} us")
