"""At which row counts do the token-on-the-lane kernels (up-projection + SwiGLU, QKV + RoPE, the MLP backward chain) beat the tiled kernels
they replace?  One process times one routing of the two in-library switches (FK_MLP_UP_FUSED / FK_QKV_FUSED are read once); the backward
pair is timed both ways in every process.  d = 384, H = 1536, 6 heads of 64; M from SimpleMAE's encoder (38 400) to cfg2 (196 608)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frankenstein_amd import kernels as K
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
d, H, HD, D = 384, 1536, 384, 64
def rnd(*s): return (torch.randn(*s, device=dev, generator=g) * 0.5).to(torch.bfloat16)
def t(f, n=7):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[n // 2] * 1e3
w13, w2t, w13t, wqkv = rnd(2 * H, d), rnd(H, d), rnd(d, 2 * H), rnd(3 * HD, d)
tag = f"UP={os.environ.get('FK_MLP_UP_FUSED', '1')} QKV={os.environ.get('FK_QKV_FUSED', '1')}"
for M in (19200, 32768, 38400, 49152, 65536, 76800, 98304, 131072, 153600, 196608):
    T = 600 if M % 600 == 0 else (150 if M % 150 == 0 else 1024)
    ang = torch.arange(T, device=dev)[:, None] * torch.arange(D // 2, device=dev)[None, :] * 0.01
    pair = torch.stack([torch.stack([ang.cos(), ang.sin()], -1), torch.stack([ang.cos() * 0.18, ang.sin() * 0.18], -1)]).float().contiguous()
    x, dy, h13 = rnd(M, d), rnd(M, d), rnd(M, 2 * H)
    up = t(lambda: K.gemm_nt_swiglu(x, w13))
    qkv = t(lambda: K.gemm_nt_rope(x, wqkv, None, pair[0], T, 0, D, 2 * HD, q_cols=HD, q_table=pair[1]))
    fb = t(lambda: K.mlp_bwd_fused(dy, w2t, h13, w13t))
    two = t(lambda: K.gemm_nt(K.gemm_nt_dswiglu(dy, w2t, h13), w13t))
    print(f"{tag}  M={M:7d}  up+swiglu {up:7.1f} us  qkv+rope {qkv:7.1f} us  | mlp bwd: fused {fb:7.1f} us, two launches {two:7.1f} us", flush=True)
    del x, dy, h13
