"""Do a matrix-bound launch and a traffic-bound launch overlap when they run on two streams?  
cfg2 shapes on one MI355X: A = fk_attn_bwd of one encoder layer (3 ms, HBM nearly idle), B = ONE long HBM-bound launch on a second stream —
(i) a 15-GB elementwise add (small workgroups, no LDS: co-resides with anything), (ii) the up-projection + SwiGLU GEMM (8 waves and the whole
160 KiB of LDS per CU: cannot share a CU with an attention workgroup).  Prints t(A), t(B), t(A || B)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frankenstein_amd import kernels as K

dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
B, H, N, D, Cb = 32, 6, 6144, 64, 256
M, d = B * N, H * D
qkv = (torch.randn(B, N, 3 * d, device=dev, generator=g) * 0.5).bfloat16()
q, k, v = (qkv[..., i * d:(i + 1) * d].unflatten(-1, (H, D)) for i in range(3))
mask = K.Mask(K.MASK_BLOCK_CAUSAL, Cb)
o, lse = K.attn_fwd(q, k, v, mask, q_prescaled=True)
do = (torch.randn(B, N, H, D, device=dev, generator=g) * 0.5).bfloat16()
dqkv = torch.empty_like(qkv)
dq, dk, dv = (dqkv[..., i * d:(i + 1) * d].unflatten(-1, (H, D)) for i in range(3))
x = (torch.randn(M, d, device=dev, generator=g)).bfloat16()
gam, bet = torch.ones(d, device=dev), torch.zeros(d, device=dev)
w13 = (torch.randn(2 * 1536, d, device=dev, generator=g) * 0.05).bfloat16()


def attn():
    K.attn_bwd(q, k, v, o, do, lse, dq, dk, dv, mask, q_prescaled=True)


def norms(n=60):
    for _ in range(n):
        K.norm_fwd(x, gam, bet, 1e-5)


def gemms(n=4):
    for _ in range(n):
        K.gemm_nt_swiglu(x, w13)


def timed(fa, fb, reps=5):
    """fa on one stream, fb on another (either may be None): eager launches, so each side is ONE or two long kernels"""
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cur = torch.cuda.current_stream()
        e0.record()
        sa.wait_stream(cur)
        sb.wait_stream(cur)
        if fb is not None:
            with torch.cuda.stream(sb):
                fb()
        if fa is not None:
            with torch.cuda.stream(sa):
                fa()
        cur.wait_stream(sa)
        cur.wait_stream(sb)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


big_a = torch.ones(2_500_000_000, dtype=torch.bfloat16, device=dev)
big_b = torch.ones_like(big_a)
big_o = torch.empty_like(big_a)


def stream_add():                                  # one launch: reads 10 GB, writes 5 GB
    from frankenstein_amd._lib import call
    call("fk_add", big_a.data_ptr(), big_b.data_ptr(), big_o.data_ptr(), big_a.numel(), K.fk_dtype(big_a), K._stream())


def one_gemm():
    K.gemm_nt_swiglu(x, w13)


for f in (attn, stream_add, one_gemm):
    f()
torch.cuda.synchronize()
for name, fb in (("one 15-GB elementwise add (small workgroups, no LDS)", stream_add), ("one up-projection + SwiGLU (whole-CU grid)", one_gemm)):
    ta, tb, tab = timed(attn, None), timed(None, fb), timed(attn, fb)
    print(f"A = fk_attn_bwd {ta:.3f} ms | B = {name} {tb:.3f} ms | A || B {tab:.3f} ms  (sum {ta + tb:.3f}, max {max(ta, tb):.3f})")
