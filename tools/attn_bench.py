"""Isolated timing of the cfg2 attention calls (B=32, H=6, N=6144, D=64, block-causal 256, bf16): fwd and bwd."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frankenstein_amd import kernels as K

B, H, N, D, Cb = 32, 6, 6144, 64, 256
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
qkv = (torch.randn(B, N, 3 * H * D, device=dev, generator=g) * 0.5).to(torch.bfloat16)
q, k, v = (qkv[..., i * H * D:(i + 1) * H * D].unflatten(-1, (H, D)) for i in range(3))
mask = K.Mask(K.MASK_BLOCK_CAUSAL, Cb)
o, lse = K.attn_fwd(q, k, v, mask)
do = (torch.randn(B, N, H, D, device=dev, generator=g) * 0.5).to(torch.bfloat16)
dqkv = torch.empty_like(qkv)
dq, dk, dv = (dqkv[..., i * H * D:(i + 1) * H * D].unflatten(-1, (H, D)) for i in range(3))
vis = 0.5 * N * N * (1 + Cb / N)
PS = os.environ.get("ATTN_PS", "1") != "0"      # the pre-scaled-Q kernels (what the training step runs); ATTN_PS=0: the generic ones
if PS:
    o, lse = K.attn_fwd(q, k, v, mask, q_prescaled=True)
cases = {"attn_fwd": (lambda: K.attn_fwd(q, k, v, mask, q_prescaled=PS), 4 * B * H * vis * D),
         "attn_bwd": (lambda: K.attn_bwd(q, k, v, o, do, lse, dq, dk, dv, mask, q_prescaled=PS), 10 * B * H * vis * D)}
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
print("pre-scaled Q kernels" if PS else "generic kernels")
for f, _ in cases.values(): f()
torch.cuda.synchronize()
for name, (f, fl) in cases.items():
    ts = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); f(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 2)
    t = sorted(ts)[len(ts) // 2]
    print(f"{name}: {t:.3f} ms  {fl / t / 1e9:.1f} TF/s")
