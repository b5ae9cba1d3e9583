"""Run every cfg2-shaped forward/backward kernel twice on the same inputs and compare the outputs bit for bit."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frankenstein_amd import kernels as K
M, d, H = 32 * 6144, 384, 1536
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
def rnd(*s): return (torch.randn(*s, device=dev, generator=g) * 0.5).to(torch.bfloat16)
x, dy = rnd(M, d), rnd(M, d)
qkv, h13, gg, dh13 = rnd(M, 3 * d), rnd(M, 2 * H), rnd(M, H), rnd(M, 2 * H)
w_qkv, w_proj, w13, w2 = rnd(3 * d, d), rnd(d, d), rnd(2 * H, d), rnd(d, H)
w13t, w2t, w_qkvt = rnd(d, 2 * H), rnd(H, d), rnd(d, 3 * d)
table = torch.randn(6144, 32, 2, device=dev)
gam, bet = torch.randn(d, device=dev), torch.randn(d, device=dev)
B, Hh, N, D = 32, 6, 6144, 64
q3 = qkv.view(B, N, 3 * d)
q, k, v = (q3[..., i * d:(i + 1) * d].unflatten(-1, (Hh, D)) for i in range(3))
mask = K.Mask(K.MASK_BLOCK_CAUSAL, 256)
o, lse = K.attn_fwd(q, k, v, mask)
do = rnd(B, N, Hh, D)
def bwd():
    dqkv = torch.empty_like(q3)
    dq, dk, dv = (dqkv[..., i * d:(i + 1) * d].unflatten(-1, (Hh, D)) for i in range(3))
    K.attn_bwd(q, k, v, o, do, lse, dq, dk, dv, mask)
    return dqkv
cases = {
    "gemm_nt_rope qkv": lambda: K.gemm_nt_rope(x, w_qkv, None, table, 6144, 0, 64, 2 * d),
    "gemm_nt proj+res": lambda: K.gemm_nt(x, w_proj, None, residual=dy),
    "gemm_nt_swiglu": lambda: torch.cat([t.reshape(-1) for t in K.gemm_nt_swiglu(x, w13)]),
    "gemm_nt down+res": lambda: K.gemm_nt(gg, w2, None, residual=dy),
    "gemm_nt_dswiglu": lambda: K.gemm_nt_dswiglu(dy, w2t, h13),
    "gemm_nt d_up": lambda: K.gemm_nt(dh13, w13t),
    "gemm_nt d_qkv": lambda: K.gemm_nt(qkv, w_qkvt),
    "gemm_tn dW_qkv": lambda: K.gemm_tn(qkv, x),
    "gemm_tn dW_up": lambda: K.gemm_tn(dh13, x),
    "gemm_tn dW_down": lambda: K.gemm_tn(dy, gg),
    "gemm_tn dW_proj": lambda: K.gemm_tn(dy, x),
    "attn_fwd": lambda: K.attn_fwd(q, k, v, mask)[0],
    "attn_bwd": bwd,
    "norm_fwd": lambda: K.norm_fwd(x, gam, bet, 1e-5)[0],
    "norm_bwd": lambda: K.norm_bwd(dy, x, gam, *K.norm_fwd(x, gam, bet, 1e-5)[1:], dres=dy)[0],
}
for name, f in cases.items():
    ref = f().clone(); torch.cuda.synchronize()
    bad = 0; mx = 0.0
    for _ in range(4):
        out = f(); torch.cuda.synchronize()
        if not torch.equal(out, ref):
            bad += 1; mx = max(mx, float((out.float() - ref.float()).abs().max()))
    print(f"{name:22s} {'DETERMINISTIC' if bad == 0 else f'DIFFERS in {bad}/4 repeats, max abs diff {mx:.4g}'}", flush=True)
