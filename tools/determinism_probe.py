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
    "mlp_bwd_fused": lambda: torch.cat([t_.reshape(-1) for t_ in K.mlp_bwd_fused(dy, w2t, h13, w13t)]),
    "gemm_nt d_qkv": lambda: K.gemm_nt(qkv, w_qkvt),
    "gemm_nt fp32 out": lambda: K.gemm_nt(x[:16384], w_qkv, out_dtype=torch.float32),
    "gemm_nt fp32 out 256-tile": lambda: K.gemm_nt(x[:16384], w13, out_dtype=torch.float32),
    "gemm_nt bias+res": lambda: K.gemm_nt(x, w_proj, gam.bfloat16(), residual=dy),
    "gemm_tn dW_qkv": lambda: K.gemm_tn(qkv, x),
    "gemm_tn dW_up": lambda: K.gemm_tn(dh13, x),
    "gemm_tn dW_down": lambda: K.gemm_tn(dy, gg),
    "gemm_tn dW_proj": lambda: K.gemm_tn(dy, x),
    "attn_fwd": lambda: K.attn_fwd(q, k, v, mask)[0],
    "attn_bwd": bwd,
    "norm_fwd": lambda: K.norm_fwd(x, gam, bet, 1e-5)[0],
    "norm_bwd": lambda: K.norm_bwd(dy, x, gam, *K.norm_fwd(x, gam, bet, 1e-5)[1:], dres=dy)[0],
}
# full-size correctness against an independent implementation (rocBLAS through torch, fp32 accumulation of the same bf16 inputs):
# a deterministic wrong tile would pass the bit-for-bit repeat check below
def rel_err(out, ref):
    return float((out.float() - ref).abs().max() / ref.abs().max())
xf, dyf = x.float(), dy.float()
checks = {
    "gemm_nt proj+res vs rocBLAS": (K.gemm_nt(x, w_proj, None, residual=dy), xf @ w_proj.float().t() + dyf),
    "gemm_nt down+res vs rocBLAS": (K.gemm_nt(gg, w2, None, residual=dy), gg.float() @ w2.float().t() + dyf),
    "gemm_nt d_up vs rocBLAS": (K.gemm_nt(dh13, w13t), dh13.float() @ w13t.float().t()),
    "gemm_nt d_qkv vs rocBLAS": (K.gemm_nt(qkv, w_qkvt), qkv.float() @ w_qkvt.float().t()),
    "gemm_tn dW_up vs rocBLAS": (K.gemm_tn(dh13, x), dh13.float().t() @ xf),
    "gemm_tn dW_qkv vs rocBLAS": (K.gemm_tn(qkv, x), qkv.float().t() @ xf),
}
fd, fx = K.mlp_bwd_fused(dy, w2t, h13, w13t)          # at full size: the fused data-gradient chain IS the two launches it replaces, bit for bit
d_two = K.gemm_nt_dswiglu(dy, w2t, h13)
print(f"{'mlp_bwd_fused vs dswiglu + d_up (bits)':40s} {'OK' if torch.equal(fd, d_two) and torch.equal(fx, K.gemm_nt(d_two, w13t)) else 'MISMATCH'}", flush=True)
del fd, fx, d_two
h13o, go = K.gemm_nt_swiglu(x, w13)
h13r = xf @ w13.float().t()
checks["gemm_nt_swiglu h13 vs rocBLAS"] = (h13o, h13r)
hr = h13o.float().view(M, H // 4, 2, 4)
checks["gemm_nt_swiglu g vs its own h13"] = (go, (torch.nn.functional.silu(hr[:, :, 0]) * hr[:, :, 1]).reshape(M, H))
qr = xf @ w_qkv.float().t()
qq = qr[:, :2 * d].reshape(32, 6144, 12, 32, 2)
cs = table[None, :, None]
rot = torch.stack([qq[..., 0] * cs[..., 0] - qq[..., 1] * cs[..., 1], qq[..., 0] * cs[..., 1] + qq[..., 1] * cs[..., 0]], -1).reshape(M, 2 * d)
checks["gemm_nt_rope qkv vs rocBLAS + rotation"] = (K.gemm_nt_rope(x, w_qkv, None, table, 6144, 0, 64, 2 * d), torch.cat([rot, qr[:, 2 * d:]], 1))
for name, (out, ref) in checks.items():
    e = rel_err(out, ref)
    print(f"{name:40s} {'OK' if e < 2e-2 else 'MISMATCH'}  max err / max |ref| = {e:.2e}", flush=True)
# bf16 LDS-DMA attention kernels at the full N = 6144 against the fp32 kernels (a different code path, itself pinned to the reference at
# this N by the cfg2_b1 golden) on the same bf16-representable inputs, 2 samples
def attn_pair(dtype):
    qs, ks, vs, dos = (t[:2].to(dtype).contiguous() for t in (q, k, v, do))
    oo, ll = K.attn_fwd(qs, ks, vs, mask)
    dq_, dk_, dv_ = torch.empty_like(qs), torch.empty_like(ks), torch.empty_like(vs)
    K.attn_bwd(qs, ks, vs, oo, dos, ll, dq_, dk_, dv_, mask)
    return oo, dq_, dk_, dv_
lo, hi = attn_pair(torch.bfloat16), attn_pair(torch.float32)
for nm, a_, b_ in zip(("attn_fwd o", "attn_bwd dq", "attn_bwd dk", "attn_bwd dv"), lo, hi):
    e = rel_err(a_, b_.float())
    print(f"{nm + ' bf16 vs fp32 kernels, N=6144':40s} {'OK' if e < 3e-2 else 'MISMATCH'}  max err / max |ref| = {e:.2e}", flush=True)
del checks, h13r, hr, qr, qq, rot, xf, dyf, lo, hi
torch.cuda.empty_cache()

for name, f in cases.items():
    ref = f().clone(); torch.cuda.synchronize()
    bad = 0; mx = 0.0
    for _ in range(4):
        out = f(); torch.cuda.synchronize()
        if not torch.equal(out, ref):
            bad += 1; mx = max(mx, float((out.float() - ref.float()).abs().max()))
    print(f"{name:22s} {'DETERMINISTIC' if bad == 0 else f'DIFFERS in {bad}/4 repeats, max abs diff {mx:.4g}'}", flush=True)
