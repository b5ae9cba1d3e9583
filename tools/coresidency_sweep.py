"""Which kernels return different bits when another kernel shares their CUs, and which co-resident kernel does it?

Part 1 (victims): every entry point at a size that leaves room on the CUs, run once on a quiet device and REPS times with the small
weight-gradient GEMM (the occupant that made fk_attn_bwd's RoPE epilogue return wrong dQ / dK values, tools/dq_coresidency_probe.py)
looping on a second stream; outputs compared bit for bit.
Part 2 (occupants): the known victim (fk_attn_bwd with the fused inverse RoPE) beside different occupants.
    FRANKEN_HIP_LIB=... python tools/coresidency_sweep.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frankenstein_amd import kernels as K

REPS = int(os.environ.get("PROBE_REPS", "4"))
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)


def rnd(*s, dt=torch.bfloat16):
    return (torch.randn(*s, device=dev, generator=g) * 0.5).to(dt)


M, d, Hh = 14592, 320, 840
x, dy = rnd(M, d), rnd(M, d)
w_qkv, w_proj, w13, w2 = rnd(3 * d, d), rnd(d, d), rnd(2 * 848, d), rnd(d, 848)
h13, gg = rnd(M, 2 * 848), rnd(M, 848)
gam, bet = torch.randn(d, device=dev, generator=g), torch.randn(d, device=dev, generator=g)
B, H, N, D = 3, 5, 4864, 64
table = torch.randn(N, D // 2, 2, device=dev, generator=g)
qkv = rnd(B * N, 3 * d)
q3 = qkv.view(B, N, 3 * d)
q, k, v = (q3[..., i * d:(i + 1) * d].unflatten(-1, (H, D)) for i in range(3))
mask = K.Mask(K.MASK_BLOCK_CAUSAL, 256)
o, lse = K.attn_fwd(q, k, v, mask)
do = rnd(B, N, H, D)
xf32 = torch.randn(M, d, device=dev, generator=g)
p32, g32, m32, v32 = (torch.randn(1 << 22, device=dev, generator=g) for _ in range(4))
v32 = v32.abs()
logits = rnd(4096, 3000, dt=torch.float32)
tgt = torch.randint(0, 3000, (4096,), device=dev, generator=g)


def bwd(rope, prescaled=False):
    dqkv = torch.empty_like(q3)
    dq, dk, dv = (dqkv[..., i * d:(i + 1) * d].unflatten(-1, (H, D)) for i in range(3))
    if rope:
        K.attn_bwd(q, k, v, o, do, lse, dq, dk, dv, mask, rope_table=table, rope_off=0, q_prescaled=prescaled)
    else:
        K.attn_bwd(q, k, v, o, do, lse, dq, dk, dv, mask, q_prescaled=prescaled)
    return dqkv


def rope_inplace():
    t = q3.clone()
    K.rope_(t, 2 * H, D, table, 0)
    return t


def adamw():
    p, gr, m_, v_ = p32.clone(), g32.clone(), m32.clone(), v32.clone()
    K.adamw_step_(p, gr, m_, v_, 3, 1e-3, clip=1.0, zero_grad=True)
    return torch.cat([p, m_, v_])


def ce():
    loss2, l = K.ce_loss_fwd(logits, tgt, -100)
    return torch.cat([loss2.flatten(), l.flatten()])


nm = K.norm_fwd(x, gam, bet, 1e-5)
victims = {
    "attn_bwd + inverse rope (generic)": lambda: bwd(True),
    "attn_bwd + inverse rope (prescaled)": lambda: bwd(True, True),
    "attn_bwd no rope": lambda: bwd(False),
    "attn_fwd": lambda: K.attn_fwd(q, k, v, mask)[0],
    "gemm_nt_rope qkv": lambda: K.gemm_nt_rope(x, w_qkv, None, table, N, 0, D, 2 * d),
    "gemm_nt proj + residual": lambda: K.gemm_nt(x, w_proj, None, residual=dy),
    "gemm_nt bias + residual": lambda: K.gemm_nt(x, w_proj, gam.bfloat16(), residual=dy),
    "gemm_nt_swiglu": lambda: torch.cat([t.reshape(-1) for t in K.gemm_nt_swiglu(x, w13)]),
    "gemm_nt_dswiglu": lambda: K.gemm_nt_dswiglu(dy, w2.t().contiguous(), h13),
    "gemm_nt fp32 out": lambda: K.gemm_nt(x, w_qkv, out_dtype=torch.float32),
    "gemm_nt fp32 operands": lambda: K.gemm_nt(xf32[:4096], w_proj.float()),
    "gemm_tn": lambda: K.gemm_tn(dy, x),
    "norm_fwd": lambda: K.norm_fwd(x, gam, bet, 1e-5)[0],
    "norm_fwd rms": lambda: K.norm_fwd(x, gam, None, 1e-6, K.NORM_RMS)[0],
    "norm_bwd + residual": lambda: K.norm_bwd(dy, x, gam, nm[1], nm[2], dres=dy)[0],
    "rope_": rope_inplace,
    "swiglu_fwd": lambda: K.swiglu_fwd(h13),
    "swiglu_bwd": lambda: K.swiglu_bwd(h13, gg),
    "gelu_fwd": lambda: K.gelu_fwd(h13),
    "gelu_bwd": lambda: K.gelu_bwd(h13, h13),
    "adamw_step": adamw,
    "ce_loss_fwd": ce,
    "colsum": lambda: K.colsum(dy),
    "cast bf16->f32": lambda: K.cast(x, torch.float32),
}

side = torch.cuda.Stream()
ga, gb = rnd(M, 320), rnd(M, 840)


def occ_tn():
    for _ in range(10):
        K.gemm_tn(ga, gb)


def occ_attn_fwd():
    for _ in range(3):
        K.attn_fwd(q, k, v, mask)


def occ_norm():
    for _ in range(60):
        K.norm_bwd(dy, x, gam, nm[1], nm[2], dres=dy)


def occ_swiglu():
    for _ in range(60):
        K.swiglu_bwd(h13, gg)


def occ_nt_small():
    for _ in range(60):
        K.gemm_nt(x[:2048], w_qkv)


def occ_tn_fp32():
    a32, b32 = ga[:4096].float(), gb[:4096].float()
    for _ in range(10):
        K.gemm_tn(a32, b32)


def contended(fn, occ):
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        occ()
    out = fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    return out


print("== victims beside the small weight-gradient GEMM", flush=True)
for name, fn in victims.items():
    try:
        ref = fn().clone()
        torch.cuda.synchronize()
        again = fn()
        torch.cuda.synchronize()
        quiet_ok = torch.equal(ref, again)
        bad, nel = 0, 0
        for _ in range(REPS):
            out = contended(fn, occ_tn)
            if not torch.equal(out, ref):
                bad += 1
                nel = max(nel, int((out != ref).sum()))
        print(f"{name:40s} quiet repeat {'same' if quiet_ok else 'DIFFERS'};  contended: "
              f"{'same bits' if bad == 0 else f'DIFFERS in {bad}/{REPS} runs (up to {nel} elements)'}", flush=True)
    except Exception as e:                                   # a case that does not apply to this build: say so and go on
        print(f"{name:40s} skipped: {type(e).__name__}: {str(e)[:120]}", flush=True)

print("== occupants beside fk_attn_bwd with the fused inverse RoPE", flush=True)
fn = victims["attn_bwd + inverse rope (generic)"]
ref = fn().clone()
torch.cuda.synchronize()
for name, occ in (("gemm_tn small (bf16)", occ_tn), ("gemm_tn small (fp32)", occ_tn_fp32), ("attn_fwd", occ_attn_fwd), ("norm_bwd", occ_norm),
                  ("swiglu_bwd", occ_swiglu), ("gemm_nt small M", occ_nt_small)):
    bad, nel = 0, 0
    for _ in range(REPS):
        out = contended(fn, occ)
        if not torch.equal(out, ref):
            bad += 1
            nel = max(nel, int((out != ref).sum()))
    print(f"occupant {name:24s}: {'same bits' if bad == 0 else f'DIFFERS in {bad}/{REPS} runs (up to {nel} elements)'}", flush=True)
