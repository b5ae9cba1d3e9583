"""Does fk_attn_bwd return the same bits when another stream keeps the CUs busy?  (DESIGN.md 5.1 'side-stream dQ differences')

One forward + backward of the ragged-shape model (B = 3, N = 4864 tokens, 5 heads x 64: tests/test_models_gpu.py::
test_bf16_odd_shapes_train_steps) records the arguments of its LAST attention backward (encoder layer 0).  That call is then replayed
on resident operands: quiet (nothing else on the device) and with an occupant looping on a second stream — the small weight-gradient
GEMM, a plain copy (HBM traffic only), a VALU-only kernel (torch elementwise chain).  Before every replay the output buffer is filled with
a sentinel, so an element the kernels did not write shows up as the sentinel; every differing element is mapped to (batch, token,
head, d) -> (workgroup tile, wave, lane group, accumulator register) and compared with an fp64 evaluation of the same formula, in
units of its bf16 ulp: a rounding-boundary flip (tiny perturbation) and a missing / stale contribution look different there."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import frankenstein_amd as fa
from frankenstein_amd import kernels as K
from frankenstein_amd.models import brainformer as bf

REPS = int(os.environ.get("PROBE_REPS", "12"))
fa.set_compute_dtype("bf16")
enc = bf.MAEConfig(window_size=475, n_electrodes=256, patch_size=25, dim=320, n_layers=2, head_dim=64, hidden_dim=840, n_heads=5, n_kv_heads=5)
cfg = bf.Config(encoder=enc, n_output_tokens=9, output_dim=70, dim=320, n_layers=1, head_dim=64, hidden_dim=328, n_heads=5, n_kv_heads=5)
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.randn(3, 475, 256, device="cuda", generator=g)
y = torch.randn(3, 9, 70, device="cuda", generator=g)
torch.manual_seed(0)
m = bf.BrainFormer(cfg).cuda()

calls = []
orig = K.attn_bwd


def rec(*a, **kw):
    calls.append((a, kw))
    return orig(*a, **kw)


K.attn_bwd = rec
loss, _ = m(x, y, date_info=None)
loss.backward()
torch.cuda.synchronize()
K.attn_bwd = orig
args, kw = [c for c in calls if c[0][0].shape[1] == 4864][-1]
q, k, v, o, do, lse, dq, dk, dv, mask = args[:10]
B, N, H, D = q.shape
HD = H * D
print(f"replaying attn_bwd B={B} N={N} H={H} D={D} mask kind {mask.kind} c={mask.c} prescaled={kw.get('q_prescaled')} "
      f"rope={'yes' if kw.get('rope_table') is not None else 'no'}", flush=True)
keep = [t.clone() for t in (q, k, v, o, do, lse)]          # the operands must not change either


def replay(sentinel: float):
    dqkv = torch.full((B, N, 3 * HD), sentinel, dtype=q.dtype, device=q.device)
    d3 = [dqkv[..., i * HD:(i + 1) * HD].unflatten(-1, (H, D)) for i in range(3)]
    orig(q, k, v, o, do, lse, d3[0], d3[1], d3[2], mask, **kw)
    return dqkv


base = replay(7.0)
torch.cuda.synchronize()
for i in range(2):
    again = replay(-3.0 - i)
    torch.cuda.synchronize()
    print("quiet replay", i, "identical to the first:", bool(torch.equal(base, again)), flush=True)

# ---- occupants on a second stream
side = torch.cuda.Stream()
M = B * N
ga = (torch.randn(M, 320, device="cuda", generator=g) * 0.5).bfloat16()
gb = (torch.randn(M, 840, device="cuda", generator=g) * 0.5).bfloat16()
big = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
big2 = torch.empty_like(big)
small = torch.randn(1 << 22, device="cuda")


def occ_tn():
    for _ in range(12):
        K.gemm_tn(ga, gb)


def occ_copy():
    for _ in range(8):
        big2.copy_(big)


def occ_valu():
    t = small
    for _ in range(40):
        t = torch.sin(t) * 1.0001 + 0.5


def occ_nt():
    for _ in range(12):
        K.gemm_nt(ga, gb[:840 * 8].view(-1, 320)[:840])


cs_tab = kw.get("rope_table")
rope_off = kw.get("rope_off", 0)
prescaled = bool(kw.get("q_prescaled"))
LOG2E = 1.4426950408889634
scale = 1.0 / (D ** 0.5)


def fp64_rows(b, h, rows):
    """dQ rows (after the inverse RoPE) for batch b, head h, query rows `rows`, evaluated in fp64 from the bf16 operands; dS is
    rounded to bf16 like the kernel's MFMA operand."""
    rows_t = torch.tensor(rows, device="cuda")
    qd = q[b, rows_t, h].double()
    kd, vd = k[b, :, h].double(), v[b, :, h].double()
    od, dod = o[b, rows_t, h].double(), do[b, rows_t, h].double()
    ls = lse[b, h, rows_t].double()
    s = qd @ kd.t()
    if prescaled:
        p = torch.exp2(s - ls[:, None] * LOG2E)
    else:
        p = torch.exp(s * scale - ls[:, None])
    kidx = torch.arange(N, device="cuda")
    vis = (kidx[None, :] // mask.c) <= (rows_t[:, None] // mask.c) if mask.kind == K.MASK_BLOCK_CAUSAL else torch.ones_like(p, dtype=torch.bool)
    p = torch.where(vis, p, torch.zeros_like(p))
    dp = dod @ vd.t()
    delta = (dod * od).sum(-1, keepdim=True)
    ds = (p * (dp - delta)).float().bfloat16().double()
    dqr = (ds @ kd) * scale
    if cs_tab is not None:
        t = cs_tab[rope_off + rows_t].double() if cs_tab.dim() == 3 else cs_tab[b, rope_off + rows_t].double()      # [r, D/2, 2]
        c, s_ = t[..., 0], t[..., 1]
        a0, a1 = dqr[:, 0::2], dqr[:, 1::2]
        out = torch.empty_like(dqr)
        out[:, 0::2] = a0 * c + a1 * s_
        out[:, 1::2] = -a0 * s_ + a1 * c
        dqr = out
    return dqr


def bf16_ulp(x):
    e = torch.floor(torch.log2(x.abs().clamp_min(1e-45)))
    return torch.exp2(e - 7)


def report(name, got, rep):
    if torch.equal(got, base):
        return 0
    diff = (got != base)
    nd = int(diff.sum())
    idx = diff.nonzero()
    print(f"[{name} rep {rep}] {nd} differing elements", flush=True)
    part = {0: "dQ", 1: "dK", 2: "dV"}
    seen = set()
    for b_, n_, c_ in idx[:4096].tolist():
        third, col = c_ // HD, c_ % HD
        h_, d_ = col // D, col % D
        key = (b_, n_ // 16, third, h_, d_)
        if key in seen:
            continue
        seen.add(key)
        rows = [r for r in range(n_ // 16 * 16, n_ // 16 * 16 + 16)]
        line = (f"   {part[third]} b={b_} rows {rows[0]}..{rows[-1]} (tile q0={n_ // 128 * 128}, wave {(n_ % 128) // 32}, rows-in-wave {(n_ % 32) // 16 * 16}+) "
                f"head {h_} d={d_} (dt={d_ // 32}, g={(d_ % 32) // 8}, lh={(d_ % 8) // 4}, j={d_ % 4}); ")
        bv, gv = base[b_, rows, c_].double(), got[b_, rows, c_].double()
        line += f"differing rows in the group: {int((bv != gv).sum())}; sentinel hit: {bool((gv == SENT).any())}; "
        if third == 0:
            ref = fp64_rows(b_, h_, rows)[:, d_]
            ulp = bf16_ulp(ref)
            eb, eg = (bv - ref) / ulp, (gv - ref) / ulp
            msk = bv != gv
            line += (f"|value| {float(ref.abs().min()):.2e}..{float(ref.abs().max()):.2e}; (quiet - fp64)/ulp on differing rows: "
                     f"{[round(float(t), 2) for t in eb[msk][:6]]}; (contended - fp64)/ulp: {[round(float(t), 2) for t in eg[msk][:6]]}; "
                     f"max |err|/ulp over the 16 rows quiet {float(eb.abs().max()):.2f} contended {float(eg.abs().max()):.2f}")
        print(line, flush=True)
        if len(seen) >= 12:
            break
    return nd


SENT = 0.0
total = {}
OCCS = os.environ.get("PROBE_OCC", "tn,copy,valu,nt").split(",")
for name, occ in (("tn", occ_tn), ("copy", occ_copy), ("valu", occ_valu), ("nt", occ_nt)):
    if name not in OCCS:
        continue
    cnt = 0
    for rep in range(REPS):
        SENT = 11.0 + rep
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            occ()
        got = replay(SENT)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        cnt += report(name, got, rep) > 0
    total[name] = cnt
    print(f"occupant {name}: {cnt} of {REPS} contended replays differ from the quiet one", flush=True)
for t_, k_ in zip((q, k, v, o, do, lse), keep):
    assert torch.equal(t_, k_), "an operand changed"
final = replay(5.0)
torch.cuda.synchronize()
print("quiet replay after the contended ones identical:", bool(torch.equal(final, base)))
print("summary", total)
