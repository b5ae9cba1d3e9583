"""Kernel-level parity on the MI355X: every C-ABI entry point against the CPU oracle primitives
(oracle/ref_models.py, oracle/ref_train.py) on seeded inputs.  fp32 mode is held to ~1e-5,
bf16 mode to bf16 rounding of a reference computed from the same bf16-rounded inputs."""
import math

import numpy as np
import pytest
import torch

from oracle import ref_models as R
from oracle import ref_train as RT

pytestmark = pytest.mark.gpu

DT = [torch.float32, torch.bfloat16]


@pytest.fixture(scope="module")
def K():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from frankenstein_amd import kernels
    return kernels


def dev(t, dtype=None):
    t = t.to("cuda")
    return t.to(dtype) if dtype is not None else t


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def q(t, dtype):
    """round-trip through the compute dtype so the reference sees the same operand values"""
    return t.to(dtype).float().clone()


def close(got, want, dtype, atol32=2e-5, rtol32=2e-5, atol16=None, rtol16=2e-2):
    got = got.detach().float().cpu()
    want = want.detach().float().cpu()
    if dtype == torch.float32:
        torch.testing.assert_close(got, want, atol=atol32, rtol=rtol32)
    else:
        a = atol16 if atol16 is not None else 2e-2 * max(1.0, float(want.abs().max()))
        torch.testing.assert_close(got, want, atol=a, rtol=rtol16)


# ----------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("dtype", DT)
def test_gemm_nt_exact_integers(K, dtype):
    # asymmetric small-integer operands: products/sums exact in both dtypes -> catches any fragment / C-layout swap
    M, N, Kd = 200, 136, 72
    a = torch.randint(-3, 4, (M, Kd), generator=torch.Generator().manual_seed(1)).float()
    w = torch.randint(-3, 4, (N, Kd), generator=torch.Generator().manual_seed(2)).float()
    w[:, 0] += torch.arange(N) % 5
    got = K.gemm_nt(dev(a, dtype), dev(w, dtype), out_dtype=torch.float32)
    assert torch.equal(got.cpu(), a @ w.t())


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shape", [(300, 200, 64), (128, 128, 384), (1000, 1003, 128), (64, 32, 32), (777, 384, 1536)])
def test_gemm_nt(K, dtype, shape):
    M, N, Kd = shape
    a, w = rnd(M, Kd, seed=1), rnd(N, Kd, seed=2, scale=1 / math.sqrt(Kd))
    bias, res = rnd(N, seed=3), rnd(M, N, seed=4)
    ref = q(a, dtype) @ q(w, dtype).t()
    close(K.gemm_nt(dev(a, dtype), dev(w, dtype)), ref, dtype)
    close(K.gemm_nt(dev(a, dtype), dev(w, dtype), bias=dev(bias, dtype), residual=dev(res, dtype)),
          ref + q(bias, dtype) + q(res, dtype), dtype)
    tab = rnd(7, N, seed=5)
    want = ref + q(tab, dtype)[torch.arange(M) % 7]
    close(K.gemm_nt(dev(a, dtype), dev(w, dtype), residual=dev(tab, dtype), res_rows=7, out_dtype=torch.float32), want, dtype)


@pytest.mark.parametrize("dtype", DT)
def test_gemm_nt_strided_views(K, dtype):
    big = dev(rnd(100, 3 * 64, seed=1), dtype)
    a = big[:, 64:128]
    w = dev(rnd(48, 64, seed=2), dtype)
    out = torch.zeros(100, 96, device="cuda", dtype=dtype)
    K.gemm_nt(a, w, out=out[:, 48:])
    close(out[:, 48:], a.float().cpu() @ w.float().cpu().t(), dtype)
    assert float(out[:, :48].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", DT)
def test_gemm_tn_exact_integers(K, dtype):
    M, N1, N2 = 333, 136, 72
    a = torch.randint(-2, 3, (M, N1), generator=torch.Generator().manual_seed(1)).float()
    b = torch.randint(-2, 3, (M, N2), generator=torch.Generator().manual_seed(2)).float()
    b[:, 1] += torch.arange(M) % 3
    got = K.gemm_tn(dev(a, dtype), dev(b, dtype))
    assert torch.equal(got.cpu(), a.t() @ b)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shape", [(1000, 136, 264), (4113, 384, 32), (70, 64, 64), (20000, 128, 512)])
def test_gemm_tn(K, dtype, shape):
    M, N1, N2 = shape
    a, b = rnd(M, N1, seed=1), rnd(M, N2, seed=2)
    ref = q(a, dtype).t() @ q(b, dtype)
    got = K.gemm_tn(dev(a, dtype), dev(b, dtype))
    close(got, ref, torch.float32, atol32=2e-3 if dtype == torch.float32 else 5e-2 * math.sqrt(M / 1000), rtol32=1e-4 if dtype == torch.float32 else 2e-2)
    acc = torch.ones(N1, N2, device="cuda")
    K.gemm_tn(dev(a, dtype), dev(b, dtype), out=acc, accumulate=True)
    close(acc, ref + 1, torch.float32, atol32=2e-3 if dtype == torch.float32 else 5e-2 * math.sqrt(M / 1000), rtol32=1e-4 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("shape", [(16384 + 192, 384, 256), (24576, 768, 128), (16384, 384, 384), (16384 + 32, 384, 128), (16384 + 96, 768, 256),
                                   # the 384 x 192 tile (512-byte-pitch B image with padding): taken for N2 % 192 == 0 with N1 >= 2048, or when 128 does not divide N2
                                   (16384 + 64, 2304, 384), (32768, 384, 1536), (16384 + 32, 768, 192)])
def test_gemm_tn_large_tile_exact_integers(K, shape):
    """the 384x128-tile LDS-DMA weight-gradient kernel (bf16, M % 32 == 0, N1 % 384 == 0, N2 % 128 == 0; 32-row stages in a 4-slot
    ring, so splits with an odd and an even number of stages and fewer stages than slots in the tail are all here): asymmetric integer
    operands make the result exact, so any fragment / swizzle / split mix-up shows up as a hard mismatch; strided views too."""
    M, N1, N2 = shape
    g = torch.Generator().manual_seed(5)
    a = torch.randint(-2, 3, (M, N1 + 8), generator=g).float()
    b = torch.randint(-2, 3, (M, N2), generator=g).float()
    a[:, 3] += (torch.arange(M) % 5).float()
    b[:, 1] += (torch.arange(M) % 3).float()
    ad = dev(a, torch.bfloat16)[:, :N1]                     # row stride N1 + 8
    bd = dev(b, torch.bfloat16)
    ref = (a[:, :N1].double().t() @ b.double()).float()
    assert float(ref.abs().max()) < 2 ** 24
    got = K.gemm_tn(ad, bd)
    assert torch.equal(got.cpu(), ref)
    acc = torch.full((N1, N2), 3.0, device="cuda")
    K.gemm_tn(ad, bd, out=acc, accumulate=True)
    assert torch.equal(acc.cpu(), ref + 3)


RING_SHAPES = [(4096, 384, 64), (4168, 128, 128), (8200, 1152, 384), (4104, 256, 64), (70000, 256, 128), (66000, 384, 192),
               (4096, 512, 128), (33000, 768, 64),
               # long K, N = 384 class: the kernel with split request waves and the 4-slot A ring (one / several tiles per block, ragged rows)
               (4224, 384, 1536), (8200, 128, 1024), (70000, 384, 1088), (4096, 640, 3072)]


@pytest.mark.parametrize("routing", ["ring", "default"])
@pytest.mark.parametrize("shape", RING_SHAPES)
def test_gemm_nt_ring_kernels_exact_integers(K, shape, routing, monkeypatch):
    """the ring-buffered wide-projection kernels (bf16, M >= 4096, N % 128 == 0, K % 64 == 0): one and several tiles per persistent
    block (the stage stream crosses tile boundaries), a single k-stage, fewer stages than ring slots, ragged last row tile, 256- and
    128-column tiles.  Integer operands make every sum exact: fp32 output must be bit-equal to the reference, bf16 output to its
    single rounding; bias / residual / periodic residual ride on the same epilogue.  routing "ring": every shape goes to the ring
    kernels (FK_NT_RING_MIN_TILES=0; by default grids under 128 tiles of 256 rows take the 128 x 128 kernels, routing "default")."""
    if routing == "ring":
        monkeypatch.setenv("FK_NT_RING_MIN_TILES", "0")
    else:
        monkeypatch.delenv("FK_NT_RING_MIN_TILES", raising=False)
    M, N, Kd = shape
    g = torch.Generator().manual_seed(7)
    a = torch.randint(-2, 3, (M, Kd + 8), generator=g).float()
    w = torch.randint(-2, 3, (N, Kd), generator=g).float()
    a[:, 5] += (torch.arange(M) % 7).float()
    w[:, 2] += (torch.arange(N) % 3).float()
    ad, wd = dev(a, torch.bfloat16)[:, :Kd], dev(w, torch.bfloat16)          # A with row stride Kd + 8
    ref = a[:, :Kd] @ w.t()
    assert float(ref.abs().max()) < 2 ** 24
    assert torch.equal(K.gemm_nt(ad, wd, out_dtype=torch.float32).cpu(), ref)
    assert torch.equal(K.gemm_nt(ad, wd).float().cpu(), ref.to(torch.bfloat16).float())
    bias = torch.randint(-4, 5, (N,), generator=g).float()
    res = torch.randint(-4, 5, (M, N), generator=g).float()
    got = K.gemm_nt(ad, wd, bias=dev(bias, torch.bfloat16), residual=dev(res, torch.bfloat16), out_dtype=torch.float32)
    assert torch.equal(got.cpu(), ref + bias + res)
    tab = torch.randint(-4, 5, (24, N), generator=g).float()
    got = K.gemm_nt(ad, wd, residual=dev(tab, torch.bfloat16), res_rows=24)
    assert torch.equal(got.float().cpu(), (ref + tab[torch.arange(M) % 24]).to(torch.bfloat16).float())


def test_gemm_nt_ring_fused_epilogues_match_small_kernel(K, monkeypatch):
    """SwiGLU-forward and RoPE epilogues on the ring kernels (M >= 4096) against the same rows computed in < 4096-row pieces,
    which take the 128x128 kernel: bit-identical (same accumulation order over k, same epilogue code)."""
    monkeypatch.setenv("FK_NT_RING_MIN_TILES", "0")            # these grids are under the default threshold of the ring kernels
    M, d, H = 4096 + 520, 128, 256
    x, w13 = dev(rnd(M, d, seed=1), torch.bfloat16), dev(rnd(2 * H, d, seed=2, scale=0.2), torch.bfloat16)
    h13, gq = K.gemm_nt_swiglu(x, w13)
    for lo in range(0, M, 2000):
        h, g2 = K.gemm_nt_swiglu(x[lo:lo + 2000], w13)
        assert torch.equal(h13[lo:lo + 2000], h) and torch.equal(gq[lo:lo + 2000], g2)
    T, D, Hh = 577, 32, 4                                      # M = 8 * 577 = 4616 rows, N = 3 * 128 = 384
    ang = R.rope_angles(D, 640, 10000.0)
    table = dev(torch.stack([torch.cos(ang), torch.sin(ang)], -1).contiguous())
    xq, wq = dev(rnd(8 * T, d, seed=3), torch.bfloat16), dev(rnd(3 * Hh * D, d, seed=4, scale=0.2), torch.bfloat16)
    full = K.gemm_nt_rope(xq, wq, None, table, T, 10, D, 2 * Hh * D)
    for b in range(0, 8, 4):
        part = K.gemm_nt_rope(xq[b * T:(b + 4) * T], wq, None, table, T, 10, D, 2 * Hh * D)
        assert torch.equal(full[b * T:(b + 4) * T], part)


@pytest.mark.parametrize("dtype", DT)
def test_colsum(K, dtype):
    x = rnd(5000, 200, seed=3)
    close(K.colsum(dev(x, dtype)), q(x, dtype).sum(0), torch.float32, atol32=1e-3, rtol32=1e-4)


# ----------------------------------------------------------------------------------------------- attention
def ref_attn(qh, kh, vh, mask):
    """q,k,v [B,N,H,D] fp32 cpu -> o [B,Nq,H,D]"""
    o = R.sdpa(qh.transpose(1, 2), kh.transpose(1, 2), vh.transpose(1, 2), mask)
    return o.transpose(1, 2)


def mask_tensor(kind, c, Nq, Nk, q_off=0, k_off=0):
    if kind == 0:
        return None
    qi = torch.arange(Nq)[:, None] + q_off
    ki = torch.arange(Nk)[None, :] + k_off
    return (ki <= qi) if kind == 1 else (ki // c) <= (qi // c)


ATTN_CASES = [
    # B, H, Nq, Nk, D, kind, c
    (2, 3, 128, 128, 64, 0, 0),
    (1, 2, 200, 200, 64, 2, 8),
    (2, 2, 57, 57, 32, 1, 0),
    (1, 4, 32, 300, 64, 0, 0),
    (2, 4, 128, 128, 16, 2, 16),
    (1, 2, 8, 8, 16, 0, 0),
    (3, 4, 8, 8, 8, 0, 0),
    (2, 4, 40, 128, 8, 2, 16),
    (1, 2, 512, 512, 64, 2, 256),
    (1, 1, 320, 320, 32, 1, 0),
]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", ATTN_CASES)
def test_attention_fwd_bwd(K, dtype, case):
    B, H, Nq, Nk, D, kind, c = case
    qkv_q = rnd(B, Nq, H * D, seed=1)
    qkv_k = rnd(B, Nk, 2 * H * D, seed=2)
    do = rnd(B, Nq, H, D, seed=3)
    # device: q in its own buffer, k/v packed side by side (exercises strides)
    qd = dev(qkv_q, dtype).view(B, Nq, H, D)
    kvd = dev(qkv_k, dtype)
    kd, vd = kvd[..., : H * D].unflatten(-1, (H, D)), kvd[..., H * D:].unflatten(-1, (H, D))
    m = K.Mask(kind, c)
    o, lse = K.attn_fwd(qd, kd, vd, m)
    qr = q(qkv_q, dtype).view(B, Nq, H, D).requires_grad_(True)
    kr = q(qkv_k[..., : H * D], dtype).reshape(B, Nk, H, D).requires_grad_(True)
    vr = q(qkv_k[..., H * D:], dtype).reshape(B, Nk, H, D).requires_grad_(True)
    mt = mask_tensor(kind, c, Nq, Nk)
    oref = ref_attn(qr, kr, vr, mt)
    close(o, oref, dtype, atol32=2e-5, atol16=2e-2)
    s = (qr.transpose(1, 2) @ kr.transpose(1, 2).transpose(-1, -2)) / math.sqrt(D)
    if mt is not None:
        s = s.masked_fill(~mt, float("-inf"))
    close(lse, torch.logsumexp(s, -1), torch.float32, atol32=1e-4 if dtype == torch.float32 else 3e-2, rtol32=1e-4)
    oref.backward(q(do, dtype))
    dod = dev(do, dtype)
    dq = torch.empty_like(qd)
    dkv = torch.empty_like(kvd)
    dk, dv = dkv[..., : H * D].unflatten(-1, (H, D)), dkv[..., H * D:].unflatten(-1, (H, D))
    K.attn_bwd(qd, kd, vd, o, dod, lse, dq, dk, dv, m)
    close(dq, qr.grad, dtype, atol32=5e-5, atol16=4e-2)
    close(dk, kr.grad, dtype, atol32=5e-5, atol16=4e-2)
    close(dv, vr.grad, dtype, atol32=5e-5, atol16=4e-2)


PS_CASES = [
    # B, H, Nq, Nk, kind, c, spike      (bf16, D = 64: the FK_ATTN_Q_PRESCALED kernels)
    (2, 3, 256, 256, 0, 0, 0.0),
    (1, 2, 200, 200, 2, 8, 0.0),          # ragged, boundary sub-tiles everywhere
    (2, 2, 333, 333, 1, 0, 0.0),          # causal, ragged
    (1, 4, 32, 300, 0, 0, 0.0),           # cross-shaped (Nq << Nk)
    (1, 2, 512, 512, 2, 256, 0.0),        # the benchmark's block size: no boundary tiles at all
    (1, 2, 640, 640, 2, 256, 40.0),       # a late key that outgrows reference 0 by far: lean loop -> fallback loop
    (1, 2, 384, 384, 2, 128, -3.0e3),     # every score hugely negative: the window test keeps the classic loop
    # shapes the generated dK/dV stream takes (all tiles fully visible and aligned): 1 / 2 / 7 query tiles per key block (ring of 3)
    (1, 2, 64, 128, 0, 0, 0.0),
    (2, 1, 128, 256, 0, 0, 0.0),
    (1, 3, 448, 128, 0, 0, 0.0),
    (1, 2, 768, 768, 2, 128, 0.0),        # block-causal, 128-key blocks: 12, 10, .. 2 tiles
    # and the generated dQ stream (128-query workgroups, 64-key tiles): 1 / 2 / 5 key tiles
    (1, 2, 128, 64, 0, 0, 0.0),
    (2, 1, 256, 128, 0, 0, 0.0),
    (1, 2, 128, 320, 0, 0, 0.0),
    # the generated forward streams: warm-up tile + first / steady / drain steps at every ring phase (2 .. 9 tiles), and a late spike that
    # must send the whole workgroup back through the classic loop
    (1, 1, 128, 128, 0, 0, 0.0),
    (1, 2, 128, 192, 0, 0, 0.0),
    (1, 2, 256, 384, 0, 0, 0.0),
    (1, 1, 128, 448, 0, 0, 0.0),
    (1, 2, 128, 576, 0, 0, 0.0),
    (1, 2, 1024, 1024, 2, 256, 0.0),
    (1, 2, 768, 768, 0, 0, 40.0),
]


@pytest.mark.parametrize("case", PS_CASES)
def test_attention_prescaled_q(K, case):
    """FK_ATTN_Q_PRESCALED (Q' = scale * log2(e) * q stored in bf16 by the projection epilogue): forward, LSE and all three
    gradients (w.r.t. the UNSCALED q) against the fp32 oracle evaluated at q = Q' / (scale * log2 e)."""
    B, H, Nq, Nk, kind, c, spike = case
    D, dtype = 64, torch.bfloat16
    cq = (1.0 / math.sqrt(D)) * 1.4426950408889634
    qp = q(rnd(B, Nq, H * D, seed=1) * cq * 2.0, dtype)                 # Q' as the device sees it
    kv = rnd(B, Nk, 2 * H * D, seed=2)
    if spike > 0:
        # key 600 of head 0 strongly aligned with query 610: its score outgrows everything seen before by far
        kv[0, 600, :D] = spike * (qp[0, 610, :D] / cq) / 8.0
    if spike < 0:
        # head 0: a component shared by all keys puts every score near spike * 8 / sqrt(D) (far below the window of reference 0)
        kv[..., 0] = 8.0
        qp[..., 0] = q(torch.tensor(spike * cq), dtype)
    kv = q(kv, dtype)
    do = rnd(B, Nq, H, D, seed=3)
    qd = dev(qp, dtype).view(B, Nq, H, D)
    kvd = dev(kv, dtype)
    kd, vd = kvd[..., : H * D].unflatten(-1, (H, D)), kvd[..., H * D:].unflatten(-1, (H, D))
    m = K.Mask(kind, c)
    o, lse = K.attn_fwd(qd, kd, vd, m, q_prescaled=True)
    qr = (qp.view(B, Nq, H, D) / cq).requires_grad_(True)
    kr = q(kv[..., : H * D], dtype).reshape(B, Nk, H, D).requires_grad_(True)
    vr = q(kv[..., H * D:], dtype).reshape(B, Nk, H, D).requires_grad_(True)
    mt = mask_tensor(kind, c, Nq, Nk)
    oref = ref_attn(qr, kr, vr, mt)
    close(o, oref, dtype, atol16=2e-2)
    sfull = (qr.transpose(1, 2) @ kr.transpose(1, 2).transpose(-1, -2)) / math.sqrt(D)
    if mt is not None:
        sfull = sfull.masked_fill(~mt, float("-inf"))
    lref = torch.logsumexp(sfull, -1)
    torch.testing.assert_close(lse.cpu(), lref.detach(), atol=3e-2, rtol=2e-3)
    oref.backward(q(do, dtype))
    dq = torch.empty_like(qd)
    dkv = torch.empty_like(kvd)
    dk, dv = dkv[..., : H * D].unflatten(-1, (H, D)), dkv[..., H * D:].unflatten(-1, (H, D))
    K.attn_bwd(qd, kd, vd, o, dev(do, dtype), lse, dq, dk, dv, m, q_prescaled=True)
    close(dq, qr.grad, dtype, atol16=4e-2 * max(1.0, float(qr.grad.abs().max()) / 4))       # (the spike case has |dq|, |dk| >> 1)
    close(dk, kr.grad, dtype, atol16=4e-2 * max(1.0, float(kr.grad.abs().max()) / 4))
    close(dv, vr.grad, dtype, atol16=4e-2)
    # the two forms of the same problem agree with each other at bf16 rounding level (not the spiked ones: Q' and q round differently,
    # and a score of several hundred turns that 2^-8 relative difference into a different softmax)
    if spike == 0:
        qun = dev(qr.detach().reshape(B, Nq, H, D), dtype)
        o2, lse2 = K.attn_fwd(qun, kd, vd, m)
        assert float((o2.float() - o.float()).abs().max()) < 6e-2


def test_attention_prescaled_q_masks_from_tables(K):
    """prefix (MAE sub-mask) and key-padding tables on the pre-scaled path, incl. fully masked query rows (-> 0, LSE = +inf)."""
    B, H, N, n, D, Cb = 2, 2, 700, 260, 64, 16
    dtype = torch.bfloat16
    cq = (1.0 / math.sqrt(D)) * 1.4426950408889634
    ids = torch.stack([torch.randperm(N, generator=torch.Generator().manual_seed(10 + i))[:n].sort()[0] for i in range(B)])
    dense = ((ids[:, None, :] // Cb) <= (ids[:, :, None] // Cb))[:, None]            # [B, 1, n, n]
    valid = torch.ones(B, n, dtype=torch.bool)
    valid[0, 200:] = False
    valid[1, 37:41] = False
    pad = (valid[:, None, :, None] & valid[:, None, None, :])
    for mask, mt in ((K.Mask.from_token_ids(dev(ids), dev(ids), Cb), dense),
                     (K.Mask(4, 0, 0, 0, dev(valid.to(torch.int32)), dev(valid.to(torch.int32))), pad)):
        qp = q(rnd(B, n, H, D, seed=1) * cq * 2.0, dtype)
        kv, vv, do = (rnd(B, n, H, D, seed=s_) for s_ in (2, 3, 4))
        qd, kd, vd = dev(qp, dtype), dev(kv, dtype), dev(vv, dtype)
        o, lse = K.attn_fwd(qd, kd, vd, mask, q_prescaled=True)
        qr = (qp / cq).requires_grad_(True)
        kr, vr = (q(t_, dtype).requires_grad_(True) for t_ in (kv, vv))
        # fully masked rows: 0 (torch >= 2.1 semantics)
        oref = R.sdpa_zero_fully_masked(qr.transpose(1, 2), kr.transpose(1, 2), vr.transpose(1, 2), mt.expand(B, 1, n, n)).transpose(1, 2)
        close(o, oref, dtype, atol16=2e-2)
        rows_dead = ~mt.expand(B, 1, n, n).any(-1)[:, 0]                        # [B, n]
        assert bool(torch.isinf(lse.cpu()[rows_dead[:, None, :].expand(B, H, n)]).all())
        (oref * q(do, dtype)).sum().backward()
        dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
        K.attn_bwd(qd, kd, vd, o, dev(do, dtype), lse, dq, dk, dv, mask, q_prescaled=True)
        close(dq, qr.grad, dtype, atol16=4e-2)
        close(dk, kr.grad, dtype, atol16=4e-2)
        close(dv, vr.grad, dtype, atol16=4e-2)


def test_attention_keypad_mask_free_tiles(K):
    """A key-padding mask whose rows are (mostly) valid: the pre-scaled kernels send the tiles in front of a sample's first padded key
    down the mask-free path (keypad_valid_prefix, attention.hip).  N > 1024 takes the scan's second round; sample 0 is all valid and must
    give the bits of a launch without a mask, sample 1 is padded at the tail, sample 2 has holes (one in the first tile: nothing free)."""
    B, H, N, D = 3, 2, 1100, 64
    dtype = torch.bfloat16
    cq = (1.0 / math.sqrt(D)) * 1.4426950408889634
    valid = torch.ones(B, N, dtype=torch.bool)
    valid[1, 1050:] = False
    valid[2, 1030:1033] = False
    valid[2, 20] = False
    pad = (valid[:, None, :, None] & valid[:, None, None, :])
    mask = K.Mask.from_padding(dev(valid), dev(valid))
    qp = q(rnd(B, N, H, D, seed=1) * cq * 2.0, dtype)
    kv, vv, do = (rnd(B, N, H, D, seed=s_) for s_ in (2, 3, 4))
    qd, kd, vd, dod = dev(qp, dtype), dev(kv, dtype), dev(vv, dtype), dev(do, dtype)
    o, lse = K.attn_fwd(qd, kd, vd, mask, q_prescaled=True)
    qr = (qp / cq).requires_grad_(True)
    kr, vr = (q(t_, dtype).requires_grad_(True) for t_ in (kv, vv))
    oref = R.sdpa_zero_fully_masked(qr.transpose(1, 2), kr.transpose(1, 2), vr.transpose(1, 2), pad.expand(B, 1, N, N)).transpose(1, 2)
    close(o, oref, dtype, atol16=2e-2)
    assert bool(torch.isinf(lse.cpu()[(~valid)[:, None, :].expand(B, H, N)]).all())
    (oref * q(do, dtype)).sum().backward()
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    K.attn_bwd(qd, kd, vd, o, dod, lse, dq, dk, dv, mask, q_prescaled=True)
    close(dq, qr.grad, dtype, atol16=4e-2)
    close(dk, kr.grad, dtype, atol16=4e-2)
    close(dv, vr.grad, dtype, atol16=4e-2)
    # sample 0 alone without a mask: the same bits
    none = K.Mask(0, 0)
    q0, k0, v0, g0 = (t_[:1].contiguous() for t_ in (qd, kd, vd, dod))
    o0, lse0 = K.attn_fwd(q0, k0, v0, none, q_prescaled=True)
    assert torch.equal(o0, o[:1]) and torch.equal(lse0, lse[:1])
    dq0, dk0, dv0 = torch.empty_like(q0), torch.empty_like(k0), torch.empty_like(v0)
    K.attn_bwd(q0, k0, v0, o0, g0, lse0, dq0, dk0, dv0, none, q_prescaled=True)
    assert torch.equal(dq0, dq[:1]) and torch.equal(dk0, dk[:1]) and torch.equal(dv0, dv[:1])


def test_attention_mask_offsets_and_spike(K):
    # sliced mask (t_q < t_k, models/brainformer.py:160-162) and a forced running-max jump (online softmax rescale)
    B, H, Nq, Nk, D = 1, 2, 40, 200, 32
    qv, kv, vv = rnd(B, Nq, H, D, seed=1), rnd(B, Nk, H, D, seed=2), rnd(B, Nk, H, D, seed=3)
    kv[0, 150, 0] = 6.0 * qv[0, 7, 0]     # late key strongly aligned with query 7 -> max jumps at the 3rd tile
    m = K.Mask(2, 8).sliced(256, 256, Nq, Nk)
    o, lse = K.attn_fwd(dev(qv), dev(kv), dev(vv), m)
    mt = mask_tensor(2, 8, 256, 256)[-Nq:, -Nk:]
    close(o, ref_attn(qv, kv, vv, mt), torch.float32)


# ----------------------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("kind", [0, 1])
@pytest.mark.parametrize("shape", [(1000, 384), (37, 64), (5, 2048)])
def test_norm(K, dtype, kind, shape):
    rows, dim = shape
    x, g, b = rnd(rows, dim, seed=1) * 2 + 0.5, 1 + 0.1 * rnd(dim, seed=2), 0.1 * rnd(dim, seed=3)
    dy, dres = rnd(rows, dim, seed=4), rnd(rows, dim, seed=5)
    xr, gr, br = q(x, dtype).requires_grad_(True), g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yref = R.layer_norm(xr, gr, br) if kind == 0 else R.rms_norm(xr, gr)
    y, mean, rstd = K.norm_fwd(dev(x, dtype), dev(g), dev(b) if kind == 0 else None, 1e-5 if kind == 0 else 1e-6, kind)
    close(y, yref, dtype, atol32=1e-5)
    yref.backward(q(dy, dtype))
    dx, dg, db = K.norm_bwd(dev(dy, dtype), dev(x, dtype), dev(g), mean, rstd, dres=dev(dres, dtype), kind=kind, want_beta=(kind == 0))
    close(dx, xr.grad + q(dres, dtype), dtype, atol32=2e-5)
    close(dg, gr.grad, torch.float32, atol32=2e-3 if dtype == torch.float32 else 5e-2, rtol32=1e-3)
    if kind == 0:
        close(db, br.grad, torch.float32, atol32=2e-3 if dtype == torch.float32 else 5e-2, rtol32=1e-3)


# ----------------------------------------------------------------------------------------------- pointwise
@pytest.mark.parametrize("dtype", DT)
def test_rope(K, dtype):
    B, T, H, D = 2, 50, 3, 16
    ang = R.rope_angles(D, 64, 10000.0)
    table = torch.stack([torch.cos(ang), torch.sin(ang)], -1).contiguous()
    x = rnd(B, T, 3 * H * D, seed=1)
    xd = dev(x, dtype)
    K.rope_(xd, 2 * H, D, dev(table), pos_off=64 - T)       # rotate q and k (first 2H heads), leave v
    want = q(x, dtype).clone()
    want[..., : 2 * H * D] = R.apply_rope(want[..., : 2 * H * D].reshape(B, T, 2 * H, D), ang).reshape(B, T, -1)
    close(xd, want, dtype, atol32=1e-6, atol16=2e-2)
    if dtype == torch.float32:
        K.rope_(xd, 2 * H, D, dev(table), pos_off=64 - T, conj=True)
        close(xd, x, dtype, atol32=1e-5)
    ang3 = torch.stack([ang[3:3 + T], ang[10:10 + T]])
    t3 = torch.stack([torch.cos(ang3), torch.sin(ang3)], -1).contiguous()
    yd = dev(x, dtype)
    K.rope_(yd, 2 * H, D, dev(t3), pos_off=0)
    want3 = q(x, dtype).clone()
    want3[..., : 2 * H * D] = R.apply_rope(want3[..., : 2 * H * D].reshape(B, T, 2 * H, D), ang3).reshape(B, T, -1)
    close(yd, want3, dtype, atol32=1e-6, atol16=2e-2)


@pytest.mark.parametrize("dtype", DT)
def test_patchify(K, dtype):
    x = rnd(3, 100, 48, seed=1)
    tok = K.patchify(dev(x), 25, 32, dtype)
    want = torch.zeros(3 * 4 * 48, 32)
    want[:, :25] = R.to_patches(x, 25).reshape(-1, 25)
    close(tok, q(want, dtype), dtype, atol32=0, rtol32=0, atol16=0, rtol16=0)


@pytest.mark.parametrize("dtype", DT)
def test_swiglu_gelu(K, dtype):
    rows, H = 333, 96
    h13, dg = rnd(rows, 2 * H, seed=1), rnd(rows, H, seed=2)
    hr = q(h13, dtype).requires_grad_(True)
    gref = R.silu(hr[:, :H]) * hr[:, H:]
    close(K.swiglu_fwd(dev(h13, dtype)), gref, dtype, atol32=1e-6)
    gref.backward(q(dg, dtype))
    close(K.swiglu_bwd(dev(h13, dtype), dev(dg, dtype)), hr.grad, dtype, atol32=2e-6)
    x = rnd(rows, H, seed=3) * 2
    xr = q(x, dtype).requires_grad_(True)
    yref = R.gelu_erf(xr)
    close(K.gelu_fwd(dev(x, dtype)), yref, dtype, atol32=1e-6)
    yref.backward(q(dg, dtype))
    close(K.gelu_bwd(dev(x, dtype), dev(dg, dtype)), xr.grad, dtype, atol32=2e-6)


@pytest.mark.parametrize("dtype", DT)
def test_cast_pack_add_copy(K, dtype):
    w = rnd(50, 25, seed=1)
    d = torch.zeros(50, 32, device="cuda", dtype=dtype)
    K.cast_pack(dev(w), d)
    assert torch.equal(d[:, :25].float().cpu(), q(w, dtype)) and float(d[:, 25:].abs().max()) == 0
    dT = torch.zeros(25, 56, device="cuda", dtype=dtype)
    K.cast_pack(dev(w), dT, transpose=True)
    assert torch.equal(dT[:, :50].float().cpu(), q(w, dtype).t())
    a, b = rnd(1000, seed=2), rnd(1000, seed=3)
    close(K.add(dev(a, dtype), dev(b, dtype)), q(a, dtype) + q(b, dtype), dtype, atol32=0, atol16=2e-2)
    assert torch.equal(K.cast(dev(a), dtype).float().cpu(), q(a, dtype))
    src = dev(rnd(6, 40, seed=4), dtype)
    dst = torch.zeros(6, 64, device="cuda", dtype=dtype)
    K.copy2d(src[:, 8:24], dst[:, 32:48])
    assert torch.equal(dst[:, 32:48], src[:, 8:24])


# ----------------------------------------------------------------------------------------------- losses / optimizer
@pytest.mark.parametrize("dtype", DT)
def test_l1_mse(K, dtype):
    p, t = rnd(32, 32, 128, seed=1), rnd(32, 32, 128, seed=2)
    go = torch.tensor([0.5], device="cuda")
    for sq in (False, True):
        pr = q(p, dtype).requires_grad_(True)
        d = pr - q(t, dtype)
        ref = (d * d).mean() if sq else d.abs().mean()
        close(K.l1_loss_fwd(dev(p, dtype), dev(t, dtype), sq)[0], ref, torch.float32, atol32=1e-5)
        (0.5 * ref).backward()
        close(K.l1_loss_bwd(dev(p, dtype), dev(t, dtype), go, sq), pr.grad, dtype, atol32=1e-9, rtol32=1e-5, atol16=1e-7, rtol16=1e-2)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("V", [300, 50257])
def test_cross_entropy(K, dtype, V):
    rows = 21
    lg = rnd(rows, V, seed=1) * 2
    tg = torch.randint(0, V, (rows,), generator=torch.Generator().manual_seed(2))
    tg[3] = -100
    tg[20] = -100
    lr_ = q(lg, dtype).requires_grad_(True)
    ref = R.cross_entropy(lr_, tg)
    loss2, lse = K.ce_loss_fwd(dev(lg, dtype), dev(tg))
    close(loss2[0], ref, torch.float32, atol32=2e-5)
    assert float(loss2[1]) == rows - 2
    ref.backward()
    go = torch.ones(1, device="cuda")
    dl = K.ce_loss_bwd(dev(lg, dtype), dev(tg), lse, loss2, go, torch.empty(rows, V, device="cuda", dtype=dtype))
    close(dl, lr_.grad, dtype, atol32=1e-7, rtol32=1e-4, atol16=1e-3, rtol16=2e-2)


def test_adamw_matches_reference_trajectory(K, golden):
    z = golden("ops")
    p = dev(torch.from_numpy(z["adamw_p0"]).clone())
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for i in range(3):
        g = dev(torch.from_numpy(z["adamw_g"][i]).clone())
        K.adamw_step_(p, g, m, v, i + 1, float(z["adamw_lrs"][i]), weight_decay=1e-5, clip=1.0, zero_grad=(i == 1))
        np.testing.assert_allclose(p.cpu().numpy(), z["adamw_traj"][i], rtol=2e-6, atol=2e-7)
        assert (float(g.abs().max()) == 0.0) == (i == 1)
    # grad_scale (DP mean) and no-clip path against the oracle formula
    p0, g0 = rnd(1001, seed=1), rnd(1001, seed=2) * 3
    pr, mr, vr = RT.adamw_step(p0, g0 * 0.25, torch.zeros(1001), torch.zeros(1001), 1, 1e-3, 1e-5)
    pd, md, vd = dev(p0.clone()), torch.zeros(1001, device="cuda"), torch.zeros(1001, device="cuda")
    K.adamw_step_(pd, dev(g0.clone()), md, vd, 1, 1e-3, weight_decay=1e-5, clip=0.0, grad_scale=0.25)
    torch.testing.assert_close(pd.cpu(), pr, rtol=2e-6, atol=2e-7)
    torch.testing.assert_close(vd.cpu(), vr, rtol=2e-6, atol=1e-9)


@pytest.mark.parametrize("dtype", DT)
def test_gpt_embed(K, dtype):
    B, tc, tw, d, V = 3, 5, 9, 64, 211
    idx = torch.randint(0, V, (B, tw), generator=torch.Generator().manual_seed(1))
    prefix, wte, wpe = rnd(B, tc, d, seed=2), rnd(V, d, seed=3), rnd(32, d, seed=4)
    out = K.gpt_embed_fwd(dev(idx), dev(prefix, dtype), dev(wte), dev(wpe), dtype)
    want = torch.cat([q(prefix, dtype), wte[idx]], 1) + wpe[: tc + tw]
    close(out, want, dtype, atol32=1e-6, atol16=3e-2)
    out0 = K.gpt_embed_fwd(dev(idx), None, dev(wte), dev(wpe), dtype)
    close(out0, wte[idx] + wpe[:tw], dtype, atol32=1e-6, atol16=3e-2)
    dout = rnd(B, tc + tw, d, seed=5)
    dw = torch.zeros(V, d, device="cuda")
    K.gpt_embed_bwd_wte(dev(idx), dev(dout, dtype), dw, tc)
    ref = torch.zeros(V, d)
    ref.index_put_((idx.reshape(-1),), q(dout, dtype)[:, tc:].reshape(-1, d), accumulate=True)
    close(dw, ref, torch.float32, atol32=1e-5)


def test_errors_are_loud(K):
    from frankenstein_amd._lib import FrankenHipError
    a = torch.zeros(8, 12, device="cuda")   # K=12 fp32 ok (multiple of 4)
    w = torch.zeros(8, 12, device="cuda")
    K.gemm_nt(a, w)
    with pytest.raises(FrankenHipError, match="multiples"):
        K.gemm_nt(torch.zeros(8, 12, device="cuda", dtype=torch.bfloat16), torch.zeros(8, 12, device="cuda", dtype=torch.bfloat16))
    with pytest.raises(FrankenHipError, match="head_dim"):
        K.attn_fwd(torch.zeros(1, 8, 1, 24, device="cuda"), torch.zeros(1, 8, 1, 24, device="cuda"), torch.zeros(1, 8, 1, 24, device="cuda"))


@pytest.mark.parametrize("dtype", DT)
def test_fused_swiglu_gemms(K, dtype):
    """up-projection + SwiGLU and down-projection dgrad + SwiGLU backward fused in the GEMM epilogues (interleaved layout)."""
    from frankenstein_amd import engine as E
    E.set_compute_dtype(dtype)
    try:
        M, d, H = 300, 64, 96
        x, w1, w3, w2 = rnd(M, d, seed=1), rnd(H, d, seed=2, scale=0.2), rnd(H, d, seed=3, scale=0.2), rnd(d, H, seed=4, scale=0.2)
        dy = rnd(M, d, seed=5)
        w1d, w3d, w2d = dev(w1), dev(w3), dev(w2)
        h13, g = K.gemm_nt_swiglu(dev(x, dtype), E.shadow_swiglu(w1d, w3d))
        xr = q(x, dtype)
        h1, h3 = xr @ q(w1, dtype).t(), xr @ q(w3, dtype).t()
        close(g, R.silu(h1) * h3, dtype, atol32=2e-5)
        il = h13.float().cpu().view(M, H // 4, 2, 4)
        close(il[:, :, 0].reshape(M, H), h1, dtype, atol32=2e-5)
        close(il[:, :, 1].reshape(M, H), h3, dtype, atol32=2e-5)
        # backward: dh13 from dy, w2^T and the saved (interleaved) h13
        dh13 = K.gemm_nt_dswiglu(dev(dy, dtype), E.shadow([w2d], transpose=True), h13)
        h1r = il[:, :, 0].reshape(M, H).clone().requires_grad_(True)
        h3r = il[:, :, 1].reshape(M, H).clone().requires_grad_(True)
        (R.silu(h1r) * h3r).backward(q(dy, dtype) @ q(w2, dtype))
        dil = dh13.float().cpu().view(M, H // 4, 2, 4)
        close(dil[:, :, 0].reshape(M, H), h1r.grad, dtype, atol32=5e-5, atol16=4e-2)
        close(dil[:, :, 1].reshape(M, H), h3r.grad, dtype, atol32=5e-5, atol16=4e-2)
        a, b = E._deinterleave_rows(dev(torch.arange(2 * H * 8, dtype=torch.float32).view(2 * H, 8)), H)
        ref = torch.arange(2 * H * 8, dtype=torch.float32).view(H // 4, 2, 4, 8)
        assert torch.equal(a.cpu(), ref[:, 0].reshape(H, 8)) and torch.equal(b.cpu(), ref[:, 1].reshape(H, 8))
    finally:
        E.set_compute_dtype("bf16")


@pytest.mark.parametrize("dtype", DT)
def test_fused_rope_projection_and_backward(K, dtype):
    """RoPE fused into the q|k|v projection epilogue, and its inverse fused into the attention-backward dQ/dK stores."""
    B, T, H, D, d = 2, 40, 3, 16, 64
    ang = R.rope_angles(D, 64, 10000.0)
    table = torch.stack([torch.cos(ang), torch.sin(ang)], -1).contiguous()
    x, w, bias = rnd(B * T, d, seed=1), rnd(3 * H * D, d, seed=2, scale=0.2), rnd(3 * H * D, seed=3, scale=0.1)
    got = K.gemm_nt_rope(dev(x, dtype), dev(w, dtype), dev(bias, dtype), dev(table), T, 64 - T, D, 2 * H * D)
    ref = (q(x, dtype) @ q(w, dtype).t() + q(bias, dtype)).view(B, T, 3 * H * D)
    ref[..., : 2 * H * D] = R.apply_rope(ref[..., : 2 * H * D].reshape(B, T, 2 * H, D), ang).reshape(B, T, -1)
    close(got.view(B, T, -1), ref, dtype, atol32=2e-5, atol16=3e-2)
    # backward: attn_bwd with rope_table == attn_bwd then conj-rope on dq, dk
    qkv = dev(rnd(B, T, 3 * H * D, seed=4), dtype)
    qv, kv, vv = (qkv[..., i * H * D:(i + 1) * H * D].unflatten(-1, (H, D)) for i in range(3))
    o, lse = K.attn_fwd(qv, kv, vv, K.Mask(1))
    do = dev(rnd(B, T, H, D, seed=5), dtype)
    d1, d2 = torch.empty_like(qkv), torch.empty_like(qkv)
    views = lambda t: [t[..., i * H * D:(i + 1) * H * D].unflatten(-1, (H, D)) for i in range(3)]
    K.attn_bwd(qv, kv, vv, o, do, lse, *views(d1), K.Mask(1))
    K.rope_(d1, 2 * H, D, dev(table), 64 - T, conj=True)
    K.attn_bwd(qv, kv, vv, o, do, lse, *views(d2), K.Mask(1), rope_table=dev(table), rope_off=64 - T)
    close(d2, d1.float().cpu(), dtype, atol32=1e-6, atol16=3e-2)


@pytest.mark.parametrize("shape", [(3, 40, 11, 24), (5, 300, 150, 384), (2, 64, 33, 256), (2, 50, 20, 520), (3, 40, 11, 20)])
@pytest.mark.parametrize("dtype", DT)
def test_gather_scatter_rows(K, dtype, shape):
    """fk_gather_rows / fk_scatter_add_rows (MAE token subsets, models/brainformer.py:429-472, models/simple_mae).  Rows of whole 8-element
    chunks take the vector kernel (16 bytes per lane, one index load per row: 24 = three chunks and 21 rows per wave, 384 = one row on 48
    lanes, 256 = two rows per wave, 520 = more chunks than lanes), 20 stays on the element-wise one."""
    B, N, n, W = shape
    src = rnd(B, N, W, seed=1)
    idx = torch.stack([torch.randperm(N, generator=torch.Generator().manual_seed(i))[:n].sort()[0] for i in range(B)])
    g = K.gather_rows(dev(src, dtype), dev(idx))
    want = q(src, dtype)[torch.arange(B)[:, None], idx]
    assert torch.equal(g.float().cpu(), want)
    d = torch.zeros(B, N, W, device="cuda", dtype=dtype)
    K.scatter_rows_(d, dev(idx), g)
    ref = torch.zeros(B, N, W).index_put((torch.arange(B)[:, None], idx), want)
    assert torch.equal(d.float().cpu(), ref)
    table = rnd(7, W, seed=2)                                   # shared fp32 table, index modulo, cast on the fly
    t = K.gather_rows(dev(table), dev(idx), out_dtype=dtype, idx_mod=7)
    assert torch.equal(t.float().cpu(), q(table[idx % 7], dtype))
    acc = torch.zeros(7, W, device="cuda")
    K.scatter_add_rows_(acc, dev(idx), g, idx_mod=7)
    ref = torch.zeros(7, W).index_put(((idx % 7).reshape(-1),), want.reshape(-1, W), accumulate=True)
    close(acc, ref, torch.float32, atol32=1e-5)


@pytest.mark.parametrize("dtype", DT)
def test_attention_prefix_mask_from_token_ids(K, dtype):
    """MAE sub-mask: mask[i, j] = (ids[j] // C) <= (ids[i] // C) at gathered (sorted) token ids, models/brainformer.py:392-413."""
    B, H, N, n, D, Cb = 2, 2, 512, 200, 32, 16
    ids = torch.stack([torch.randperm(N, generator=torch.Generator().manual_seed(10 + i))[:n].sort()[0] for i in range(B)])
    m = K.Mask.from_token_ids(dev(ids), dev(ids), Cb)
    dense = (ids[:, None, :] // Cb) <= (ids[:, :, None] // Cb)                 # [B, n, n]
    assert torch.equal(m.limits.cpu().long(), dense.sum(-1))
    qv, kv, vv, do = (rnd(B, n, H, D, seed=s) for s in (1, 2, 3, 4))
    qd, kd, vd = dev(qv, dtype), dev(kv, dtype), dev(vv, dtype)
    o, lse = K.attn_fwd(qd, kd, vd, m)
    qr, kr, vr = (q(t_, dtype).requires_grad_(True) for t_ in (qv, kv, vv))
    oref = ref_attn(qr, kr, vr, dense[:, None])
    close(o, oref, dtype, atol32=2e-5, atol16=2e-2)
    oref.backward(q(do, dtype))
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    K.attn_bwd(qd, kd, vd, o, dev(do, dtype), lse, dq, dk, dv, m)
    close(dq, qr.grad, dtype, atol32=5e-5, atol16=4e-2)
    close(dk, kr.grad, dtype, atol32=5e-5, atol16=4e-2)
    close(dv, vr.grad, dtype, atol32=5e-5, atol16=4e-2)


DENSE_CASES = [
    # B, H, Nq, Nk, D, one mask per sample, one mask per head
    (2, 2, 200, 200, 64, True, False),
    (1, 3, 57, 300, 32, False, False),
    (2, 2, 128, 256, 16, True, False),
    (3, 2, 8, 8, 8, False, False),
    (2, 1, 320, 192, 64, False, False),
    (2, 3, 130, 200, 32, True, True),
    (2, 2, 64, 96, 64, False, True),
]


@pytest.mark.parametrize("case", [(12, 4096, 6, True), (49, 1000, 3, False), (13, 4096, 2, True)])
def test_qkv_projection_with_rope_token_on_the_lane_equals_the_tiled_kernel_bit_for_bit(K, case):
    """fk_gemm_nt_rope routes wide bf16 projections at d = 384, head_dim 64 (no bias; row counts whose 256-token workgroups fill at least
    70 % of their rounds of 256: from 45 825 rows on) to the token-on-the-lane kernel (qkv_rope_fused_kernel: rotation in registers, two
    waves per SIMD).  Its output must be the bits of the tiled kernel, which the same entry point runs below the threshold: the reference is the same call on two ranges of whole samples — with and without the
    pre-scaled query table, with a position offset, heads != 6 — and the rotation against the fp32 formula."""
    B, T, H, prescaled = case
    d, D = 384, 64
    M, N = B * T, 3 * H * D
    g = torch.Generator().manual_seed(B * T + H)
    x = (torch.randn(M, d, generator=g) * 0.7).bfloat16().cuda()
    w = (torch.randn(N, d, generator=g) / math.sqrt(d)).bfloat16().cuda()
    Tc = T + 5
    ang = torch.rand(Tc, D // 2, generator=g) * 6.28
    both = torch.stack([torch.stack([ang.cos(), ang.sin()], -1), torch.stack([ang.cos(), ang.sin()], -1) * 0.1803]).contiguous().cuda()
    tab, qtab = both[0], both[1]
    kw = dict(q_cols=H * D, q_table=qtab) if prescaled else {}
    out = K.gemm_nt_rope(x, w, None, tab, T, 5, D, 2 * H * D, **kw)
    cut = (B // 2) * T
    assert M >= 45825 and cut < 45825 and M - cut < 45825
    ref = torch.cat([K.gemm_nt_rope(x[a:b], w, None, tab, T, 5, D, 2 * H * D, **kw) for a, b in ((0, cut), (cut, M))])
    assert torch.equal(out, ref), float((out.float() - ref.float()).abs().max())
    y = (x.float().cpu() @ w.float().cpu().t()).view(B, T, 3 * H, D // 2, 2)
    cs = torch.stack([ang.cos(), ang.sin()], -1)[5:5 + T][None, :, None]
    rot = torch.stack([y[..., 0] * cs[..., 0] - y[..., 1] * cs[..., 1], y[..., 0] * cs[..., 1] + y[..., 1] * cs[..., 0]], -1)
    want = torch.cat([rot[:, :, :2 * H], y[:, :, 2 * H:]], 2)
    if prescaled:
        want[:, :, :H] *= 0.1803
    torch.testing.assert_close(out.float().cpu().view(B, T, 3 * H, D // 2, 2), want, atol=4e-2, rtol=2e-2)


@pytest.mark.parametrize("case", [(49152, 64), (50_000 + 77, 1536), (49_000, 96)])
def test_swiglu_up_projection_token_on_the_lane_equals_the_tiled_kernel_bit_for_bit(K, case):
    """fk_gemm_nt_swiglu routes wide bf16 MLPs at d = 384 (from 45 825 rows on, see mu_grid_fills in csrc/mlp_fused.hip) to the token-on-the-lane kernel (mlp_up_fused_kernel: SwiGLU in registers,
    two waves per SIMD): H13 and G must be the bits of the tiled kernel, which the same entry point runs below the threshold — rows are
    independent, so the reference is the same call on two row ranges under it — and both against the fp32 formula."""
    M, H = case
    d = 384
    g = torch.Generator().manual_seed(M + 3 * H)
    x = (torch.randn(M, d, generator=g) * 0.7).bfloat16().cuda()
    w13 = (torch.randn(2 * H, d, generator=g) / math.sqrt(d)).bfloat16().cuda()
    h13, gg = K.gemm_nt_swiglu(x, w13)
    cut = M // 2
    assert M >= 45825 and M - cut < 45825
    ref = [K.gemm_nt_swiglu(x[a:b], w13) for a, b in ((0, cut), (cut, M))]
    h13_ref, g_ref = torch.cat([r[0] for r in ref]), torch.cat([r[1] for r in ref])
    assert torch.equal(h13, h13_ref), float((h13.float() - h13_ref.float()).abs().max())
    assert torch.equal(gg, g_ref), float((gg.float() - g_ref.float()).abs().max())
    want = x.float().cpu() @ w13.float().cpu().t()
    torch.testing.assert_close(h13.float().cpu(), want, atol=3e-2, rtol=2e-2)
    hh = h13.float().cpu().view(M, H // 4, 2, 4)
    torch.testing.assert_close(gg.float().cpu(), (torch.nn.functional.silu(hh[:, :, 0]) * hh[:, :, 1]).reshape(M, H), atol=3e-2, rtol=2e-2)


@pytest.mark.parametrize("case", [(256, 64), (1000, 96), (4096 + 37, 1536), (128, 32), (4096, 1536), (1024, 96), (384, 160), (33 * 1024, 1536)])
def test_mlp_backward_fused_equals_the_two_gemm_kernels_bit_for_bit(K, case):
    """fk_mlp_bwd_fused (the SwiGLU MLP's data-gradient chain in one launch, token on the lane, dh13 handed from the first product to the
    second in registers) against the two launches it replaces — fk_gemm_nt_dswiglu, then fk_gemm_nt on the dh13 it wrote: same products,
    operand slots and summation order, so dh13 AND dx must be identical bits (ragged row counts: the last workgroup and wave are partial),
    and both against the fp32 oracle formula.  Row counts that are whole 128-token tiles take mlp_bwd_fused_asm_kernel (the second
    product's step as a generated instruction stream with the next chunk's SwiGLU derivative in its gaps): 1, 2, 3, 5 and 48 chunks, more
    workgroups than CUs."""
    M, H = case
    d = 384
    g = torch.Generator().manual_seed(M + H)
    dy = (torch.randn(M, d, generator=g) * 0.5).bfloat16().cuda()
    h13 = (torch.randn(M, 2 * H, generator=g)).bfloat16().cuda()
    w2t = (torch.randn(H, d, generator=g) / math.sqrt(d)).bfloat16().cuda()
    w13t = (torch.randn(d, 2 * H, generator=g) / math.sqrt(H)).bfloat16().cuda()
    dh_ref = K.gemm_nt_dswiglu(dy, w2t, h13)
    dx_ref = K.gemm_nt(dh_ref, w13t)
    dh, dx = K.mlp_bwd_fused(dy, w2t, h13, w13t)
    assert torch.equal(dh, dh_ref), float((dh.float() - dh_ref.float()).abs().max())
    assert torch.equal(dx, dx_ref), float((dx.float() - dx_ref.float()).abs().max())
    # the oracle's formula in fp32 (interleaved hidden layout: per 4 units 4 x h1, then 4 x h3)
    dg = dy.float().cpu() @ w2t.float().cpu().t()
    hh = h13.float().cpu().view(M, H // 4, 2, 4)
    a1, a3 = hh[:, :, 0].reshape(M, H), hh[:, :, 1].reshape(M, H)
    sg = torch.sigmoid(a1)
    d1, d3 = dg * sg * a3 * (1 + a1 * (1 - sg)), dg * sg * a1
    want = torch.stack([d1.view(M, H // 4, 4), d3.view(M, H // 4, 4)], dim=2).reshape(M, 2 * H)
    torch.testing.assert_close(dh.float().cpu(), want, atol=3e-2, rtol=2e-2)
    torch.testing.assert_close(dx.float().cpu(), dh.float().cpu() @ w13t.float().cpu().t(), atol=3e-2, rtol=2e-2)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", DENSE_CASES)
def test_attention_dense_boolean_mask(K, dtype, case):
    """FK_MASK_DENSE: any boolean mask, as the reference's attention hands it to SDPA (models/brainformer.py:160-168); heads share it,
    the batch may or may not.  Rows whose first key tiles are wholly masked and keys no query sees are part of the case."""
    B, H, Nq, Nk, D, per_sample, per_head = case
    g = torch.Generator().manual_seed(Nq * 7 + Nk)
    mt = torch.rand(B if per_sample else 1, H if per_head else 1, Nq, Nk, generator=g) < 0.35
    mt[..., : min(96, Nk - 1)] &= (torch.arange(Nq) % 3 != 0)[:, None]      # every third query: nothing visible in the first key tiles
    mt[..., Nk // 2] = False                                                 # a key without any query
    mt[..., Nk - 1] = True                                                   # and no empty row (NaN in the reference)
    m = K.Mask.from_dense(mt.cuda(), Nq, Nk)
    hm = H if per_head and H > 1 else 1
    assert m.limits.dtype == torch.uint8 and m.c == (hm * Nq * Nk if per_sample and B > 1 else 0) and m.q_off == (Nq * Nk if hm > 1 else 0)
    qv, kv, vv, do = rnd(B, Nq, H, D, seed=1), rnd(B, Nk, H, D, seed=2), rnd(B, Nk, H, D, seed=3), rnd(B, Nq, H, D, seed=4)
    qd, kd, vd = dev(qv, dtype), dev(kv, dtype), dev(vv, dtype)
    o, lse = K.attn_fwd(qd, kd, vd, m)
    qr, kr, vr = (q(t_, dtype).requires_grad_(True) for t_ in (qv, kv, vv))
    oref = ref_attn(qr, kr, vr, mt)
    close(o, oref, dtype, atol32=2e-5, atol16=2e-2)
    oref.backward(q(do, dtype))
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    K.attn_bwd(qd, kd, vd, o, dev(do, dtype), lse, dq, dk, dv, m)
    close(dq, qr.grad, dtype, atol32=5e-5, atol16=4e-2)
    close(dk, kr.grad, dtype, atol32=5e-5, atol16=4e-2)
    close(dv, vr.grad, dtype, atol32=5e-5, atol16=4e-2)
    assert float(dk[:, Nk // 2].abs().max()) == 0.0 and float(dv[:, Nk // 2].abs().max()) == 0.0


def test_attention_dense_mask_agrees_with_the_analytic_kinds(K):
    """The same visibility given as a table and as a rule: same kernels, same tile order -> the same bits (fp32), and a wholly masked
    query row gives 0 (the documented difference from SDPA's NaN) with zero gradients."""
    B, H, N, D, Cb = 2, 2, 192, 32, 16
    qv, kv, vv, do = (dev(rnd(B, N, H, D, seed=s), torch.float32) for s in (1, 2, 3, 4))
    mt = mask_tensor(2, Cb, N, N)
    outs = []
    for m in (K.Mask(2, Cb), K.Mask.from_dense(mt.cuda(), N, N)):
        o, lse = K.attn_fwd(qv, kv, vv, m)
        dq, dk, dv = torch.empty_like(qv), torch.empty_like(kv), torch.empty_like(vv)
        K.attn_bwd(qv, kv, vv, o, do, lse, dq, dk, dv, m)
        outs.append((o, lse, dq, dk, dv))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    mt = mt.clone()
    mt[5] = False
    m = K.Mask.from_dense(mt.cuda(), N, N)
    o, lse = K.attn_fwd(qv, kv, vv, m)
    dq, dk, dv = torch.empty_like(qv), torch.empty_like(kv), torch.empty_like(vv)
    K.attn_bwd(qv, kv, vv, o, do, lse, dq, dk, dv, m)
    assert float(o[:, 5].abs().max()) == 0.0 and float(dq[:, 5].abs().max()) == 0.0
    assert all(bool(torch.isfinite(t_).all()) for t_ in (o, dq, dk, dv))
    with pytest.raises(NotImplementedError, match="at most"):
        K.Mask.from_dense(torch.ones(2, 2, 3, N, N, dtype=torch.bool, device="cuda"), N, N)
    from frankenstein_amd._lib import FrankenHipError
    with pytest.raises(FrankenHipError, match="dense mask"):
        K.attn_fwd(qv.bfloat16().repeat(1, 1, 1, 2), kv.bfloat16().repeat(1, 1, 1, 2), vv.bfloat16().repeat(1, 1, 1, 2), m, q_prescaled=True)


@pytest.mark.parametrize("shape", ["b11k", "11q1", "1hq1", "kk_sliced", "b1qk_long"])
def test_attention_dense_mask_broadcast_forms(K, shape):
    """Masks with size-1 query / key axes — the common key-padding form [B, 1, 1, N_k], a query-only [1, 1, N_q, 1], per-head rows —
    are legal in the reference (mask[..., -t_q:, -t_k:] leaves a size-1 axis whole, SDPA broadcasts it: models/brainformer.py:160-168);
    a mask longer than the call is sliced from the END of both axes.  A batch / head extent that is neither 1 nor the call's is refused."""
    B, H, Nq, Nk, D = 3, 2, 70, 150, 32
    g = torch.Generator().manual_seed(11)
    if shape == "b11k":
        mt = torch.rand(B, 1, 1, Nk, generator=g) < 0.6
        mt[..., 0] = True
    elif shape == "11q1":
        mt = torch.ones(1, 1, Nq, 1, dtype=torch.bool)
    elif shape == "1hq1":
        mt = torch.ones(1, H, Nq, 1, dtype=torch.bool)
    elif shape == "kk_sliced":
        mt = torch.rand(Nk + 20, Nk + 9, generator=g) < 0.5
        mt[:, -1] = True
    else:
        mt = torch.rand(B, 1, Nq + 5, Nk + 3, generator=g) < 0.5
        mt[..., -1] = True
    m = K.Mask.from_dense(mt.cuda(), Nq, Nk)
    assert m.limits.shape[-2:] == (Nq, Nk) and m.limits.is_contiguous()
    full = mt[..., max(0, mt.shape[-2] - Nq):, max(0, mt.shape[-1] - Nk):]
    qv, kv, vv, do = rnd(B, Nq, H, D, seed=1), rnd(B, Nk, H, D, seed=2), rnd(B, Nk, H, D, seed=3), rnd(B, Nq, H, D, seed=4)
    qd, kd, vd = (dev(t_, torch.float32) for t_ in (qv, kv, vv))
    o, lse = K.attn_fwd(qd, kd, vd, m)
    qr, kr, vr = (q(t_, torch.float32).requires_grad_(True) for t_ in (qv, kv, vv))
    oref = ref_attn(qr, kr, vr, full)
    close(o, oref, torch.float32, atol32=2e-5, atol16=2e-2)
    oref.backward(q(do, torch.float32))
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    K.attn_bwd(qd, kd, vd, o, dev(do, torch.float32), lse, dq, dk, dv, m)
    close(dq, qr.grad, torch.float32, atol32=5e-5, atol16=4e-2)
    close(dk, kr.grad, torch.float32, atol32=5e-5, atol16=4e-2)
    close(dv, vr.grad, torch.float32, atol32=5e-5, atol16=4e-2)
    bad = K.Mask.from_dense(torch.ones(2, 1, Nq, Nk, dtype=torch.bool, device="cuda"), Nq, Nk)      # two samples' masks for a batch of three
    with pytest.raises(ValueError, match="does not broadcast"):
        K.attn_fwd(qd, kd, vd, bad)
    with pytest.raises(ValueError, match="does not broadcast"):
        K.attn_bwd(qd, kd, vd, o, dev(do, torch.float32), lse, dq, dk, dv, bad)
    with pytest.raises(ValueError, match="does not broadcast"):
        K.Mask.from_dense(torch.ones(B, 1, 7, Nk, dtype=torch.bool, device="cuda"), Nq, Nk)


@pytest.mark.parametrize("case", [(2, 3, 32, 6144, True), (1, 2, 20, 1100, False), (2, 1, 8, 1024, False), (1, 2, 32, 2500, True)])
def test_attention_few_queries_long_context(K, case):
    """the perceiver read-out shape (models/brainformer.py:204-215: 32 queries x 6144 keys, no mask; bf16, D = 64, Nq <= 32, Nk >= 1024):
    the kernels whose eight waves split the KEYS and merge their partial softmax states — forward, LSE and all three gradients against
    the oracle, with ragged key counts (a partial last tile, waves with different tile counts) and fewer than 32 queries; a spike in a
    late key tile makes the waves' maxima differ by far."""
    B, H, Nq, Nk, spike = case
    D, dtype = 64, torch.bfloat16
    qv, kv, vv, do = rnd(B, Nq, H, D, seed=1), rnd(B, Nk, H, D, seed=2), rnd(B, Nk, H, D, seed=3), rnd(B, Nq, H, D, seed=4)
    if spike:
        kv[:, Nk - 40] *= 6.0
    qd, kd, vd = dev(qv, dtype), dev(kv, dtype), dev(vv, dtype)
    o, lse = K.attn_fwd(qd, kd, vd)
    qr, kr, vr = (q(t_, dtype).requires_grad_(True) for t_ in (qv, kv, vv))
    oref = ref_attn(qr, kr, vr, None)
    close(o, oref, dtype, atol16=2e-2)
    s = (qr.transpose(1, 2) @ kr.transpose(1, 2).transpose(-1, -2)) / math.sqrt(D)
    close(lse, torch.logsumexp(s, -1), torch.float32, atol32=3e-2, rtol32=1e-4)
    oref.backward(q(do, dtype))
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    K.attn_bwd(qd, kd, vd, o, dev(do, dtype), lse, dq, dk, dv)
    close(dq, qr.grad, dtype, atol16=4e-2)
    close(dk, kr.grad, dtype, atol16=4e-2)
    close(dv, vr.grad, dtype, atol16=4e-2)


# ----------------------------------------------------------------------------------------------- dropout
def drop_words(seed, step):
    return torch.tensor([seed, step], dtype=torch.int32, device="cuda")


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("p", [0.1, 0.5])
def test_dropout_elementwise_is_the_documented_stream(K, dtype, p):
    """fk_dropout = nn.Dropout in training mode (models/gpt2_model.py:75,91,190): every keep / drop decision equals the host restatement
    (tests/dropout_ref.py), kept values are x / (1 - p), the residual form adds, and the backward is the same mask."""
    from tests import dropout_ref as DR
    n, seed, step, site = 3 * 1000 * 64, 1234567, 5, 3
    x = rnd(n, seed=1) + 3.0                                              # no zeros: a zero output is a dropped element
    words = drop_words(seed, step)
    y = K.dropout(dev(x, dtype), p, words, site)
    keep = torch.from_numpy(DR.keep_flat(seed, step, site, n, p))
    assert torch.equal(y.cpu() != 0, keep)
    assert abs(float(keep.float().mean()) - (1 - p)) < 4.0 * math.sqrt(p * (1 - p) / n)
    want = torch.where(keep, q(x, dtype) * (1.0 / (1.0 - float(np.float32(p)))), torch.zeros(()))
    close(y, want, dtype, atol32=1e-6, rtol32=1e-6, atol16=0.0, rtol16=8e-3)
    res = rnd(n, seed=2)
    y2 = K.dropout(dev(x, dtype), p, words, site, residual=dev(res, dtype))
    close(y2, want + q(res, dtype), dtype, atol32=1e-6, rtol32=1e-6, atol16=4e-2, rtol16=8e-3)
    # another site, another step, another seed: other masks
    for w, st_ in ((words, site + 1), (drop_words(seed, step + 1), site), (drop_words(seed + 1, step), site)):
        other = K.dropout(dev(x, dtype), p, w, st_).cpu() != 0
        assert 0.2 * min(p, 1 - p) < float((other != keep).float().mean()) < 2.2 * p * (1 - p) + 0.05
    from frankenstein_amd._lib import FrankenHipError
    with pytest.raises(FrankenHipError, match="outside"):
        K.dropout(dev(x, dtype), 1.0, words, site)


ATTN_DROP_CASES = [
    # B, H, Nq, Nk, D, mask kind, c, p
    (2, 2, 200, 200, 64, 1, 0, 0.1),
    (1, 3, 57, 300, 32, 0, 0, 0.5),
    (2, 2, 128, 128, 16, 2, 16, 0.25),
]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", ATTN_DROP_CASES)
def test_attention_dropout_matches_the_oracle_with_the_same_draw(K, dtype, case):
    """fk_attn_*_dropout = SDPA(dropout_p) in training mode (models/gpt2_model.py:64): with the mask predicted on the host, forward and
    all three gradients equal the oracle's softmax -> dropout -> P V."""
    from tests import dropout_ref as DR
    B, H, Nq, Nk, D, kind, c, p = case
    seed, step, site = 424242, 9, 2
    words = drop_words(seed, step)
    keep = torch.from_numpy(DR.keep_attention(seed, step, site, B, H, Nq, Nk, p))
    ks = keep.float() * (1.0 / (1.0 - float(np.float32(p))))
    qv, kv, vv, do = rnd(B, Nq, H, D, seed=1), rnd(B, Nk, H, D, seed=2), rnd(B, Nk, H, D, seed=3), rnd(B, Nq, H, D, seed=4)
    qd, kd, vd = dev(qv, dtype), dev(kv, dtype), dev(vv, dtype)
    m = K.Mask(kind, c)
    o, lse = K.attn_fwd(qd, kd, vd, m, dropout=(p, words, site))
    qr, kr, vr = (q(t_, dtype).requires_grad_(True) for t_ in (qv, kv, vv))
    oref = R.sdpa_dropout(qr.transpose(1, 2), kr.transpose(1, 2), vr.transpose(1, 2), mask_tensor(kind, c, Nq, Nk), ks).transpose(1, 2)
    close(o, oref, dtype, atol32=2e-5, atol16=3e-2)
    o0, lse0 = K.attn_fwd(qd, kd, vd, m)
    assert torch.equal(lse, lse0)                                         # the softmax statistics do not see the dropout
    oref.backward(q(do, dtype))
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    K.attn_bwd(qd, kd, vd, o, dev(do, dtype), lse, dq, dk, dv, m, dropout=(p, words, site))
    close(dq, qr.grad, dtype, atol32=5e-5, atol16=5e-2)
    close(dk, kr.grad, dtype, atol32=5e-5, atol16=5e-2)
    close(dv, vr.grad, dtype, atol32=5e-5, atol16=5e-2)


def test_poison_allocator_sees_overruns_and_unwritten_outputs(K):
    """tests/poison.py (FK_TEST_POISON=1 runs the whole suite under it): its guard bands catch a write one element past a buffer, its
    fill makes an unwritten output NaN; and when the mode is on, the product's allocations really go through it."""
    import os
    from tests import poison
    t = poison.Tracker()
    x = t.alloc((10,), torch.float32, "cuda", poison=True, fallback=None)
    assert bool(torch.isnan(x).all())
    x.fill_(1.0)
    assert t.check() == []
    y = t.alloc((3, 8), torch.bfloat16, "cuda", poison=False, fallback=None)
    assert float(y.float().abs().max()) == 0.0
    torch.as_strided(y, (25,), (1,))[24] = 1.0                      # one element past the end
    assert t.check() == [(0, 48, "above")]
    z = t.alloc((4,), torch.int32, "cuda", poison=True, fallback=None)
    torch.as_strided(z, (1,), (1,), storage_offset=z.storage_offset() - 1).fill_(7)   # one element below the start
    assert t.check() == [(0, 16, "below")]
    if os.environ.get("FK_TEST_POISON") == "1":
        assert isinstance(K.torch, poison._Proxy)
        n0 = K.torch._t.count
        K.gelu_fwd(torch.zeros(64, device="cuda"))
        assert K.torch._t.count == n0 + 1


def test_shadow_refresh_single_launch(K):
    """fk_cast_pack_multi (engine.refresh_shadows): every weight shadow re-packed in one launch == the per-weight packs."""
    import frankenstein_amd as fa
    from frankenstein_amd import engine as E
    fa.set_compute_dtype("bf16")
    try:
        g = torch.Generator().manual_seed(11)
        ws = [torch.nn.Parameter(torch.randn(n, 40, generator=g).cuda()) for n in (24, 56, 8)]
        w1, w3 = (torch.nn.Parameter(torch.randn(32, 40, generator=g).cuda()) for _ in range(2))
        b = torch.nn.Parameter(torch.randn(77, generator=g).cuda())
        sh = [E.shadow(ws), E.shadow(ws, transpose=True), E.shadow(ws[:1], pad_k=48), E.shadow_swiglu(w1, w3),
              E.shadow_swiglu(w1, w3, transpose=True), E.shadow([b])]
        with torch.no_grad():
            for p in (*ws, w1, w3, b):
                p.view(-1)[:] = torch.randn(p.numel(), generator=g).cuda()     # in-place through a view, like the optimizer
        E.bump_weight_epoch()
        E.refresh_shadows([*ws, w1, w3, b])
        cat = torch.cat([w.detach() for w in ws]).bfloat16()
        assert torch.equal(sh[0], cat) and torch.equal(sh[1], cat.t())
        assert torch.equal(sh[2][:, :40], ws[0].detach().bfloat16()) and float(sh[2][:, 40:].abs().max()) == 0.0
        inter = torch.stack([w1.detach().view(8, 4, 40), w3.detach().view(8, 4, 40)], 1).reshape(64, 40).bfloat16()
        assert torch.equal(sh[3], inter) and torch.equal(sh[4], inter.t())
        assert torch.equal(sh[5], b.detach().bfloat16())
        assert E.shadow(ws) is sh[0] and E.shadow_swiglu(w1, w3) is sh[3]            # lookups, no re-pack
    finally:
        fa.set_compute_dtype("fp32")


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cfg", [(3, 1, 1), (3, 1, 2), (4, 2, 1), (5, 1, 1), (2, 1, 1)])
def test_im2col_col2im_causal(K, dtype, cfg):
    """fk_im2col1d vs an explicit left-padded unfold; fk_col2im1d is its adjoint (<cols, g> == <x, col2im(g)>)."""
    ks, stride, dil = cfg
    B, T, C = 2, 11, 16
    x = rnd(B * T, C, seed=4).view(B, T, C)
    pad = dil * (ks - 1)
    xp = torch.nn.functional.pad(q(x, dtype), (0, 0, pad, 0))
    tout = (T - 1) // stride + 1
    ref = torch.stack([torch.cat([xp[:, t * stride + k * dil] for k in range(ks)], -1) for t in range(tout)], 1).reshape(B * tout, ks * C)
    cols = K.im2col1d(dev(x, dtype), ks, stride, dil)
    assert torch.equal(cols.float().cpu(), ref)
    g = rnd(B * tout, ks * C, seed=5)
    dx = K.col2im1d(dev(g, torch.float32), B, T, C, ks, stride, dil)
    lhs = (ref.double() * g.double()).sum()
    rhs = (q(x, dtype).double() * dx.cpu().double()).sum()
    assert abs(float(lhs - rhs)) < 1e-3 * max(1.0, abs(float(lhs)))


@pytest.mark.parametrize("dtype", DT)
def test_elu_and_argmax(K, dtype):
    x = rnd(37, 50, seed=6) * 3
    xd = dev(x, dtype)
    close(K.elu_fwd(xd), torch.nn.functional.elu(q(x, dtype)), dtype, atol32=1e-6)
    dy = rnd(37, 50, seed=7)
    ref = q(dy, dtype) * torch.where(q(x, dtype) > 0, torch.ones_like(x), torch.exp(q(x, dtype)))
    close(K.elu_bwd(xd, dev(dy, dtype)), ref, dtype, atol32=1e-5)
    a = q(x, dtype).clone()
    a[3, 7] = a[3, 40] = 100.0                 # tie: first index wins
    a[5] = -5.0                                # constant row
    idx = K.argmax_rows(dev(a, dtype))
    assert torch.equal(idx.cpu(), a.argmax(-1)) and int(idx[3]) == 7 and int(idx[5]) == 0


def test_full_size_kernels_are_deterministic():
    """Every cfg2-shaped GEMM / attention / norm launch twice on the same inputs: bit-identical outputs.  (A prefetch placement in
    the persistent GEMMs once made the RoPE epilogue return a handful of wrong elements out of 2e8, different from run to run;
    small-shape parity tests cannot see that.)"""
    import runpy, io, contextlib, os
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        runpy.run_path(os.path.join(os.path.dirname(__file__), "..", "tools", "determinism_probe.py"), run_name="__main__")
    out = buf.getvalue()
    assert "DETERMINISTIC" in out and "DIFFERS" not in out and "MISMATCH" not in out and out.count(" OK ") >= 13, out


@pytest.mark.parametrize("dtype", DT)
def test_degenerate_shapes(K, dtype):
    """Smallest legal problems: one row / one key / one column vector — nothing may read or write out of bounds."""
    a, w = rnd(1, 8, seed=1), rnd(8, 8, seed=2)
    close(K.gemm_nt(dev(a, dtype), dev(w, dtype)), q(a, dtype) @ q(w, dtype).t(), dtype, atol32=1e-5)
    close(K.gemm_tn(dev(a, dtype), dev(a, dtype)), q(a, dtype).t() @ q(a, dtype), torch.float32, atol32=1e-5)
    x = rnd(1, 8, seed=3)
    y, mean, rstd = K.norm_fwd(dev(x, dtype), torch.ones(8, device="cuda"), torch.zeros(8, device="cuda"), 1e-5)
    close(y, torch.nn.functional.layer_norm(q(x, dtype), (8,)), dtype, atol32=1e-5)
    close(K.colsum(dev(x, dtype)), q(x, dtype)[0], torch.float32, atol32=1e-6)
    qq = rnd(1, 1, 16, seed=4).view(1, 1, 1, 16)
    o, lse = K.attn_fwd(dev(qq, dtype), dev(qq, dtype), dev(qq, dtype), K.Mask(K.MASK_CAUSAL))
    close(o, q(qq, dtype), dtype, atol32=1e-6)                          # a single key: softmax weight 1
    dq, dk, dv = (torch.empty_like(o) for _ in range(3))
    K.attn_bwd(dev(qq, dtype), dev(qq, dtype), dev(qq, dtype), o, torch.ones_like(o), lse, dq, dk, dv, K.Mask(K.MASK_CAUSAL))
    close(dv, torch.ones_like(qq), dtype, atol32=1e-6)
    assert float(dq.abs().max()) < 1e-5 and float(dk.abs().max()) < 1e-5     # softmax over one key is constant


def test_sample_topk_on_device(K):
    """fk_sample_topk = logits / T -> top-k crop (ties kept) -> softmax -> one multinomial draw (models/gpt2_model.py:340-351): the
    empirical distribution of 40 000 draws matches the exact one, draws never leave the top-k set, top_k = 1 is the argmax, and the
    device-side step / position counters advance once per launch whatever the number of rows."""
    B, V, k, T = 3, 1000, 40, 0.8
    g = torch.Generator().manual_seed(3)
    wide = torch.randn(B, 1024, generator=g) * 2.0
    logits = wide[:, :V].contiguous()
    lg = wide.cuda()[:, :V]                                             # row stride 1024 > V
    assert lg.stride(0) == 1024
    ref = logits / T
    kth = ref.topk(k, dim=-1).values[:, -1:]
    probs = torch.softmax(ref.masked_fill(ref < kth, float("-inf")), dim=-1)
    state = K.SampleState(lg.device, seed=1234)
    pos = torch.tensor([7], dtype=torch.int32, device="cuda")
    n = 40000
    out = torch.empty((B, n), dtype=torch.int64, device="cuda")
    cur = torch.empty(B, dtype=torch.int64, device="cuda")
    for _ in range(n):
        K.sample_topk(lg, T, k, state, cur=cur, out=out, pos_inc=pos)
    assert int(state.step) == n and int(pos) == 7 + n and int(state.ticket) == 0
    assert torch.equal(out[:, -1], cur)
    o = out.cpu()
    for b in range(B):
        cnt = torch.bincount(o[b], minlength=V).double()
        assert int((cnt[probs[b] == 0] > 0).sum()) == 0                 # never outside the top-k set
        tv = 0.5 * float((cnt / n - probs[b].double()).abs().sum())
        assert tv < 0.03, tv                                            # total variation; sampling noise at n = 40 000, k = 40 is ~0.012
    # different rows draw independently (same step, different Philox counter)
    assert float((o[0] == o[1]).double().mean()) < 0.2
    # reproducible from the seed
    st2 = K.SampleState(lg.device, seed=1234)
    again = torch.stack([K.sample_topk(lg, T, k, st2).clone() for _ in range(50)], 1)
    assert torch.equal(again.cpu(), o[:, :50])
    # greedy and untruncated forms
    st3 = K.SampleState(lg.device, seed=5)
    assert torch.equal(K.sample_topk(lg, 1.0, 1, st3).cpu(), logits.argmax(-1))
    full = torch.stack([K.sample_topk(lg, 1.0, None, st3).clone() for _ in range(4000)], 1).cpu()
    pf = torch.softmax(logits, -1)
    for b in range(B):
        cnt = torch.bincount(full[b], minlength=V).double()
        assert 0.5 * float((cnt / 4000 - pf[b].double()).abs().sum()) < 0.25
    # a step counter that has run past the output buffer (a reused state, one graph replay too many) must not write past it
    st4 = K.SampleState(lg.device, seed=9)
    buf = torch.full((B, 8), -7, dtype=torch.int64, device="cuda")
    small = buf[:, :3]                                                  # [B, 3] view, row stride 8: columns 3.. are NOT the kernel's
    for _ in range(6):
        K.sample_topk(lg, T, k, st4, cur=cur, out=small)
    assert int(st4.step) == 6 and bool((buf[:, 3:] == -7).all()) and bool((buf[:, :3] >= 0).all())


# ----------------------------------------------------------------------------------------------- fused head + cross entropy
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("rows,V,Kd,bias", [(203, 211, 64, True), (130, 1000, 96, False), (64, 129, 32, True), (333, 300, 32, False)])   # 129, 300: an odd number of 64-column parts
def test_head_ce_fused_kernels(K, dtype, rows, V, Kd, bias):
    """fk_head_ce_fwd / fk_head_ce_bwd (lm_head + F.cross_entropy, models/gpt2_model.py:205-210, without the logits) against torch on the
    same rounded operands: loss, row log-sum-exp, the transposed d-logits (padding rows / columns zero) and the bias gradient; ragged
    rows / V (last tiles partly or, for a wave, wholly past V), ignored rows, fk_transpose2d."""
    vec = 8 if dtype == torch.bfloat16 else 4
    npad = (V + vec - 1) // vec * vec
    h = q(rnd(rows, Kd, seed=1), dtype)
    w = q(rnd(V, Kd, seed=2, scale=0.5), dtype)
    b = q(rnd(V, seed=3), dtype) if bias else None
    tg = torch.randint(0, V, (rows,), generator=torch.Generator().manual_seed(4))
    tg[::7] = -100
    tg[5] = V - 1
    tg[6] = 0
    wpad = torch.zeros(npad, Kd)
    wpad[:V] = w
    hd, wd = dev(h, dtype), dev(wpad, dtype)
    bd = None if b is None else dev(b, dtype)
    tgd = tg.cuda()
    loss2, lse = K.head_ce_fwd(hd, wd, bd, tgd, V)
    hr = h.clone().requires_grad_(True)
    logits = hr @ w.t() + (b if b is not None else 0.0)
    want = torch.nn.functional.cross_entropy(logits, tg, ignore_index=-100)
    tol = 1e-5 if dtype == torch.float32 else 2e-3
    assert abs(float(loss2[0]) - float(want)) < tol * max(1.0, abs(float(want))) and int(loss2[1]) == int((tg != -100).sum())
    torch.testing.assert_close(lse.cpu(), torch.logsumexp(logits.detach(), -1), atol=tol * 5, rtol=tol)
    gout = torch.tensor([0.7], device="cuda")
    rows_pad = (rows + 63) // 64 * 64
    dlT, db = K.head_ce_bwd(hd, wd, bd, tgd, lse, loss2, gout, V, npad, rows_pad, bias)
    logits.retain_grad()
    (want * 0.7).backward()
    full = torch.zeros(npad, rows_pad)
    full[:V, :rows] = logits.grad.t()
    close(dlT, full, dtype, atol32=1e-6, rtol32=1e-4, atol16=2e-5, rtol16=2e-2)
    assert float(dlT[V:].abs().max()) == 0.0 if npad > V else True
    assert float(dlT[:, rows:].abs().max()) == 0.0 if rows_pad > rows else True
    if bias:
        torch.testing.assert_close(db.cpu(), logits.grad.sum(0), atol=2e-5 if dtype == torch.float32 else 2e-4, rtol=1e-2)
    # the two gradient products as engine.HeadCrossEntropy forms them
    dh = K.gemm_tn(dlT, wd)[:rows]
    close(dh, hr.grad, dtype, atol32=1e-5, rtol32=1e-4, atol16=2e-3 * float(hr.grad.abs().max()) + 1e-6, rtol16=3e-2)
    hT = torch.zeros((Kd, rows_pad), dtype=dtype, device="cuda")
    K.transpose2d(hd, out=hT)
    assert torch.equal(hT[:, :rows].t().contiguous(), hd) and float(hT[:, rows:].abs().max() if rows_pad > rows else 0.0) == 0.0
    dw = K.gemm_nt(dlT, hT, out_dtype=torch.float32)[:V]
    want_dw = logits.grad.t() @ h
    close(dw, want_dw, dtype, atol32=1e-5, rtol32=1e-4, atol16=2e-3 * float(want_dw.abs().max()) + 1e-6, rtol16=3e-2)


@pytest.mark.parametrize("dtype", DT)
def test_split_key_attention_equals_the_single_pass(K, dtype):
    """fk_attn_combine: a few queries against a long unmasked context (perceiver read-out, models/brainformer.py:204-215) with the keys
    split into S ranges folded into the batch dimension — forward output / LSE and all three gradients equal the unsplit kernels
    (the way engine.CrossAttnBranch runs it), and the fp32 result matches torch's SDPA."""
    from frankenstein_amd import engine as E
    B, T, H, D, Nc, S = 2, 32, 4, 64, 2048, 8
    HD = H * D
    qf = q(rnd(B, T, H, D, seed=1), dtype)
    kvf = q(rnd(B, Nc, 2 * HD, seed=2), dtype)
    dof = q(rnd(B, T, H, D, seed=3), dtype)
    qd, kv, dod = dev(qf, dtype), dev(kvf, dtype), dev(dof, dtype)
    kv3 = kv.view(B, Nc, 2 * HD)
    k_, v_ = kv3[..., :HD].unflatten(-1, (H, D)), kv3[..., HD:].unflatten(-1, (H, D))
    o1, l1 = K.attn_fwd(qd, k_, v_)
    kvs = kv.view(B * S, Nc // S, 2 * HD)
    ks, vs = kvs[..., :HD].unflatten(-1, (H, D)), kvs[..., HD:].unflatten(-1, (H, D))
    o_s, l_s = K.attn_fwd(E._replicate(qd, S), ks, vs)
    o2, l2 = K.attn_combine(o_s.view(B, S, T, H, D), l_s.view(B, S, H, T))
    close(o2, o1, dtype, atol32=2e-6, rtol32=1e-5, atol16=1e-2)
    torch.testing.assert_close(l2, l1, atol=1e-5, rtol=1e-5)
    if dtype == torch.float32:
        want = torch.nn.functional.scaled_dot_product_attention(qf.transpose(1, 2), kvf[..., :HD].view(B, Nc, H, D).transpose(1, 2),
                                                                kvf[..., HD:].view(B, Nc, H, D).transpose(1, 2)).transpose(1, 2)
        close(o2, want, dtype, atol32=2e-5, rtol32=1e-4)
    # backward: unsplit vs split with the GLOBAL statistics
    dq1, dkv1 = torch.empty_like(qd), torch.empty_like(kv)
    d3 = dkv1.view(B, Nc, 2 * HD)
    K.attn_bwd(qd, k_, v_, o1, dod, l1, dq1, d3[..., :HD].unflatten(-1, (H, D)), d3[..., HD:].unflatten(-1, (H, D)))
    dkv2 = torch.empty_like(kv)
    d2 = dkv2.view(B * S, Nc // S, 2 * HD)
    dq_s = torch.empty((B * S, T, H, D), dtype=dtype, device="cuda")
    K.attn_bwd(E._replicate(qd, S), ks, vs, E._replicate(o1, S), E._replicate(dod, S), E._replicate(l1, S), dq_s,
               d2[..., :HD].unflatten(-1, (H, D)), d2[..., HD:].unflatten(-1, (H, D)))
    dq2 = K.attn_combine(dq_s.view(B, S, T, H, D))[0]
    close(dq2, dq1, dtype, atol32=2e-6, rtol32=1e-4, atol16=2e-2 * float(dq1.float().abs().max()))
    close(dkv2, dkv1, dtype, atol32=2e-6, rtol32=1e-4, atol16=2e-2 * float(dkv1.float().abs().max()))
