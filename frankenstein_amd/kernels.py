"""Tensor-level wrappers over the C ABI (shape checks, output/scratch allocation through
PyTorch's caching allocator, current HIP stream).  No math happens here and there is no
fallback: every function ends in a libfranken_hip.so call.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import FK_BF16, FK_F32, MASK_BLOCK_CAUSAL, MASK_CAUSAL, MASK_DENSE, MASK_KEYPAD, MASK_NONE, MASK_PREFIX, NORM_LAYER, NORM_RMS, ATTN_Q_PRESCALED, call, lib

Tensor = torch.Tensor


# optional live instrumentation (bench.py): name -> list of (start_event, end_event) on the launch stream
TIMERS = None
TIMER_PREFIX = None   # e.g. "attn_": only kernels whose timer name starts with it are bracketed (an event pair costs ~0.6 us of GPU time)


class _timed:
    __slots__ = ("name", "ev")

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.ev = None
        if TIMERS is not None and (TIMER_PREFIX is None or self.name.startswith(TIMER_PREFIX)):
            self.ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self.ev[0].record()

    def __exit__(self, *a):
        if self.ev is not None and TIMERS is not None:
            self.ev[1].record()
            TIMERS.setdefault(self.name, []).append(self.ev)


def fk_dtype(t) -> int:
    dt = t if isinstance(t, torch.dtype) else t.dtype
    if dt == torch.bfloat16:
        return FK_BF16
    if dt == torch.float32:
        return FK_F32
    raise TypeError(f"frankenstein_amd kernels support float32 / bfloat16, got {dt}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


def _ws(nbytes: int, device) -> Tuple[Optional[Tensor], int]:
    if nbytes == 0:
        return None, 0
    w = torch.empty(nbytes, dtype=torch.uint8, device=device)
    return w, nbytes


def _as2d(t: Tensor) -> Tensor:
    assert t.stride(-1) == 1, "innermost dim must be contiguous"
    if t.dim() == 2:
        return t
    return t.reshape(-1, t.shape[-1])


# ------------------------------------------------------------------------------------------- GEMM
def gemm_nt(a: Tensor, w: Tensor, bias: Optional[Tensor] = None, residual: Optional[Tensor] = None,
            res_rows: int = 0, out_dtype: Optional[torch.dtype] = None, out: Optional[Tensor] = None) -> Tensor:
    """out[M,N] = a[M,K] @ w[N,K]^T (+ bias) (+ residual[m % res_rows])."""
    assert a.dim() == 2 and w.dim() == 2 and a.shape[1] == w.shape[1], (a.shape, w.shape)
    assert a.dtype == w.dtype and a.stride(1) == 1 and w.stride(1) == 1
    M, K = a.shape
    N = w.shape[0]
    odt = out_dtype or a.dtype
    if out is None:
        out = torch.empty((M, N), dtype=odt, device=a.device)
    assert out.shape == (M, N) and out.stride(1) == 1 and out.dtype == odt
    if bias is not None:
        assert bias.dtype == a.dtype and bias.numel() == N and bias.is_contiguous()
    ldr = 0
    if residual is not None:
        assert residual.dtype == a.dtype and residual.dim() == 2 and residual.stride(1) == 1 and residual.shape[1] == N
        ldr = residual.stride(0)
    with _timed(f"gemm_nt:{M}x{N}x{K}"):
        call("fk_gemm_nt", a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(), out.stride(0), M, N, K,
             _ptr(bias), _ptr(residual), ldr, res_rows, fk_dtype(a), fk_dtype(odt), _stream())
    return out


def gemm_nt_rope(a: Tensor, w: Tensor, bias: Optional[Tensor], table: Tensor, T: int, pos_off: int, D: int,
                 rot_cols: int, q_cols: int = 0, q_table: Optional[Tensor] = None) -> Tensor:
    """out[M,N] = a @ w^T (+ bias) with RoPE applied to the first rot_cols columns (heads of width D); rows are B x T tokens.
    q_table: table * (softmax_scale * log2 e), in the SAME allocation as table; the first q_cols columns (the queries) are rotated
    with it and so leave the projection pre-scaled for attn_fwd(..., q_prescaled=True)."""
    assert a.dim() == 2 and w.dim() == 2 and a.shape[1] == w.shape[1] and a.dtype == w.dtype and a.stride(1) == 1 and w.stride(1) == 1
    assert table.dtype == torch.float32 and table.is_contiguous() and table.shape[-1] == 2 and table.shape[-2] == D // 2
    M, Kd = a.shape
    N = w.shape[0]
    tbs = table.stride(0) if table.dim() == 4 else 0
    assert pos_off >= 0 and pos_off + T <= table.shape[-3] and M % T == 0
    out = torch.empty((M, N), dtype=a.dtype, device=a.device)
    q_off = 0
    if q_table is not None and q_cols:
        assert q_table.shape == table.shape and q_table.dtype == torch.float32 and q_table.is_contiguous() and q_table.stride() == table.stride()
        q_off = (q_table.data_ptr() - table.data_ptr()) // 4
    else:
        q_cols = 0
    with _timed(f"gemm_nt_rope:{M}x{N}x{Kd}"):
        call("fk_gemm_nt_rope", a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(), N, M, N, Kd, _ptr(bias),
             table.data_ptr(), tbs, T, pos_off, D, rot_cols, q_cols, q_off, fk_dtype(a), _stream())
    return out


def gemm_nt_swiglu(a: Tensor, w13: Tensor) -> Tuple[Tensor, Tensor]:
    """(h13 [M, 2H] interleaved, g [M, H]) = fused up-projection + SwiGLU; w13 is the interleaved [2H, K] shadow."""
    assert a.dim() == 2 and w13.dim() == 2 and a.shape[1] == w13.shape[1] and a.dtype == w13.dtype
    assert a.stride(1) == 1 and w13.is_contiguous()
    M, Kd = a.shape
    H = w13.shape[0] // 2
    h13 = torch.empty((M, 2 * H), dtype=a.dtype, device=a.device)
    g = torch.empty((M, H), dtype=a.dtype, device=a.device)
    with _timed(f"gemm_nt_swiglu:{M}x{2 * H}x{Kd}"):
        call("fk_gemm_nt_swiglu", a.data_ptr(), a.stride(0), w13.data_ptr(), w13.stride(0), h13.data_ptr(), 2 * H,
             g.data_ptr(), H, M, H, Kd, fk_dtype(a), _stream())
    return h13, g


def gemm_nt_dswiglu(dy: Tensor, w2t: Tensor, h13: Tensor) -> Tensor:
    """dh13 [M, 2H] (interleaved) from dy [M, d], the transposed down-projection shadow w2t [H, d] and the saved h13."""
    assert dy.dim() == 2 and w2t.dim() == 2 and dy.shape[1] == w2t.shape[1] and dy.dtype == w2t.dtype == h13.dtype
    assert dy.stride(1) == 1 and w2t.is_contiguous() and h13.is_contiguous()
    M, Kd = dy.shape
    H = w2t.shape[0]
    assert h13.shape == (M, 2 * H)
    dh13 = torch.empty_like(h13)
    with _timed(f"gemm_nt_dswiglu:{M}x{H}x{Kd}"):
        call("fk_gemm_nt_dswiglu", dy.data_ptr(), dy.stride(0), w2t.data_ptr(), w2t.stride(0), h13.data_ptr(), 2 * H,
             dh13.data_ptr(), 2 * H, M, H, Kd, fk_dtype(dy), _stream())
    return dh13


def mlp_bwd_fused(dy: Tensor, w2t: Tensor, h13: Tensor, w13t: Tensor) -> Tuple[Tensor, Tensor]:
    """(dh13 [M, 2H], dx [M, d]) = the SwiGLU MLP's data-gradient chain in one launch (fk_mlp_bwd_fused): what gemm_nt_dswiglu(dy, w2t, h13)
    followed by gemm_nt(dh13, w13t) return, bit for bit, without reading dh13 back.  bf16, d = 384, H % 32 == 0."""
    M, d = dy.shape
    H = w2t.shape[0]
    assert dy.dtype == torch.bfloat16 == w2t.dtype == h13.dtype == w13t.dtype and dy.stride(1) == 1
    assert w2t.shape == (H, d) and w2t.is_contiguous() and h13.shape == (M, 2 * H) and h13.is_contiguous() and w13t.shape == (d, 2 * H) and w13t.is_contiguous()
    dh13 = torch.empty_like(h13)
    dx = torch.empty((M, d), dtype=dy.dtype, device=dy.device)
    with _timed(f"mlp_bwd_fused:{M}x{H}x{d}"):
        call("fk_mlp_bwd_fused", dy.data_ptr(), dy.stride(0), w2t.data_ptr(), w2t.stride(0), h13.data_ptr(), 2 * H, w13t.data_ptr(), w13t.stride(0),
             dh13.data_ptr(), 2 * H, dx.data_ptr(), d, M, H, d, fk_dtype(dy), _stream())
    return dh13, dx


def gemm_tn(a: Tensor, b: Tensor, out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    """out[N1,N2] (fp32) (+)= a[M,N1]^T @ b[M,N2]."""
    assert a.dim() == 2 and b.dim() == 2 and a.shape[0] == b.shape[0] and a.dtype == b.dtype
    assert a.stride(1) == 1 and b.stride(1) == 1
    M, N1 = a.shape
    N2 = b.shape[1]
    if out is None:
        assert not accumulate
        out = torch.empty((N1, N2), dtype=torch.float32, device=a.device)
    assert out.shape == (N1, N2) and out.dtype == torch.float32 and out.stride(1) == 1
    ws, nb = _ws(lib().fk_gemm_tn_workspace_bytes(M, N1, N2, fk_dtype(a)), a.device)
    with _timed(f"gemm_tn:{M}x{N1}x{N2}"):
        call("fk_gemm_tn", a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0), M, N1, N2,
             int(accumulate), fk_dtype(a), _ptr(ws), nb, _stream())
    return out


def colsum(x: Tensor, out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    assert x.dim() == 2 and x.stride(1) == 1
    rows, cols = x.shape
    if out is None:
        assert not accumulate
        out = torch.empty(cols, dtype=torch.float32, device=x.device)
    assert out.numel() == cols and out.dtype == torch.float32 and out.is_contiguous()
    ws, nb = _ws(lib().fk_colsum_workspace_bytes(rows, cols), x.device)
    call("fk_colsum", x.data_ptr(), x.stride(0), out.data_ptr(), rows, cols, int(accumulate), fk_dtype(x), _ptr(ws), nb,
         _stream())
    return out


# ------------------------------------------------------------------------------------------- attention
class Mask:
    """Analytic attention mask: kind in {none, causal, block_causal(C)} with position offsets
    (the reference slices its [N,N] buffer as mask[..., -t_q:, -t_k:], models/brainformer.py:160-162)."""
    __slots__ = ("kind", "c", "q_off", "k_off", "limits", "qfirst")

    def __init__(self, kind: int = MASK_NONE, c: int = 0, q_off: int = 0, k_off: int = 0, limits=None, qfirst=None):
        self.kind, self.c, self.q_off, self.k_off, self.limits, self.qfirst = kind, c, q_off, k_off, limits, qfirst

    def sliced(self, n_full_q: int, n_full_k: int, t_q: int, t_k: int) -> "Mask":
        assert self.kind not in (MASK_PREFIX, MASK_KEYPAD, MASK_DENSE) or (n_full_q == t_q and n_full_k == t_k), "table masks cannot be sliced"
        return Mask(self.kind, self.c, self.q_off + n_full_q - t_q, self.k_off + n_full_k - t_k, self.limits, self.qfirst)

    @staticmethod
    def from_dense(mask: Tensor, t_q: int, t_k: int) -> "Mask":
        """Any boolean mask (True = attend) broadcastable to [B, H, N_q, N_k] — [N, N], [B, 1, N, N], [1, H, N, N], [B, H, N, N]: the
        reference's attention takes whatever tensor it is given and slices it as mask[..., -t_q:, -t_k:] (models/brainformer.py:160-168).
        Stored as uint8 [Bm, Hm, t_q, t_k] for the per-element path of the generic kernels (c = batch stride, q_off = head stride)."""
        assert mask.dim() >= 2
        m = mask[..., max(0, mask.shape[-2] - t_q):, max(0, mask.shape[-1] - t_k):]      # a size-1 axis stays whole, like the Python slice
        if m.dim() > 4:
            if any(d != 1 for d in m.shape[:-4]):
                raise NotImplementedError(f"dense attention mask {tuple(mask.shape)}: at most [B, H, N_q, N_k]")
            m = m.reshape(m.shape[-4:])
        while m.dim() < 4:
            m = m.unsqueeze(0)
        if m.shape[-2] not in (1, t_q) or m.shape[-1] not in (1, t_k):      # SDPA would refuse to broadcast it too
            raise ValueError(f"attention mask {tuple(mask.shape)} does not broadcast to {t_q} queries x {t_k} keys")
        m = m.expand(m.shape[0], m.shape[1], t_q, t_k)          # key-padding [B, 1, 1, N_k] and query-only [.., N_q, 1] forms: SDPA broadcasts them
        bm, hm = m.shape[0], m.shape[1]
        u8 = m.to(torch.uint8).contiguous()
        return Mask(MASK_DENSE, 0 if bm == 1 else hm * t_q * t_k, 0 if hm == 1 else t_q * t_k, 0, u8, None)

    @staticmethod
    def from_padding(q_valid: Tensor, k_valid: Tensor) -> "Mask":
        """visible(i, j) = q_valid[b, i] & k_valid[b, j]  (create_attention_mask_from_padding, models/simple_mae:228-236)."""
        return Mask(MASK_KEYPAD, 0, 0, 0, q_valid.to(torch.int32).contiguous(), k_valid.to(torch.int32).contiguous())

    @staticmethod
    def from_token_ids(q_ids: Tensor, k_ids: Tensor, block: int) -> "Mask":
        """Sub-mask of the block-causal mask at (ascending) token indices: visible(i, j) = k_ids[j] // block <= q_ids[i] // block
        (MAE.get_sub_att_matrix, models/brainformer.py:392-413), as prefix tables instead of a [B,1,n,n] tensor."""
        assert q_ids.dtype == torch.int64 and k_ids.dtype == torch.int64 and q_ids.dim() == 2 and k_ids.dim() == 2
        B, nq = q_ids.shape
        nk = k_ids.shape[1]
        limits = torch.empty((B, nq), dtype=torch.int32, device=q_ids.device)
        qfirst = torch.empty((B, nk), dtype=torch.int32, device=q_ids.device)
        call("fk_prefix_mask", q_ids.contiguous().data_ptr(), k_ids.contiguous().data_ptr(), block, limits.data_ptr(),
             qfirst.data_ptr(), B, nq, nk, _stream())
        return Mask(MASK_PREFIX, block, 0, 0, limits, qfirst)


NO_MASK = Mask()


def _bnhd(t: Tensor):
    assert t.dim() == 4 and t.stride(3) == 1 and t.stride(2) == t.shape[3], "need [B,N,H,D] with heads packed in a row"
    return t.stride(0), t.stride(1)


def _check_dense(mask: "Mask", B: int, H: int, Nq: int, Nk: int) -> None:
    """A dense table is read at b * c + h * q_off + q * Nk + k: its batch / head extents must be 1 (broadcast) or the call's."""
    if mask.kind == MASK_DENSE:
        t = mask.limits
        assert t is not None and t.dim() == 4 and t.dtype == torch.uint8 and t.is_contiguous()
        if t.shape[0] not in (1, B) or t.shape[1] not in (1, H) or t.shape[2] != Nq or t.shape[3] != Nk:
            raise ValueError(f"dense attention mask {tuple(t.shape)} does not broadcast to [B={B}, H={H}, {Nq}, {Nk}]")


def attn_fwd(q: Tensor, k: Tensor, v: Tensor, mask: Mask = NO_MASK, scale: Optional[float] = None,
             out: Optional[Tensor] = None, q_prescaled: bool = False, dropout: Optional[tuple] = None) -> Tuple[Tensor, Tensor]:
    """q [B,Nq,H,D], k/v [B,Nk,H,D] (strided views ok) -> (o [B,Nq,H,D], lse [B,H,Nq] fp32).
    q_prescaled: q already holds scale * log2(e) * q (gemm_nt_rope's q_table; bf16, D = 64 only).
    dropout = (p, seed words [2] int32 on the device, site): SDPA's dropout_p in training mode (fk_attn_fwd_dropout)."""
    B, Nq, H, D = q.shape
    Nk = k.shape[1]
    assert k.shape == (B, Nk, H, D) and v.shape == (B, Nk, H, D) and q.dtype == k.dtype == v.dtype
    _check_dense(mask, B, H, Nq, Nk)
    if out is None:
        out = torch.empty((B, Nq, H, D), dtype=q.dtype, device=q.device)
    lse = torch.empty((B, H, Nq), dtype=torch.float32, device=q.device)
    (qb, qr), (kb, kr), (vb, vr), (ob, orr) = _bnhd(q), _bnhd(k), _bnhd(v), _bnhd(out)
    sc = scale if scale is not None else 1.0 / math.sqrt(D)
    with _timed(f"attn_fwd:{B}x{H}x{Nq}x{Nk}x{D}:m{mask.kind}"):
      if dropout is None:
        call("fk_attn_fwd", q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr(), B, H, Nq, Nk, D,
             qb, qr, kb, kr, vb, vr, ob, orr, mask.kind, mask.c, mask.q_off, mask.k_off, _ptr(mask.limits), _ptr(mask.qfirst),
             sc, ATTN_Q_PRESCALED if q_prescaled else 0, fk_dtype(q), _stream())
      else:
        call("fk_attn_fwd_dropout", q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr(), B, H, Nq, Nk, D,
             qb, qr, kb, kr, vb, vr, ob, orr, mask.kind, mask.c, mask.q_off, mask.k_off, _ptr(mask.limits), _ptr(mask.qfirst),
             sc, ATTN_Q_PRESCALED if q_prescaled else 0, dropout[0], dropout[1].data_ptr(), dropout[2], fk_dtype(q), _stream())
    return out, lse


def attn_bwd(q: Tensor, k: Tensor, v: Tensor, o: Tensor, do: Tensor, lse: Tensor, dq: Tensor, dk: Tensor, dv: Tensor,
             mask: Mask = NO_MASK, scale: Optional[float] = None, rope_table: Optional[Tensor] = None,
             rope_off: int = 0, q_prescaled: bool = False, dropout: Optional[tuple] = None) -> None:
    """Writes dq/dk/dv (same strides as q/k/v); do must have o's strides.  rope_table: also un-rotate dq/dk (RoPE backward).
    dropout: the forward's (p, seed words, site) — the mask is regenerated, not stored."""
    B, Nq, H, D = q.shape
    Nk = k.shape[1]
    (qb, qr), (kb, kr), (vb, vr), (ob, orr) = _bnhd(q), _bnhd(k), _bnhd(v), _bnhd(o)
    assert _bnhd(dq) == (qb, qr) and _bnhd(dk) == (kb, kr) and _bnhd(dv) == (vb, vr) and _bnhd(do) == (ob, orr)
    assert do.dtype == q.dtype and dq.dtype == q.dtype
    _check_dense(mask, B, H, Nq, Nk)
    delta = torch.empty(2 * B * H * ((Nq + 63) // 64 * 64), dtype=torch.float32, device=q.device)     # scratch: row statistics for dK/dV
    sc = scale if scale is not None else 1.0 / math.sqrt(D)
    with _timed(f"attn_bwd:{B}x{H}x{Nq}x{Nk}x{D}:m{mask.kind}"):
      head = (q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(),
              dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), delta.data_ptr(), B, H, Nq, Nk, D, qb, qr, kb, kr, vb, vr, ob, orr,
              mask.kind, mask.c, mask.q_off, mask.k_off, _ptr(mask.limits), _ptr(mask.qfirst), sc, _ptr(rope_table),
              0 if rope_table is None or rope_table.dim() == 3 else rope_table.stride(0), rope_off,
              ATTN_Q_PRESCALED if q_prescaled else 0)
      if dropout is None:
        call("fk_attn_bwd", *head, fk_dtype(q), _stream())
      else:
        call("fk_attn_bwd_dropout", *head, dropout[0], dropout[1].data_ptr(), dropout[2], fk_dtype(q), _stream())


def attn_combine(parts: Tensor, lse_parts: Optional[Tensor] = None, want_lse: bool = True):
    """parts [B, S, T, H, D] (+ lse_parts [B, S, H, T]) -> (out [B, T, H, D], lse [B, H, T] or None): fk_attn_combine."""
    B, S, T, H, D = parts.shape
    assert parts.is_contiguous()
    out = torch.empty((B, T, H, D), dtype=parts.dtype, device=parts.device)
    lse = None
    if lse_parts is not None:
        assert lse_parts.dtype == torch.float32 and lse_parts.is_contiguous() and lse_parts.shape == (B, S, H, T)
        lse = torch.empty((B, H, T), dtype=torch.float32, device=parts.device) if want_lse else None
    call("fk_attn_combine", parts.data_ptr(), _ptr(lse_parts), out.data_ptr(), _ptr(lse), B, S, T, H, D, fk_dtype(parts), _stream())
    return out, lse


# ------------------------------------------------------------------------------------------- norms
def norm_fwd(x: Tensor, gamma: Tensor, beta: Optional[Tensor], eps: float, kind: int = NORM_LAYER):
    x2 = _as2d(x)
    assert x2.is_contiguous() and gamma.dtype == torch.float32
    rows, dim = x2.shape
    y = torch.empty_like(x2)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    call("fk_norm_fwd", x2.data_ptr(), gamma.data_ptr(), _ptr(beta), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
         rows, dim, eps, kind, fk_dtype(x), _stream())
    return y.view(x.shape), mean, rstd


def norm_bwd(dy: Tensor, x: Tensor, gamma: Tensor, mean: Tensor, rstd: Tensor, dres: Optional[Tensor] = None,
             kind: int = NORM_LAYER, want_beta: bool = True, dgamma: Optional[Tensor] = None, dbeta: Optional[Tensor] = None,
             accumulate: bool = False):
    """returns (dx, dgamma, dbeta) with dx = dres + norm_bwd(dy); given dgamma / dbeta buffers are written (+= when accumulate)."""
    x2, dy2 = _as2d(x), _as2d(dy)
    assert x2.is_contiguous() and dy2.is_contiguous() and dy2.dtype == x2.dtype
    rows, dim = x2.shape
    dx = torch.empty_like(x2)
    if dgamma is None:
        assert not accumulate
        dgamma = torch.empty(dim, dtype=torch.float32, device=x.device)
    if dbeta is None and want_beta:
        assert not accumulate
        dbeta = torch.empty(dim, dtype=torch.float32, device=x.device)
    if not want_beta:
        dbeta = None
    for t in (dgamma, dbeta):
        assert t is None or (t.dtype == torch.float32 and t.numel() == dim and t.is_contiguous())
    dres2 = None
    if dres is not None:
        dres2 = _as2d(dres)
        assert dres2.is_contiguous() and dres2.dtype == x2.dtype
    ws, nb = _ws(lib().fk_norm_bwd_workspace_bytes(rows, dim), x.device)
    call("fk_norm_bwd", dy2.data_ptr(), x2.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _ptr(dres2),
         dx.data_ptr(), dgamma.data_ptr(), _ptr(dbeta), rows, dim, kind, int(accumulate), fk_dtype(x), _ptr(ws), nb, _stream())
    return dx.view(x.shape), dgamma, dbeta


# ------------------------------------------------------------------------------------------- pointwise
def rope_(x: Tensor, nheads: int, D: int, table: Tensor, pos_off: int = 0, conj: bool = False) -> Tensor:
    """In place on the first nheads*D columns of each row of x [B,T,ld]; table fp32 [Tc, D/2, 2] or [B, Tc, D/2, 2]."""
    assert x.dim() == 3 and x.stride(2) == 1 and x.stride(0) == x.shape[1] * x.stride(1)
    assert table.dtype == torch.float32 and table.is_contiguous() and table.shape[-1] == 2 and table.shape[-2] == D // 2
    B, T, _ = x.shape
    tbs = 0
    if table.dim() == 4:
        assert table.shape[0] == B
        tbs = table.stride(0)
    assert pos_off >= 0 and pos_off + T <= table.shape[-3], "rope cache shorter than the sequence"
    call("fk_rope", x.data_ptr(), B, T, x.stride(1), nheads, D, table.data_ptr(), tbs, pos_off, int(conj), fk_dtype(x), _stream())
    return x


def patchify(x: Tensor, P: int, ldp: int, dtype: torch.dtype) -> Tensor:
    assert x.dtype == torch.float32 and x.dim() == 3 and x.is_contiguous()
    B, T, Cc = x.shape
    tok = torch.empty((B * (T // P) * Cc, ldp), dtype=dtype, device=x.device)
    call("fk_patchify", x.data_ptr(), tok.data_ptr(), B, T, Cc, P, ldp, fk_dtype(dtype), _stream())
    return tok


def swiglu_fwd(h13: Tensor) -> Tensor:
    assert h13.dim() == 2 and h13.is_contiguous()
    rows, H2 = h13.shape
    g = torch.empty((rows, H2 // 2), dtype=h13.dtype, device=h13.device)
    call("fk_swiglu_fwd", h13.data_ptr(), g.data_ptr(), rows, H2 // 2, fk_dtype(h13), _stream())
    return g


def swiglu_bwd(h13: Tensor, dg: Tensor) -> Tensor:
    assert h13.is_contiguous() and dg.is_contiguous()
    rows, H2 = h13.shape
    d = torch.empty_like(h13)
    call("fk_swiglu_bwd", h13.data_ptr(), dg.data_ptr(), d.data_ptr(), rows, H2 // 2, fk_dtype(h13), _stream())
    return d


def gelu_fwd(x: Tensor) -> Tensor:
    assert x.is_contiguous()
    y = torch.empty_like(x)
    call("fk_gelu_fwd", x.data_ptr(), y.data_ptr(), x.numel(), fk_dtype(x), _stream())
    return y


def dropout(x: Tensor, p: float, seed: Tensor, site: int, residual: Optional[Tensor] = None, out: Optional[Tensor] = None) -> Tensor:
    """y = [residual +] keep ? x / (1 - p) : 0 (fk_dropout; nn.Dropout in training mode).  seed: int32 [2] on the device (seed, step).
    The backward is the same call on dy (same p / seed / site) without residual."""
    assert x.is_contiguous() and (residual is None or (residual.is_contiguous() and residual.shape == x.shape and residual.dtype == x.dtype))
    assert seed.dtype == torch.int32 and seed.numel() == 2 and seed.device == x.device
    y = torch.empty_like(x) if out is None else out
    call("fk_dropout", x.data_ptr(), _ptr(residual), y.data_ptr(), x.numel(), p, seed.data_ptr(), site, fk_dtype(x), _stream())
    return y


def gelu_bwd(x: Tensor, dy: Tensor) -> Tensor:
    assert x.is_contiguous() and dy.is_contiguous()
    dx = torch.empty_like(x)
    call("fk_gelu_bwd", x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), fk_dtype(x), _stream())
    return dx


def cast_pack(src: Tensor, dst: Tensor, transpose: bool = False) -> Tensor:
    """dst[r, c] = src[r, c] (or dst[c, r]) ; src fp32 [rows, cols], dst pre-allocated (may be wider: zero padded by caller)."""
    assert src.dtype == torch.float32 and src.dim() == 2 and src.stride(1) == 1 and dst.dim() == 2 and dst.stride(1) == 1
    rows, cols = src.shape
    call("fk_cast_pack", src.data_ptr(), src.stride(0), dst.data_ptr(), dst.stride(0), rows, cols, int(transpose),
         fk_dtype(dst), _stream())
    return dst


def cast_pack_rows(src: Tensor, dst: Tensor, transpose: bool, rblk: int, rstride: int, roff: int) -> Tensor:
    """cast_pack with the row map j -> (j // rblk) * rstride + j % rblk + roff (interleaved SwiGLU weight shadows)."""
    assert src.dtype == torch.float32 and src.dim() == 2 and src.stride(1) == 1 and dst.dim() == 2 and dst.stride(1) == 1
    rows, cols = src.shape
    call("fk_cast_pack_rows", src.data_ptr(), src.stride(0), dst.data_ptr(), dst.stride(0), rows, cols, int(transpose),
         rblk, rstride, roff, fk_dtype(dst), _stream())
    return dst


def cast_pack_multi(jobs: Tensor, njobs: int, total_chunks: int, dtype: torch.dtype) -> None:
    """jobs: device uint8 tensor holding njobs fk_pack_job records (see include/franken_hip.h)."""
    assert jobs.dtype == torch.uint8 and jobs.is_cuda and jobs.numel() >= 64 * njobs
    call("fk_cast_pack_multi", jobs.data_ptr(), njobs, total_chunks, fk_dtype(dtype), _stream())


def cast(src: Tensor, dtype: torch.dtype) -> Tensor:
    assert src.is_contiguous()
    dst = torch.empty(src.shape, dtype=dtype, device=src.device)
    call("fk_cast", src.data_ptr(), fk_dtype(src), dst.data_ptr(), fk_dtype(dtype), src.numel(), _stream())
    return dst


def add(a: Tensor, b: Tensor) -> Tensor:
    assert a.is_contiguous() and b.is_contiguous() and a.shape == b.shape and a.dtype == b.dtype
    y = torch.empty_like(a)
    call("fk_add", a.data_ptr(), b.data_ptr(), y.data_ptr(), a.numel(), fk_dtype(a), _stream())
    return y


def gather_rows(src: Tensor, idx: Tensor, out_dtype: Optional[torch.dtype] = None, idx_mod: int = 0) -> Tensor:
    """out[b, i, :] = src[b, idx[b,i] (% idx_mod), :]; src [B, N, W] or a shared table [N, W]; idx int64 [B, n]."""
    assert idx.dtype == torch.int64 and idx.dim() == 2 and idx.is_contiguous() and src.is_contiguous()
    B, n = idx.shape
    W = src.shape[-1]
    sbs = src.shape[-2] * W if src.dim() == 3 else 0
    assert src.dim() == 2 or src.shape[0] == B
    odt = out_dtype or src.dtype
    out = torch.empty((B, n, W), dtype=odt, device=src.device)
    call("fk_gather_rows", src.data_ptr(), sbs, fk_dtype(src), idx.data_ptr(), idx_mod, out.data_ptr(), n * W, fk_dtype(odt),
         B, n, W, 0, _stream())
    return out


def scatter_rows_(dst: Tensor, idx: Tensor, src: Tensor) -> Tensor:
    """dst[b, idx[b,i], :] = src[b, i, :]   (dst [B, N, W] contiguous, updated in place)."""
    assert idx.dtype == torch.int64 and idx.is_contiguous() and src.is_contiguous() and dst.is_contiguous()
    B, n = idx.shape
    W = src.shape[-1]
    assert src.shape == (B, n, W) and dst.dim() == 3 and dst.shape[0] == B and dst.shape[2] == W
    call("fk_gather_rows", src.data_ptr(), n * W, fk_dtype(src), idx.data_ptr(), 0, dst.data_ptr(), dst.shape[1] * W,
         fk_dtype(dst), B, n, W, 1, _stream())
    return dst


def scatter_add_rows_(table: Tensor, idx: Tensor, src: Tensor, idx_mod: int = 0) -> Tensor:
    """table[idx[r] (% idx_mod), :] += src[r, :]  (fp32 table; atomics)."""
    assert table.dtype == torch.float32 and table.is_contiguous() and src.is_contiguous() and idx.is_contiguous()
    W = table.shape[-1]
    rows = idx.numel()
    assert src.numel() == rows * W
    call("fk_scatter_add_rows", src.data_ptr(), fk_dtype(src), idx.data_ptr(), idx_mod, table.data_ptr(), rows, W, _stream())
    return table


def copy2d(src: Tensor, dst: Tensor) -> Tensor:
    assert src.dim() == 2 and dst.shape == src.shape and src.stride(1) == 1 and dst.stride(1) == 1 and src.dtype == dst.dtype
    call("fk_copy2d", src.data_ptr(), src.stride(0), dst.data_ptr(), dst.stride(0), src.shape[0], src.shape[1],
         fk_dtype(src), _stream())
    return dst


def add2d_(dst: Tensor, src: Tensor) -> Tensor:
    """dst += src  (fp32 2-D, row strides allowed)."""
    assert dst.dtype == torch.float32 and src.dtype == torch.float32 and dst.shape == src.shape and dst.dim() == 2
    assert dst.stride(1) == 1 and src.stride(1) == 1
    call("fk_add2d", src.data_ptr(), src.stride(0), dst.data_ptr(), dst.stride(0), dst.shape[0], dst.shape[1], _stream())
    return dst


# ------------------------------------------------------------------------------------------- decode step (device-side position)
def gpt_embed_step(idx: Tensor, wte: Tensor, wpe: Tensor, pos: Tensor, dtype: torch.dtype) -> Tensor:
    """x[b] = wte[idx[b]] + wpe[pos[0]]; idx int64 [B], pos int32 [1] on the device."""
    assert idx.dtype == torch.int64 and idx.is_contiguous() and pos.dtype == torch.int32 and pos.numel() == 1
    assert wte.dtype == torch.float32 and wpe.dtype == torch.float32 and wte.is_contiguous() and wpe.is_contiguous()
    B, dim = idx.numel(), wte.shape[1]
    out = torch.empty((B, dim), dtype=dtype, device=idx.device)
    call("fk_gpt_embed_step", idx.data_ptr(), wte.data_ptr(), wpe.data_ptr(), pos.data_ptr(), out.data_ptr(), B, dim, wte.shape[0],
         fk_dtype(dtype), _stream())
    return out


def kv_append_(qkv: Tensor, kv: Tensor, pos: Tensor) -> None:
    """kv[b, pos[0], :] = qkv[b, d:3d]; qkv [B, 3d] contiguous, kv [B, Tmax, 2d] contiguous."""
    B, d3 = qkv.shape
    assert qkv.is_contiguous() and kv.is_contiguous() and kv.shape[0] == B and kv.shape[2] * 3 == d3 * 2 and kv.dtype == qkv.dtype
    call("fk_kv_append", qkv.data_ptr(), kv.data_ptr(), pos.data_ptr(), B, d3 // 3, kv.shape[1], fk_dtype(qkv), _stream())


class SampleState:
    """Device-side state of fk_sample_topk: Philox seed, step counter (= column of `out` the next draw goes to), ticket word."""

    def __init__(self, device, seed: Optional[int] = None, step: int = 0):
        if seed is None:                                   # follows torch.manual_seed like torch.multinomial would
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        self.seed = torch.tensor([seed], dtype=torch.int64, device=device)
        self.step = torch.tensor([step], dtype=torch.int64, device=device)
        self.ticket = torch.zeros(1, dtype=torch.int32, device=device)


def sample_topk(logits: Tensor, temperature: float, top_k: Optional[int], state: SampleState, cur: Optional[Tensor] = None,
                out: Optional[Tensor] = None, pos_inc: Optional[Tensor] = None) -> Tensor:
    """One token per row of fp32 logits [B, V] (row stride >= V): temperature, top-k crop, softmax, multinomial — one launch
    (models/gpt2_model.py:340-351).  Writes cur [B] int64 (returned), out[:, step] if given, then step += 1 (and pos_inc += 1)."""
    assert logits.dim() == 2 and logits.dtype == torch.float32 and logits.stride(1) == 1
    B, V = logits.shape
    if cur is None:
        cur = torch.empty(B, dtype=torch.int64, device=logits.device)
    assert cur.dtype == torch.int64 and cur.is_contiguous() and cur.numel() == B
    out_ld = out_cols = 0
    if out is not None:
        assert out.dtype == torch.int64 and out.dim() == 2 and out.shape[0] == B and out.stride(1) == 1
        out_ld, out_cols = out.stride(0), out.shape[1]          # the kernel stops writing at column out_cols, whatever the step counter says
    if pos_inc is not None:
        assert pos_inc.dtype == torch.int32 and pos_inc.numel() == 1
    call("fk_sample_topk", logits.data_ptr(), logits.stride(0), B, V, float(temperature), int(top_k or 0), state.seed.data_ptr(),
         state.step.data_ptr(), _ptr(pos_inc), cur.data_ptr(), _ptr(out), out_ld, out_cols, state.ticket.data_ptr(), _stream())
    return cur


def attn_decode(qkv: Tensor, kv: Tensor, pos: Tensor, n_head: int) -> Tensor:
    """one causal query per sample against the cache rows 0..pos[0]: q = qkv[:, :d] -> o [B, d]."""
    B, d3 = qkv.shape
    d = d3 // 3
    D = d // n_head
    out = torch.empty((B, d), dtype=qkv.dtype, device=qkv.device)
    call("fk_attn_decode", qkv.data_ptr(), qkv.stride(0), kv.data_ptr(), kv.stride(0), kv.stride(1), out.data_ptr(), d, pos.data_ptr(),
         B, n_head, D, 1.0 / math.sqrt(D), fk_dtype(qkv), _stream())
    return out


# ------------------------------------------------------------------------------------------- conv (VQ-VAE tokenizer)
def im2col1d(x: Tensor, ksize: int, stride: int = 1, dil: int = 1) -> Tensor:
    """x [B, T, C] -> cols [B * Tout, ksize * C] with the causal left padding dil * (ksize - 1); Tout = (T - 1) // stride + 1."""
    assert x.dim() == 3 and x.is_contiguous()
    B, T, C = x.shape
    tout = (T - 1) // stride + 1
    cols = torch.empty((B * tout, ksize * C), dtype=x.dtype, device=x.device)
    call("fk_im2col1d", x.data_ptr(), cols.data_ptr(), B, T, C, ksize, stride, dil, fk_dtype(x), _stream())
    return cols


def col2im1d(dcols: Tensor, B: int, T: int, C: int, ksize: int, stride: int = 1, dil: int = 1) -> Tensor:
    tout = (T - 1) // stride + 1
    assert dcols.is_contiguous() and dcols.shape == (B * tout, ksize * C)
    dx = torch.empty((B, T, C), dtype=dcols.dtype, device=dcols.device)
    call("fk_col2im1d", dcols.data_ptr(), dx.data_ptr(), B, T, C, ksize, stride, dil, fk_dtype(dcols), _stream())
    return dx


def elu_fwd(x: Tensor) -> Tensor:
    assert x.is_contiguous()
    y = torch.empty_like(x)
    call("fk_elu_fwd", x.data_ptr(), y.data_ptr(), x.numel(), fk_dtype(x), _stream())
    return y


def elu_bwd(x: Tensor, dy: Tensor) -> Tensor:
    assert x.is_contiguous() and dy.is_contiguous() and dy.dtype == x.dtype and dy.shape == x.shape
    dx = torch.empty_like(x)
    call("fk_elu_bwd", x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), fk_dtype(x), _stream())
    return dx


def argmax_rows(x: Tensor) -> Tensor:
    assert x.dim() == 2 and x.stride(1) == 1
    idx = torch.empty(x.shape[0], dtype=torch.int64, device=x.device)
    call("fk_argmax_rows", x.data_ptr(), x.stride(0), idx.data_ptr(), x.shape[0], x.shape[1], fk_dtype(x), _stream())
    return idx


# ------------------------------------------------------------------------------------------- losses
def l1_loss_fwd(pred: Tensor, target: Tensor, squared: bool = False, row_weight: Optional[Tensor] = None) -> Tensor:
    """-> loss2 fp32[2] = {mean (weighted) |d| or d^2, weight sum}; row_weight: fp32 [rows] with rows = numel / last dim."""
    assert pred.is_contiguous() and target.is_contiguous() and pred.shape == target.shape and pred.dtype == target.dtype
    loss2 = torch.empty(2, dtype=torch.float32, device=pred.device)
    row_len = pred.shape[-1]
    if row_weight is not None:
        assert row_weight.dtype == torch.float32 and row_weight.is_contiguous() and row_weight.numel() * row_len == pred.numel()
    ws, nb = _ws(lib().fk_loss_workspace_bytes(pred.numel()), pred.device)
    call("fk_l1_loss_fwd", pred.data_ptr(), target.data_ptr(), loss2.data_ptr(), pred.numel(), int(squared), _ptr(row_weight),
         row_len, fk_dtype(pred), _ptr(ws), nb, _stream())
    return loss2


def l1_loss_bwd(pred: Tensor, target: Tensor, gout: Tensor, squared: bool = False, row_weight: Optional[Tensor] = None,
                loss2: Optional[Tensor] = None) -> Tensor:
    assert gout.dtype == torch.float32 and gout.numel() == 1
    d = torch.empty_like(pred)
    call("fk_l1_loss_bwd", pred.data_ptr(), target.data_ptr(), gout.data_ptr(), d.data_ptr(), pred.numel(), int(squared),
         _ptr(row_weight), pred.shape[-1], _ptr(loss2), fk_dtype(pred), _stream())
    return d


def ce_loss_fwd(logits: Tensor, targets: Tensor, ignore_index: int = -100):
    """logits [rows, V] (row stride ok), targets int64 [rows] -> (loss2 fp32[2] = {mean nll, count}, row_lse)."""
    assert logits.dim() == 2 and logits.stride(1) == 1 and targets.dtype == torch.int64 and targets.is_contiguous()
    rows, V = logits.shape
    assert targets.numel() == rows
    loss2 = torch.empty(2, dtype=torch.float32, device=logits.device)
    lse = torch.empty(rows, dtype=torch.float32, device=logits.device)
    ws, nb = _ws(lib().fk_ce_workspace_bytes(rows), logits.device)
    call("fk_ce_loss_fwd", logits.data_ptr(), logits.stride(0), targets.data_ptr(), loss2.data_ptr(), lse.data_ptr(),
         rows, V, ignore_index, fk_dtype(logits), _ptr(ws), nb, _stream())
    return loss2, lse


def ce_loss_bwd(logits: Tensor, targets: Tensor, lse: Tensor, loss2: Tensor, gout: Tensor, dlogits: Tensor,
                ignore_index: int = -100) -> Tensor:
    rows, V = logits.shape
    assert dlogits.shape == logits.shape and dlogits.stride(1) == 1 and dlogits.dtype == logits.dtype
    call("fk_ce_loss_bwd", logits.data_ptr(), logits.stride(0), targets.data_ptr(), lse.data_ptr(), loss2.data_ptr(),
         gout.data_ptr(), dlogits.data_ptr(), dlogits.stride(0), rows, V, ignore_index, fk_dtype(logits), _stream())
    return dlogits


class CeChunkState:
    """per-row running statistics of the chunked cross entropy (fk_ce_chunk_*)"""

    def __init__(self, rows: int, device):
        self.m = torch.empty(rows, dtype=torch.float32, device=device)
        self.s = torch.empty(rows, dtype=torch.float32, device=device)
        self.t = torch.empty(rows, dtype=torch.float32, device=device)
        self.first = True


def ce_chunk_fwd(logits: Tensor, targets: Tensor, col0: int, st: CeChunkState) -> None:
    assert logits.dim() == 2 and logits.dtype == torch.float32 and logits.stride(1) == 1 and targets.dtype == torch.int64 and targets.is_contiguous()
    rows, cw = logits.shape
    call("fk_ce_chunk_fwd", logits.data_ptr(), logits.stride(0), targets.data_ptr(), col0, st.m.data_ptr(), st.s.data_ptr(), st.t.data_ptr(),
         rows, cw, int(st.first), _stream())
    st.first = False


def ce_chunk_finish(st: CeChunkState, targets: Tensor, V: int, ignore_index: int = -100):
    rows = targets.numel()
    loss2 = torch.empty(2, dtype=torch.float32, device=targets.device)
    lse = torch.empty(rows, dtype=torch.float32, device=targets.device)
    ws, nb = _ws(lib().fk_ce_workspace_bytes(rows), targets.device)
    call("fk_ce_chunk_finish", st.m.data_ptr(), st.s.data_ptr(), st.t.data_ptr(), targets.data_ptr(), lse.data_ptr(), loss2.data_ptr(),
         rows, V, ignore_index, _ptr(ws), nb, _stream())
    return loss2, lse


def ce_chunk_bwd(logits: Tensor, targets: Tensor, col0: int, cw_valid: int, lse: Tensor, loss2: Tensor, gout: Tensor, dl: Tensor, V: int,
                 ignore_index: int = -100) -> Tensor:
    rows, cw = dl.shape
    assert logits.dtype == torch.float32 and logits.stride(1) == 1 and dl.stride(1) == 1 and logits.shape[0] == rows
    call("fk_ce_chunk_bwd", logits.data_ptr(), logits.stride(0), targets.data_ptr(), col0, lse.data_ptr(), loss2.data_ptr(), gout.data_ptr(),
         dl.data_ptr(), dl.stride(0), rows, cw, cw_valid, V, ignore_index, fk_dtype(dl), _stream())
    return dl


def head_ce_fwd(h: Tensor, w: Tensor, bias: Optional[Tensor], targets: Tensor, V: int, ignore_index: int = -100):
    """(loss2, row_lse) of mean CE(h @ w[:V]^T (+ bias), targets) without the [rows, V] logits (fk_head_ce_fwd + fk_ce_chunk_finish)."""
    assert h.dim() == 2 and w.dim() == 2 and h.shape[1] == w.shape[1] and h.dtype == w.dtype and h.stride(1) == 1 and w.stride(1) == 1
    assert w.shape[0] >= V and targets.dtype == torch.int64 and targets.is_contiguous() and targets.numel() == h.shape[0]
    rows, Kd = h.shape
    if bias is not None:
        assert bias.dtype == h.dtype and bias.is_contiguous() and bias.numel() >= V
    st = CeChunkState(rows, h.device)
    ws, nb = _ws(lib().fk_head_ce_workspace_bytes(rows, V), h.device)
    with _timed(f"head_ce_fwd:{rows}x{V}x{Kd}"):
        call("fk_head_ce_fwd", h.data_ptr(), h.stride(0), w.data_ptr(), w.stride(0), w.shape[0], _ptr(bias), targets.data_ptr(),
             st.m.data_ptr(), st.s.data_ptr(), st.t.data_ptr(), rows, V, Kd, fk_dtype(h), _ptr(ws), nb, _stream())
    st.first = False
    return ce_chunk_finish(st, targets, V, ignore_index)


def head_ce_bwd(h: Tensor, w: Tensor, bias: Optional[Tensor], targets: Tensor, lse: Tensor, loss2: Tensor, gout: Tensor, V: int,
                vpad: int, rows_pad: int, want_bias: bool, ignore_index: int = -100):
    """dlT [vpad, rows_pad] (transposed d-logits, zero padded) and, for a head with bias, its gradient [V] (fk_head_ce_bwd)."""
    rows, Kd = h.shape
    dlT = torch.empty((vpad, rows_pad), dtype=h.dtype, device=h.device)
    dbpart = torch.empty((2 * ((rows + 127) // 128), vpad), dtype=torch.float32, device=h.device) if want_bias else None
    with _timed(f"head_ce_bwd:{rows}x{V}x{Kd}"):
        call("fk_head_ce_bwd", h.data_ptr(), h.stride(0), w.data_ptr(), w.stride(0), w.shape[0], _ptr(bias), targets.data_ptr(),
             lse.data_ptr(), loss2.data_ptr(), gout.data_ptr(), dlT.data_ptr(), dlT.stride(0), rows_pad, vpad, _ptr(dbpart), rows, V, Kd,
             ignore_index, fk_dtype(h), _stream())
    db = colsum(dbpart)[:V] if want_bias else None
    return dlT, db


def transpose2d(src: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """out[c, r] = src[r, c]; a given `out` may be larger (its padding is left as it is)."""
    assert src.dim() == 2 and src.stride(1) == 1
    rows, cols = src.shape
    if out is None:
        out = torch.empty((cols, rows), dtype=src.dtype, device=src.device)
    assert out.dtype == src.dtype and out.stride(1) == 1 and out.shape[0] >= cols and out.shape[1] >= rows
    call("fk_transpose2d", src.data_ptr(), src.stride(0), out.data_ptr(), out.stride(0), rows, cols, fk_dtype(src), _stream())
    return out


# ------------------------------------------------------------------------------------------- GPT embedding
def gpt_embed_fwd(idx: Tensor, prefix: Optional[Tensor], wte: Tensor, wpe: Tensor, dtype: torch.dtype) -> Tensor:
    B, t_words = idx.shape
    t_ctx = 0 if prefix is None else prefix.shape[1]
    dim = wte.shape[1]
    assert idx.dtype == torch.int64 and idx.is_contiguous() and wte.dtype == torch.float32 and wpe.dtype == torch.float32
    assert wte.is_contiguous() and wpe.is_contiguous() and wpe.shape[0] >= t_ctx + t_words
    if prefix is not None:
        assert prefix.is_contiguous() and prefix.dtype == dtype and prefix.shape == (B, t_ctx, dim)
    out = torch.empty((B, t_ctx + t_words, dim), dtype=dtype, device=idx.device)
    call("fk_gpt_embed_fwd", idx.data_ptr(), _ptr(prefix), wte.data_ptr(), wpe.data_ptr(), out.data_ptr(), B, t_ctx,
         t_words, dim, wte.shape[0], fk_dtype(dtype), _stream())
    return out


def gpt_embed_bwd_wte(idx: Tensor, dout: Tensor, dwte: Tensor, t_ctx: int) -> None:
    B, t_words = idx.shape
    assert dout.is_contiguous() and dwte.dtype == torch.float32 and dwte.is_contiguous()
    call("fk_gpt_embed_bwd_wte", idx.data_ptr(), dout.data_ptr(), dwte.data_ptr(), B, t_ctx, t_words, dwte.shape[1],
         dwte.shape[0], fk_dtype(dout), _stream())


# ------------------------------------------------------------------------------------------- optimizer
def adamw_step_(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, beta1: float = 0.9,
                beta2: float = 0.999, eps: float = 1e-8, weight_decay: float = 1e-2, clip: float = 0.0,
                grad_scale: float = 1.0, zero_grad: bool = False) -> None:
    for t in (p, g, m, v):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == p.numel()
    call("fk_adamw_step", p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, beta1, beta2, eps,
         weight_decay, step, clip, grad_scale, int(zero_grad), _stream())
