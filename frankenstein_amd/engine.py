"""Autograd glue between the nn.Module surface (models/) and the HIP kernels (kernels.py).

Design (MI355X-first, not a translation of the reference's op graph):
  * master parameters stay fp32; kernels consume compute-dtype *shadows* (bf16 or fp32) that are laid
    out for the GEMMs: q/k/v (and w1/w3) concatenated into one [sum N, K] operand, plus a transposed
    copy [K, sum N] so every activation GEMM — forward and dgrad — is the same K-contiguous NT kernel.
  * the unit of autograd is the pre-norm residual branch ("sub-block"):  y = x + f(LN(x)).
    One torch.autograd.Function per branch keeps exactly the tensors the hand-written backward
    needs and fuses the residual-gradient add into the LayerNorm backward kernel.
  * no torch math op is executed on the hot path: only allocation, views and kernel launches.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import math
import os
import weakref

import torch

from . import kernels as K
from .kernels import Mask, NO_MASK

Tensor = torch.Tensor

_COMPUTE_DTYPE = torch.bfloat16
_EPOCH = 0


def set_compute_dtype(dtype) -> None:
    """'bf16' (throughput mode: bf16 operands, fp32 accumulate) or 'fp32' (exact-fp32 MFMA parity mode)."""
    global _COMPUTE_DTYPE
    if isinstance(dtype, str):
        dtype = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp32": torch.float32, "float32": torch.float32}[dtype]
    assert dtype in (torch.bfloat16, torch.float32)
    _COMPUTE_DTYPE = dtype


def compute_dtype() -> torch.dtype:
    return _COMPUTE_DTYPE


def bump_weight_epoch() -> None:
    """Called by the optimizer (which updates masters through raw pointers) to invalidate all shadows."""
    global _EPOCH
    _EPOCH += 1


# --------------------------------------------------------------------------------------------- weight-gradient stream
# dW = dY^T X GEMMs do not feed the rest of the backward: with a flat gradient arena (train_utils.ParamArena) they are
# accumulated straight into the arena (no temporary, no autograd add).  Optionally (FusedAdamW(overlap_wgrad=True) or
# FK_WGRAD_STREAM=1) they run on a SECOND HIP stream beside the dX chain; the optimizer joins that stream before the update and
# DP buckets are all-reduced from it (GradSync).  OFF by default: the large-tile GEMMs are one 128-KiB-LDS block per CU, and two
# such grids from two streams split the CUs unevenly — measured 65 ms steps turning into 80-160 ms ones at random.  (The run-to-run
# differences round 2 saw in this mode were not the stream's doing: compiler-packed fp32 in the RoPE epilogues beside ANY co-resident
# bf16 GEMM, fixed in the build flags — DESIGN.md 5.4, tests/test_coresidency_gpu.py.)
_WGRAD_STREAM = None


def enable_wgrad_stream(on: bool = True):
    global _WGRAD_STREAM
    _WGRAD_STREAM = torch.cuda.Stream() if on else None
    return _WGRAD_STREAM


def wgrad_stream():
    return _WGRAD_STREAM


def _arena_grad(p: Tensor):
    g = p.grad
    return g if (g is not None and getattr(p, "_fk_arena", False) and g.is_contiguous()) else None


def wgrad(a: Tensor, b: Tensor, params: Sequence[Tensor], swiglu_interleaved: bool = False):
    """Gradients of the row-concatenated weights `params` = a^T b ([sum N, K], fp32).  Returns a list aligned with params:
    tensors (caller hands them to autograd) or Nones when the result was accumulated directly into the arena (on the
    weight-gradient stream when one is enabled, else on the current stream)."""
    side = _WGRAD_STREAM
    grads = [_arena_grad(p) for p in params]
    if any(g is None for g in grads):
        dw = K.gemm_tn(a, b)
        if swiglu_interleaved:
            return list(_deinterleave_rows(dw, params[0].shape[0]))
        return _split_rows(dw, params)
    if side is not None:
        side.wait_stream(torch.cuda.current_stream())
        a.record_stream(side)
        b.record_stream(side)
    with torch.cuda.stream(side if side is not None else torch.cuda.current_stream()):
        Kd = b.shape[1]
        adjacent = all(grads[i + 1].data_ptr() == grads[i].data_ptr() + grads[i].numel() * 4 for i in range(len(grads) - 1))
        if swiglu_interleaved:
            dw = K.gemm_tn(a, b)                                   # [2H, K] interleaved rows
            H = params[0].shape[0]
            v = dw.view(H // 4, 8 * Kd)
            K.add2d_(grads[0].view(H // 4, 4 * Kd), v[:, :4 * Kd])
            K.add2d_(grads[1].view(H // 4, 4 * Kd), v[:, 4 * Kd:])
        elif adjacent:
            n = sum(p.shape[0] for p in params)
            out = torch.as_strided(grads[0], (n, Kd), (Kd, 1))
            K.gemm_tn(a, b, out=out, accumulate=True)
        else:
            dw = K.gemm_tn(a, b)
            for g, part in zip(grads, _split_rows(dw, params)):
                K.add2d_(g.view(part.shape), part)
    # autograd still runs the parameters' post-accumulate-grad hooks (GradSync) after this Function returns, once per
    # backward and after the last use of a shared weight, so bucket readiness needs no extra signalling here
    return [None] * len(params)


def norm_bwd(dh: Tensor, x2: Tensor, ln_w: Tensor, ln_b: Optional[Tensor], mean: Tensor, rstd: Tensor, dres: Optional[Tensor] = None,
             kind: int = K.NORM_LAYER):
    """K.norm_bwd whose dgamma / dbeta are accumulated straight into the gradient arena when the parameters live there
    (returns Nones for them: no temporary, no autograd add); plain tensors otherwise."""
    gw = _arena_grad(ln_w)
    gb = _arena_grad(ln_b) if ln_b is not None else None
    if gw is not None and (ln_b is None or gb is not None):
        dx, _, _ = K.norm_bwd(dh, x2, ln_w.detach(), mean, rstd, dres=dres, kind=kind, want_beta=ln_b is not None,
                              dgamma=gw.view(-1), dbeta=None if gb is None else gb.view(-1), accumulate=True)
        return dx, None, None
    return K.norm_bwd(dh, x2, ln_w.detach(), mean, rstd, dres=dres, kind=kind, want_beta=ln_b is not None)


# --------------------------------------------------------------------------------------------- shadows
class _Shadow:
    """One compute-dtype weight copy.  `jobs` are the fk_cast_pack_rows calls that fill `tensor` from the masters
    (src, dst view, transpose, rblk, rstride, roff); `params` are weak references (a shadow dies with its model)."""
    __slots__ = ("stamp", "tensor", "jobs", "params", "dtype", "ptrs", "repack")

    def __init__(self):
        self.stamp, self.tensor, self.jobs, self.params, self.dtype, self.ptrs = None, None, [], (), None, ()
        self.repack = None       # shadows whose layout is not a cast_pack job (convolution weights): callable that refills `tensor` in place

    def alive(self) -> bool:
        """masters still exist and still live where the pack jobs read them (ParamArena moves parameter storage)"""
        return all(r() is not None and r().data_ptr() == q for r, q in zip(self.params, self.ptrs))

    def current_stamp(self):
        return (_EPOCH, tuple(r()._version for r in self.params))


_SHADOWS: dict = {}
_REFRESH = {"sig": None, "table": None, "njobs": 0, "chunks": 0}


def _run_jobs(ent: "_Shadow") -> None:
    for src, dst, tr, rblk, rstride, roff in ent.jobs:
        if rblk:
            K.cast_pack_rows(src, dst, tr, rblk, rstride, roff)
        else:
            K.cast_pack(src, dst, transpose=tr)


def _shadow_entry(key, params) -> "_Shadow":
    ent = _SHADOWS.get(key)
    if ent is not None and not ent.alive():       # the address was reused by another model's parameter
        ent = None
    if ent is None:
        ent = _SHADOWS[key] = _Shadow()
        # a view (space_embedding.view(C, d), ...) is tracked through its base: the view object dies with the call
        owners = [p._base if p._base is not None else p for p in params]
        ent.params = tuple(weakref.ref(o) for o in owners)
        ent.ptrs = tuple(o.data_ptr() for o in owners)
        ent.dtype = _COMPUTE_DTYPE
    return ent


def shadow(params: Sequence[Tensor], transpose: bool = False, pad_k: int = 0, pad_n: int = 0) -> Tensor:
    """Compute-dtype copy of cat(params, dim=0) ([sum N, K]); transposed -> [K, sum N]; pad_k / pad_n zero-pad
    the K / N extents (16-byte GEMM operand alignment).  1-D params (biases) are concatenated as vectors.
    Re-packed only when a master changed (tensor version counter) or the optimizer bumped the epoch; the optimizer
    re-packs every shadow of its parameters in one launch (refresh_shadows), so in a training loop this is a lookup."""
    dt = _COMPUTE_DTYPE
    key = (tuple((p.data_ptr(), tuple(p.shape)) for p in params), transpose, pad_k, pad_n, dt)
    ent = _shadow_entry(key, params)
    stamp = ent.current_stamp()
    if ent.stamp == stamp:
        return ent.tensor
    if ent.tensor is not None and ent.jobs:        # same masters, new values: re-pack in place
        _run_jobs(ent)
        ent.stamp = stamp
        return ent.tensor
    p0 = params[0]
    for p in params:
        assert p.dtype == torch.float32, "master parameters must be float32 (compute dtype is set with set_compute_dtype)"
    jobs = []
    if p0.dim() == 1:
        n = sum(p.numel() for p in params)
        if dt == torch.float32 and len(params) == 1:
            out = p0.detach()
        else:
            out = torch.empty(n, dtype=dt, device=p0.device)
            off = 0
            for p in params:
                jobs.append((p.detach().view(1, -1), out[off:off + p.numel()].view(1, -1), False, 0, 0, 0))
                off += p.numel()
    else:
        Kd = p0.shape[1]
        kp = max(Kd, pad_k)
        n = sum(p.shape[0] for p in params)
        npad = max(n, pad_n)
        if dt == torch.float32 and len(params) == 1 and not transpose and kp == Kd and npad == n and p0.is_contiguous():
            out = p0.detach()
        else:
            shape = (kp, npad) if transpose else (npad, kp)
            out = (torch.zeros if (kp != Kd or npad != n) else torch.empty)(shape, dtype=dt, device=p0.device)
            off = 0
            for p in params:
                src = p.detach()
                assert src.dim() == 2 and src.shape[1] == Kd and src.is_contiguous()
                dst = out[:Kd, off:off + src.shape[0]] if transpose else out[off:off + src.shape[0], :Kd]
                jobs.append((src, dst, transpose, 0, 0, 0))
                off += src.shape[0]
    ent.jobs, ent.tensor = jobs, out
    _run_jobs(ent)
    ent.stamp = stamp
    return out


def shadow_swiglu(w1: Tensor, w3: Tensor, transpose: bool = False) -> Tensor:
    """Interleaved SwiGLU up-projection shadow: per 4 hidden units, 4 rows of w1 then 4 rows of w3 ([2H, K]; transposed
    [K, 2H]) — the layout the fused GEMM epilogues (fk_gemm_nt_swiglu / fk_gemm_nt_dswiglu) expect."""
    dt = _COMPUTE_DTYPE
    key = (("swiglu", w1.data_ptr(), w3.data_ptr(), tuple(w1.shape)), transpose, dt)
    ent = _shadow_entry(key, (w1, w3))
    stamp = ent.current_stamp()
    if ent.stamp == stamp:
        return ent.tensor
    if ent.tensor is None:
        H, Kd = w1.shape
        assert w3.shape == (H, Kd) and H % 8 == 0 and w1.dtype == torch.float32 and w1.is_contiguous() and w3.is_contiguous()
        out = torch.empty((Kd, 2 * H) if transpose else (2 * H, Kd), dtype=dt, device=w1.device)
        ent.jobs = [(w1.detach(), out, transpose, 4, 8, 0), (w3.detach(), out, transpose, 4, 8, 4)]
        ent.tensor = out
    _run_jobs(ent)
    ent.stamp = stamp
    return ent.tensor


def refresh_shadows(params: Sequence[Tensor]) -> None:
    """Re-pack, in ONE launch (fk_cast_pack_multi), every shadow built so far whose masters all belong to `params` — called by
    the optimizer right after it updated the masters (one launch per step instead of one or two per weight)."""
    ids = {id(p) for p in params}
    for k in [k for k, e in _SHADOWS.items() if not e.alive()]:
        del _SHADOWS[k]
    # shadows with their own in-place re-pack (convolution weights): refreshed here too, so that a captured graph, which keeps
    # reading the shadow's storage, sees every optimizer step (they used to be rebuilt lazily into NEW storage by the eager forward only)
    for e in _SHADOWS.values():
        if e.repack is not None and e.tensor is not None and e.dtype == _COMPUTE_DTYPE and all(id(r()) in ids for r in e.params):
            e.repack()
            e.stamp = e.current_stamp()
    ents = [e for e in _SHADOWS.values()
            if e.jobs and e.dtype == _COMPUTE_DTYPE and e.tensor.is_cuda and all(id(r()) in ids for r in e.params)]
    if not ents:
        return
    sig = tuple(id(e) for e in ents)
    if _REFRESH["sig"] != sig:
        import numpy as np
        rec = np.dtype([("src", "<u8"), ("dst", "<u8"), ("lds", "<i8"), ("ldd", "<i8"), ("rows", "<i4"), ("cols", "<i4"),
                        ("transpose", "<i4"), ("rblk", "<i4"), ("rstride", "<i4"), ("roff", "<i4"), ("chunk_begin", "<i8")])
        assert rec.itemsize == 64
        rows, chunks = [], 0
        for e in ents:
            for src, dst, tr, rblk, rstride, roff in e.jobs:
                r, c = src.shape
                rows.append((src.data_ptr(), dst.data_ptr(), src.stride(0), dst.stride(0), r, c, int(tr), rblk, rstride, roff, chunks))
                chunks += ((r + 31) // 32) * ((c + 31) // 32) if tr else (r * c + 1023) // 1024
        tab = torch.from_numpy(np.array(rows, dtype=rec).view(np.uint8).copy()).to(ents[0].tensor.device)
        _REFRESH.update(sig=sig, table=tab, njobs=len(rows), chunks=chunks, ents=ents)
    K.cast_pack_multi(_REFRESH["table"], _REFRESH["njobs"], _REFRESH["chunks"], _COMPUTE_DTYPE)
    for e in ents:
        e.stamp = e.current_stamp()


def _deinterleave_rows(t: Tensor, H: int):
    """[2H, K] fp32 in the interleaved row order -> (rows of w1, rows of w3), each [H, K] contiguous."""
    Kd = t.shape[1]
    a = torch.empty((H, Kd), dtype=t.dtype, device=t.device)
    b = torch.empty((H, Kd), dtype=t.dtype, device=t.device)
    v = t.view(H // 4, 8 * Kd)
    K.copy2d(v[:, :4 * Kd], a.view(H // 4, 4 * Kd))
    K.copy2d(v[:, 4 * Kd:], b.view(H // 4, 4 * Kd))
    return a, b


def _split_rows(t: Tensor, params: Sequence[Tensor]) -> List[Tensor]:
    out, off = [], 0
    for p in params:
        n = p.shape[0]
        out.append(t[off:off + n].view(p.shape))
        off += n
    return out


def to_compute(x: Tensor) -> Tensor:
    """fp32 / bf16 activation -> compute dtype (kernel cast, no autograd: inputs only)."""
    if x.dtype == _COMPUTE_DTYPE:
        return x.contiguous()
    return K.cast(x.contiguous(), _COMPUTE_DTYPE)


# --------------------------------------------------------------------------------------------- rope spec
_ROPE_PAIRS: dict = {}


class Rope:
    """fp32 (cos, sin) table [Tc, D/2, 2] (or per-sample [B, Tc, D/2, 2]); the LAST T rows are used
    (models/brainformer.py:80,82)."""
    __slots__ = ("table", "_src")

    def __init__(self, cache: Tensor):
        t = torch.view_as_real(cache) if cache.is_complex() else cache
        assert t.dtype == torch.float32 and t.shape[-1] == 2
        self.table = t.contiguous()
        self._src = cache

    def pos_off(self, T: int) -> int:
        return self.table.shape[-3] - T

    def pair(self, c: float):
        """(table, table * c) as the two halves of ONE allocation: the second is what fk_gemm_nt_rope rotates the query columns
        with, so that Q leaves the projection multiplied by c = softmax_scale * log2(e) (FK_ATTN_Q_PRESCALED).  Cached per cache
        tensor (the model hands the same tensor to every layer); the entry dies with the tensor."""
        src, key = self._src, id(self._src)
        ent = _ROPE_PAIRS.get(key)
        stamp = (src._version, float(c), src.data_ptr())
        if ent is not None and ent[0]() is src and ent[1] == stamp:
            return ent[2][0], ent[2][1]
        both = torch.stack([self.table, self.table * float(c)]).contiguous()
        _ROPE_PAIRS[key] = (weakref.ref(src, lambda _r, k=key: _ROPE_PAIRS.pop(k, None)), stamp, both)
        return both[0], both[1]


_IDENT_PAIRS: dict = {}


def identity_pair(D: int, c: float, device):
    """(table, q_table) for attention blocks WITHOUT RoPE (SimpleMAE's / MAE's decoder, GPT-2 at head_dim 64): eight identical rows of
    (cos, sin) = (1, 0) and (c, 0).  Through fk_gemm_nt_rope's epilogue the key columns stay bit-exact (x * 1 - y * 0) and the query
    columns leave the projection as c * q with one rounding, i.e. pre-scaled for the lean attention kernels (FK_ATTN_Q_PRESCALED) —
    without it those blocks ran the generic kernels (SimpleMAE decoder at B = 32: 208 + 157 + 160 us per layer for 18 GFLOP)."""
    key = (D, float(c), str(device))
    ent = _IDENT_PAIRS.get(key)
    if ent is None:
        one = torch.zeros((8, D // 2, 2), dtype=torch.float32)
        one[..., 0] = 1.0
        ent = _IDENT_PAIRS[key] = torch.stack([one, one * float(c)]).contiguous().to(device)
    return ent[0], ent[1]


def _attn_prescale(D: int) -> bool:
    """Queries pre-scaled by scale * log2(e) in the projection epilogue + the lean attention kernels: bf16, head_dim 64."""
    return _COMPUTE_DTYPE == torch.bfloat16 and D == 64 and os.environ.get("FK_ATTN_NO_PRESCALE") is None


# --------------------------------------------------------------------------------------------- dropout
# nn.Dropout / SDPA dropout_p of the GPT-2 decoder in training mode (models/gpt2_model.py:40,64,75,85,91,129).  No mask tensor exists:
# every application is (p, seed words, site) and the kernels regenerate keep / drop from the element index (include/franken_hip.h,
# fk_dropout).  The MASTER seed words live on the device — [torch.initial_seed(), a step counter advanced once per forward by
# dropout_begin()] — so a captured training step draws new masks on every replay; `site` numbers the applications within one forward.
# Every forward works on its own SNAPSHOT of the two words (one 8-byte device copy, captured in graphs too): a backward that runs after
# another training-mode forward (two forwards, then (l1 + l2).backward(); a recompute) regenerates the masks ITS forward drew, like
# torch's dropout.  No words tensor is ever replaced: a captured graph holds raw pointers to them, so a new seed is written in place.
_DROP_WORDS: dict = {}
_DROP_CUR: dict = {}
_DROP_SITE = [0]


def dropout_words(device) -> Tensor:
    """The master [seed, step] words of `device` (int32, device memory; the same tensor for the life of the process)."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    seed = torch.initial_seed() & 0x7FFFFFFF
    ent = _DROP_WORDS.get(device)
    if ent is None:
        ent = _DROP_WORDS[device] = [seed, torch.tensor([seed, 0], dtype=torch.int32, device=device)]
    elif ent[0] != seed and not torch.cuda.is_current_stream_capturing():
        ent[0] = seed
        ent[1].copy_(torch.tensor([seed, 0], dtype=torch.int32))        # a new torch.manual_seed: restart the stream, in place
    return ent[1]


def dropout_begin(device) -> None:
    """Start of a training forward with dropout > 0: the next step's masks (master step word + 1 on the device), this forward's
    snapshot of the words, sites numbered from 0."""
    words = dropout_words(device)
    words[1:].add_(1)
    _DROP_CUR[words.device] = words.clone()
    _DROP_SITE[0] = 0


def dropout_snapshot(device) -> Tensor:
    """The words the current forward draws with (dropout_begin's snapshot; the master words before any forward began)."""
    words = dropout_words(device)
    return _DROP_CUR.get(words.device, words)


def dropout_site() -> int:
    _DROP_SITE[0] += 1
    return _DROP_SITE[0] - 1


def drop_spec(p: float, device, n_sites: int):
    """(p, seed words, site, ...) for one module call, or None when p == 0."""
    if not p:
        return None
    assert 0.0 < p < 1.0, f"dropout p = {p}"
    return (float(p), dropout_snapshot(device)) + tuple(dropout_site() for _ in range(n_sites))


class Dropout(torch.autograd.Function):
    """y = nn.Dropout(p)(x) in training mode (embedding dropout, models/gpt2_model.py:129,190)."""

    @staticmethod
    def forward(ctx, x, drop):
        ctx.drop = drop
        return K.dropout(x.contiguous(), drop[0], drop[1], drop[2])

    @staticmethod
    def backward(ctx, dy):
        d = ctx.drop
        return K.dropout(dy.contiguous(), d[0], d[1], d[2]), None


# --------------------------------------------------------------------------------------------- functions
class AttnBranch(torch.autograd.Function):
    """y = [x +] proj(SDPA(rope(q), rope(k), v)) with q,k,v = Linear([LN](x));  one pre-norm attention branch.

    Reference: Block.forward attention half (models/brainformer.py:243 with :147-173) and
    gpt2 Block (models/gpt2_model.py:104 with :52-76)."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, pw, pb, qkv_b, spec, *qkv_w):
        H, D, mask, rope, residual, eps, nkind = spec[:7]
        drop = spec[7] if len(spec) > 7 else None       # (p, seed words, site of the attention dropout, site of the residual dropout)
        B, N, d = x.shape
        M = B * N
        x2 = x.view(M, d)
        has_ln = ln_w is not None
        if has_ln:
            h, mean, rstd = K.norm_fwd(x2, ln_w.detach(), None if ln_b is None else ln_b.detach(), eps, nkind)
        else:
            h, mean, rstd = x2, None, None
        HD = H * D
        bq = None if qkv_b is None else shadow([qkv_b])
        prescale = False
        if rope is not None and D % 8 == 0 and (3 * HD) % 8 == 0:     # RoPE fused into the projection epilogue
            prescale = _attn_prescale(D) and mask.kind != K.MASK_DENSE and drop is None     # dense masks / dropout: the generic kernels
            if prescale:
                tab, qtab = rope.pair((1.0 / math.sqrt(D)) * 1.4426950408889634)
                qkv = K.gemm_nt_rope(h, shadow(qkv_w), bq, tab, N, rope.pos_off(N), D, 2 * HD, q_cols=HD, q_table=qtab)
            else:
                qkv = K.gemm_nt_rope(h, shadow(qkv_w), bq, rope.table, N, rope.pos_off(N), D, 2 * HD)
            qkv3 = qkv.view(B, N, 3 * HD)
        elif (rope is None and N >= 64 and M % 8 == 0 and (3 * HD) % 8 == 0 and _attn_prescale(D) and mask.kind != K.MASK_DENSE and drop is None
              and os.environ.get("FK_ATTN_NO_IDENT_PRESCALE") is None):
            # no RoPE, but the lean kernels want pre-scaled queries: the same epilogue with an identity "rotation" (identity_pair)
            tab, qtab = identity_pair(D, (1.0 / math.sqrt(D)) * 1.4426950408889634, x.device)
            qkv = K.gemm_nt_rope(h, shadow(qkv_w), bq, tab, 8, 0, D, 2 * HD, q_cols=HD, q_table=qtab)
            qkv3 = qkv.view(B, N, 3 * HD)
            prescale = True
        else:
            qkv = K.gemm_nt(h, shadow(qkv_w), bias=bq)
            qkv3 = qkv.view(B, N, 3 * HD)
            if rope is not None:
                K.rope_(qkv3, 2 * H, D, rope.table, rope.pos_off(N))
        q, k, v = (qkv3[..., i * HD:(i + 1) * HD].unflatten(-1, (H, D)) for i in range(3))
        o, lse = K.attn_fwd(q, k, v, mask, q_prescaled=prescale, dropout=None if drop is None else drop[:3])
        if drop is None:
            y = K.gemm_nt(o.view(M, HD), shadow([pw]), bias=None if pb is None else shadow([pb]),
                          residual=x2 if residual else None)
        else:                                           # x + resid_dropout(c_proj(y))  (models/gpt2_model.py:75,104)
            y = K.gemm_nt(o.view(M, HD), shadow([pw]), bias=None if pb is None else shadow([pb]))
            K.dropout(y, drop[0], drop[1], drop[3], residual=x2 if residual else None, out=y)
        ctx.spec, ctx.has_ln, ctx.nw, ctx.prescale = spec, has_ln, len(qkv_w), prescale
        ctx.flags = (ln_b is not None, pb is not None, qkv_b is not None)
        ctx.ln_b = ln_b
        ctx.save_for_backward(x, ln_w, pw, *qkv_w, h if has_ln else None, mean, rstd, qkv, o, lse)
        return y.view(B, N, -1)

    @staticmethod
    def backward(ctx, dy):
        H, D, mask, rope, residual, eps, nkind = ctx.spec[:7]
        drop = ctx.spec[7] if len(ctx.spec) > 7 else None
        sv = ctx.saved_tensors
        x, ln_w, pw = sv[0], sv[1], sv[2]
        qkv_w = sv[3:3 + ctx.nw]
        h, mean, rstd, qkv, o, lse = sv[3 + ctx.nw:]
        has_lnb, has_pb, has_qb = ctx.flags
        B, N, d = x.shape
        M, HD = B * N, H * D
        x2 = x.view(M, d)
        if h is None:
            h = x2
        dy2 = dy.contiguous().view(M, -1)
        dyp = dy2 if drop is None else K.dropout(dy2, drop[0], drop[1], drop[3])        # the gradient behind the residual dropout
        do = K.gemm_nt(dyp, shadow([pw], transpose=True))
        (dpw,) = wgrad(dyp, o.view(M, HD), [pw])
        dpb = K.colsum(dyp) if has_pb else None
        dqkv = torch.empty_like(qkv)
        qkv3, dqkv3 = qkv.view(B, N, 3 * HD), dqkv.view(B, N, 3 * HD)
        q, k, v = (qkv3[..., i * HD:(i + 1) * HD].unflatten(-1, (H, D)) for i in range(3))
        dq, dk, dv = (dqkv3[..., i * HD:(i + 1) * HD].unflatten(-1, (H, D)) for i in range(3))
        if rope is not None and D % 4 == 0:       # inverse RoPE fused into the dQ / dK stores
            K.attn_bwd(q, k, v, o, do.view(B, N, H, D), lse, dq, dk, dv, mask, rope_table=rope.table, rope_off=rope.pos_off(N),
                       q_prescaled=ctx.prescale, dropout=None if drop is None else drop[:3])
        else:
            K.attn_bwd(q, k, v, o, do.view(B, N, H, D), lse, dq, dk, dv, mask, q_prescaled=ctx.prescale, dropout=None if drop is None else drop[:3])
            if rope is not None:
                K.rope_(dqkv3, 2 * H, D, rope.table, rope.pos_off(N), conj=True)
        dh = K.gemm_nt(dqkv, shadow(qkv_w, transpose=True))
        dws = wgrad(dqkv, h, list(qkv_w))
        dqb = K.colsum(dqkv) if has_qb else None
        if ctx.has_ln:
            dx, dg, db = norm_bwd(dh, x2, ln_w, ctx.ln_b, mean, rstd, dres=dy2 if residual else None, kind=nkind)
        else:
            dx, dg, db = (K.add(dh, dy2) if residual else dh), None, None
        return (dx.view(B, N, d), dg, db, dpw, dpb, dqb, None, *dws)


_ZERO_IDX: dict = {}


def _replicate(x: Tensor, S: int) -> Tensor:
    """x [B, ...] -> [B * S, ...] with every sample repeated S times (one gather launch; queries / outputs of split-key attention)"""
    B = x.shape[0]
    key = (B, S, x.device)
    idx = _ZERO_IDX.get(key)
    if idx is None:
        idx = _ZERO_IDX[key] = torch.zeros((B, S), dtype=torch.int64, device=x.device)
    return K.gather_rows(x.reshape(B, 1, -1), idx).view(B * S, *x.shape[1:])


def _key_splits(T: int, Nc: int, mask: Mask) -> int:
    """Split-key attention for a few queries against a long unmasked context (the perceiver's read-out): S key ranges folded into the
    batch dimension give S times the workgroups, fk_attn_combine merges the partial results.  OPT-IN (FK_ATTN_SPLIT=1): measured on cfg2
    (32 queries x 6144 keys, B = 32) the forward goes 137 -> 100 us and the query gradient 127 -> 82 us per layer — both sit on the
    60 us it takes to stream K and V (302 MB) once — and the five replicate / combine launches per layer give the 0.16 ms per step back
    (profiles/r03_split_perceiver.md): no net gain, so the single-pass form stays the default."""
    if os.environ.get("FK_ATTN_SPLIT") != "1" or mask.kind != K.MASK_NONE or T > 128 or Nc < 1024:
        return 1
    for S in (8, 4, 2):
        if Nc % (S * 128) == 0:
            return S
    return 1


class CrossAttnBranch(torch.autograd.Function):
    """y = x + proj(SDPA(q = Wq LN(x), k = Wk ctx, v = Wv ctx))  (models/brainformer.py:262 with :198-219)."""

    @staticmethod
    def forward(ctx, x, context, ln_w, ln_b, qw, kw, vw, pw, spec):
        H, D, mask, eps = spec
        B, T, d = x.shape
        Nc = context.shape[1]
        HD = H * D
        x2, c2 = x.view(B * T, d), context.view(B * Nc, d)
        h, mean, rstd = K.norm_fwd(x2, ln_w.detach(), ln_b.detach(), eps)
        ctx.ln_b = ln_b
        q = K.gemm_nt(h, shadow([qw]))
        kv = K.gemm_nt(c2, shadow([kw, vw]))
        S = _key_splits(T, Nc, mask)
        if S > 1:
            kvs = kv.view(B * S, Nc // S, 2 * HD)
            o_s, lse_s = K.attn_fwd(_replicate(q.view(B, T, H, D), S), kvs[..., :HD].unflatten(-1, (H, D)), kvs[..., HD:].unflatten(-1, (H, D)), mask)
            o, lse = K.attn_combine(o_s.view(B, S, T, H, D), lse_s.view(B, S, H, T))
        else:
            kv3 = kv.view(B, Nc, 2 * HD)
            o, lse = K.attn_fwd(q.view(B, T, H, D), kv3[..., :HD].unflatten(-1, (H, D)), kv3[..., HD:].unflatten(-1, (H, D)), mask)
        y = K.gemm_nt(o.view(B * T, HD), shadow([pw]), residual=x2)
        ctx.spec, ctx.S = spec, S
        ctx.save_for_backward(x, context, ln_w, qw, kw, vw, pw, h, mean, rstd, q, kv, o, lse)
        return y.view(B, T, d)

    @staticmethod
    def backward(ctx, dy):
        H, D, mask, eps = ctx.spec
        x, context, ln_w, qw, kw, vw, pw, h, mean, rstd, q, kv, o, lse = ctx.saved_tensors
        B, T, d = x.shape
        Nc, HD, S = context.shape[1], H * D, ctx.S
        x2, c2 = x.view(B * T, d), context.view(B * Nc, d)
        dy2 = dy.contiguous().view(B * T, d)
        do = K.gemm_nt(dy2, shadow([pw], transpose=True))
        (dpw,) = wgrad(dy2, o.view(B * T, HD), [pw])
        dkv = torch.empty_like(kv)
        if S > 1:
            # the same key ranges as the forward, every range against the replicated queries / outputs and the GLOBAL row statistics: dK / dV
            # land in their rows of dkv, the S partial query gradients are summed by fk_attn_combine
            kvs, dkvs = kv.view(B * S, Nc // S, 2 * HD), dkv.view(B * S, Nc // S, 2 * HD)
            dq_s = torch.empty((B * S, T, H, D), dtype=q.dtype, device=q.device)
            K.attn_bwd(_replicate(q.view(B, T, H, D), S), kvs[..., :HD].unflatten(-1, (H, D)), kvs[..., HD:].unflatten(-1, (H, D)),
                       _replicate(o, S), _replicate(do.view(B, T, H, D), S), _replicate(lse, S), dq_s,
                       dkvs[..., :HD].unflatten(-1, (H, D)), dkvs[..., HD:].unflatten(-1, (H, D)), mask)
            dq = K.attn_combine(dq_s.view(B, S, T, H, D))[0].view(B * T, HD)
        else:
            dq = torch.empty_like(q)
            kv3, dkv3 = kv.view(B, Nc, 2 * HD), dkv.view(B, Nc, 2 * HD)
            K.attn_bwd(q.view(B, T, H, D), kv3[..., :HD].unflatten(-1, (H, D)), kv3[..., HD:].unflatten(-1, (H, D)), o,
                       do.view(B, T, H, D), lse, dq.view(B, T, H, D), dkv3[..., :HD].unflatten(-1, (H, D)),
                       dkv3[..., HD:].unflatten(-1, (H, D)), mask)
        dh = K.gemm_nt(dq, shadow([qw], transpose=True))
        (dqw,) = wgrad(dq, h, [qw])
        dctx = K.gemm_nt(dkv, shadow([kw, vw], transpose=True))
        dkw, dvw = wgrad(dkv, c2, [kw, vw])
        dx, dg, db = norm_bwd(dh, x2, ln_w, ctx.ln_b, mean, rstd, dres=dy2)
        return dx.view(B, T, d), dctx.view(B, Nc, d), dg, db, dqw, dkw, dvw, dpw, None


_MLP_BWD_FUSED = os.environ.get("FK_MLP_BWD_FUSED", "1") != "0"


class MlpBranch(torch.autograd.Function):
    """y = [x +] W2 act(W1 [LN](x)):  SwiGLU (w1,w3 -> silu*gate -> w2, models/brainformer.py:115-124,244) when
    ``gate_w`` is given, GELU-erf MLP with biases (models/gpt2_model.py:87-92,105) otherwise."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, up_w, up_b, gate_w, down_w, down_b, spec):
        residual, eps, nkind = spec[:3]
        drop = spec[3] if len(spec) > 3 else None       # (p, seed words, site): dropout on the down-projection's output
        shp = x.shape
        d = shp[-1]
        x2 = x.reshape(-1, d)
        has_ln = ln_w is not None
        if has_ln:
            h, mean, rstd = K.norm_fwd(x2, ln_w.detach(), None if ln_b is None else ln_b.detach(), eps, nkind)
        else:
            h, mean, rstd = x2, None, None
        fused = gate_w is not None and up_b is None and up_w.shape[0] % 8 == 0
        if fused:     # up-projection + SwiGLU in one kernel (interleaved hidden layout)
            a, g = K.gemm_nt_swiglu(h, shadow_swiglu(up_w, gate_w))
        else:
            ups = [up_w] if gate_w is None else [up_w, gate_w]
            a = K.gemm_nt(h, shadow(ups), bias=None if up_b is None else shadow([up_b]))
            g = K.gelu_fwd(a) if gate_w is None else K.swiglu_fwd(a)
        if drop is None:
            y = K.gemm_nt(g, shadow([down_w]), bias=None if down_b is None else shadow([down_b]),
                          residual=x2 if residual else None)
        else:                                           # x + dropout(c_proj(gelu(c_fc(x))))  (models/gpt2_model.py:87-92,105)
            y = K.gemm_nt(g, shadow([down_w]), bias=None if down_b is None else shadow([down_b]))
            K.dropout(y, drop[0], drop[1], drop[2], residual=x2 if residual else None, out=y)
        ctx.spec, ctx.has_ln, ctx.fused = spec, has_ln, fused
        ctx.flags = (ln_b is not None, up_b is not None, gate_w is not None, down_b is not None)
        ctx.ln_b = ln_b
        ctx.save_for_backward(x, ln_w, up_w, gate_w, down_w, h if has_ln else None, mean, rstd, a, g)
        return y.view(*shp[:-1], y.shape[-1])

    @staticmethod
    def backward(ctx, dy):
        residual, eps, nkind = ctx.spec[:3]
        drop = ctx.spec[3] if len(ctx.spec) > 3 else None
        x, ln_w, up_w, gate_w, down_w, h, mean, rstd, a, g = ctx.saved_tensors
        has_lnb, has_ub, gated, has_db = ctx.flags
        shp = x.shape
        d = shp[-1]
        x2 = x.reshape(-1, d)
        if h is None:
            h = x2
        dy2 = dy.contiguous().view(x2.shape[0], -1)
        ups = [up_w, gate_w] if gated else [up_w]
        dyd = dy2 if drop is None else K.dropout(dy2, drop[0], drop[1], drop[2])         # the gradient behind the dropout
        (ddown,) = wgrad(dyd, g, [down_w])
        ddb = K.colsum(dyd) if has_db else None
        if (ctx.fused and _MLP_BWD_FUSED and dyd.dtype == torch.bfloat16 and d == 384 and g.shape[1] % 32 == 0 and dyd.shape[0] >= 32768          # 128 tokens per workgroup: below ~256 workgroups the two tiled GEMMs fill the chip better
                and dyd.shape[0] * a.shape[1] * 2 < 2 ** 32):
            # both products of the data-gradient chain in ONE attention-shaped launch, dh13 handed over in registers: the same bits as the
            # two launches below, -0.3 ... -0.4 ms per cfg2 step; FK_MLP_BWD_FUSED=0 keeps the two launches
            da, dh = K.mlp_bwd_fused(dyd, shadow([down_w], transpose=True), a, shadow_swiglu(up_w, gate_w, transpose=True))
            dups = wgrad(da, h, [up_w, gate_w], swiglu_interleaved=True)
        elif ctx.fused:   # down-projection dgrad + SwiGLU backward in one kernel; dg is never materialised
            da = K.gemm_nt_dswiglu(dyd, shadow([down_w], transpose=True), a)
            dh = K.gemm_nt(da, shadow_swiglu(up_w, gate_w, transpose=True))
            dups = wgrad(da, h, [up_w, gate_w], swiglu_interleaved=True)
        else:
            dg_ = K.gemm_nt(dyd, shadow([down_w], transpose=True))
            da = K.swiglu_bwd(a, dg_) if gated else K.gelu_bwd(a, dg_)
            dh = K.gemm_nt(da, shadow(ups, transpose=True))
            dups = wgrad(da, h, ups)
        dub = K.colsum(da) if has_ub else None
        if ctx.has_ln:
            dx, dgam, dbet = norm_bwd(dh, x2, ln_w, ctx.ln_b, mean, rstd, dres=dy2 if residual else None, kind=nkind)
        else:
            dx, dgam, dbet = (K.add(dh, dy2) if residual else dh), None, None
        return (dx.view(shp), dgam, dbet, dups[0], dub, dups[1] if gated else None, ddown, ddb, None)


class NormLinear(torch.autograd.Function):
    """y = Linear([LN](x)) (+ bias);  heads (perceiver.ln_f -> to_motion / to_words, lm_head) and plain Linears."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, w, b, eps, out_fp32):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        has_ln = ln_w is not None
        if has_ln:
            h, mean, rstd = K.norm_fwd(x2, ln_w.detach(), None if ln_b is None else ln_b.detach(), eps)
        else:
            h, mean, rstd = x2, None, None
        y = K.gemm_nt(h, shadow([w]), bias=None if b is None else shadow([b]),
                      out_dtype=torch.float32 if out_fp32 else None)
        ctx.has_ln, ctx.flags = has_ln, (ln_b is not None, b is not None)
        ctx.ln_b = ln_b
        ctx.save_for_backward(x, ln_w, w, h if has_ln else None, mean, rstd)
        return y.view(*shp[:-1], y.shape[-1])

    @staticmethod
    def backward(ctx, dy):
        x, ln_w, w, h, mean, rstd = ctx.saved_tensors
        has_lnb, has_b = ctx.flags
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        if h is None:
            h = x2
        dy2 = dy.contiguous().view(x2.shape[0], -1)
        if dy2.dtype != x2.dtype:
            dy2 = K.cast(dy2, x2.dtype)
        nout = dy2.shape[1]
        vec = 8 if dy2.dtype == torch.bfloat16 else 4
        npad = (nout + vec - 1) // vec * vec
        if npad != nout:   # e.g. vocab 50257: zero-pad the reduction dim of the dgrad / the rows of the wgrad operand
            dyp = torch.zeros((dy2.shape[0], npad), dtype=dy2.dtype, device=dy2.device)
            K.copy2d(dy2, dyp[:, :nout])
        else:
            dyp = dy2
        dh = K.gemm_nt(dyp, shadow([w], transpose=True, pad_n=npad))
        dw = K.gemm_tn(dyp, h)[:nout]
        db = K.colsum(dy2) if has_b else None
        if ctx.has_ln:
            dx, dg, dbe = norm_bwd(dh, x2, ln_w, ctx.ln_b, mean, rstd)
        else:
            dx, dg, dbe = dh, None, None
        return dx.view(shp), dg, dbe, dw, db, None, None


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, eps, kind):
        y, mean, rstd = K.norm_fwd(x.contiguous(), w.detach(), None if b is None else b.detach(), eps, kind)
        ctx.kind, ctx.has_b, ctx.ln_b = kind, b is not None, b
        ctx.save_for_backward(x, w, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, mean, rstd = ctx.saved_tensors
        dx, dg, db = norm_bwd(dy.contiguous(), x.contiguous(), w, ctx.ln_b, mean, rstd, kind=ctx.kind)
        return dx, dg, db, None, None


class PatchEmbed(torch.autograd.Function):
    """tokens = Linear(patch -> dim)(to_patches(x)) + space_embedding[c]  (models/brainformer.py:338-343).
    x is data (no gradient); K (= patch size) is zero-padded to a 16-byte multiple for the GEMM."""

    @staticmethod
    def forward(ctx, x, emb_w, emb_b, space, P):
        B, T, Cn = x.shape
        d = emb_w.shape[0]
        dt = _COMPUTE_DTYPE
        kp = (P + 31) // 32 * 32
        xin = x if x.dtype == torch.float32 else K.cast(x.contiguous(), torch.float32)
        tok = K.patchify(xin.contiguous(), P, kp, dt)
        sp = shadow([space.view(Cn, d)])
        h = K.gemm_nt(tok, shadow([emb_w], pad_k=kp), bias=shadow([emb_b]), residual=sp, res_rows=Cn)
        ctx.dims = (B, T // P, Cn, d, P)
        ctx.save_for_backward(tok)
        return h.view(B, (T // P) * Cn, d)

    @staticmethod
    def backward(ctx, dh):
        (tok,) = ctx.saved_tensors
        B, nT, Cn, d, P = ctx.dims
        dh2 = dh.contiguous().view(-1, d)
        dw = K.gemm_tn(dh2, tok)[:, :P].contiguous()
        db = K.colsum(dh2)
        dspace = K.colsum(dh2.view(B * nT, Cn * d)).view(1, Cn, d)
        return None, dw, db, dspace, None


class ExpandQueries(torch.autograd.Function):
    """learnable_queries [1,M,d] (fp32 master) -> [B,M,d] compute dtype (models/brainformer.py:545)."""

    @staticmethod
    def forward(ctx, qparam, B):
        _, M, d = qparam.shape
        row = shadow([qparam.view(1, M * d)])
        out = torch.empty((B, M * d), dtype=row.dtype, device=row.device)
        K.copy2d(row.expand(B, M * d), out)
        ctx.shape = (M, d)
        return out.view(B, M, d)

    @staticmethod
    def backward(ctx, dy):
        M, d = ctx.shape
        return K.colsum(dy.contiguous().view(-1, M * d)).view(1, M, d), None


class L1Loss(torch.autograd.Function):
    """mean |d| (L1) or d^2 (MSE); optional per-row weights give the masked mean of SimpleMAE (models/simple_mae:393-395)."""

    @staticmethod
    def forward(ctx, pred, target, squared, row_weight=None):
        tgt = target if target.dtype == pred.dtype else K.cast(target.contiguous(), pred.dtype)
        p = pred.contiguous()
        loss2 = K.l1_loss_fwd(p, tgt.contiguous(), squared, row_weight)
        ctx.squared = squared
        ctx.save_for_backward(p, tgt.contiguous(), row_weight, loss2)
        return loss2[0]

    @staticmethod
    def backward(ctx, gout):
        p, tgt, row_weight, loss2 = ctx.saved_tensors
        g = gout.reshape(1).float().contiguous()
        return K.l1_loss_bwd(p, tgt, g, ctx.squared, row_weight, loss2), None, None, None


class CrossEntropy(torch.autograd.Function):
    """mean NLL over rows whose target != ignore_index; logits [rows, V] (any row stride)."""

    @staticmethod
    def forward(ctx, logits, targets, ignore_index):
        loss2, lse = K.ce_loss_fwd(logits, targets.contiguous(), ignore_index)
        ctx.ignore = ignore_index
        ctx.save_for_backward(logits, targets, lse, loss2)
        return loss2[0]

    @staticmethod
    def backward(ctx, gout):
        logits, targets, lse, loss2 = ctx.saved_tensors
        g = gout.reshape(1).float().contiguous()
        d = torch.empty(logits.shape, dtype=logits.dtype, device=logits.device)
        return K.ce_loss_bwd(logits, targets.contiguous(), lse, loss2, g, d, ctx.ignore), None, None


def cross_entropy(logits: Tensor, targets: Tensor, ignore_index: int = -100) -> Tensor:
    V = logits.shape[-1]
    lg = logits.reshape(-1, V) if logits.is_contiguous() else logits
    if lg.dim() != 2:
        lg = logits.contiguous().view(-1, V)
    return CrossEntropy.apply(lg, targets.reshape(-1), ignore_index)


class HeadCrossEntropy(torch.autograd.Function):
    """loss = mean CE(Linear([LN](x)), targets) over the rows whose target != ignore_index, WITHOUT a [rows, V] tensor in either
    direction: lm_head + F.cross_entropy of models/gpt2_model.py:205-210 and the `to_words` head of the notebook CE BrainFormer, used when
    the caller does not need the logits (`fuse_head_loss`).
    forward : ONE product — fk_head_ce_fwd keeps every 128 x 128 logits tile in the MFMA accumulators and emits per-row (max, sum exp,
              target logit); fk_ce_chunk_finish turns them into the row log-sum-exp and the loss.
    backward: the product again with the vocabulary on the lane, its epilogue writing the TRANSPOSED d-logits dlT [V, rows] (bf16 in the
              throughput mode: 80 MB at rows = 800, V = 50257), so that dH = dlT^T W is a contraction over the vocabulary (fk_gemm_tn, split
              slabs) and dW = dlT H (fk_gemm_nt over the rows) — no GEMM with K = 50257 and 21 workgroups.
    Targets outside [0, V) other than ignore_index contribute nothing to the loss' numerator but count as valid rows in neither
    (F.cross_entropy would raise): callers pass token ids.  `chunk` is accepted for compatibility and unused."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, w, b, targets, eps, ignore_index, chunk):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        V = w.shape[0]
        has_ln = ln_w is not None
        if has_ln:
            h, mean, rstd = K.norm_fwd(x2, ln_w.detach(), None if ln_b is None else ln_b.detach(), eps)
        else:
            h, mean, rstd = to_compute(x2), None, None
        vec = 8 if h.dtype == torch.bfloat16 else 4
        npad = (V + vec - 1) // vec * vec
        wsh = shadow([w], pad_n=npad)
        bsh = None if b is None else shadow([b])
        tg = targets.reshape(-1).contiguous()
        loss2, lse = K.head_ce_fwd(h, wsh, bsh, tg, V, ignore_index)
        ctx.cfg = (has_ln, b is not None, ignore_index, npad)
        ctx.ln_b = ln_b
        ctx.save_for_backward(x, ln_w, w, b, h, mean, rstd, tg, lse, loss2)
        return loss2[0]

    @staticmethod
    def backward(ctx, gout):
        x, ln_w, w, b, h, mean, rstd, tg, lse, loss2 = ctx.saved_tensors
        has_ln, has_b, ignore_index, npad = ctx.cfg
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        rows, d, V = h.shape[0], h.shape[1], w.shape[0]
        g = gout.reshape(1).float().contiguous()
        wsh = shadow([w], pad_n=npad)
        bsh = shadow([b]) if has_b else None
        rows_pad = (rows + 63) // 64 * 64                                   # K of the weight-gradient product: whole 64-element k-tiles
        dlT, db = K.head_ce_bwd(h, wsh, bsh, tg, lse, loss2, g, V, npad, rows_pad, has_b, ignore_index)
        dh32 = K.gemm_tn(dlT, wsh)                                          # [rows_pad, d] = sum_v dlT[v, m] W[v, :]
        dh = dh32[:rows] if h.dtype == torch.float32 else K.cast(dh32[:rows], h.dtype)
        hT = torch.zeros((d, rows_pad), dtype=h.dtype, device=h.device)
        K.transpose2d(h, out=hT)
        dw = K.gemm_nt(dlT, hT, out_dtype=torch.float32)[:V]                # [V, d] = sum_m dlT[v, m] H[m, :]
        if has_ln:
            dx, dg, dbe = norm_bwd(dh, x2, ln_w, ctx.ln_b, mean, rstd)
        else:
            dx, dg, dbe = (dh if dh.dtype == x2.dtype else K.cast(dh, x2.dtype)), None, None
        return dx.view(shp), dg, dbe, dw, db, None, None, None, None


def head_cross_entropy(x: Tensor, ln_w, ln_b, w: Tensor, b, targets: Tensor, eps: float = 1e-5, ignore_index: int = -100,
                       chunk: int = 8192) -> Tensor:
    return HeadCrossEntropy.apply(x, ln_w, ln_b, w, b, targets, eps, ignore_index, chunk)


def l1_loss(pred: Tensor, target: Tensor) -> Tensor:
    return L1Loss.apply(pred, target, False)


def mse_loss(pred: Tensor, target: Tensor, row_weight: Optional[Tensor] = None) -> Tensor:
    return L1Loss.apply(pred, target, True, row_weight)


# --------------------------------------------------------------------------------------------- conv stack
def _conv_shadow(w: Tensor, mode: str, transpose: bool) -> Tensor:
    """GEMM-shaped compute-dtype view of a convolution weight (re-packed lazily when the master changes):
    mode "conv":  nn.Conv1d weight [Cout, Cin, K]          -> W' [Cout, K*Cin],   W'[o, k*Cin + c] = W[o, c, k]
    mode "convT": nn.ConvTranspose1d weight [Cin, Cout, 2s] -> W' [s*Cout, 2*Cin], row (r, o), taps (x[j-1], x[j]):
                  W'[(r,o), c] = W[c, o, r + s],  W'[(r,o), Cin + c] = W[c, o, r]      (models/vq_brain.py:31-45)."""
    key = (("conv", mode, w.data_ptr(), tuple(w.shape)), transpose, _COMPUTE_DTYPE)
    ent = _shadow_entry(key, (w,))
    stamp = ent.current_stamp()
    if ent.stamp == stamp:
        return ent.tensor
    wref = ent.params[0]

    def pack():
        wd = wref().detach()
        if mode == "conv":
            g = wd.permute(0, 2, 1).reshape(wd.shape[0], -1)
        else:
            cin, cout, k2 = wd.shape
            s_ = k2 // 2
            g = torch.cat([wd[:, :, s_:].permute(2, 1, 0), wd[:, :, :s_].permute(2, 1, 0)], dim=2).reshape(s_ * cout, 2 * cin)
        g = g.t() if transpose else g
        if ent.tensor is None:
            ent.tensor = torch.empty(g.shape, dtype=_COMPUTE_DTYPE, device=wd.device)
        ent.tensor.copy_(g)                 # in place: the storage a captured GEMM reads stays the same across optimizer steps

    ent.repack = pack
    pack()
    ent.stamp = stamp
    return ent.tensor


def _conv_wgrad_to_master(dwp: Tensor, w: Tensor, mode: str) -> Tensor:
    """inverse of the _conv_shadow layout for the fp32 weight gradient"""
    if mode == "conv":
        cout, cin, k = w.shape
        return dwp.view(cout, k, cin).permute(0, 2, 1).contiguous()
    cin, cout, k2 = w.shape
    s_ = k2 // 2
    t = dwp.view(s_, cout, 2, cin)                       # [r, o, tap, c]
    return torch.cat([t[:, :, 1].permute(2, 1, 0), t[:, :, 0].permute(2, 1, 0)], dim=2).contiguous()   # [c, o, (r | r + s)]


class CausalConv1dFn(torch.autograd.Function):
    """Channels-last causal convolution as im2col + MFMA GEMM (bias and an optional residual fused in the GEMM epilogue).
    mode "conv": y[b, t, :] = sum_k W[:, :, k] x[b, t*stride + k*dil - dil*(K-1), :] + bias  (CausalConv1d, models/vq_brain.py:22-28)
    mode "convT": CausalConvTranspose1d(kernel 2s, stride s): [B, T, Cin] -> [B, s*T, Cout]  (:31-45)."""

    @staticmethod
    def forward(ctx, x, w, b, stride, dil, mode, residual):
        B, T, cin = x.shape
        if mode == "conv":
            cout, _, ks = w.shape
            st, n_rep = stride, 1
        else:
            _, cout, k2 = w.shape
            ks, st, n_rep = 2, 1, k2 // 2
            assert k2 == 2 * stride and dil == 1, "CausalConvTranspose1d is built for kernel_size = 2 * stride"
        x = x.contiguous()
        cols = x.view(B * T, cin) if (ks == 1 and st == 1) else K.im2col1d(x, ks, st, dil)
        bias = None
        if b is not None:
            bias = shadow([b]) if n_rep == 1 else K.cast(b.detach().repeat(n_rep), _COMPUTE_DTYPE)
        res2 = None if residual is None else residual.contiguous().view(cols.shape[0], -1)
        y = K.gemm_nt(cols, _conv_shadow(w, mode, False), bias, residual=res2)
        tout = (T - 1) // st + 1
        ctx.cfg = (B, T, cin, ks, st, dil, mode, n_rep, b is not None, residual is not None)
        ctx.save_for_backward(x, w)
        return y.view(B, tout * n_rep, cout)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        B, T, cin, ks, st, dil, mode, n_rep, has_b, has_res = ctx.cfg
        dy = dy.contiguous()
        dy2 = dy.view(-1, dy.shape[-1] * n_rep)
        cols = x.view(B * T, cin) if (ks == 1 and st == 1) else K.im2col1d(x, ks, st, dil)
        dcols = K.gemm_nt(dy2, _conv_shadow(w, mode, True))
        dx = dcols.view(B, T, cin) if (ks == 1 and st == 1) else K.col2im1d(dcols, B, T, cin, ks, st, dil)
        dw = _conv_wgrad_to_master(K.gemm_tn(dy2, cols), w, mode)
        db = None
        if has_b:
            db = K.colsum(dy2)
            if n_rep > 1:
                db = db.view(n_rep, -1).sum(0)
        return dx, dw, db, None, None, None, (dy if has_res else None)


class EluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.save_for_backward(x)
        return K.elu_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return K.elu_bwd(x, dy.contiguous())


class StraightThrough(torch.autograd.Function):
    """forward: the quantized vectors; backward: the gradient goes to the encoder output unchanged (x + (q - x).detach())."""

    @staticmethod
    def forward(ctx, x, q):
        return q

    @staticmethod
    def backward(ctx, dq):
        return dq, None


# --------------------------------------------------------------------------------------------- MAE pieces
class GatherRows(torch.autograd.Function):
    """out[b, i, :] = src[b, idx[b, i], :]  (models/brainformer.py:441,468); idx rows are unique per sample."""

    @staticmethod
    def forward(ctx, src, idx):
        ctx.n_src = src.shape[1]
        ctx.save_for_backward(idx)
        return K.gather_rows(src.contiguous(), idx)

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        dy = dy.contiguous()
        d = torch.zeros((dy.shape[0], ctx.n_src, dy.shape[2]), dtype=dy.dtype, device=dy.device)
        K.scatter_rows_(d, idx, dy)
        return d, None


class MaskedPatchEmbed(torch.autograd.Function):
    """tokens = emb(patches[b, unmasked]) + spatial_pos_embedding[b, unmasked]  (models/brainformer.py:429-444)."""

    @staticmethod
    def forward(ctx, tok_all, emb_w, emb_b, space, unmasked, P):
        B, N, kp = tok_all.shape
        Cn, d = space.shape[1], space.shape[2]
        tok_u = K.gather_rows(tok_all, unmasked)                                   # [B, n, kp]
        sp = K.gather_rows(shadow([space.view(Cn, d)]), unmasked, idx_mod=Cn)     # spatial rows repeat every Cn tokens
        n = unmasked.shape[1]
        h = K.gemm_nt(tok_u.view(B * n, kp), shadow([emb_w], pad_k=kp), bias=shadow([emb_b]), residual=sp.view(B * n, d))
        ctx.dims = (B, n, Cn, d, P)
        ctx.save_for_backward(tok_u, unmasked)
        return h.view(B, n, d)

    @staticmethod
    def backward(ctx, dh):
        tok_u, unmasked = ctx.saved_tensors
        B, n, Cn, d, P = ctx.dims
        dh2 = dh.contiguous().view(B * n, d)
        dw = K.gemm_tn(dh2, tok_u.view(B * n, -1))[:, :P].contiguous()
        db = K.colsum(dh2)
        dspace = torch.zeros((Cn, d), dtype=torch.float32, device=dh.device)
        K.scatter_add_rows_(dspace, unmasked, dh2, idx_mod=Cn)
        return None, dw, db, dspace.view(1, Cn, d), None, None


class AssembleDecoder(torch.autograd.Function):
    """dec[b, unmasked] = tokens, dec[b, masked] = mask_token, + decoder_pos_emb[cat(unmasked, masked)] added in
    CONCATENATION order (the reference's quirk, models/brainformer.py:455-460)."""

    @staticmethod
    def forward(ctx, tokens, mask_token, pos_table, unmasked, masked):
        B, n, dd = tokens.shape
        N = n + masked.shape[1]
        row = shadow([mask_token.view(1, dd)])
        dec = torch.empty((B, N, dd), dtype=tokens.dtype, device=tokens.device)
        K.copy2d(row.expand(B * N, dd), dec.view(B * N, dd))                       # every row = mask_token ...
        K.scatter_rows_(dec, unmasked, tokens.contiguous())                        # ... then the visible tokens
        order = torch.cat([unmasked, masked], 1).contiguous()
        pos = K.gather_rows(pos_table.detach(), order, out_dtype=tokens.dtype)     # [B, N, dd] in concatenation order
        ctx.save_for_backward(unmasked, masked, order)
        ctx.tshape = pos_table.shape
        return K.add(dec, pos)

    @staticmethod
    def backward(ctx, dy):
        unmasked, masked, order = ctx.saved_tensors
        dy = dy.contiguous()
        dd = dy.shape[2]
        dtok = K.gather_rows(dy, unmasked)
        dmask = K.colsum(K.gather_rows(dy, masked).view(-1, dd))
        dpos = torch.zeros(ctx.tshape, dtype=torch.float32, device=dy.device)
        K.scatter_add_rows_(dpos, order, dy)
        return dtok, dmask, dpos, None, None
