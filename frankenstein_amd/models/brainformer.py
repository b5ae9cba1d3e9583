"""MI355X-native brainformer: same module surface as the reference's ``models/brainformer.py``
(config dataclasses, class names, constructor/forward signatures, parameter names = state-dict keys),
with every forward/backward executed by the hand-written HIP kernels of libfranken_hip.so.

Reference map (file:line relative to the reference root):
  MAEConfig / Config                 models/brainformer.py:17-53
  build_complex_rope_cache           :56-68        apply_rope            :70-91
  build_advanced_causal_mask         :93-111       MLP (SwiGLU)          :115-124
  CausalSelfAttention                :126-173      CausalCrossAttention  :175-219
  RMSNorm                            :221-232      Block / CrossBlock    :234-268
  Encoder                            :271-352      BrainFormer           :488-574

Differences by design: parameters stay fp32 masters, compute runs in the dtype chosen with
``frankenstein_amd.set_compute_dtype`` ('bf16' default, 'fp32' parity mode); the [N,N] boolean
``attn_mask`` buffer is kept only for state-dict compatibility — the attention kernel evaluates
the same predicate analytically ((j // C) <= (i // C)); there is no CPU / PyTorch fallback.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn

from .. import engine as E
from .. import kernels as K
from ..kernels import MASK_BLOCK_CAUSAL, Mask, NO_MASK


class Serializable:  # the reference derives its configs from simple_parsing's Serializable (only a base class)
    pass


@dataclass
class MAEConfig(Serializable):
    # data params
    window_size: int = 1024
    n_electrodes: int = 256
    patch_size: int = 48
    # encoder
    dim: int = 256
    n_layers: int = 4
    head_dim: int = 32
    hidden_dim: int = 1024
    n_heads: int = 8
    n_kv_heads: int = 8
    rope_theta: int = 10000
    # decoder
    n_dec_layers: Optional[int] = 4
    decoder_dim: Optional[int] = 256


@dataclass
class Config(Serializable):
    encoder: MAEConfig
    # perceiver
    n_output_tokens: int = 32
    output_dim: int = 1024
    dim: int = 256
    n_layers: int = 2
    head_dim: int = 16
    hidden_dim: int = 512
    n_heads: int = 4
    n_kv_heads: int = 4
    rope_theta: int = 10_000


# ------------------------------------------------------------------------------------------- rope / mask helpers
def build_complex_rope_cache(dim: int, seq_len: int, theta: float) -> torch.Tensor:
    """complex64 [seq_len, dim//2] = exp(i * t * theta^(-2k/dim)); init-time, computed on the host exactly
    like the reference so both sides rotate by bit-identical (cos, sin) tables."""
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2).float() / dim))
    ang = torch.outer(torch.arange(seq_len), freqs).float()
    cache = torch.polar(torch.ones_like(ang), ang)
    cache.requires_grad = False
    return cache


def register_mask(t: torch.Tensor, mask: Mask) -> None:
    """Tell the attention path that boolean tensor ``t`` IS the analytic mask ``mask``.  The tag lives on the tensor object itself
    (no address-keyed registry: a freed mask's address can be recycled by an unrelated tensor of the same shape)."""
    t._fk_mask = mask


def resolve_mask(attn_mask, t_q: int, t_k: int) -> Mask:
    if attn_mask is None:
        return NO_MASK
    if isinstance(attn_mask, Mask):
        return attn_mask
    m = getattr(attn_mask, "_fk_mask", None)
    if m is None:
        # an arbitrary boolean tensor (the reference hands whatever it is given to SDPA, models/brainformer.py:160-168): the per-element
        # path of the generic kernels reads it as uint8.  The masks the reference itself builds are tagged and never come here.
        assert attn_mask.dtype == torch.bool, "attention masks are boolean (True = attend)"
        dm = Mask.from_dense(attn_mask, t_q, t_k)
        if not dm.limits.is_cuda and torch.cuda.is_available():
            dm.limits = dm.limits.cuda()
        return dm
    return m.sliced(attn_mask.shape[-2], attn_mask.shape[-1], t_q, t_k)   # mask[..., -t_q:, -t_k:]


def build_advanced_causal_mask(block_size: int, tok_per_time: int) -> torch.Tensor:
    """bool [block_size, block_size], True = attend: lower-triangular OR same time block, i.e.
    (j // tok_per_time) <= (i // tok_per_time).  Returned for API/state-dict compatibility and registered
    so the kernels use the analytic form."""
    blk = torch.arange(block_size) // tok_per_time
    m = blk[None, :] <= blk[:, None]
    register_mask(m, Mask(MASK_BLOCK_CAUSAL, tok_per_time))
    return m


class _CastFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src = x.dtype
        return K.cast(x.contiguous(), dtype)

    @staticmethod
    def backward(ctx, dy):
        return K.cast(dy.contiguous(), ctx.src), None


def _prep(x: torch.Tensor) -> torch.Tensor:
    """activation -> compute dtype, contiguous (kernel cast; differentiable)."""
    dt = E.compute_dtype()
    if x.dtype != dt:
        x = _CastFn.apply(x, dt)
    return x if x.is_contiguous() else _Contig.apply(x)


class _Contig(torch.autograd.Function):
    """Contiguous copy of a row-sliced activation ([B, T', d] view of a [B, T, d] buffer) via the copy kernel."""

    @staticmethod
    def forward(ctx, x):
        out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
        if x.dim() == 3 and x.stride(2) == 1 and x.stride(1) == x.shape[2]:
            Bn, Tn, dn = x.shape
            K.copy2d(torch.as_strided(x, (Bn, Tn * dn), (x.stride(0), 1)), out.view(Bn, Tn * dn))
        elif x.dim() == 2 and x.stride(1) == 1:
            K.copy2d(x, out)
        else:
            raise NotImplementedError(f"unsupported activation layout: shape {tuple(x.shape)} strides {x.stride()}")
        return out

    @staticmethod
    def backward(ctx, dy):
        return dy


def apply_rope(x: torch.Tensor, rope: torch.Tensor) -> torch.Tensor:
    """x [b, t, n_h, dim], rope complex [T, dim//2] or [B, T, dim//2] (last t rows used).  Out of place."""
    B, T, H, D = x.shape
    y = _prep(x).clone().view(B, T, H * D)
    r = E.Rope(rope)
    K.rope_(y, H, D, r.table, r.pos_off(T))
    return y.view(B, T, H, D).to(x.dtype)


# ------------------------------------------------------------------------------------------- blocks
class Linear(nn.Linear):
    """nn.Linear whose forward/backward run on the MFMA GEMM kernels."""

    def forward(self, x):
        return E.NormLinear.apply(_prep(x), None, None, self.weight, self.bias, 0.0, False)


class LayerNorm(nn.LayerNorm):
    def forward(self, x):
        return E.LayerNormFn.apply(_prep(x), self.weight, self.bias, self.eps, K.NORM_LAYER)


class RMSNorm(nn.Module):
    def __init__(self, dim: int, eps: float = 1e-6):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))

    def forward(self, x):
        return E.LayerNormFn.apply(_prep(x), self.weight, None, self.eps, K.NORM_RMS)


def _norm_kind(ln) -> int:
    return K.NORM_RMS if isinstance(ln, RMSNorm) else K.NORM_LAYER


class MLP(nn.Module):
    """SwiGLU: w2(silu(w1 x) * w3 x), no biases."""

    def __init__(self, config):
        super().__init__()
        self.w1 = Linear(config.dim, config.hidden_dim, bias=False)
        self.w2 = Linear(config.hidden_dim, config.dim, bias=False)
        self.w3 = Linear(config.dim, config.hidden_dim, bias=False)

    def branch(self, x, ln: Optional[nn.LayerNorm], residual: bool):
        return E.MlpBranch.apply(x, None if ln is None else ln.weight, None if ln is None else getattr(ln, 'bias', None),
                                 self.w1.weight, None, self.w3.weight, self.w2.weight, None,
                                 (residual, 0.0 if ln is None else ln.eps, _norm_kind(ln)))

    def forward(self, x) -> torch.Tensor:
        return self.branch(_prep(x), None, False)


class CausalSelfAttention(nn.Module):
    def __init__(self, config, is_causal=True):
        super().__init__()
        assert config.n_heads == config.n_kv_heads, "n_heads should be equal n_kv_heads"
        self.n_heads = config.n_heads
        self.n_kv_heads = config.n_heads
        self.repeats = self.n_heads // self.n_kv_heads
        self.head_dim = config.head_dim
        self.qw = Linear(config.dim, config.head_dim * config.n_heads, bias=False)
        self.kw = Linear(config.dim, config.head_dim * config.n_kv_heads, bias=False)
        self.vw = Linear(config.dim, config.head_dim * config.n_kv_heads, bias=False)
        self.project = Linear(config.head_dim * config.n_heads, config.dim, bias=False)

    def branch(self, x, attn_mask, rope, ln: Optional[nn.LayerNorm], residual: bool):
        T = x.shape[1]
        spec = (self.n_heads, self.head_dim, resolve_mask(attn_mask, T, T),
                None if rope is None else E.Rope(rope), residual, 0.0 if ln is None else ln.eps, _norm_kind(ln))
        return E.AttnBranch.apply(x, None if ln is None else ln.weight, None if ln is None else getattr(ln, 'bias', None),
                                  self.project.weight, None, None, spec, self.qw.weight, self.kw.weight, self.vw.weight)

    def forward(self, x, attn_mask, rope, kv_cache=None):
        return self.branch(_prep(x), attn_mask, rope, None, False)


class CausalCrossAttention(nn.Module):
    def __init__(self, config, is_causal=True):
        super().__init__()
        assert config.n_heads == config.n_kv_heads, "n_heads should be equal n_kv_heads"
        self.n_heads = config.n_heads
        self.n_kv_heads = config.n_heads
        self.repeats = self.n_heads // self.n_kv_heads
        self.head_dim = config.head_dim
        self.qw = Linear(config.dim, config.head_dim * config.n_heads, bias=False)
        self.kw = Linear(config.dim, config.head_dim * config.n_kv_heads, bias=False)
        self.vw = Linear(config.dim, config.head_dim * config.n_kv_heads, bias=False)
        self.project = Linear(config.head_dim * config.n_heads, config.dim, bias=False)
        self.kv_cache = None

    def branch(self, x, context, attn_mask, ln: nn.LayerNorm):
        spec = (self.n_heads, self.head_dim, resolve_mask(attn_mask, x.shape[1], context.shape[1]), ln.eps)
        return E.CrossAttnBranch.apply(x, context, ln.weight, ln.bias, self.qw.weight, self.kw.weight,
                                       self.vw.weight, self.project.weight, spec)

    def forward(self, x, context, attn_mask=None, use_kv_cache=None):
        """Stand-alone use (no pre-norm, no residual): composed from the projection and attention kernels."""
        x, context = _prep(x), _prep(context)
        B, T, _ = x.shape
        H, D = self.n_heads, self.head_dim
        q = self.qw(x).view(B, T, H, D)
        k = self.kw(context).view(B, -1, H, D)
        v = self.vw(context).view(B, -1, H, D)
        o = _SDPA.apply(q, k, v, resolve_mask(attn_mask, T, context.shape[1]))
        return self.project(o.view(B, T, H * D))


class _SDPA(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, mask):
        o, lse = K.attn_fwd(q, k, v, mask)
        ctx.mask = mask
        ctx.save_for_backward(q, k, v, o, lse)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        K.attn_bwd(q, k, v, o, do.contiguous(), lse, dq, dk, dv, ctx.mask)
        return dq, dk, dv, None


class Block(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.ln_1 = LayerNorm(config.dim)
        self.attn = CausalSelfAttention(config)
        self.ln_2 = LayerNorm(config.dim)
        self.mlp = MLP(config)

    def forward(self, x, attn_mask=None, rope=None, kv_cache=False):
        x = _prep(x)
        x = self.attn.branch(x, attn_mask, rope, self.ln_1, True)      # x + attn(ln_1(x))
        return self.mlp.branch(x, self.ln_2, True)                     # x + mlp(ln_2(x))


class CrossBlock(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.sa_block = Block(config)
        self.ln_1 = LayerNorm(config.dim)
        self.cross_attn = CausalCrossAttention(config)
        self.ln_2 = LayerNorm(config.dim)
        self.mlp = MLP(config)

    def forward(self, x, context, self_attn_mask=None, cross_attn_mask=None, sa_rope=None):
        x, context = _prep(x), _prep(context)
        x = self.cross_attn.branch(x, context, cross_attn_mask, self.ln_1)
        x = self.mlp.branch(x, self.ln_2, True)
        return self.sa_block(x, attn_mask=self_attn_mask, rope=sa_rope)


# ------------------------------------------------------------------------------------------- models
class Encoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.patch_size = config.patch_size
        self.n_electrodes = config.n_electrodes
        self.n_patches_per_channel = config.window_size // config.patch_size
        self.block_size = self.n_patches_per_channel * config.n_electrodes
        self.transformer = nn.ModuleDict(dict(
            emb=Linear(config.patch_size, config.dim),
            h=nn.ModuleList([Block(config) for _ in range(config.n_layers)]),
            ln_f=LayerNorm(config.dim),
        ))
        self.space_embedding = nn.Parameter(torch.randn(1, config.n_electrodes, config.dim), requires_grad=True)
        self.precompute_rope_cash = build_complex_rope_cache(dim=config.head_dim, seq_len=self.block_size,
                                                             theta=config.rope_theta)
        # The reference registers the [N, N] boolean mask as a buffer (models/brainformer.py:298): 37.7 MB at N = 6144 that no kernel
        # here reads.  It is therefore built lazily, on the host, only when somebody asks for `.attn_mask` or for a state dict (whose
        # `attn_mask` key is kept for safetensors load_model / save_model compatibility); the forward uses the analytic form.
        self._attn_mask_host = None
        self._register_state_dict_hook(Encoder._state_dict_mask_hook)
        self._register_load_state_dict_pre_hook(Encoder._load_state_dict_mask_hook)
        print("Encoder: number of parameters: %.2fM" % (self.get_num_params() / 1e6,))

    @property
    def attn_mask(self) -> torch.Tensor:
        if self._attn_mask_host is None:
            self._attn_mask_host = build_advanced_causal_mask(block_size=self.block_size, tok_per_time=self.n_electrodes)
        return self._attn_mask_host

    @staticmethod
    def _state_dict_mask_hook(module, state_dict, prefix, local_metadata):
        state_dict[prefix + "attn_mask"] = module.attn_mask

    @staticmethod
    def _load_state_dict_mask_hook(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        state_dict.pop(prefix + "attn_mask", None)          # a function of the config, not a learned value

    @property
    def dtype(self) -> torch.dtype:
        return next(self.parameters()).dtype

    @property
    def device(self) -> torch.device:
        return next(self.parameters()).device

    @property
    def rope_cache(self) -> torch.Tensor:
        if self.precompute_rope_cash.device != self.device:
            self.precompute_rope_cash = self.precompute_rope_cash.to(device=self.device)
        return self.precompute_rope_cash

    @property
    def spatial_pos_embedding(self):
        return self.space_embedding.repeat((1, self.n_patches_per_channel, 1))

    def to_patches(self, x):
        """'b (t p1) c -> b (t c) p1' (kernel transpose; returned in the compute dtype)."""
        B, T, C = x.shape
        P = self.patch_size
        xin = x if x.dtype == torch.float32 else x.float()
        return K.patchify(xin.contiguous(), P, P, E.compute_dtype()).view(B, (T // P) * C, P)

    def get_num_params(self):
        return sum(p.numel() for p in self.parameters())

    def forward(self, x, kv_cache=None):
        """x: [B, T, C] feature frames -> [B, (T/patch)*C, dim] tokens."""
        assert x.shape[2] == self.n_electrodes and x.shape[1] % self.patch_size == 0
        h = E.PatchEmbed.apply(x, self.transformer.emb.weight, self.transformer.emb.bias, self.space_embedding,
                               self.patch_size)
        T = h.shape[1]
        mask = Mask(MASK_BLOCK_CAUSAL, self.n_electrodes).sliced(self.block_size, self.block_size, T, T)   # attn_mask[-T:, -T:]
        rope = self.rope_cache
        for block in self.transformer.h:
            h = block(h, attn_mask=mask, rope=rope, kv_cache=kv_cache)
        return self.transformer.ln_f(h)


class MAE(nn.Module):
    """Masked auto-encoder pretraining (models/brainformer.py:354-486): encode a random 1 - masking_ratio subset of the
    patch tokens (per-sample RoPE rows and the gathered block-causal sub-mask, evaluated as prefix tables), decode all
    tokens with mask tokens + positional embedding, MSE on the masked patches.  ``indices=(masked, unmasked)`` lets a
    caller (parity tests) supply the random index sets; otherwise they are drawn like the reference does."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.decoder_dim = config.decoder_dim
        self.encoder = Encoder(config)
        self.decoder = nn.ModuleDict(dict(
            emb=nn.Identity(),
            h=nn.ModuleList([Block(config) for _ in range(config.n_dec_layers)]),
        ))
        self.mask_token = nn.Parameter(torch.randn(config.dim))
        self.decoder_pos_emb = nn.Embedding(self.encoder.block_size, config.decoder_dim)
        self.to_signals = Linear(config.decoder_dim, config.patch_size)
        print("MAE: number of parameters: %.2fM" % (self.get_num_params() / 1e6))

    def get_num_params(self, non_embedding=True):
        return sum(p.numel() for p in self.parameters())

    def get_masking_indices(self, masking_ratio, x):
        b, n_tokens, _ = x.shape
        num_masked = int(masking_ratio * n_tokens)
        rand_indices = torch.rand(b, n_tokens, device=x.device).argsort(dim=-1)
        masked, unmasked = rand_indices[:, :num_masked], rand_indices[:, num_masked:]
        return torch.sort(masked, dim=1)[0], torch.sort(unmasked, dim=1)[0]

    def to_signal_shape(self, tok):
        """'b (t c) p -> b (t p) c'  (host-side view math on the small return_preds outputs)."""
        B, N, P = tok.shape
        Cn = self.config.n_electrodes
        return tok.view(B, N // Cn, Cn, P).permute(0, 1, 3, 2).reshape(B, (N // Cn) * P, Cn)

    def forward(self, x, targets=None, date_info=None, masking_ratio=0.75, return_preds=False, indices=None):
        cfg, enc = self.config, self.encoder
        B, T, Cn = x.shape
        P = cfg.patch_size
        kp = (P + 31) // 32 * 32
        xin = x if x.dtype == torch.float32 else x.float()
        tok_all = K.patchify(xin.contiguous(), P, kp, E.compute_dtype()).view(B, (T // P) * Cn, kp)
        n_tokens = tok_all.shape[1]
        if indices is None:
            masked, unmasked = self.get_masking_indices(masking_ratio, tok_all)
        else:
            masked, unmasked = indices
        masked, unmasked = masked.contiguous(), unmasked.contiguous()
        # per-sample rope rows and the sub-mask of the block-causal mask at the kept tokens
        table = torch.view_as_real(enc.rope_cache).reshape(enc.block_size, -1)
        rope = K.gather_rows(table.contiguous(), unmasked).view(B, unmasked.shape[1], -1, 2)
        mask = Mask.from_token_ids(unmasked, unmasked, enc.n_electrodes)
        tokens = E.MaskedPatchEmbed.apply(tok_all, enc.transformer.emb.weight, enc.transformer.emb.bias,
                                          enc.space_embedding, unmasked, P)
        for block in enc.transformer.h:
            tokens = block(tokens, attn_mask=mask, rope=rope)
        tokens = enc.transformer.ln_f(tokens)
        dec = E.AssembleDecoder.apply(self.decoder.emb(tokens), self.mask_token, self.decoder_pos_emb.weight, unmasked, masked)
        for block in self.decoder.h:
            dec = block(dec)
        pred = self.to_signals(E.GatherRows.apply(dec, masked))                    # [B, n_masked, P]
        tok_p = tok_all if kp == P else K.patchify(xin.contiguous(), P, P, E.compute_dtype()).view(B, n_tokens, P)
        target = K.gather_rows(tok_p, masked)                                      # [B, n_masked, P] (data, no gradient)
        loss = E.mse_loss(pred, target)
        if return_preds:
            with torch.no_grad():
                rec = K.cast(tok_p, torch.float32)
                K.scatter_rows_(rec, masked, pred.detach().float().contiguous())
                bm = torch.zeros_like(rec)
                K.scatter_rows_(bm, masked, torch.ones_like(pred, dtype=torch.float32))
            return loss, self.to_signal_shape(rec), self.to_signal_shape(bm)
        return (loss, None)


def _mae_get_sub_att_matrix(self, attn_mask, unmasked_indices):
    """Dense form of the gathered block-causal sub-mask, [b, 1, n, n] bool (models/brainformer.py:392-413).  Kept for API parity and
    for checking: the forward itself passes the same mask analytically (kernels.Mask.from_token_ids), never as a tensor."""
    am = attn_mask.to(unmasked_indices.device)
    sub = am[unmasked_indices[:, :, None], unmasked_indices[:, None, :]]
    return sub[:, None]


MAE.get_sub_att_matrix = _mae_get_sub_att_matrix


class BrainFormer(nn.Module):
    config = Config
    head_name = 'to_motion'

    def __init__(self, config: Config):
        super().__init__()
        self.config = config
        self.encoder = Encoder(config.encoder)
        self.n_output_tokens = config.n_output_tokens
        self.learnable_queries = nn.Parameter(torch.zeros(1, config.n_output_tokens, config.dim))
        self.perceiver = nn.ModuleDict({
            'h': nn.ModuleList([CrossBlock(config) for _ in range(config.n_layers)]),
            'ln_f': LayerNorm(config.dim),
            self.head_name: Linear(config.dim, config.output_dim)})
        self.register_buffer('cross_attn_mask', None)
        self.register_buffer('self_attn_mask', None)
        self.precompute_rope_cash = build_complex_rope_cache(dim=config.head_dim, seq_len=config.n_output_tokens,
                                                             theta=config.rope_theta)
        print("Full HandFormer: number of parameters: %.2fM" % (self.get_num_params() / 1e6,))

    def get_num_params(self):
        return sum(p.numel() for p in self.parameters())

    @property
    def dtype(self) -> torch.dtype:
        return next(self.parameters()).dtype

    @property
    def device(self) -> torch.device:
        return next(self.parameters()).device

    @property
    def rope_cache(self) -> torch.Tensor:
        if self.precompute_rope_cash.device != self.device:
            self.precompute_rope_cash = self.precompute_rope_cash.to(device=self.device)
        return self.precompute_rope_cash

    def queries_out(self, x):
        """encoder -> learnable queries -> perceiver CrossBlocks (before ln_f and the head)."""
        b = x.shape[0]
        ctx = self.encoder(x)
        q = E.ExpandQueries.apply(self.learnable_queries, b)
        for cross_block in self.perceiver.h:
            q = cross_block(q, ctx, self.self_attn_mask, self.cross_attn_mask, sa_rope=self.rope_cache)
        return q

    def features(self, x):
        """encoder -> learnable queries -> perceiver CrossBlocks -> ln_f -> head."""
        q = self.queries_out(x)
        head = self.perceiver[self.head_name]
        ln = self.perceiver.ln_f
        return E.NormLinear.apply(q, ln.weight, ln.bias, head.weight, head.bias, ln.eps, False)

    def forward(self, x, targets=None, date_info=None):
        pred = self.features(x)
        if targets is None:
            return None, pred
        return E.l1_loss(pred, targets), pred

    @torch.no_grad()
    def inference(self, myo, date_info):
        x = torch.from_numpy(myo)[None].to(self.device).float()
        pred = self.forward(x, targets=None)[1]
        return pred[0].float().cpu().numpy().T



def default_generation(model, emg, stride=8):
    """Sliding-window latency loop of the reference (models/brainformer.py:579-597): emg [Time, n_channels]."""
    ws = model.config.window_size
    sample = emg[:ws]
    for i in range(int((emg.size(0) - ws) // stride)):
        model(sample[None, ...])
        sample = emg[i * stride:i * stride + ws]
    return 'Completed'


@torch.no_grad()
def cache_generation(model, emg, stride=8):
    """The reference's variant (:599-618) passes use_kv_cache=True to a forward that has no such argument (stale code); the same
    windows are run through the plain forward here."""
    return default_generation(model, emg, stride)
