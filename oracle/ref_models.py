"""CPU oracle: functional restatement of the reference models (TEST INFRASTRUCTURE).

All functions take a flat ``sd`` (state-dict style ``name -> torch.Tensor``) and plain
config objects (any object with the reference's dataclass field names).  Math is written
out explicitly (no nn.Module, no F.scaled_dot_product_attention, no F.layer_norm, no
F.cross_entropy) so that this is an independent statement of the algorithm; torch is only
the tensor library + autograd that produces the oracle gradients.

Reference citations are relative to /root/reference.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Dict, Optional

import torch

Tensor = torch.Tensor


# --------------------------------------------------------------------------- configs
def mae_config(**kw) -> SimpleNamespace:
    """Field names/defaults of models/brainformer.py:17-37 (MAEConfig)."""
    d = dict(window_size=1024, n_electrodes=256, patch_size=48, dim=256, n_layers=4,
             head_dim=32, hidden_dim=1024, n_heads=8, n_kv_heads=8, rope_theta=10000,
             n_dec_layers=4, decoder_dim=256)
    d.update(kw)
    return SimpleNamespace(**d)


def perceiver_config(encoder, **kw) -> SimpleNamespace:
    """Field names/defaults of models/brainformer.py:39-53 (Config)."""
    d = dict(encoder=encoder, n_output_tokens=32, output_dim=1024, dim=256, n_layers=2,
             head_dim=16, hidden_dim=512, n_heads=4, n_kv_heads=4, rope_theta=10000)
    d.update(kw)
    return SimpleNamespace(**d)


def gpt_config(**kw) -> SimpleNamespace:
    """Field names/defaults of models/gpt2_model.py:108-116 (GPTConfig)."""
    d = dict(block_size=1024, vocab_size=50304, n_layer=12, n_head=12, n_embd=768,
             dropout=0.0, bias=True)
    d.update(kw)
    return SimpleNamespace(**d)


# --------------------------------------------------------------------------- primitives
def rope_angles(head_dim: int, seq_len: int, theta: float) -> Tensor:
    """models/brainformer.py:56-68: angle[t, i] = t * theta^(-2i/head_dim), float32 [T, hd/2].
    (the reference stores polar(1, angle) as complex64; we keep the angle and use cos/sin)."""
    freqs = 1.0 / (theta ** (torch.arange(0, head_dim, 2).float() / head_dim))
    t = torch.arange(seq_len)
    return torch.outer(t, freqs).float()


def apply_rope(x: Tensor, ang: Tensor) -> Tensor:
    """models/brainformer.py:70-91.  x [B,T,H,hd]; ang [Tc,hd/2] or [B,Tc,hd/2]; the LAST T rows
    of the cache are used (``rope[-T:]``, :80/:82).  Interleaved pairs (x[2i], x[2i+1]) are
    rotated as the complex number x[2i] + i x[2i+1] times e^{i ang}, in fp32, cast back."""
    T = x.shape[1]
    ang = ang[-T:] if ang.dim() == 2 else ang[:, -T:]
    ang = ang.unsqueeze(-2)                                   # [..., T, 1, hd/2]
    c, s = torch.cos(ang), torch.sin(ang)
    xf = x.float().reshape(*x.shape[:-1], -1, 2)
    re, im = xf[..., 0], xf[..., 1]
    out = torch.stack((re * c - im * s, re * s + im * c), dim=-1).flatten(3)
    return out.type_as(x)


def block_causal_mask(block_size: int, tok_per_time: int) -> Tensor:
    """models/brainformer.py:93-111: tril OR same-time-block == (j // C) <= (i // C)."""
    i = torch.arange(block_size)
    return (i[None, :] // tok_per_time) <= (i[:, None] // tok_per_time)


def layer_norm(x: Tensor, w: Tensor, b: Optional[Tensor], eps: float = 1e-5) -> Tensor:
    """nn.LayerNorm(dim) (models/brainformer.py:237) / F.layer_norm (models/gpt2_model.py:27):
    biased variance over the last dim."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    y = (x - mu) * torch.rsqrt(var + eps) * w
    return y + b if b is not None else y


def rms_norm(x: Tensor, w: Tensor, eps: float = 1e-6) -> Tensor:
    """models/brainformer.py:221-232 / models/simple_mae:181-192."""
    xf = x.float()
    y = xf * torch.rsqrt((xf * xf).mean(-1, keepdim=True) + eps)
    return y.type_as(x) * w


def linear(x: Tensor, w: Tensor, b: Optional[Tensor] = None) -> Tensor:
    y = x @ w.t()
    return y + b if b is not None else y


def sdpa(q: Tensor, k: Tensor, v: Tensor, mask: Optional[Tensor]) -> Tensor:
    """F.scaled_dot_product_attention(q,k,v,attn_mask) semantics (models/brainformer.py:168,215;
    models/gpt2_model.py:64): q,k,v [B,H,T,hd]; bool mask True = attend; scale 1/sqrt(hd)."""
    s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(q.shape[-1]))
    if mask is not None:
        s = s.masked_fill(~mask, float("-inf"))
    p = torch.softmax(s, dim=-1)
    return p @ v


def silu(x: Tensor) -> Tensor:
    return x * torch.sigmoid(x)


def gelu_erf(x: Tensor) -> Tensor:
    """nn.GELU() default = exact erf form (models/gpt2_model.py:83)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def cross_entropy(logits: Tensor, target: Tensor, ignore_index: int = -100) -> Tensor:
    """F.cross_entropy(logits[N,V], target[N], ignore_index) with mean over non-ignored rows
    (models/gpt2_model.py:210; notebooks_trainer/train_brainformer.ipynb cell 3)."""
    lse = torch.logsumexp(logits.float(), dim=-1)
    valid = target != ignore_index
    tgt = torch.where(valid, target, torch.zeros_like(target))
    picked = logits.float().gather(-1, tgt[:, None])[:, 0]
    nll = (lse - picked) * valid
    return nll.sum() / valid.sum()


# --------------------------------------------------------------------------- brainformer
def swiglu_mlp(sd, p: str, x: Tensor) -> Tensor:
    """models/brainformer.py:115-124: w2(silu(w1 x) * w3 x), no bias."""
    return linear(silu(linear(x, sd[p + "w1.weight"])) * linear(x, sd[p + "w3.weight"]),
                  sd[p + "w2.weight"])


def self_attention(sd, p: str, x: Tensor, n_heads: int, head_dim: int,
                   mask: Optional[Tensor], ang: Optional[Tensor]) -> Tensor:
    """models/brainformer.py:147-173."""
    B, T, _ = x.shape
    q = linear(x, sd[p + "qw.weight"]).view(B, T, n_heads, head_dim)
    k = linear(x, sd[p + "kw.weight"]).view(B, T, n_heads, head_dim)
    v = linear(x, sd[p + "vw.weight"]).view(B, T, n_heads, head_dim)
    if ang is not None:
        q, k = apply_rope(q, ang), apply_rope(k, ang)
    if mask is not None:
        mask = mask[..., -T:, -T:]
    o = sdpa(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), mask)
    o = o.transpose(1, 2).reshape(B, T, n_heads * head_dim)
    return linear(o, sd[p + "project.weight"])


def cross_attention(sd, p: str, x: Tensor, ctx: Tensor, n_heads: int, head_dim: int,
                    mask: Optional[Tensor] = None) -> Tensor:
    """models/brainformer.py:198-219 (no RoPE, mask always None in use)."""
    B, T, _ = x.shape
    N = ctx.shape[1]
    q = linear(x, sd[p + "qw.weight"]).view(B, T, n_heads, head_dim).transpose(1, 2)
    k = linear(ctx, sd[p + "kw.weight"]).view(B, N, n_heads, head_dim).transpose(1, 2)
    v = linear(ctx, sd[p + "vw.weight"]).view(B, N, n_heads, head_dim).transpose(1, 2)
    if mask is not None:
        mask = mask[..., -T:, -N:]
    o = sdpa(q, k, v, mask).transpose(1, 2).reshape(B, T, n_heads * head_dim)
    return linear(o, sd[p + "project.weight"])


def block(sd, p: str, x: Tensor, cfg, mask, ang) -> Tensor:
    """models/brainformer.py:242-245."""
    h = layer_norm(x, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"])
    x = x + self_attention(sd, p + "attn.", h, cfg.n_heads, cfg.head_dim, mask, ang)
    h = layer_norm(x, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"])
    return x + swiglu_mlp(sd, p + "mlp.", h)


def cross_block(sd, p: str, x: Tensor, ctx: Tensor, cfg, sa_mask, ca_mask, sa_ang) -> Tensor:
    """models/brainformer.py:257-268."""
    h = layer_norm(x, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"])
    x = x + cross_attention(sd, p + "cross_attn.", h, ctx, cfg.n_heads, cfg.head_dim, ca_mask)
    h = layer_norm(x, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"])
    x = x + swiglu_mlp(sd, p + "mlp.", h)
    return block(sd, p + "sa_block.", x, cfg, sa_mask, sa_ang)


def to_patches(x: Tensor, patch: int) -> Tensor:
    """Rearrange('b (t p1) c -> b (t c) p1') (models/brainformer.py:282)."""
    B, T, C = x.shape
    return x.view(B, T // patch, patch, C).permute(0, 1, 3, 2).reshape(B, (T // patch) * C, patch)


def encoder_forward(sd, p: str, x: Tensor, cfg) -> Tensor:
    """models/brainformer.py:333-352.  ``p`` is the key prefix of the Encoder ('encoder.')."""
    n_t = cfg.window_size // cfg.patch_size
    block_size = n_t * cfg.n_electrodes
    tok = to_patches(x, cfg.patch_size)
    n_tokens = tok.shape[1]
    h = linear(tok, sd[p + "transformer.emb.weight"], sd[p + "transformer.emb.bias"])
    space = sd[p + "space_embedding"].repeat(1, n_t, 1)          # :321-327
    h = h + space[:, -n_tokens:]
    mask = block_causal_mask(block_size, cfg.n_electrodes)
    ang = rope_angles(cfg.head_dim, block_size, cfg.rope_theta)
    for i in range(cfg.n_layers):
        h = block(sd, f"{p}transformer.h.{i}.", h, cfg, mask, ang)
    return layer_norm(h, sd[p + "transformer.ln_f.weight"], sd[p + "transformer.ln_f.bias"])


def perceiver_forward(sd, x: Tensor, cfg, head: str, p: str = "") -> Tensor:
    """encoder -> learnable queries -> CrossBlocks -> ln_f -> head (models/brainformer.py:532-552;
    notebook variants use head='to_words').  Returns pred/logits [B, M, output_dim]."""
    B = x.shape[0]
    ctx = encoder_forward(sd, p + "encoder.", x, cfg.encoder)
    q = sd[p + "learnable_queries"].expand(B, cfg.n_output_tokens, -1)
    ang = rope_angles(cfg.head_dim, cfg.n_output_tokens, cfg.rope_theta)
    for i in range(cfg.n_layers):
        q = cross_block(sd, f"{p}perceiver.h.{i}.", q, ctx, cfg, None, None, ang)
    q = layer_norm(q, sd[p + "perceiver.ln_f.weight"], sd[p + "perceiver.ln_f.bias"])
    return linear(q, sd[f"{p}perceiver.{head}.weight"], sd[f"{p}perceiver.{head}.bias"])


def brainformer_l1(sd, x: Tensor, targets: Optional[Tensor], cfg):
    """models/brainformer.py:532-558 (file class): (loss, pred), L1 mean loss."""
    pred = perceiver_forward(sd, x, cfg, "to_motion")
    if targets is None:
        return None, pred
    return (pred - targets).abs().mean(), pred


def brainformer_ce(sd, x: Tensor, targets: Optional[Tensor], cfg):
    """notebooks_trainer/train_brainformer.ipynb cell 3: logits + CE(ignore_index=-100 default)."""
    logits = perceiver_forward(sd, x, cfg, "to_words")
    if targets is None:
        return None, logits
    return cross_entropy(logits.reshape(-1, logits.shape[-1]), targets.reshape(-1)), logits


# --------------------------------------------------------------------------- GPT-2
def sdpa_dropout(q: Tensor, k: Tensor, v: Tensor, mask: Optional[Tensor], keep_scaled: Tensor) -> Tensor:
    """F.scaled_dot_product_attention(..., dropout_p=p) in training mode (models/gpt2_model.py:64) with the draw given: the softmax over
    the visible keys, then keep_scaled = keep / (1 - p) applied to the probabilities ([B, H, T, T])."""
    s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(q.shape[-1]))
    if mask is not None:
        s = s.masked_fill(~mask, float("-inf"))
    return (torch.softmax(s, dim=-1) * keep_scaled) @ v


def gpt_attention(sd, p: str, x: Tensor, n_head: int, masks=None) -> Tensor:
    """models/gpt2_model.py:52-76 (fused c_attn, causal).  masks: None (dropout 0 / eval) or an iterator of keep / (1 - p) tensors, one
    per dropout application in call order (attention probabilities :64, resid_dropout :75)."""
    B, T, C = x.shape
    qkv = linear(x, sd[p + "c_attn.weight"], sd.get(p + "c_attn.bias"))
    q, k, v = qkv.split(C, dim=2)
    hs = C // n_head
    q = q.view(B, T, n_head, hs).transpose(1, 2)
    k = k.view(B, T, n_head, hs).transpose(1, 2)
    v = v.view(B, T, n_head, hs).transpose(1, 2)
    causal = torch.ones(T, T, dtype=torch.bool).tril()
    if masks is None:
        y = sdpa(q, k, v, causal).transpose(1, 2).reshape(B, T, C)
        return linear(y, sd[p + "c_proj.weight"], sd.get(p + "c_proj.bias"))
    y = sdpa_dropout(q, k, v, causal, next(masks)).transpose(1, 2).reshape(B, T, C)
    return linear(y, sd[p + "c_proj.weight"], sd.get(p + "c_proj.bias")) * next(masks)


def gpt_block(sd, p: str, x: Tensor, n_head: int, masks=None) -> Tensor:
    """models/gpt2_model.py:103-106, 87-92 (MLP dropout :91 when masks are given)."""
    h = layer_norm(x, sd[p + "ln_1.weight"], sd.get(p + "ln_1.bias"))
    x = x + gpt_attention(sd, p + "attn.", h, n_head, masks)
    h = layer_norm(x, sd[p + "ln_2.weight"], sd.get(p + "ln_2.bias"))
    h = gelu_erf(linear(h, sd[p + "mlp.c_fc.weight"], sd.get(p + "mlp.c_fc.bias")))
    h = linear(h, sd[p + "mlp.c_proj.weight"], sd.get(p + "mlp.c_proj.bias"))
    return x + (h if masks is None else h * next(masks))


def gpt_forward(sd, idx: Tensor, prefix: Optional[Tensor], targets: Optional[Tensor], cfg,
                p: str = "", masks=None):
    """models/gpt2_model.py:178-216.  lm_head weight is tied to wte (:138).  masks: training mode with dropout > 0 — an iterator of
    keep / (1 - p) tensors in the order the reference applies its dropouts (transformer.drop :190, then per block the attention
    probabilities :64, resid_dropout :75, the MLP's :91)."""
    wte = sd[p + "transformer.wte.weight"]
    t_words = idx.shape[1]
    tok = wte[idx]
    if prefix is not None:
        tok = torch.cat([prefix, tok], dim=1)
    t_full = tok.shape[1]
    x = tok + sd[p + "transformer.wpe.weight"][:t_full]
    if masks is not None:
        x = x * next(masks)
    for i in range(cfg.n_layer):
        x = gpt_block(sd, f"{p}transformer.h.{i}.", x, cfg.n_head, masks)
    x = x[:, -t_words:]
    x = layer_norm(x, sd[p + "transformer.ln_f.weight"], sd.get(p + "transformer.ln_f.bias"))
    if targets is not None:
        logits = linear(x, wte)
        loss = cross_entropy(logits[:, :-1].reshape(-1, logits.shape[-1]),
                             targets[:, 1:].reshape(-1), -100)
    else:
        logits = linear(x[:, [-1], :], wte)
        loss = None
    return loss, logits


def franky_forward(sd, x: Tensor, targets: Tensor, bcfg, gcfg):
    """notebooks_trainer/franky_baseline_gpt2.ipynb cell 4 (Franky.forward): brain features as
    GPT prefix; -100 in idx replaced by 50256.  Keys: brain_model.* / llm_model.*"""
    feats = perceiver_forward(sd, x, bcfg, "to_words", p="brain_model.")
    idx = targets.clone()
    idx[idx == -100] = 50256
    return gpt_forward(sd, idx, feats, targets, gcfg, p="llm_model.")


# --------------------------------------------------------------------------- key/shape tables
def encoder_shapes(cfg, p: str = "encoder.") -> Dict[str, tuple]:
    d, hd, H = cfg.dim, cfg.head_dim * cfg.n_heads, cfg.hidden_dim
    s = {p + "space_embedding": (1, cfg.n_electrodes, d),
         p + "transformer.emb.weight": (d, cfg.patch_size), p + "transformer.emb.bias": (d,),
         p + "transformer.ln_f.weight": (d,), p + "transformer.ln_f.bias": (d,)}
    for i in range(cfg.n_layers):
        s.update(block_shapes(f"{p}transformer.h.{i}.", d, hd, H))
    return s


def block_shapes(q: str, d: int, hd: int, H: int, attn: str = "attn") -> Dict[str, tuple]:
    return {q + "ln_1.weight": (d,), q + "ln_1.bias": (d,), q + "ln_2.weight": (d,), q + "ln_2.bias": (d,),
            q + f"{attn}.qw.weight": (hd, d), q + f"{attn}.kw.weight": (hd, d), q + f"{attn}.vw.weight": (hd, d),
            q + f"{attn}.project.weight": (d, hd),
            q + "mlp.w1.weight": (H, d), q + "mlp.w2.weight": (d, H), q + "mlp.w3.weight": (H, d)}


def brainformer_shapes(cfg, head: str, p: str = "") -> Dict[str, tuple]:
    """State-dict keys of BrainFormer / BrainEncoder, minus the attn_mask buffer."""
    s = {p + "learnable_queries": (1, cfg.n_output_tokens, cfg.dim)}
    s.update(encoder_shapes(cfg.encoder, p + "encoder."))
    d, hd, H = cfg.dim, cfg.head_dim * cfg.n_heads, cfg.hidden_dim
    for i in range(cfg.n_layers):
        q = f"{p}perceiver.h.{i}."
        s.update(block_shapes(q + "sa_block.", d, hd, H))
        s.update(block_shapes(q, d, hd, H, attn="cross_attn"))
    s.update({p + "perceiver.ln_f.weight": (d,), p + "perceiver.ln_f.bias": (d,),
              f"{p}perceiver.{head}.weight": (cfg.output_dim, d), f"{p}perceiver.{head}.bias": (cfg.output_dim,)})
    return s


def gpt_shapes(cfg, p: str = "") -> Dict[str, tuple]:
    """State-dict keys of GPT (lm_head.weight omitted: tied to transformer.wte.weight)."""
    d = cfg.n_embd
    s = {p + "transformer.wte.weight": (cfg.vocab_size, d), p + "transformer.wpe.weight": (cfg.block_size, d),
         p + "transformer.ln_f.weight": (d,)}
    if cfg.bias:
        s[p + "transformer.ln_f.bias"] = (d,)
    for i in range(cfg.n_layer):
        q = f"{p}transformer.h.{i}."
        s.update({q + "ln_1.weight": (d,), q + "ln_2.weight": (d,),
                  q + "attn.c_attn.weight": (3 * d, d), q + "attn.c_proj.weight": (d, d),
                  q + "mlp.c_fc.weight": (4 * d, d), q + "mlp.c_proj.weight": (d, 4 * d)})
        if cfg.bias:
            s.update({q + "ln_1.bias": (d,), q + "ln_2.bias": (d,), q + "attn.c_attn.bias": (3 * d,),
                      q + "attn.c_proj.bias": (d,), q + "mlp.c_fc.bias": (4 * d,), q + "mlp.c_proj.bias": (d,)})
    return s


# --------------------------------------------------------------------------- MAE
def mae_shapes(cfg, p: str = "") -> Dict[str, tuple]:
    """State-dict keys of brainformer.MAE (models/brainformer.py:354-374), minus the attn_mask buffer."""
    s = encoder_shapes(cfg, p + "encoder.")
    d, hd, H = cfg.dim, cfg.head_dim * cfg.n_heads, cfg.hidden_dim
    for i in range(cfg.n_dec_layers):
        s.update(block_shapes(f"{p}decoder.h.{i}.", d, hd, H))
    n_tok = (cfg.window_size // cfg.patch_size) * cfg.n_electrodes
    s.update({p + "mask_token": (cfg.dim,), p + "decoder_pos_emb.weight": (n_tok, cfg.decoder_dim),
              p + "to_signals.weight": (cfg.patch_size, cfg.decoder_dim), p + "to_signals.bias": (cfg.patch_size,)})
    return s


def mae_forward(sd, x: Tensor, cfg, masked: Tensor, unmasked: Tensor, p: str = ""):
    """models/brainformer.py:415-473 with the random index sets given (they are inputs of the parity fixture).
    Returns (mse loss, predictions for the masked patches [B, n_masked, patch])."""
    B = x.shape[0]
    n_t = cfg.window_size // cfg.patch_size
    n_tok = n_t * cfg.n_electrodes
    tok = to_patches(x, cfg.patch_size)                                   # [B, N, P]
    br = torch.arange(B)[:, None]
    space = sd[p + "encoder.space_embedding"].repeat(1, n_t, 1).expand(B, -1, -1)[br, unmasked]
    ang = rope_angles(cfg.head_dim, n_tok, cfg.rope_theta).expand(B, -1, -1)[br, unmasked]      # per-sample rows (:430-434)
    full = block_causal_mask(n_tok, cfg.n_electrodes)
    sub = full.expand(B, -1, -1)[torch.arange(B)[:, None, None], unmasked[..., None], unmasked[:, None, :]][:, None]
    h = linear(tok[br, unmasked], sd[p + "encoder.transformer.emb.weight"], sd[p + "encoder.transformer.emb.bias"]) + space
    for i in range(cfg.n_layers):
        h = block(sd, f"{p}encoder.transformer.h.{i}.", h, cfg, sub, ang)
    h = layer_norm(h, sd[p + "encoder.transformer.ln_f.weight"], sd[p + "encoder.transformer.ln_f.bias"])
    dec = torch.zeros(B, n_tok, cfg.decoder_dim)
    dec = dec.index_put((br, unmasked), h)
    dec = dec.index_put((br, masked), sd[p + "mask_token"].expand(B, masked.shape[1], -1))
    # quirk kept from the reference (:459-460): positional rows are added in CONCATENATION order, not scattered
    dec = dec + sd[p + "decoder_pos_emb.weight"][torch.cat([unmasked, masked], 1)]
    for i in range(cfg.n_dec_layers):
        dec = block(sd, f"{p}decoder.h.{i}.", dec, cfg, None, None)
    pred = linear(dec[br, masked], sd[p + "to_signals.weight"], sd[p + "to_signals.bias"])
    loss = ((pred - tok[br, masked]) ** 2).mean()
    return loss, pred


# --------------------------------------------------------------------------- SimpleMAE (BASELINE.json configs[4])
def simple_encoder_config(**kw) -> SimpleNamespace:
    """notebooks/simple_mae.ipynb cell 1 (SimpleEncoderConfig)."""
    d = dict(block_size=768, patch_size=128, n_layers=6, dim=256, hidden_dim=1024, head_dim=32, n_heads=4, n_kv_heads=4,
             rope_theta=10000)
    d.update(kw)
    return SimpleNamespace(**d)


def simple_mae_config(**kw) -> SimpleNamespace:
    """notebooks/simple_mae.ipynb cell 1 (SimpleMAEConfig)."""
    d = dict(n_layers=2, dim=256, hidden_dim=1024, head_dim=32, n_heads=8, n_kv_heads=8, rope_theta=10000)
    d.update(kw)
    return SimpleNamespace(**d)


def rms_block_shapes(q: str, cfg) -> Dict[str, tuple]:
    d, hd, H = cfg.dim, cfg.head_dim * cfg.n_heads, cfg.hidden_dim
    return {q + "ln_1.weight": (d,), q + "ln_2.weight": (d,), q + "attn.qw.weight": (hd, d), q + "attn.kw.weight": (hd, d),
            q + "attn.vw.weight": (hd, d), q + "attn.project.weight": (d, hd), q + "mlp.w1.weight": (H, d),
            q + "mlp.w2.weight": (d, H), q + "mlp.w3.weight": (H, d)}


def simple_mae_shapes(ecfg, mcfg) -> Dict[str, tuple]:
    s = {"encoder.transformer.emb.weight": (ecfg.dim, ecfg.patch_size), "encoder.transformer.emb.bias": (ecfg.dim,),
         "encoder.transformer.ln_f.weight": (ecfg.dim,), "encoder.transformer.ln_f.bias": (ecfg.dim,),
         "decoder.emb.weight": (mcfg.dim, ecfg.dim), "decoder.emb.bias": (mcfg.dim,), "mask_token": (mcfg.dim,),
         "decoder_pos_emb.weight": (ecfg.block_size, mcfg.dim), "to_signals.weight": (ecfg.patch_size, mcfg.dim),
         "to_signals.bias": (ecfg.patch_size,)}
    for i in range(ecfg.n_layers):
        s.update(rms_block_shapes(f"encoder.transformer.h.{i}.", ecfg))
    for i in range(mcfg.n_layers):
        s.update(rms_block_shapes(f"decoder.h.{i}.", mcfg))
    return s


def rms_block(sd, p: str, x: Tensor, cfg, mask, ang) -> Tensor:
    """models/simple_mae:194-205 (RMSNorm eps 1e-6, no bias)."""
    h = rms_norm(x, sd[p + "ln_1.weight"])
    x = x + self_attention(sd, p + "attn.", h, cfg.n_heads, cfg.head_dim, mask, ang)
    h = rms_norm(x, sd[p + "ln_2.weight"])
    return x + swiglu_mlp(sd, p + "mlp.", h)


def sdpa_zero_fully_masked(q, k, v, mask):
    """torch >= 2.1 CPU SDPA gives 0 (not NaN) for query rows whose keys are all masked (padding rows)."""
    s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(q.shape[-1]))
    s = s.masked_fill(~mask, float("-inf"))
    p = torch.softmax(s, dim=-1)
    return torch.nan_to_num(p, nan=0.0) @ v


def simple_mae_forward(sd, x: Tensor, ecfg, mcfg, masked: Tensor, unmasked: Tensor):
    """models/simple_mae:338-407 with the random index sets given.  Returns (loss, predictions at the masked frames)."""
    B, T, _ = x.shape
    br = torch.arange(B)[:, None]
    valid = ~(x == 0).all(dim=2)
    m_all = (valid[:, None, :] & valid[:, :, None])[:, None]
    vu = valid[br, unmasked]
    m_u = (vu[:, None, :] & vu[:, :, None])[:, None]
    ang = rope_angles(ecfg.head_dim, ecfg.block_size, ecfg.rope_theta).expand(B, -1, -1)[br, unmasked]
    global sdpa
    keep = sdpa
    sdpa = sdpa_zero_fully_masked
    try:
        h = linear(x[br, unmasked], sd["encoder.transformer.emb.weight"], sd["encoder.transformer.emb.bias"])
        for i in range(ecfg.n_layers):
            h = rms_block(sd, f"encoder.transformer.h.{i}.", h, ecfg, m_u, ang)
        h = layer_norm(h, sd["encoder.transformer.ln_f.weight"], sd["encoder.transformer.ln_f.bias"])
        h = linear(h, sd["decoder.emb.weight"], sd["decoder.emb.bias"])
        dec = torch.zeros(B, T, mcfg.dim).index_put((br, unmasked), h)
        dec = dec.index_put((br, masked), sd["mask_token"].expand(B, masked.shape[1], -1))
        dec = dec + sd["decoder_pos_emb.weight"][torch.cat([unmasked, masked], 1)]
        for i in range(mcfg.n_layers):
            dec = rms_block(sd, f"decoder.h.{i}.", dec, mcfg, m_all, None)
    finally:
        sdpa = keep
    pred_all = linear(dec, sd["to_signals.weight"], sd["to_signals.bias"])
    pred, real = pred_all[br, masked], x[br, masked]
    w = valid[br, masked]
    loss = (((pred - real) ** 2) * w[..., None]).sum() / (w.sum() * x.shape[2])
    return loss, pred
