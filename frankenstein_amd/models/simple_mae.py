"""MI355X-native SimpleMAE (the reference's ``models/simple_mae`` file, which has no .py suffix, and its config
dataclasses that exist only in ``notebooks/simple_mae.ipynb`` cell 1): per-frame tokens (Linear(n_channels -> dim)),
RMSNorm blocks with per-sample RoPE rows, padding-aware attention (all-zero frames are padding), masked-token MSE
over the non-padded masked frames.  BASELINE.json configs[4].

Reference map: RMSNorm models/simple_mae:181-192, Block :194-205, create_attention_mask_from_padding :228-236,
SimpleEncoder :238-297, SimpleMAE :301-407.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.nn as nn

from .. import engine as E
from .. import kernels as K
from ..kernels import Mask
from .brainformer import (MLP, CausalCrossAttention, CausalSelfAttention, CrossBlock, LayerNorm, Linear, RMSNorm,  # noqa: F401
                          Serializable, _prep, apply_rope, build_advanced_causal_mask, build_complex_rope_cache)
# (the reference file carries private copies of these building blocks, models/simple_mae:1-227; here they are the shared ones)


@dataclass
class SimpleEncoderConfig(Serializable):
    block_size: int = 768
    patch_size: int = 128
    n_layers: int = 6
    dim: int = 256
    hidden_dim: int = 1024
    head_dim: int = 32
    n_heads: int = 4
    n_kv_heads: int = 4
    rope_theta: int = 10000


@dataclass
class SimpleMAEConfig(Serializable):
    n_layers: int = 2
    dim: int = 256
    hidden_dim: int = 1024
    head_dim: int = 32
    n_heads: int = 8
    n_kv_heads: int = 8
    rope_theta: int = 10000


def create_attention_mask_from_padding(x, pad_value=0) -> Mask:
    """Frames whose channels all equal pad_value are padding: mask[b, i, j] = valid[b, i] & valid[b, j], returned as the
    analytic key-padding mask the attention kernels take (never materialised as [B, T, T])."""
    valid = ~(x == pad_value).all(dim=2)
    return Mask.from_padding(valid, valid)


class Block(nn.Module):
    """Pre-RMSNorm residual block (eps 1e-6, no bias)."""

    def __init__(self, config):
        super().__init__()
        self.ln_1 = RMSNorm(config.dim)
        self.attn = CausalSelfAttention(config)
        self.ln_2 = RMSNorm(config.dim)
        self.mlp = MLP(config)

    def forward(self, x, attn_mask=None, rope=None, kv_cache=False):
        x = self.attn.branch(_prep(x), attn_mask, rope, self.ln_1, True)
        return self.mlp.branch(x, self.ln_2, True)


class SimpleEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.transformer = nn.ModuleDict(dict(
            emb=Linear(config.patch_size, config.dim),
            h=nn.ModuleList([Block(config) for _ in range(config.n_layers)]),
            ln_f=LayerNorm(config.dim),
        ))
        self.precompute_rope_cash = build_complex_rope_cache(dim=config.head_dim, seq_len=config.block_size,
                                                             theta=config.rope_theta)
        self.attn_mask = None      # the reference keeps an all-True [block, block] tensor here; None == attend to everything
        print("Encoder: number of parameters: %.2fM" % (self.get_num_params() / 1e6,))

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

    def get_num_params(self):
        return sum(p.numel() for p in self.parameters())

    def forward(self, x, attn_mask=None, rope_cache=None):
        """x [B, T, C] frames (C = patch_size).  The reference slices the 2-D cache as rope[:T] (models/simple_mae:40)."""
        attn_mask = self.attn_mask if attn_mask is None else attn_mask
        if rope_cache is None:
            rope_cache = self.rope_cache[: x.shape[1]]
        h = self.transformer.emb(x)
        for block in self.transformer.h:
            h = block(h, attn_mask=attn_mask, rope=rope_cache)
        return self.transformer.ln_f(h)


class SimpleMAE(nn.Module):
    def __init__(self, encoder_config, mae_config):
        super().__init__()
        self.encoder_config = encoder_config
        self.encoder = SimpleEncoder(encoder_config)
        self.dim = mae_config.dim
        self.decoder = nn.ModuleDict(dict(
            emb=Linear(encoder_config.dim, mae_config.dim),
            h=nn.ModuleList([Block(mae_config) for _ in range(mae_config.n_layers)]),
        ))
        self.mask_token = nn.Parameter(torch.randn(mae_config.dim))
        self.decoder_pos_emb = nn.Embedding(encoder_config.block_size, mae_config.dim)
        self.to_signals = Linear(mae_config.dim, encoder_config.patch_size)
        print("MAE: number of parameters: %.2fM" % (self.get_num_params() / 1e6))

    def get_num_params(self, non_embedding=True):
        return sum(p.numel() for p in self.parameters())

    def get_masking_indices(self, masking_ratio, x):
        b, n_tokens, _ = x.shape
        num_masked = int(masking_ratio * n_tokens)
        rand_indices = torch.rand(b, n_tokens, device=x.device).argsort(dim=-1)
        masked, unmasked = rand_indices[:, :num_masked], rand_indices[:, num_masked:]
        return torch.sort(masked, dim=1)[0], torch.sort(unmasked, dim=1)[0]

    def forward(self, x, targets=None, date_info=None, masking_ratio=0.75, return_preds=False, indices=None):
        """x [B, T, C] fp32 frames.  ``indices=(masked, unmasked)`` supplies the random index sets (parity tests)."""
        B, T, Cn = x.shape
        xin = x if x.dtype == torch.float32 else x.float()
        masked, unmasked = self.get_masking_indices(masking_ratio, xin) if indices is None else indices
        masked, unmasked = masked.contiguous(), unmasked.contiguous()
        valid = ~(xin == 0).all(dim=2)                                           # [B, T] index prep (host-side glue)
        valid_u = torch.gather(valid, 1, unmasked)
        mask_u, mask_all = Mask.from_padding(valid_u, valid_u), Mask.from_padding(valid, valid)
        table = torch.view_as_real(self.encoder.rope_cache).reshape(self.encoder_config.block_size, -1)
        rope_u = K.gather_rows(table.contiguous(), unmasked).view(B, unmasked.shape[1], -1, 2)
        xc = E.to_compute(xin)                                                   # frames in the compute dtype
        tokens = self.encoder(K.gather_rows(xc, unmasked), attn_mask=mask_u, rope_cache=rope_u)
        dec = E.AssembleDecoder.apply(self.decoder.emb(tokens), self.mask_token, self.decoder_pos_emb.weight, unmasked, masked)
        for block in self.decoder.h:
            dec = block(dec, mask_all)
        pred = self.to_signals(E.GatherRows.apply(dec, masked))                  # [B, n_masked, C]
        target = K.gather_rows(xc, masked)
        w = torch.gather(valid, 1, masked).to(torch.float32).reshape(-1).contiguous()   # loss only on non-padded frames
        loss = E.mse_loss(pred, target, w)
        if return_preds:
            with torch.no_grad():
                rec = torch.zeros_like(xin)
                K.scatter_rows_(rec, unmasked, K.gather_rows(xin.contiguous(), unmasked))
                K.scatter_rows_(rec, masked, pred.detach().float().contiguous())
                bm = torch.zeros_like(xin)
                K.scatter_rows_(bm, masked, torch.ones_like(pred, dtype=torch.float32))
            return loss, rec, bm
        return loss, None
