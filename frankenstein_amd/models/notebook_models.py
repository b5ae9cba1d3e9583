"""The model classes the reference defines only inside its trainer notebooks, on the HIP kernels:

  BrainEncoder   notebooks_trainer/franky_baseline_gpt2.ipynb cell 3  (encoder + perceiver -> ``to_words`` features)
  Franky         notebooks_trainer/franky_baseline_gpt2.ipynb cell 4  (brain features as GPT prefix, CE loss)
  BrainFormerCE  notebooks_trainer/train_brainformer.ipynb cell 3     (``BrainFormer`` there: vocab head + CE)

Same constructor / forward signatures and state-dict keys as the notebook classes, so a notebook can
``from frankenstein_amd.models.notebook_models import BrainEncoder, Franky`` instead of defining them inline.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import engine as E
from .brainformer import BrainFormer as _FileBrainFormer
from .brainformer import Config


class BrainEncoder(_FileBrainFormer):
    """forward(x) -> logits/features [B, n_output_tokens, output_dim] (no loss)."""
    config = Config
    head_name = 'to_words'

    def forward(self, x, targets=None, date_info=None):
        return self.features(x)


class BrainFormerCE(_FileBrainFormer):
    """forward(x, targets) -> (CE loss over all output tokens with ignore_index=-100, logits)."""
    config = Config
    head_name = 'to_words'

    def forward(self, x, targets=None, date_info=None):
        if targets is not None and getattr(self, "fuse_head_loss", False):
            # loss only (train_utils.enable_fused_head_loss): perceiver.ln_f -> to_words -> CE without the [B, tokens, V] logits
            q = self.queries_out(x)
            head, ln = self.perceiver[self.head_name], self.perceiver.ln_f
            return E.head_cross_entropy(q, ln.weight, ln.bias, head.weight, head.bias, targets, ln.eps, -100, getattr(self, "head_chunk", 8192)), None
        logits = self.features(x)
        if targets is None:
            return None, logits
        return E.cross_entropy(logits, targets, -100), logits


class Franky(nn.Module):
    """Brain features -> GPT prefix; targets' -100 padding is replaced by token 50256 for the input ids."""

    def __init__(self, brain_model, llm_model, tokenizer=None):
        super().__init__()
        self.brain_model = brain_model
        self.llm_model = llm_model
        self.tokenizer = tokenizer
        print("Full Franky: number of parameters: %.2fM" % (self.get_num_params() / 1e6,))

    def get_num_params(self):
        return sum(p.numel() for p in self.parameters())

    @property
    def dtype(self) -> torch.dtype:
        return next(self.parameters()).dtype

    @property
    def device(self) -> torch.device:
        return next(self.parameters()).device

    def forward(self, x, targets=None, date_info=None):
        features = self.brain_model(x)
        new_idx = targets.clone()
        new_idx[new_idx == -100] = 50256
        return self.llm_model.forward(idx=new_idx, prefix=features, targets=targets)

    @torch.no_grad()
    def generate(self, x, max_new_tokens=25, temperature=1.0, top_k=10, eot=50256):
        """x: numpy [T, C].  Returns generated token ids (the notebook's version is unfinished; this one runs)."""
        xin = torch.from_numpy(x[None]).to(self.device).float()
        prefix = self.brain_model(xin)
        ids = torch.full((1, 1), eot, dtype=torch.long, device=self.device)
        return self.llm_model.generate(ids, max_new_tokens, prefix=prefix, temperature=temperature, top_k=top_k)
