"""MI355X-native GPT-2 decoder with the reference's ``models/gpt2_model.py`` surface: ``GPTConfig``,
``GPT(config).forward(idx, prefix=None, targets=None) -> (loss, logits)`` with brain-feature *prefix*
embeddings, tied ``lm_head``/``wte``, and the same state-dict keys.  Forward/backward run on the HIP
kernels (fused c_attn GEMM + causal flash attention + GELU MLP + CE).

Reference map: LayerNorm models/gpt2_model.py:18-27, CausalSelfAttention :29-76, MLP :78-92,
Block :94-106, GPTConfig :108-116, GPT :118-216 (+ crop_block_size :218-227, from_pretrained :229-284,
configure_optimizers :286-310, estimate_mfu :312-326, generate :328-353).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn as nn

from .. import engine as E
from .. import kernels as K
from ..kernels import MASK_CAUSAL, Mask
from .brainformer import Linear, _prep

CAUSAL = Mask(MASK_CAUSAL)


class LayerNorm(nn.Module):
    """LayerNorm with an optional bias (eps 1e-5)."""

    def __init__(self, ndim, bias):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(ndim))
        self.bias = nn.Parameter(torch.zeros(ndim)) if bias else None
        self.eps = 1e-5

    def forward(self, input):
        return E.LayerNormFn.apply(_prep(input), self.weight, self.bias, self.eps, K.NORM_LAYER)


class CausalSelfAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        assert config.n_embd % config.n_head == 0
        self.c_attn = Linear(config.n_embd, 3 * config.n_embd, bias=config.bias)
        self.c_proj = Linear(config.n_embd, config.n_embd, bias=config.bias)
        self.n_head = config.n_head
        self.n_embd = config.n_embd
        self.dropout = config.dropout

    def branch(self, x, ln, residual: bool):
        spec = (self.n_head, self.n_embd // self.n_head, CAUSAL, None, residual, 0.0 if ln is None else ln.eps, K.NORM_LAYER)
        if self.dropout and self.training:          # SDPA dropout_p + resid_dropout (models/gpt2_model.py:64,75): two sites
            spec = spec + (E.drop_spec(self.dropout, x.device, 2),)
        return E.AttnBranch.apply(x, None if ln is None else ln.weight, None if ln is None else ln.bias,
                                  self.c_proj.weight, self.c_proj.bias, self.c_attn.bias, spec, self.c_attn.weight)

    def forward(self, x):
        return self.branch(_prep(x), None, False)


class MLP(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.c_fc = Linear(config.n_embd, 4 * config.n_embd, bias=config.bias)
        self.gelu = nn.GELU()
        self.c_proj = Linear(4 * config.n_embd, config.n_embd, bias=config.bias)
        self.dropout = nn.Dropout(config.dropout)

    def branch(self, x, ln, residual: bool):
        spec = (residual, 0.0 if ln is None else ln.eps, K.NORM_LAYER)
        if self.dropout.p and self.training:        # models/gpt2_model.py:91
            spec = spec + (E.drop_spec(self.dropout.p, x.device, 1),)
        return E.MlpBranch.apply(x, None if ln is None else ln.weight, None if ln is None else ln.bias,
                                 self.c_fc.weight, self.c_fc.bias, None, self.c_proj.weight, self.c_proj.bias, spec)

    def forward(self, x):
        return self.branch(_prep(x), None, False)


class Block(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.ln_1 = LayerNorm(config.n_embd, bias=config.bias)
        self.attn = CausalSelfAttention(config)
        self.ln_2 = LayerNorm(config.n_embd, bias=config.bias)
        self.mlp = MLP(config)

    def forward(self, x):
        x = self.attn.branch(_prep(x), self.ln_1, True)
        return self.mlp.branch(x, self.ln_2, True)

    @torch.no_grad()
    def forward_cached(self, x, kv, pos):
        """Incremental inference step: x [B, t, d] are the tokens at positions pos..pos+t-1, kv [B, Tmax, 2d] this layer's
        key|value cache (rows < pos filled by earlier calls).  Same kernels as the training forward; attention is the
        causal mask with the query offset pos, so a t = 1 step costs O(pos) instead of re-running the whole sequence
        (the reference re-forwards everything per token, models/gpt2_model.py:336-340)."""
        B, t, d = x.shape
        at, ml = self.attn, self.mlp
        H, D = at.n_head, d // at.n_head
        x2 = x.reshape(B * t, d)
        h, _, _ = K.norm_fwd(x2, self.ln_1.weight.detach(), None if self.ln_1.bias is None else self.ln_1.bias.detach(), self.ln_1.eps)
        qkv = K.gemm_nt(h, E.shadow([at.c_attn.weight]), None if at.c_attn.bias is None else E.shadow([at.c_attn.bias]))
        q3 = qkv.view(B, t, 3 * d)
        for b in range(B):                                   # append k|v rows of the new tokens
            K.copy2d(q3[b, :, d:], kv[b, pos:pos + t])
        k = kv[:, :pos + t, :d].unflatten(-1, (H, D))
        v = kv[:, :pos + t, d:].unflatten(-1, (H, D))
        o, _ = K.attn_fwd(q3[..., :d].unflatten(-1, (H, D)), k, v, Mask(MASK_CAUSAL, q_off=pos))
        x2 = K.gemm_nt(o.view(B * t, d), E.shadow([at.c_proj.weight]), None if at.c_proj.bias is None else E.shadow([at.c_proj.bias]),
                       residual=x2)
        h, _, _ = K.norm_fwd(x2, self.ln_2.weight.detach(), None if self.ln_2.bias is None else self.ln_2.bias.detach(), self.ln_2.eps)
        a = K.gemm_nt(h, E.shadow([ml.c_fc.weight]), None if ml.c_fc.bias is None else E.shadow([ml.c_fc.bias]))
        x2 = K.gemm_nt(K.gelu_fwd(a), E.shadow([ml.c_proj.weight]), None if ml.c_proj.bias is None else E.shadow([ml.c_proj.bias]),
                       residual=x2)
        return x2.view(B, t, d)


def _bias(lin):
    return None if lin.bias is None else E.shadow([lin.bias])


def _block_forward_decode(self, x, kv, pos):
    """One new token per sample with the position in a device int32 (graph-capturable): x [B, d], kv [B, Tmax, 2d]."""
    at, ml = self.attn, self.mlp
    h, _, _ = K.norm_fwd(x, self.ln_1.weight.detach(), None if self.ln_1.bias is None else self.ln_1.bias.detach(), self.ln_1.eps)
    qkv = K.gemm_nt(h, E.shadow([at.c_attn.weight]), _bias(at.c_attn))
    K.kv_append_(qkv, kv, pos)
    o = K.attn_decode(qkv, kv, pos, at.n_head)
    x = K.gemm_nt(o, E.shadow([at.c_proj.weight]), _bias(at.c_proj), residual=x)
    h, _, _ = K.norm_fwd(x, self.ln_2.weight.detach(), None if self.ln_2.bias is None else self.ln_2.bias.detach(), self.ln_2.eps)
    a = K.gemm_nt(h, E.shadow([ml.c_fc.weight]), _bias(ml.c_fc))
    return K.gemm_nt(K.gelu_fwd(a), E.shadow([ml.c_proj.weight]), _bias(ml.c_proj), residual=x)


Block.forward_decode = _block_forward_decode


@dataclass
class GPTConfig:
    block_size: int = 1024
    vocab_size: int = 50304
    n_layer: int = 12
    n_head: int = 12
    n_embd: int = 768
    dropout: float = 0.0
    bias: bool = True


class _GptEmbed(torch.autograd.Function):
    """x[b, t] = (t < t_ctx ? prefix[b, t] : wte[idx[b, t - t_ctx]]) + wpe[t]   (models/gpt2_model.py:183-196)."""

    @staticmethod
    def forward(ctx, idx, prefix, wte, wpe):
        out = K.gpt_embed_fwd(idx.contiguous(), prefix, wte.detach(), wpe.detach(), E.compute_dtype())
        ctx.t_ctx = 0 if prefix is None else prefix.shape[1]
        ctx.shapes = (wte.shape, wpe.shape)
        ctx.save_for_backward(idx)
        return out

    @staticmethod
    def backward(ctx, dx):
        (idx,) = ctx.saved_tensors
        dx = dx.contiguous()
        B, t_full, d = dx.shape
        t_ctx = ctx.t_ctx
        dprefix = None
        if t_ctx:
            dprefix = torch.empty((B, t_ctx, d), dtype=dx.dtype, device=dx.device)
            K.copy2d(dx.view(B, t_full * d)[:, :t_ctx * d], dprefix.view(B, t_ctx * d))
        dwte = torch.zeros(ctx.shapes[0], dtype=torch.float32, device=dx.device)
        K.gpt_embed_bwd_wte(idx.contiguous(), dx, dwte, t_ctx)
        dwpe = torch.zeros(ctx.shapes[1], dtype=torch.float32, device=dx.device)
        K.colsum(dx.view(B, t_full * d), out=dwpe.view(-1)[: t_full * d])
        return None, dprefix, dwte, dwpe


class GPT(nn.Module):
    def __init__(self, config):
        super().__init__()
        assert config.vocab_size is not None
        assert config.block_size is not None
        self.config = config
        self.transformer = nn.ModuleDict(dict(
            wte=nn.Embedding(config.vocab_size, config.n_embd),
            wpe=nn.Embedding(config.block_size, config.n_embd),
            drop=nn.Dropout(config.dropout),
            h=nn.ModuleList([Block(config) for _ in range(config.n_layer)]),
            ln_f=LayerNorm(config.n_embd, bias=config.bias),
        ))
        self.lm_head = Linear(config.n_embd, config.vocab_size, bias=False)
        self.transformer.wte.weight = self.lm_head.weight   # weight tying
        self.apply(self._init_weights)
        for pn, p in self.named_parameters():
            if pn.endswith('c_proj.weight'):
                torch.nn.init.normal_(p, mean=0.0, std=0.02 / math.sqrt(2 * config.n_layer))
        print("number of parameters: %.2fM" % (self.get_num_params() / 1e6,))

    def get_num_params(self, non_embedding=True):
        n_params = sum(p.numel() for p in self.parameters())
        if non_embedding:
            n_params -= self.transformer.wpe.weight.numel()
        return n_params

    @property
    def dtype(self) -> torch.dtype:
        return next(self.parameters()).dtype

    @property
    def device(self) -> torch.device:
        return next(self.parameters()).device

    def _init_weights(self, module):
        if isinstance(module, nn.Linear):
            torch.nn.init.normal_(module.weight, mean=0.0, std=0.02)
            if module.bias is not None:
                torch.nn.init.zeros_(module.bias)
        elif isinstance(module, nn.Embedding):
            torch.nn.init.normal_(module.weight, mean=0.0, std=0.02)

    def forward(self, idx, prefix=None, targets=None):
        t_words = idx.size(1)
        dropping = bool(self.transformer.drop.p) and self.training
        if dropping:
            E.dropout_begin(idx.device)               # this forward's masks: step word + 1, sites from 0
        if prefix is not None:
            prefix = _prep(prefix)
        x = _GptEmbed.apply(idx, prefix, self.transformer.wte.weight, self.transformer.wpe.weight)
        if dropping:                                  # transformer.drop(tok_emb + pos_emb), models/gpt2_model.py:190
            x = E.Dropout.apply(x, E.drop_spec(self.transformer.drop.p, x.device, 1))
        for block in self.transformer.h:
            x = block(x)
        x = _prep(x[:, -t_words:])            # keep only the text positions (strided-copy kernel)
        ln = self.transformer.ln_f
        if targets is not None:
            # CE(logits[:, :-1], targets[:, 1:], ignore_index=-100) == CE over all rows with the targets shifted
            # left and the last position ignored (ignored rows contribute nothing to the mean)
            shifted = torch.full_like(targets, -100)
            shifted[:, :-1] = targets[:, 1:]
            if getattr(self, "fuse_head_loss", False):
                # the caller only wants the loss (train_utils.enable_fused_head_loss): the [B, t, 50257] logits are never materialised
                loss = E.head_cross_entropy(x, ln.weight, ln.bias, self.lm_head.weight, None, shifted, ln.eps, -100, getattr(self, "head_chunk", 8192))
                return loss, None
            logits = E.NormLinear.apply(x, ln.weight, ln.bias, self.lm_head.weight, None, ln.eps, False)
            loss = E.cross_entropy(logits, shifted, -100)
        else:
            last = _prep(x[:, -1:, :])
            logits = E.NormLinear.apply(last, ln.weight, ln.bias, self.lm_head.weight, None, ln.eps, False)
            loss = None
        return loss, logits

    def crop_block_size(self, block_size):
        assert block_size <= self.config.block_size
        self.config.block_size = block_size
        self.transformer.wpe.weight = nn.Parameter(self.transformer.wpe.weight[:block_size])

    @classmethod
    def from_pretrained(cls, model_type, override_args=None):
        """Load OpenAI GPT-2 weights through HF transformers (needs network / a local HF cache)."""
        sizes = {'gpt2': (12, 12, 768), 'gpt2-medium': (24, 16, 1024), 'gpt2-large': (36, 20, 1280),
                 'gpt2-xl': (48, 25, 1600)}
        assert model_type in sizes
        override_args = override_args or {}
        assert all(k == 'dropout' for k in override_args)
        from transformers import GPT2LMHeadModel
        n_layer, n_head, n_embd = sizes[model_type]
        config = GPTConfig(block_size=1024, vocab_size=50257, n_layer=n_layer, n_head=n_head, n_embd=n_embd,
                           dropout=override_args.get('dropout', 0.0), bias=True)
        model = cls(config)
        sd = model.state_dict()
        hf = GPT2LMHeadModel.from_pretrained(model_type).state_dict()
        conv1d = ('attn.c_attn.weight', 'attn.c_proj.weight', 'mlp.c_fc.weight', 'mlp.c_proj.weight')
        with torch.no_grad():
            for k, v in hf.items():
                if k.endswith('.attn.masked_bias') or k.endswith('.attn.bias'):
                    continue
                src = v.t() if k.endswith(conv1d) else v   # HF stores Conv1D weights transposed
                assert sd[k].shape == src.shape, (k, sd[k].shape, src.shape)
                sd[k].copy_(src)
        return model

    def configure_optimizers(self, weight_decay, learning_rate, betas, device_type):
        decay = [p for _, p in self.named_parameters() if p.requires_grad and p.dim() >= 2]
        nodecay = [p for _, p in self.named_parameters() if p.requires_grad and p.dim() < 2]
        groups = [{'params': decay, 'weight_decay': weight_decay}, {'params': nodecay, 'weight_decay': 0.0}]
        return torch.optim.AdamW(groups, lr=learning_rate, betas=betas)

    def estimate_mfu(self, fwdbwd_per_iter, dt, peak_flops=2.5e15):
        """model flops utilisation vs the MI355X dense bf16 MFMA peak (the reference hard-codes A100 312 TF)."""
        N = self.get_num_params()
        cfg = self.config
        L, H, Q, T = cfg.n_layer, cfg.n_head, cfg.n_embd // cfg.n_head, cfg.block_size
        flops_per_iter = (6 * N + 12 * L * H * Q * T) * T * fwdbwd_per_iter
        return flops_per_iter / dt / peak_flops

    @torch.no_grad()
    def _cached_logits(self, idx_new, cache, pos, prefix=None):
        """last-position logits [B, V] after feeding prefix (first call only) + idx_new [B, t] at positions pos..; updates cache."""
        wte, wpe = self.transformer.wte.weight, self.transformer.wpe.weight
        x = K.gpt_embed_fwd(idx_new.contiguous(), prefix, wte.detach(), wpe.detach()[pos:], E.compute_dtype())
        t = x.shape[1]
        for li, block in enumerate(self.transformer.h):
            x = block.forward_cached(x, cache[li], pos)
        ln = self.transformer.ln_f
        last = x[:, -1, :].contiguous()
        h, _, _ = K.norm_fwd(last, ln.weight.detach(), None if ln.bias is None else ln.bias.detach(), ln.eps)
        V = self.config.vocab_size
        npad = (V + 7) // 8 * 8                               # 16-byte rows for the vector GEMM epilogue
        logits = K.gemm_nt(h, E.shadow([self.lm_head.weight], pad_n=npad), out_dtype=torch.float32)[:, :V]
        return logits, pos + t

    @torch.no_grad()
    def _decode_logits_dev(self, cur, cache, pos):
        """last-position logits [B, V] for the tokens `cur` [B] at device position pos (int32[1]); appends to the caches."""
        x = K.gpt_embed_step(cur, self.transformer.wte.weight.detach(), self.transformer.wpe.weight.detach(), pos, E.compute_dtype())
        for li, block in enumerate(self.transformer.h):
            x = block.forward_decode(x, cache[li], pos)
        ln = self.transformer.ln_f
        h, _, _ = K.norm_fwd(x, ln.weight.detach(), None if ln.bias is None else ln.bias.detach(), ln.eps)
        V = self.config.vocab_size
        return K.gemm_nt(h, E.shadow([self.lm_head.weight], pad_n=(V + 7) // 8 * 8), out_dtype=torch.float32)[:, :V]

    @staticmethod
    def _sample(logits, temperature, top_k, state=None):
        """temperature -> top-k crop -> softmax -> multinomial (models/gpt2_model.py:340-351).  On the device this is ONE launch
        (fk_sample_topk, Philox keyed by a seed drawn from torch's generator); the torch-op form is kept for host tensors."""
        if logits.is_cuda:
            lg = logits.float()
            lg = lg if lg.stride(-1) == 1 else lg.contiguous()
            st = state if state is not None else K.SampleState(logits.device)
            return K.sample_topk(lg, temperature, top_k, st).view(-1, 1).clone()
        logits = logits.float() / temperature
        if top_k is not None:
            v, _ = torch.topk(logits, min(top_k, logits.size(-1)))
            logits = logits.masked_fill(logits < v[:, -1:], -float('Inf'))
        return torch.multinomial(torch.softmax(logits, dim=-1), num_samples=1)

    @torch.no_grad()
    def generate(self, idx, max_new_tokens, prefix=None, temperature=1.0, top_k=None, use_cache=True, use_graph=None):
        """Sampling loop of the reference (models/gpt2_model.py:328-353: temperature, top-k crop, softmax, multinomial; returns
        the first sample's ids).  With use_cache (default) the prefix + prompt are run once and every new token is one
        incremental step against per-layer key/value caches; with use_graph (default: on from 64 new tokens, where the one-off
        capture has paid for itself)
        that step — embedding, blocks, head AND the sampling — reads its position from the device and is captured once as a
        hipGraph that is replayed per token (the step is launch-bound: ~25 small kernels).  When the sequence would outgrow
        block_size the reference's crop-and-re-forward path is used instead."""
        B, t0 = idx.shape
        t_ctx = 0 if prefix is None else prefix.shape[1]
        total = t_ctx + t0 + max_new_tokens
        cached = use_cache and total <= self.config.block_size and max_new_tokens > 0
        state = K.SampleState(idx.device) if idx.is_cuda else None      # one Philox stream per call, seeded from torch's generator
        if use_graph is None:
            use_graph = max_new_tokens >= 64
        if cached:
            d = self.config.n_embd
            cache = [torch.empty((B, total, 2 * d), dtype=E.compute_dtype(), device=idx.device) for _ in self.transformer.h]
            logits, pos = self._cached_logits(idx, cache, 0, None if prefix is None else _prep(prefix))
            if use_graph and idx.is_cuda:
                return self._generate_graph(idx, logits, cache, pos, max_new_tokens, temperature, top_k)
        for it in range(max_new_tokens):
            if cached:
                if it > 0:
                    logits, pos = self._cached_logits(idx[:, -1:], cache, pos)
            else:
                idx_cond = idx if idx.size(1) <= self.config.block_size else idx[:, -self.config.block_size:]
                _, lg = self(idx_cond, prefix=prefix)
                logits = lg[:, -1, :]
            idx = torch.cat((idx, self._sample(logits, temperature, top_k, state)), dim=1)
        return idx[0]

    @torch.no_grad()
    def _generate_graph(self, idx, logits0, cache, pos0, max_new_tokens, temperature, top_k):
        B, dev = idx.shape[0], idx.device
        out = torch.empty((B, max_new_tokens), dtype=torch.int64, device=dev)
        cur = torch.empty(B, dtype=torch.int64, device=dev)
        state = K.SampleState(dev)                       # step counter = the column of `out` the next token goes to
        lg0 = logits0.float()
        K.sample_topk(lg0 if lg0.stride(-1) == 1 else lg0.contiguous(), temperature, top_k, state, cur=cur, out=out)
        pos = torch.tensor([pos0], dtype=torch.int32, device=dev)

        def step():
            # the whole sampling tail is one launch: it writes cur / out[:, step] and advances both the step counter and `pos`
            K.sample_topk(self._decode_logits_dev(cur, cache, pos), temperature, top_k, state, cur=cur, out=out, pos_inc=pos)

        n_eager = min(2, max_new_tokens - 1)            # warm-up (allocator, lazy shadows) before the capture
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(n_eager):
                step()
            remaining = max_new_tokens - 1 - n_eager
            if remaining > 0:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):      # records the step, does not run it
                    step()
                for _ in range(remaining):
                    graph.replay()
        torch.cuda.current_stream().wait_stream(side)
        return torch.cat((idx, out), dim=1)[0]

    @torch.no_grad()
    def generate_beam_search(self, idx, max_new_tokens, prefix, temperature=1.0, topk=20, beam_width=5):
        """Stochastic beam search of the reference (models/gpt2_model.py:355-416): every step each of the `beam_width` beams draws
        `beam_width` continuations WITHOUT replacement from its `topk` most likely tokens, the `beam_width` best-scoring
        (cumulative log-probability) of the beam_width^2 candidates survive; returns the best beam's ids.  Batch size 1.
        Host-side bookkeeping around the kernel forward (one batched forward of all beams per step)."""
        if topk is None:
            topk = 2 * beam_width
        self.eval()
        beams = idx.repeat(beam_width, 1)
        scores = torch.zeros(beam_width, device=idx.device)
        prefix = prefix.expand(beam_width, -1, -1)
        for _ in range(max_new_tokens):
            _, logits = self(beams, prefix=prefix.contiguous())
            logp = torch.log_softmax(logits[:, -1, :].float() / temperature, dim=-1)
            top_lp, top_ix = logp.topk(topk, dim=-1)
            picks = torch.multinomial(top_lp.exp(), beam_width, replacement=False)           # [beam, beam_width] indices into top-k
            cand_score = (scores[:, None] + top_lp.gather(1, picks)).reshape(-1)
            cand_tok = top_ix.gather(1, picks).reshape(-1)
            cand_beam = torch.arange(beam_width, device=idx.device).repeat_interleave(beam_width)
            order = torch.sort(cand_score, descending=True, stable=True).indices[:beam_width]
            beams = torch.cat((beams[cand_beam[order]], cand_tok[order, None]), dim=1)
            scores = cand_score[order]
        return beams[scores.argmax()]

    @torch.no_grad()
    def beam_search(self, idx, max_new_tokens, prefix, temperature=1.0, topk=20, beam_width=3):
        """Deterministic beam search of the reference (models/gpt2_model.py:419-454), including its quirk: the running context
        `idx` is shared by all beams and grows by every beam's last token in turn (it is not forked per beam), so the scores are
        those of that merged sequence.  Returns the token list of the best entry [idx[0, 0], t1, t2, ...].  Batch size 1."""
        self.eval()
        _, logits = self(idx, prefix=prefix)
        lp, ix = torch.topk(torch.log_softmax(logits[:, -1, :].float(), dim=-1), beam_width)
        first = idx[0, 0].item()
        beam = [(ix[0, i], lp[0, i], [first, ix[0, i].item()]) for i in range(beam_width)]
        for _ in range(max_new_tokens - 1):
            cands = []
            for last, score, toks in beam:
                idx = torch.cat((idx, last.reshape(1, 1)), dim=-1)
                _, logits = self(idx, prefix=prefix)
                lp, ix = torch.topk(torch.log_softmax(logits[:, -1, :].float(), dim=-1), beam_width)
                for i in range(beam_width):
                    cands.append((ix[0, i], score + lp[0, i], toks + [ix[0, i].item()]))
            beam = sorted(cands, key=lambda c: float(c[1]), reverse=True)[:beam_width]
        self.last_beam_scores = [float(b[1]) for b in beam]
        return beam[0][2]
