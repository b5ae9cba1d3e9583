"""Shared parity cases: the configs the golden fixtures were generated with
(tests/golden/make_golden.py) expressed for the oracle, plus seeded inputs."""
from __future__ import annotations

import numpy as np
import torch

from frankenstein_amd import synth
from oracle import ref_models as R


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def state(shapes, seed=synth.SEED_WEIGHTS):
    return {k: t(v) for k, v in synth.make_state(shapes, seed).items()}


def enc_small():
    return R.mae_config(window_size=32, n_electrodes=16, patch_size=4, dim=64, n_layers=2, head_dim=16,
                        hidden_dim=128, n_heads=4, n_kv_heads=4)


def bf_l1_small():
    cfg = R.perceiver_config(enc_small(), n_output_tokens=8, output_dim=12, dim=64, n_layers=2, head_dim=8,
                             hidden_dim=96, n_heads=4, n_kv_heads=4)
    x = t(synth.make_inputs(3, 32, 16))
    tgt = t(synth.make_motion_targets(3, 8, 12))
    return cfg, x, tgt


def bf_ce_small():
    cfg = R.perceiver_config(enc_small(), n_output_tokens=7, output_dim=300, dim=64, n_layers=1, head_dim=16,
                             hidden_dim=96, n_heads=4, n_kv_heads=4)
    x = t(synth.make_inputs(3, 32, 16))
    tok = t(synth.make_tokens(3, 7, vocab=300))
    return cfg, x, tok


def gpt_small(bias: bool):
    cfg = R.gpt_config(block_size=64, vocab_size=211, n_layer=2, n_head=4, n_embd=64, dropout=0.0, bias=bias)
    prefix = t(synth.make_motion_targets(3, 5, 64, seed=99))
    tk = t(synth.make_tokens(3, 9, vocab=211))
    idx = tk.clone()
    idx[idx == -100] = 210
    return cfg, prefix, tk, idx


def cfg1():
    enc = R.mae_config(window_size=200, n_electrodes=256, patch_size=25, dim=128, n_layers=2, head_dim=32,
                       hidden_dim=512, n_heads=4, n_kv_heads=4)
    bcfg = R.perceiver_config(enc, n_output_tokens=32, output_dim=128, dim=128, n_layers=2, head_dim=32,
                              hidden_dim=256, n_heads=4, n_kv_heads=4)
    gcfg = R.gpt_config(block_size=1024, vocab_size=50257, n_layer=2, n_head=4, n_embd=128, dropout=0.0, bias=True)
    x = t(synth.make_inputs(4, 200, 256))
    tok = t(synth.make_tokens(4, 25))
    return bcfg, gcfg, x, tok


def cfg1_shapes(bcfg, gcfg):
    s = R.brainformer_shapes(bcfg, "to_words", p="brain_model.")
    s.update(R.gpt_shapes(gcfg, p="llm_model."))
    return s


def cfg2(B=1, head_dim_out=128):
    enc = R.mae_config(window_size=600, n_electrodes=256, patch_size=25, dim=384, n_layers=6, head_dim=64,
                       hidden_dim=1536, n_heads=6, n_kv_heads=6)
    cfg = R.perceiver_config(enc, n_output_tokens=32, output_dim=head_dim_out, dim=384, n_layers=2, head_dim=64,
                             hidden_dim=768, n_heads=6, n_kv_heads=6)
    x = t(synth.make_inputs(B, 600, 256))
    tgt = t(synth.make_motion_targets(B, 32, head_dim_out))
    return cfg, x, tgt


def summarize_rows(named):
    names, rows = [], []
    for k, v in named.items():
        a = v.detach().double().flatten().numpy()
        head = np.zeros(8)
        head[: min(8, a.size)] = a[:8]
        names.append(k)
        rows.append(np.concatenate([[a.sum(), np.abs(a).sum()], head]))
    return names, np.array(rows)


def sample_index(n: int, k: int = 256) -> np.ndarray:
    """k evenly spaced flat indices of an n-element tensor (all of them when n <= k): the gradient samples stored for the
    full-size fixtures, whose complete gradients would be tens of MB."""
    return np.arange(n) if n <= k else (np.arange(k, dtype=np.int64) * n) // k


def sample_rows(named, k: int = 256):
    """name -> float64 [k] sample of the flattened tensor (zero padded below k elements)."""
    names, rows = [], []
    for key, v in named.items():
        a = v.detach().double().flatten().numpy()
        r = np.zeros(k)
        idx = sample_index(a.size, k)
        r[: idx.size] = a[idx]
        names.append(key)
        rows.append(r)
    return names, np.array(rows)


def cfg2_ce(B=1):
    """cfg2's CE-head variant (notebook class, n_output_tokens=25, output_dim=50257)."""
    enc = R.mae_config(window_size=600, n_electrodes=256, patch_size=25, dim=384, n_layers=6, head_dim=64,
                       hidden_dim=1536, n_heads=6, n_kv_heads=6)
    cfg = R.perceiver_config(enc, n_output_tokens=25, output_dim=50257, dim=384, n_layers=2, head_dim=64,
                             hidden_dim=768, n_heads=6, n_kv_heads=6)
    x = t(synth.make_inputs(B, 600, 256))
    tok = t(synth.make_tokens(B, 25))
    return cfg, x, tok


def mae_small():
    cfg = R.mae_config(window_size=32, n_electrodes=16, patch_size=4, dim=64, n_layers=2, head_dim=16, hidden_dim=128,
                       n_heads=4, n_kv_heads=4, n_dec_layers=2, decoder_dim=64)
    return cfg, t(synth.make_inputs(3, 32, 16))


def unpatch(tok, C, P):
    """'b (t c) p -> b (t p) c' (models/brainformer.py:372)"""
    B, N, _ = tok.shape
    return tok.view(B, N // C, C, P).permute(0, 1, 3, 2).reshape(B, (N // C) * P, C)


def simple_mae_small():
    ecfg = R.simple_encoder_config(block_size=40, patch_size=24, n_layers=2, dim=64, hidden_dim=128, head_dim=16, n_heads=4, n_kv_heads=4)
    mcfg = R.simple_mae_config(n_layers=2, dim=48, hidden_dim=96, head_dim=8, n_heads=4, n_kv_heads=4)
    return ecfg, mcfg


def cfg5_simple_mae(pad_from=(600, 590, 333, 599)):
    """BASELINE configs[4] at SURVEY 8d's size (cfg5): SimpleMAE, 6-layer d = 384 encoder on 600 frame tokens, 2-layer decoder; the input
    of tests/golden/cfg5_simple_mae.npz: B = 4 synthetic samples, three of them with zero-padded tails."""
    ecfg = R.simple_encoder_config(block_size=600, patch_size=256, n_layers=6, dim=384, hidden_dim=1536, head_dim=64, n_heads=6, n_kv_heads=6)
    mcfg = R.simple_mae_config(n_layers=2, dim=384, hidden_dim=1536, head_dim=64, n_heads=6, n_kv_heads=6)
    x = t(synth.make_inputs(len(pad_from), 600, 256)).clone()
    for b, p0 in enumerate(pad_from):
        x[b, int(p0):] = 0.0
    return ecfg, mcfg, x


def train_accum(n_items: int = 10):
    """Dataset of the grad_accum fixture (tests/golden/train_accum.npz): sample i -> (x_i, y_i)."""
    cfg, _, _ = bf_l1_small()
    xs = t(synth.make_inputs(n_items, 32, 16, seed=synth.SEED_INPUT + 17))
    ys = t(synth.make_motion_targets(n_items, 8, 12, seed=synth.SEED_INPUT + 18))
    return cfg, xs, ys
