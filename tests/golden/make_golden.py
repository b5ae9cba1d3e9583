#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE (CPU, fp32).

Runs only in the build container, where /root/reference is mounted read-only.  It imports the
reference's own modules (models/brainformer.py, models/gpt2_model.py, utils/train_utils.py) and
exec()s the notebook-only classes (BrainEncoder / Franky / CE-BrainFormer) straight from the
.ipynb JSON — no reference source is copied into this repo; only inputs-by-seed and OUTPUT
numbers are stored.  Harness-side stub modules stand in for the absent `simple_parsing` and
`wandb` packages (they are only a dataclass base / a logger).

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
from __future__ import annotations

import importlib.machinery
import json
import os
import sys
import types
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
REF = Path(os.environ.get("FRANKEN_REFERENCE", "/root/reference"))
OUT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.dont_write_bytecode = True

from frankenstein_amd import synth  # noqa: E402


# ----------------------------------------------------------------------------- import the reference
def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    class Serializable:  # dataclass base only (models/brainformer.py:11,18,40)
        pass

    class ArgumentParser:  # imported, never used (utils/train_utils.py:8)
        pass

    sp = _stub("simple_parsing", ArgumentParser=ArgumentParser)
    sp.helpers = _stub("simple_parsing.helpers", Serializable=Serializable)
    _stub("wandb", init=lambda *a, **k: None, log=lambda *a, **k: None)
    sys.path.insert(0, str(REF))
    import models.brainformer as bf  # type: ignore
    import models.gpt2_model as g2  # type: ignore
    import utils.train_utils as tu  # type: ignore
    return bf, g2, tu


def notebook_class(nb: str, cell: int, ns: dict):
    d = json.load(open(REF / nb))
    src = "".join(d["cells"][cell]["source"])
    exec(compile(src, f"{nb}:cell{cell}", "exec"), ns)


# ----------------------------------------------------------------------------- helpers
def load_synth(model: torch.nn.Module, seed=synth.SEED_WEIGHTS, skip=("attn_mask",)):
    sd = model.state_dict()
    shapes = {k: tuple(v.shape) for k, v in sd.items() if v is not None}
    st = synth.make_state(shapes, seed, skip)
    # tied weights: lm_head.weight IS transformer.wte.weight -> one tensor, generated under the wte key
    for k in list(st):
        if k.endswith("lm_head.weight"):
            st[k] = st[k.replace("lm_head.weight", "transformer.wte.weight")]
    full = {k: torch.from_numpy(v) for k, v in st.items()}
    missing, unexpected = model.load_state_dict(full, strict=False)
    assert all(any(m.endswith(s) for s in skip) for m in missing), missing
    assert not unexpected, unexpected
    return shapes


def summarize(named: dict):
    """name -> [sum, abs-sum, first 8 values] float64 rows (tiny pins for big tensors)."""
    names, rows = [], []
    for k, t in named.items():
        a = t.detach().double().flatten().numpy()
        head = np.zeros(8)
        head[: min(8, a.size)] = a[:8]
        names.append(k)
        rows.append(np.concatenate([[a.sum(), np.abs(a).sum()], head]))
    return np.array(names), np.array(rows, dtype=np.float64)


def grads_of(model):
    seen, out = set(), {}
    for k, p in model.named_parameters():
        if id(p) in seen:
            continue
        seen.add(id(p))
        out[k] = p.grad if p.grad is not None else torch.zeros_like(p)
    return out


def params_of(model):
    return {k: p for k, p in model.named_parameters()}


def save(name, **arrs):
    only = os.environ.get("FK_GOLDEN_ONLY")          # regenerate a single fixture without touching the others
    if only and name not in only.split(","):
        return
    np.savez_compressed(OUT / f"{name}.npz", **arrs)
    sz = (OUT / f"{name}.npz").stat().st_size
    print(f"wrote {name}.npz  {sz/1024:.1f} KiB")


def two_steps(model, loss_fn, tu, lrs=(1e-3, 5e-4), wd=1e-5, clip=1.0):
    """Reference step body utils/train_utils.py:128-148 without accelerate: set lr, zero_grad, fwd, bwd,
    clip_grad_value_, AdamW.step."""
    opt = torch.optim.AdamW(model.parameters(), lr=lrs[0], weight_decay=wd)
    losses = []
    for lr in lrs:
        for g in opt.param_groups:
            g["lr"] = lr
        opt.zero_grad(set_to_none=True)
        loss = loss_fn()
        loss.backward()
        torch.nn.utils.clip_grad_value_(model.parameters(), clip)
        opt.step()
        losses.append(float(loss))
    return np.array(losses)


# ----------------------------------------------------------------------------- cases
def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    bf, g2, tu = import_reference()
    from einops import rearrange
    import torch.nn.functional as F
    ns = dict(torch=torch, nn=torch.nn, F=F, rearrange=rearrange, Config=bf.Config, Encoder=bf.Encoder,
              CrossBlock=bf.CrossBlock, build_complex_rope_cache=bf.build_complex_rope_cache)
    notebook_class("notebooks_trainer/franky_baseline_gpt2.ipynb", 3, ns)   # BrainEncoder
    notebook_class("notebooks_trainer/franky_baseline_gpt2.ipynb", 4, ns)   # Franky
    BrainEncoder, Franky = ns["BrainEncoder"], ns["Franky"]
    ns2 = dict(ns)
    notebook_class("notebooks_trainer/train_brainformer.ipynb", 3, ns2)     # CE BrainFormer
    BrainFormerCE = ns2["BrainFormer"]

    # ---- ops: rope / mask / lr schedule / adamw
    rng = np.random.default_rng(7)
    xq = torch.from_numpy(rng.standard_normal((2, 10, 3, 8), dtype=np.float32))
    cache = bf.build_complex_rope_cache(8, 16, 10000)
    cache3 = cache[None, 3:13].repeat(2, 1, 1).clone()
    cache3[1] = cache[None, 6:16]
    sched = tu.init_lr_scheduler(tu.TrainConfig())
    its = np.array([0, 1, 1000, 1999, 2000, 2001, 26000, 49999, 50000, 50001, 99999])
    p0 = torch.from_numpy(rng.standard_normal(257, dtype=np.float32)).requires_grad_(True)
    p_init = p0.detach().clone().numpy()
    gs = [torch.from_numpy((3.0 * rng.standard_normal(257)).astype(np.float32)) for _ in range(3)]
    opt = torch.optim.AdamW([p0], lr=1e-3, weight_decay=1e-5)
    traj = []
    for i, g in enumerate(gs):
        for grp in opt.param_groups:
            grp["lr"] = [1e-3, 7e-4, 2e-4][i]
        p0.grad = g.clone()
        torch.nn.utils.clip_grad_value_([p0], 1.0)
        opt.step()
        traj.append(p0.detach().clone().numpy())
    save("ops",
         rope_x=xq.numpy(), rope_out2d=bf.apply_rope(xq, cache).numpy(), rope_out3d=bf.apply_rope(xq, cache3).numpy(),
         rope_cache_re=cache.real.numpy(), rope_cache_im=cache.imag.numpy(),
         mask_12_4=bf.build_advanced_causal_mask(12, 4).numpy(),
         lr_its=its, lr_vals=np.array([sched(int(i)) for i in its]),
         adamw_p0=p_init, adamw_g=np.stack([g.numpy() for g in gs]), adamw_traj=np.stack(traj),
         adamw_lrs=np.array([1e-3, 7e-4, 2e-4]))

    # ---- bf_l1_small: file-class BrainFormer, h*dh != dim in the perceiver, odd sizes
    enc = bf.MAEConfig(window_size=32, n_electrodes=16, patch_size=4, dim=64, n_layers=2, head_dim=16,
                       hidden_dim=128, n_heads=4, n_kv_heads=4)
    cfg = bf.Config(encoder=enc, n_output_tokens=8, output_dim=12, dim=64, n_layers=2, head_dim=8,
                    hidden_dim=96, n_heads=4, n_kv_heads=4)
    m = bf.BrainFormer(cfg).float()
    load_synth(m)
    x = torch.from_numpy(synth.make_inputs(3, 32, 16))
    tgt = torch.from_numpy(synth.make_motion_targets(3, 8, 12))
    loss, pred = m(x, tgt)
    loss.backward()
    ctx = m.encoder(x)
    save("bf_l1_small", loss=np.array(float(loss)), pred=pred.detach().numpy(), enc_out=ctx.detach().numpy(),
         **{"grad/" + k: v.numpy() for k, v in grads_of(m).items()})

    # ---- bf_ce_small: notebook CE BrainFormer, V=300, targets with -100
    cfg = bf.Config(encoder=enc, n_output_tokens=7, output_dim=300, dim=64, n_layers=1, head_dim=16,
                    hidden_dim=96, n_heads=4, n_kv_heads=4)
    m = BrainFormerCE(cfg).float()
    load_synth(m)
    tok = torch.from_numpy(synth.make_tokens(3, 7, vocab=300))
    loss, logits = m(x, tok)
    loss.backward()
    save("bf_ce_small", loss=np.array(float(loss)), logits=logits.detach().numpy(), targets=tok.numpy(),
         **{"grad/" + k: v.numpy() for k, v in grads_of(m).items()})

    # ---- gpt_small: prefix forward, bias True / False, train + inference branch
    for bias in (True, False):
        gcfg = g2.GPTConfig(block_size=64, vocab_size=211, n_layer=2, n_head=4, n_embd=64, dropout=0.0, bias=bias)
        g = g2.GPT(gcfg).float()
        load_synth(g)
        prefix = torch.from_numpy(synth.make_motion_targets(3, 5, 64, seed=99))
        tk = torch.from_numpy(synth.make_tokens(3, 9, vocab=211))
        idx = tk.clone()
        idx[idx == -100] = 210
        prefix.requires_grad_(True)
        loss, logits = g(idx, prefix=prefix, targets=tk)
        loss.backward()
        _, last = g(idx, prefix=prefix.detach(), targets=None)
        loss_np, logits_np = g(idx, prefix=None, targets=tk)
        save(f"gpt_small_bias{int(bias)}", loss=np.array(float(loss)), logits=logits.detach().numpy(),
             last_logits=last.detach().numpy(), targets=tk.numpy(), prefix_grad=prefix.grad.numpy(),
             loss_noprefix=np.array(float(loss_np)), logits_noprefix=logits_np.detach().numpy(),
             **{"grad/" + k: v.numpy() for k, v in grads_of(g).items()})
        if bias:
            # greedy decoding (top_k=1 makes GPT.generate deterministic): tokens + the last-position logits it samples from
            g.eval()
            start = idx[:1, :4].clone()
            gen = g.generate(start.clone(), max_new_tokens=8, prefix=prefix[:1].detach(), top_k=1)
            steps, cur = [], start.clone()
            for _ in range(8):
                _, lg = g(cur, prefix=prefix[:1].detach())
                steps.append(lg[0, -1].detach().numpy())
                cur = torch.cat([cur, lg[:, -1].argmax(-1, keepdim=True)], 1)
            import contextlib, io
            with contextlib.redirect_stdout(io.StringIO()):
                bs = g.beam_search(start.clone(), 5, prefix[:1].detach(), beam_width=3)
            save("gpt_generate", start=start.numpy(), tokens=gen.numpy(), step_logits=np.stack(steps), tokens_argmax=cur[0].numpy(),
                 beam_tokens=np.array(bs, dtype=np.int64))

    # ---- cfg1: Franky(BrainEncoder + gpt2-nano), B=4, T=200, + 2 optimizer steps
    enc = bf.MAEConfig(window_size=200, n_electrodes=256, patch_size=25, dim=128, n_layers=2, head_dim=32,
                       hidden_dim=512, n_heads=4, n_kv_heads=4)
    bcfg = bf.Config(encoder=enc, n_output_tokens=32, output_dim=128, dim=128, n_layers=2, head_dim=32,
                     hidden_dim=256, n_heads=4, n_kv_heads=4)
    gcfg = g2.GPTConfig(block_size=1024, vocab_size=50257, n_layer=2, n_head=4, n_embd=128, dropout=0.0, bias=True)
    fr = Franky(BrainEncoder(bcfg), g2.GPT(gcfg)).float()
    load_synth(fr)
    x = torch.from_numpy(synth.make_inputs(4, 200, 256))
    tok = torch.from_numpy(synth.make_tokens(4, 25))
    feats = fr.brain_model(x)
    loss, logits = fr(x, tok)
    loss.backward()
    gn, gr = summarize(grads_of(fr))
    lg = logits.detach()
    fr.zero_grad(set_to_none=True)
    losses = two_steps(fr, lambda: fr(x, tok)[0], tu)
    pn, pr = summarize(params_of(fr))
    save("cfg1_franky", loss=np.array(float(loss)), features=feats.detach().numpy(),
         logits_head=lg[:, :, :64].numpy(), logits_lse=torch.logsumexp(lg, -1).numpy(),
         logits_argmax=lg.argmax(-1).numpy(), logits_tail=lg[:, :, -33:].numpy(),
         grad_names=gn, grad_rows=gr, step_losses=losses, param_names=pn, param_rows=pr)

    # ---- cfg2 at B=1: brainformer-small (6L, d=384, 6 heads x 64), N=6144 tokens, L1 head
    enc = bf.MAEConfig(window_size=600, n_electrodes=256, patch_size=25, dim=384, n_layers=6, head_dim=64,
                       hidden_dim=1536, n_heads=6, n_kv_heads=6)
    cfg = bf.Config(encoder=enc, n_output_tokens=32, output_dim=128, dim=384, n_layers=2, head_dim=64,
                    hidden_dim=768, n_heads=6, n_kv_heads=6)
    m = bf.BrainFormer(cfg).float()
    load_synth(m)
    x = torch.from_numpy(synth.make_inputs(1, 600, 256))
    tgt = torch.from_numpy(synth.make_motion_targets(1, 32, 128))
    loss, pred = m(x, tgt)
    loss.backward()
    with torch.no_grad():
        ctx = m.encoder(x)
    gn, gr = summarize(grads_of(m))
    save("cfg2_b1", loss=np.array(float(loss)), pred=pred.detach().numpy(),
         enc_rows=ctx[0, [0, 1, 255, 256, 3071, 6143]].numpy(), grad_names=gn, grad_rows=gr)
    # evenly spaced 256-element samples of every gradient of the same run (the full gradients are 81 MB): what the bf16 parity test at
    # the benchmarked shape computes its per-parameter cosine from
    from tests import cases as TC
    sn, sr = TC.sample_rows(grads_of(m))
    save("cfg2_b1_samples", loss=np.array(float(loss)), grad_names=np.array(sn), grad_samples=sr,
         enc_samples=ctx[0].double().flatten().numpy()[TC.sample_index(ctx[0].numel(), 4096)])

    # ---- cfg2's CE-head variant at B=1 (notebook CE BrainFormer, n_output_tokens=25, output_dim=50257)
    cfg = bf.Config(encoder=enc, n_output_tokens=25, output_dim=50257, dim=384, n_layers=2, head_dim=64,
                    hidden_dim=768, n_heads=6, n_kv_heads=6)
    m = BrainFormerCE(cfg).float()
    load_synth(m)
    tok = torch.from_numpy(synth.make_tokens(1, 25))
    loss, logits = m(x, tok)
    loss.backward()
    lg = logits.detach()
    gn, gr = summarize(grads_of(m))
    sn, sr = TC.sample_rows(grads_of(m))
    save("cfg2_b1_ce", loss=np.array(float(loss)), targets=tok.numpy(), logits_head=lg[:, :, :64].numpy(),
         logits_tail=lg[:, :, -33:].numpy(), logits_lse=torch.logsumexp(lg, -1).numpy(), logits_argmax=lg.argmax(-1).numpy(),
         logits_at_target=lg[0].gather(-1, tok[0].clamp(min=0)[:, None])[:, 0].numpy(),
         grad_names=gn, grad_rows=gr, grad_samples=sr)

    # ---- mae_small: brainformer.MAE (models/brainformer.py:354-486); the random index sets are recorded as inputs
    enc = bf.MAEConfig(window_size=32, n_electrodes=16, patch_size=4, dim=64, n_layers=2, head_dim=16, hidden_dim=128,
                       n_heads=4, n_kv_heads=4, n_dec_layers=2, decoder_dim=64)
    mae = bf.MAE(enc).float()
    load_synth(mae)
    x = torch.from_numpy(synth.make_inputs(3, 32, 16))
    torch.manual_seed(123)
    masked, unmasked = mae.get_masking_indices(0.75, mae.encoder.to_patches(x))
    torch.manual_seed(123)
    loss, recon, bmask = mae(x, masking_ratio=0.75, return_preds=True)
    loss.backward()
    save("mae_small", loss=np.array(float(loss)), masked=masked.numpy(), unmasked=unmasked.numpy(),
         recon=recon.detach().numpy(), binary_mask=bmask.detach().numpy(),
         **{"grad/" + k: v.numpy() for k, v in grads_of(mae).items()})

    # ---- simple_mae_small: models/simple_mae (no .py suffix) + the notebook-only config dataclasses; two samples carry a
    #      zero-padded tail (utils/data_utils.py:243-267 pads with 0) so the padding mask / masked loss are exercised
    import contextlib
    import dataclasses
    import io
    sm = importlib.machinery.SourceFileLoader("ref_simple_mae", str(REF / "models" / "simple_mae")).load_module()
    ns3 = dict(dataclass=dataclasses.dataclass, Serializable=object)
    exec("".join(json.load(open(REF / "notebooks" / "simple_mae.ipynb"))["cells"][1]["source"]), ns3)
    ecfg = ns3["SimpleEncoderConfig"](block_size=40, patch_size=24, n_layers=2, dim=64, hidden_dim=128, head_dim=16,
                                      n_heads=4, n_kv_heads=4)
    mcfg = ns3["SimpleMAEConfig"](n_layers=2, dim=48, hidden_dim=96, head_dim=8, n_heads=4, n_kv_heads=4)
    m = sm.SimpleMAE(ecfg, mcfg).float()
    load_synth(m, skip=())
    x = torch.from_numpy(synth.make_inputs(3, 40, 24))
    x[1, 33:] = 0.0
    x[2, 38:] = 0.0
    torch.manual_seed(321)
    masked, unmasked = m.get_masking_indices(0.75, x)
    torch.manual_seed(321)
    with contextlib.redirect_stdout(io.StringIO()):      # the reference prints debug shapes in forward
        loss, recon, bmask = m(x, masking_ratio=0.75, return_preds=True)
    loss.backward()
    save("simple_mae_small", loss=np.array(float(loss.detach())), masked=masked.numpy(), unmasked=unmasked.numpy(), x=x.numpy(),
         recon=recon.detach().numpy(), binary_mask=bmask.detach().numpy(),
         **{"grad/" + k: v.numpy() for k, v in grads_of(m).items()})


def pipeline_golden():
    """utils/data_utils.py process_signal + pad_truncate_brain_list on ragged synthetic trials (float64 like the .mat arrays)."""
    sys.modules.setdefault("scipy.io", __import__("scipy.io"))
    du = importlib.machinery.SourceFileLoader("ref_data_utils", str(REF / "utils" / "data_utils.py")).load_module()
    rng = np.random.default_rng(7)
    lens = [37, 52, 3, 64, 45, 9]
    blocks = np.array([5, 5, 9, 9, 5, 2])
    volt = [rng.normal(0.5 * (i % 3), 1.0 + 0.2 * i, size=(n, 8)).astype(np.float32) for i, n in enumerate(lens)]
    spk = [rng.poisson(2.0, size=(n, 8)).astype(np.float32) for n in lens]
    for v in volt:
        v[:, 3] = 1.25                                  # a dead channel: std == 0 -> 1
    proc = du.process_signal([v.astype(np.float64) for v in volt], [s.astype(np.float64) for s in spk], blocks)
    padded = np.stack(du.pad_truncate_brain_list(list(proc), 48)).astype(np.float32)
    zs = du.z_score_per_block_scaling([np.concatenate([v, s], 1).astype(np.float64) for v, s in zip(volt, spk)], list(blocks))
    save("pipeline", lens=np.array(lens), blocks=blocks, padded=padded, **{f"volt{i}": v for i, v in enumerate(volt)},
         **{f"spk{i}": s for i, s in enumerate(spk)}, **{f"z{i}": z.astype(np.float32) for i, z in enumerate(zs)})


def vq_golden():
    """models/vq_brain.py convolution stack (Encoder, Decoder, custom_l1_loss, calculate_perp) on CPU; the third-party VQ layer
    (absent, unpinned) is stubbed out and bypassed: decoder(encoder(x))."""
    for name, attrs in (("vector_quantize_pytorch", ("ResidualVQ", "VectorQuantize")), ("pytorch_model_summary", ("summary",))):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__spec__ = importlib.machinery.ModuleSpec(name, None)
            for a in attrs:
                setattr(m, a, type(a, (torch.nn.Module,), {"__init__": lambda self, *aa, **kw: torch.nn.Module.__init__(self)}))
            sys.modules[name] = m
    vq = importlib.machinery.SourceFileLoader("ref_vq_brain", str(REF / "models" / "vq_brain.py")).load_module()
    torch.manual_seed(0)
    net = vq.SoundStream(C=32, D=16, codebook_size=64, n_electrodes=16)
    load_synth(net.encoder)
    load_synth(net.decoder)
    x = torch.from_numpy(synth.make_inputs(2, 40, 16))
    x[1, 33:] = 0.0                                      # padded frames: excluded from the loss
    e = net.encoder(x)
    o = net.decoder(e)
    loss = net.custom_l1_loss(o, x)
    loss.backward()
    idx = torch.from_numpy(np.random.default_rng(3).integers(0, 64, size=(2, 10)))
    save("vq_conv_small", e=e.detach().numpy(), o=o.detach().numpy(), loss=np.array(float(loss)),
         perp=np.array(float(net.calculate_perp(idx))), perp_idx=idx.numpy(),
         **{"grad/encoder." + k: v.numpy() for k, v in grads_of(net.encoder).items()},
         **{"grad/decoder." + k: v.numpy() for k, v in grads_of(net.decoder).items()})


def vq_cfg4_golden():
    """The same convolution stack AT THE SIZE BASELINE configs[3] / SURVEY cfg4 names: SoundStream(C = 256, D = 64, codebook 1024, 256
    electrodes) on [2, 600, 256] (one padded tail) — codes [2, 150, 64], reconstruction, masked L1 loss, gradient summaries + samples.  The
    third-party VQ layer (absent, unpinned) is bypassed as in vq_conv_small: decoder(encoder(x))."""
    for name, attrs in (("vector_quantize_pytorch", ("ResidualVQ", "VectorQuantize")), ("pytorch_model_summary", ("summary",))):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__spec__ = importlib.machinery.ModuleSpec(name, None)
            for a in attrs:
                setattr(m, a, type(a, (torch.nn.Module,), {"__init__": lambda self, *aa, **kw: torch.nn.Module.__init__(self)}))
            sys.modules[name] = m
    from tests import cases as TC
    vq = importlib.machinery.SourceFileLoader("ref_vq_brain", str(REF / "models" / "vq_brain.py")).load_module()
    torch.manual_seed(0)
    net = vq.SoundStream(C=256, D=64, codebook_size=1024, n_electrodes=256)
    load_synth(net.encoder)
    load_synth(net.decoder)
    x = torch.from_numpy(synth.make_inputs(2, 600, 256))
    x[1, 541:] = 0.0
    e = net.encoder(x)
    o = net.decoder(e)
    loss = net.custom_l1_loss(o, x)
    loss.backward()
    g = {**{"encoder." + k: v for k, v in grads_of(net.encoder).items()}, **{"decoder." + k: v for k, v in grads_of(net.decoder).items()}}
    gn, gr = summarize(g)
    sn, sr = TC.sample_rows(g)
    save("vq_conv_cfg4", e=e.detach().numpy(), o_every4=o.detach().numpy()[:, ::4], o_sum=np.array(float(o.detach().double().sum())),
         loss=np.array(float(loss)), grad_names=gn, grad_rows=gr, grad_samples=sr)


def accum_golden():
    """The reference's REAL hot loop — utils/train_utils.py:93-185 run_train_model under accelerate, CPU, fp32 — with
    grad_accum = 2 on the small L1 BrainFormer: pins what gradient accumulation means in the reference (accelerate's
    AcceleratedOptimizer.zero_grad / .step act only on sync micro-steps, and the loop zeroes BEFORE the forward, :134).
    The loop never returns (the `break` at :185 leaves only the `for`), so the run is ended from a forward pre-hook after
    N_FWD forwards.  Stored: the sample indices of every micro-batch in the order the loader produced them, the loss of every
    forward, a parameter checksum at the entry of every forward (says WHEN updates happened) and the parameters at the end."""
    import tempfile
    bf, g2, tu = import_reference()
    N_FWD, N_ITEMS = 12, 10
    enc = bf.MAEConfig(window_size=32, n_electrodes=16, patch_size=4, dim=64, n_layers=2, head_dim=16,
                       hidden_dim=128, n_heads=4, n_kv_heads=4)
    cfg = bf.Config(encoder=enc, n_output_tokens=8, output_dim=12, dim=64, n_layers=2, head_dim=8,
                    hidden_dim=96, n_heads=4, n_kv_heads=4)
    m = bf.BrainFormer(cfg).float()
    load_synth(m)
    xs = torch.from_numpy(synth.make_inputs(N_ITEMS, 32, 16, seed=synth.SEED_INPUT + 17))
    ys = torch.from_numpy(synth.make_motion_targets(N_ITEMS, 8, 12, seed=synth.SEED_INPUT + 18))

    class DS(torch.utils.data.Dataset):
        def __init__(self):
            self.log = []

        def __len__(self):
            return N_ITEMS

        def __getitem__(self, i):
            self.log.append(int(i))
            return xs[i], ys[i], 0

    class Stop(BaseException):
        pass

    losses, sums = [], []
    probe = m.encoder.transformer.h[0].attn.qw.weight

    def pre(_mod, _args, _kw=None):
        if len(sums) == N_FWD:
            raise Stop()
        sums.append(float(probe.detach().double().sum()))

    def post(_mod, _args, out):
        losses.append(float(out[0].detach()))

    m.register_forward_pre_hook(pre)
    m.register_forward_hook(post)
    tc = tu.TrainConfig(exp_name="accum", batch_size=4, grad_accum=2, learning_rate=1e-3, weight_decay=1e-5, max_steps=10 ** 6,
                        eval_interval=10 ** 6, use_scheduler=True, warmup_iters=4, lr_decay_iters=20, num_workers=0,
                        pin_memory=False, grad_clip=1.0, mixed_precision=False)
    tr, va = DS(), DS()
    import contextlib, io
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            tu.run_train_model(m, (tr, va), tc, "golden", Path(tempfile.mkdtemp()))
    except Stop:
        pass
    assert len(losses) == N_FWD and len(tr.log) >= 2 * N_FWD
    order = np.array(tr.log[:2 * N_FWD]).reshape(N_FWD, 2)
    sched = tu.init_lr_scheduler(tc)
    save("train_accum", order=order, losses=np.array(losses), probe_sums=np.array(sums),
         lrs=np.array([sched(i) for i in range(N_FWD)]), n_items=np.array(N_ITEMS),
         **{"param/" + k: v.detach().numpy() for k, v in params_of(m).items()})


def cfg2_batch_golden(B=3):
    """cfg2 (brainformer-small, N = 6144 tokens, L1 head) at B = 3: the full-size shape with more than one sample and a batch that is
    not a power of two — loss, predictions, encoder rows of every sample, gradient summaries and evenly spaced gradient samples."""
    bf, g2, tu = import_reference()
    from tests import cases as TC
    enc = bf.MAEConfig(window_size=600, n_electrodes=256, patch_size=25, dim=384, n_layers=6, head_dim=64,
                       hidden_dim=1536, n_heads=6, n_kv_heads=6)
    cfg = bf.Config(encoder=enc, n_output_tokens=32, output_dim=128, dim=384, n_layers=2, head_dim=64,
                    hidden_dim=768, n_heads=6, n_kv_heads=6)
    m = bf.BrainFormer(cfg).float()
    load_synth(m)
    x = torch.from_numpy(synth.make_inputs(B, 600, 256))
    tgt = torch.from_numpy(synth.make_motion_targets(B, 32, 128))
    loss, pred = m(x, tgt)
    loss.backward()
    with torch.no_grad():
        ctx = m.encoder(x)
    gn, gr = summarize(grads_of(m))
    sn, sr = TC.sample_rows(grads_of(m))
    save(f"cfg2_b{B}", loss=np.array(float(loss)), pred=pred.detach().numpy(), enc_rows=ctx[:, [0, 1, 255, 256, 3071, 6143]].numpy(),
         grad_names=gn, grad_rows=gr, grad_samples=sr)


def _cfg2_reference_model():
    bf, g2, tu = import_reference()
    enc = bf.MAEConfig(window_size=600, n_electrodes=256, patch_size=25, dim=384, n_layers=6, head_dim=64,
                       hidden_dim=1536, n_heads=6, n_kv_heads=6)
    cfg = bf.Config(encoder=enc, n_output_tokens=32, output_dim=128, dim=384, n_layers=2, head_dim=64,
                    hidden_dim=768, n_heads=6, n_kv_heads=6)
    m = bf.BrainFormer(cfg).float()
    load_synth(m)
    return m


ENC_ROWS = [0, 1, 255, 256, 3071, 6143]


def cfg2_b32_fwd_golden(B=32):
    """The BENCHMARKED batch through the reference: cfg2 (brainformer-small, N = 6144, L1 head), B = 32, forward only under
    torch.no_grad() — loss, the [32, 32, 128] predictions and six encoder rows of every sample.  One call of the reference's forward on
    the whole batch (models/brainformer.py:532-558); the address space is capped so that a host without the memory for it fails with a
    Python error instead of being killed."""
    import resource
    resource.setrlimit(resource.RLIMIT_AS, (56 << 30, 56 << 30))
    m = _cfg2_reference_model()
    x = torch.from_numpy(synth.make_inputs(B, 600, 256))
    tgt = torch.from_numpy(synth.make_motion_targets(B, 32, 128))
    with torch.no_grad():
        loss, pred = m(x, tgt)
        ctx = m.encoder(x)
    save(f"cfg2_b{B}_fwd", loss=np.array(float(loss)), pred=pred.numpy(), enc_rows=ctx[:, ENC_ROWS].numpy(),
         note=np.array("reference forward, one call at B = 32, torch.no_grad(), CPU fp32"))


def cfg2_b8_grad_golden(B=8, MB=2):
    """cfg2 at B = 8 WITH gradients.  The reference's dense-mask attention keeps ~15 GB of fp32 scores per sample alive for the backward,
    so a B = 8 backward does not fit this container: the gradient is produced by the reference's OWN modules over four micro-batches of
    two samples, each loss scaled by MB / B before backward() so that autograd's accumulation gives the gradient of the B = 8 mean
    loss (the L1 loss is a mean over equally sized samples; this is stated in the fixture's `note`).  Loss = mean of the four losses,
    pred / encoder rows concatenated."""
    m = _cfg2_reference_model()
    from tests import cases as TC
    x = torch.from_numpy(synth.make_inputs(B, 600, 256))
    tgt = torch.from_numpy(synth.make_motion_targets(B, 32, 128))
    losses, preds, rows = [], [], []
    for i in range(0, B, MB):
        loss, pred = m(x[i:i + MB], tgt[i:i + MB])
        (loss * (MB / B)).backward()
        losses.append(float(loss))
        preds.append(pred.detach())
        with torch.no_grad():
            rows.append(m.encoder(x[i:i + MB])[:, ENC_ROWS])
    gn, gr = summarize(grads_of(m))
    sn, sr = TC.sample_rows(grads_of(m))
    save(f"cfg2_b{B}_grad", loss=np.array(float(np.mean(losses))), micro_losses=np.array(losses), pred=torch.cat(preds).numpy(),
         enc_rows=torch.cat(rows).numpy(), grad_names=gn, grad_rows=gr, grad_samples=sr,
         note=np.array(f"reference modules, {B // MB} micro-batches of {MB} samples, each loss * {MB}/{B} before backward() (autograd accumulation) "
                       f"= gradient of the B = {B} mean L1 loss; CPU fp32"))


def cfg5_simple_mae_golden(B=4):
    """BASELINE configs[4] AT THE SIZE SURVEY 8d names — SimpleMAE: 6-layer d = 384 encoder on 600 frame tokens (patch 256), 2-layer decoder,
    masking ratio 0.75 — through the reference (models/simple_mae + the notebook's config dataclasses), B = 4 with zero-padded tails of three
    samples: loss, the reconstruction at the masked frames, the binary mask, gradient summaries and 256 evenly spaced samples of every
    gradient.  The random index sets are fixture INPUTS (torch's CPU generator is not reproducible on the device)."""
    import contextlib
    import dataclasses
    import io
    from tests import cases as TC
    import_reference()                                   # the harness stubs for simple_parsing / wandb
    sm = importlib.machinery.SourceFileLoader("ref_simple_mae", str(REF / "models" / "simple_mae")).load_module()
    ns3 = dict(dataclass=dataclasses.dataclass, Serializable=object)
    exec("".join(json.load(open(REF / "notebooks" / "simple_mae.ipynb"))["cells"][1]["source"]), ns3)
    ecfg = ns3["SimpleEncoderConfig"](block_size=600, patch_size=256, n_layers=6, dim=384, hidden_dim=1536, head_dim=64, n_heads=6, n_kv_heads=6)
    mcfg = ns3["SimpleMAEConfig"](n_layers=2, dim=384, hidden_dim=1536, head_dim=64, n_heads=6, n_kv_heads=6)
    with contextlib.redirect_stdout(io.StringIO()):
        m = sm.SimpleMAE(ecfg, mcfg).float()
    load_synth(m, skip=())
    x = torch.from_numpy(synth.make_inputs(B, 600, 256))
    x[1, 590:] = 0.0
    x[2, 333:] = 0.0
    x[3, 599:] = 0.0
    torch.manual_seed(4321)
    masked, unmasked = m.get_masking_indices(0.75, x)
    torch.manual_seed(4321)
    with contextlib.redirect_stdout(io.StringIO()):
        loss, recon, bmask = m(x, masking_ratio=0.75, return_preds=True)
    loss.backward()
    gn, gr = summarize(grads_of(m))
    sn, sr = TC.sample_rows(grads_of(m))
    save("cfg5_simple_mae", loss=np.array(float(loss.detach())), masked=masked.numpy(), unmasked=unmasked.numpy(), pad_from=np.array([600, 590, 333, 599]),
         recon_every4=recon.detach().numpy().astype(np.float32)[:, ::4], recon_sum=np.array(float(recon.detach().double().sum())),
         binary_mask=bmask.detach().numpy(), grad_names=gn, grad_rows=gr, grad_samples=sr)


if __name__ == "__main__":
    if os.environ.get("FK_GOLDEN_ONLY") == "cfg5_simple_mae":
        cfg5_simple_mae_golden()
    elif os.environ.get("FK_GOLDEN_ONLY") == "cfg2_b3":
        cfg2_batch_golden()
    elif os.environ.get("FK_GOLDEN_ONLY") == "cfg2_b32_fwd":
        cfg2_b32_fwd_golden()
    elif os.environ.get("FK_GOLDEN_ONLY") == "cfg2_b8_grad":
        cfg2_b8_grad_golden()
    elif os.environ.get("FK_GOLDEN_ONLY") == "pipeline":
        pipeline_golden()
    elif os.environ.get("FK_GOLDEN_ONLY") == "vq_conv_small":
        vq_golden()
    elif os.environ.get("FK_GOLDEN_ONLY") == "vq_conv_cfg4":
        vq_cfg4_golden()
    elif os.environ.get("FK_GOLDEN_ONLY") == "train_accum":
        accum_golden()
    else:
        main()
        pipeline_golden()
        vq_golden()
        accum_golden()
        cfg2_batch_golden()
        cfg2_b8_grad_golden()
        cfg2_b32_fwd_golden()
        cfg5_simple_mae_golden()
        vq_cfg4_golden()
