"""CPU oracle: training-step restatement (TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows utils/train_utils.py of the reference:
  * get_lr            <- init_lr_scheduler, utils/train_utils.py:49-72
  * clip_grad_value   <- accelerator.clip_grad_value_(params, 1.0), utils/train_utils.py:142
  * adamw_step        <- torch.optim.AdamW(lr, weight_decay) defaults betas (.9,.999), eps 1e-8,
                          utils/train_utils.py:117-119,143 (decoupled weight decay, bias-corrected)
  * train_step        <- loop body utils/train_utils.py:128-148
"""
from __future__ import annotations

import math
from typing import Callable, Dict

import torch


def get_lr(it: int, learning_rate: float = 1e-3, warmup_iters: int = 2000, lr_decay_iters: int = 50000,
           use_scheduler: bool = True) -> float:
    min_lr = learning_rate / 10
    if not use_scheduler:
        return learning_rate
    if it < warmup_iters:
        return learning_rate * it / warmup_iters
    if it > lr_decay_iters:
        return min_lr
    ratio = (it - warmup_iters) / (lr_decay_iters - warmup_iters)
    coeff = 0.5 * (1.0 + math.cos(math.pi * ratio))
    return min_lr + coeff * (learning_rate - min_lr)


def clip_grad_value(g: torch.Tensor, clip: float) -> torch.Tensor:
    return g.clamp(-clip, clip)


def adamw_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int, lr: float,
               weight_decay: float = 1e-5, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8):
    """One AdamW update (torch.optim.AdamW single-tensor semantics), returns new (p, m, v).
    ``step`` is the 1-based count of updates including this one."""
    p = p * (1.0 - lr * weight_decay)
    m = beta1 * m + (1.0 - beta1) * g
    v = beta2 * v + (1.0 - beta2) * g * g
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


def train_step(loss_fn: Callable[[Dict[str, torch.Tensor]], torch.Tensor], sd: Dict[str, torch.Tensor],
               state: Dict[str, Dict[str, torch.Tensor]], step: int, lr: float, weight_decay: float = 1e-5,
               grad_clip: float = 1.0):
    """One optimizer step on every tensor of ``sd`` (all trainable).  ``loss_fn(sd)`` returns the scalar
    loss.  ``state[name] = {'m','v'}`` is updated in place; returns (loss, grads, new_sd)."""
    leaves = {k: t.detach().clone().requires_grad_(True) for k, t in sd.items()}
    loss = loss_fn(leaves)
    grads = torch.autograd.grad(loss, list(leaves.values()), allow_unused=True)
    new_sd, gout = {}, {}
    for (k, p), g in zip(leaves.items(), grads):
        if g is None:
            g = torch.zeros_like(p)
        gout[k] = g
        gc = clip_grad_value(g, grad_clip) if grad_clip is not None else g
        st = state.setdefault(k, {"m": torch.zeros_like(p), "v": torch.zeros_like(p)})
        np_, st["m"], st["v"] = adamw_step(p.detach(), gc, st["m"], st["v"], step, lr, weight_decay)
        new_sd[k] = np_
    return loss.detach(), gout, new_sd


def accum_sync_flags(n_micro: int, grad_accum: int, batches_per_epoch: int):
    """Which of the first n_micro micro-steps of the reference loop are "sync" steps under accelerate (utils/train_utils.py:98,133:
    Accelerator(gradient_accumulation_steps=k) + `with accelerator.accumulate(model)`): the k-th micro-step since the last reset, or
    the last batch of the loader, which also resets the count (GradientState.sync_with_dataloader)."""
    flags, count = [], 0
    for i in range(n_micro):
        if i % batches_per_epoch == batches_per_epoch - 1:
            count = 0
            flags.append(True)
        else:
            count += 1
            flags.append(count % grad_accum == 0)
    return flags


def train_loop_accum(loss_fn, sd: Dict[str, torch.Tensor], batches, grad_accum: int, batches_per_epoch: int, lr_fn,
                     weight_decay: float = 1e-5, grad_clip: float = 1.0):
    """The reference's hot loop with grad_accum > 1, restated (utils/train_utils.py:127-148 as it executes under accelerate):
    zero_grad() (:134) and optimizer.step() (:143) are no-ops except on sync micro-steps, and zero_grad comes BEFORE the forward, so
    a sync micro-step updates with the gradient of its own micro-batch alone, divided by grad_accum (accelerator.backward, :139);
    lr = scheduler(overall_step) of that micro-step (:129).  `loss_fn(sd, batch)` -> scalar loss.
    Returns (per-micro-step losses, per-micro-step sync flags, final sd)."""
    state: Dict[str, Dict[str, torch.Tensor]] = {}
    flags = accum_sync_flags(len(batches), grad_accum, batches_per_epoch)
    losses, t = [], 0
    for i, (batch, sync) in enumerate(zip(batches, flags)):
        if not sync:
            with torch.no_grad():
                losses.append(float(loss_fn(sd, batch)))
            continue
        t += 1
        loss, _, sd = train_step(lambda s: loss_fn(s, batch) / grad_accum, sd, state, t, lr_fn(i), weight_decay, grad_clip)
        losses.append(float(loss) * grad_accum)
    return losses, flags, sd
