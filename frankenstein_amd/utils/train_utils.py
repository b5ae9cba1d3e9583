"""Training driver with the reference's ``utils/train_utils.py`` surface (TrainConfig, count_parameters,
init_lr_scheduler, prepare_data_loaders, run_train_model, simple_train_model) plus the explicit
``train_step`` the north star asks for — the loop body of utils/train_utils.py:128-148:

    lr = get_lr(step) -> zero_grad -> loss, _ = model(inputs, labels, date_info) -> backward
    (-> DP gradient mean) -> clip_grad_value_(1.0) -> AdamW.step()

MI355X-first implementation: all trainable parameters live in ONE flat fp32 arena (params, grads, Adam
moments), so clip + AdamW + zero_grad is a single HBM-bound kernel launch (fk_adamw_step) and the data-parallel
gradient exchange is a handful of large contiguous RCCL all-reduces issued from autograd hooks while the
backward is still running (one process per GPU, torch.distributed 'nccl' = RCCL over xGMI).  No accelerate /
wandb dependency (accelerate's DDP wrapping, utils/train_utils.py:97-101,122, is what GradSync replaces).
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from pathlib import Path
from typing import Callable, List, Optional

import torch

from .. import engine as E
from .. import kernels as K


@dataclass
class TrainConfig():
    exp_name: str = 'default'

    batch_size: int = 256
    grad_accum: int = 1

    p_augs: float = 0.0

    learning_rate: float = 1e-3
    weight_decay: float = 1e-5

    max_steps: int = 100_000
    eval_interval: int = 1_000

    use_scheduler: bool = True
    warmup_iters: int = 2_000
    lr_decay_iters: int = 50_000

    num_workers: int = 3
    pin_memory: bool = True

    grad_clip: float = 1.0
    mixed_precision: bool = True

    visualize_predictions: bool = False


def count_parameters(model):
    n_trainable = sum(p.numel() for p in model.parameters() if p.requires_grad)
    n_total = sum(p.numel() for p in model.parameters())
    print(f"Total: {n_total/1e6:.2f}M, Trainable: {n_trainable/1e6:.2f}M")
    return n_total, n_trainable


def init_lr_scheduler(config) -> Callable[[int], float]:
    """linear warm-up to lr over warmup_iters, cosine to lr/10 at lr_decay_iters, constant after
    (utils/train_utils.py:49-72)."""
    lr, warm, decay = config.learning_rate, config.warmup_iters, config.lr_decay_iters
    min_lr = lr / 10
    constant = not config.use_scheduler

    def get_lr(it):
        if constant:
            return lr
        if it < warm:
            return lr * it / warm
        if it > decay:
            return min_lr
        ratio = (it - warm) / (decay - warm)
        assert 0 <= ratio <= 1
        return min_lr + 0.5 * (1.0 + math.cos(math.pi * ratio)) * (lr - min_lr)

    return get_lr


def prepare_data_loaders(train_dataset, val_dataset, config):
    batch_size = config.batch_size // config.grad_accum
    mk = lambda ds, shuffle: torch.utils.data.DataLoader(ds, batch_size=batch_size, shuffle=shuffle,
                                                         num_workers=config.num_workers, pin_memory=config.pin_memory)
    return mk(train_dataset, True), mk(val_dataset, False)


# ------------------------------------------------------------------------------------------------ arena + optimizer
def _unique_trainable(model) -> List[torch.nn.Parameter]:
    seen, out = set(), []
    for p in model.parameters():
        if p.requires_grad and id(p) not in seen:
            seen.add(id(p))
            out.append(p)
    return out


class ParamArena:
    """Flat fp32 storage for all trainable parameters (+ grads): every Parameter becomes a view into ``flat``
    and its ``.grad`` a view into ``grad`` (tensors padded to 16 bytes so vector kernels stay aligned)."""

    def __init__(self, model: torch.nn.Module):
        self.params = _unique_trainable(model)
        assert self.params, "model has no trainable parameters"
        dev = self.params[0].device
        self.offsets, off = [], 0
        for p in self.params:
            assert p.dtype == torch.float32 and p.device == dev, "parameters must be fp32 masters on one device"
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4
        self.numel = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = self.flat[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = self.grad[o:o + p.numel()].view(p.shape)
                p._fk_arena = True          # engine.wgrad may accumulate weight gradients straight into the arena
        E.bump_weight_epoch()

    def rebind_grads(self):
        """(re)attach .grad views, e.g. after someone called zero_grad(set_to_none=True)."""
        for p, o in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                g = self.grad[o:o + p.numel()].view(p.shape)
                if p.grad is not None:
                    g.copy_(p.grad)
                p.grad = g


class GradSync:
    """Data-parallel gradient exchange (what accelerate's DDP wrap does implicitly in the reference,
    utils/train_utils.py:122,139): contiguous buckets of the flat grad arena are all-reduced (SUM; the mean's
    1/world is folded into the optimizer kernel) as soon as every parameter of the bucket has its gradient,
    asynchronously on the process group's communication stream, so the exchange overlaps the remaining backward."""

    def __init__(self, arena: ParamArena, group=None, bucket_bytes: int = 8 << 20, always: bool = False, measure: bool = False,
                 register: bool = True):
        import torch.distributed as dist
        self.dist, self.group, self.arena = dist, group, arena
        # measure=True: bracket the wait for the buckets in finish() with timestamps on the compute stream (HIP events; wall clock for
        # CPU tensors) -> exposed_ms_mean(): how long the step sat waiting for the exchange after its last backward kernel
        self.measure, self._exposed = measure, []
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.enabled = True
        # always=True: issue the collectives also in a one-rank group (identity) — how the RCCL path is rehearsed on a one-GPU box
        self.active = self.world > 1 or (always and dist.is_available() and dist.is_initialized())
        self.buckets: List[tuple] = []           # (start, end, n_params)
        self.bucket_of: List[int] = []
        cap = max(1, bucket_bytes // 4)
        start, count = 0, 0
        for i, (p, o) in enumerate(zip(arena.params, arena.offsets)):
            end = o + (p.numel() + 3) // 4 * 4
            if count and end - start > cap:
                self.buckets.append((start, o, count))
                start, count = o, 0
            self.bucket_of.append(len(self.buckets))
            count += 1
        self.buckets.append((start, arena.numel, count))
        self._pending = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._works = []
        if self.active:
            # replicas start from rank 0's parameters: what wrapping the model in DDP does in the reference
            # (accelerator.prepare, utils/train_utils.py:122) — identical seeds are not relied upon
            self.broadcast_parameters()
            if register:                                       # (FusedAdamW calls on_grad from its own per-parameter hook instead: one
                for i, p in enumerate(arena.params):           # Python hook per parameter costs ~5 us of host time per backward)
                    hook = self._make_hook(self.bucket_of[i])
                    p.register_post_accumulate_grad_hook(hook)     # also fires when engine.wgrad wrote the gradient itself

    def _mark(self):
        if self.arena.grad.is_cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            return ev
        import time
        return time.perf_counter()

    def exposed_ms_mean(self, last: Optional[int] = None) -> float:
        """mean over the last `last` measured steps (all by default) of the compute stream's wait for the exchange, in ms; call after
        a device synchronisation"""
        pairs = self._exposed[-last:] if last else self._exposed
        if not pairs:
            return 0.0
        ms = [a.elapsed_time(b) if hasattr(a, "elapsed_time") else (b - a) * 1e3 for a, b in pairs]
        return sum(ms) / len(ms)

    def broadcast_parameters(self, src: int = 0):
        """Every rank takes the master parameters of group rank `src` (one collective over the flat arena)."""
        if self.active:
            gsrc = self.dist.get_global_rank(self.group, src) if self.group is not None else src
            self.dist.broadcast(self.arena.flat, src=gsrc, group=self.group)
            E.bump_weight_epoch()

    def on_grad(self, i: int) -> None:
        """parameter i of the arena has its gradient: counts towards its bucket, which goes out when complete"""
        if not (self.active and self.enabled):
            return
        b = self.bucket_of[i]
        self._pending[b] += 1
        if self._pending[b] == self.buckets[b][2]:
            self._launch(b)

    def _make_hook(self, b: int):
        def hook(_param):
            if not self.enabled:
                return
            self._pending[b] += 1
            if self._pending[b] == self.buckets[b][2]:
                self._launch(b)
        return hook

    def _launch(self, b: int):
        s, e, _ = self.buckets[b]
        self._launched[b] = True
        side = E.wgrad_stream()
        if side is not None:
            # the bucket may hold gradients produced on the weight-gradient stream AND on the main stream: order the
            # collective after both by issuing it from the side stream once that has caught up with the main one
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                w = self.dist.all_reduce(self.arena.grad[s:e], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            w = self.dist.all_reduce(self.arena.grad[s:e], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._works.append(w)

    def finish(self) -> float:
        """Launch whatever has not gone out (unused parameters), wait for all buckets; returns the factor that turns
        the summed gradient into the DDP mean."""
        if self.active and self.enabled:
            for b in range(len(self.buckets)):
                if not self._launched[b]:
                    self._launch(b)
            side = E.wgrad_stream()
            t0 = self._mark() if self.measure else None
            for w in self._works:
                if side is not None:
                    with torch.cuda.stream(side):
                        w.wait()
                else:
                    w.wait()
            if self.measure:
                if side is not None:
                    torch.cuda.current_stream().wait_stream(side)
                self._exposed.append((t0, self._mark()))
        self._works.clear()
        self._pending = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        return 1.0 / self.world if self.enabled else 1.0


class FusedAdamW:
    """torch.optim.AdamW(model.parameters(), lr, weight_decay) semantics (betas .9/.999, eps 1e-8, decoupled decay on
    every parameter, utils/train_utils.py:117-119) + clip_grad_value_ (:142), as one kernel over the arena."""

    def __init__(self, model: torch.nn.Module, lr: float = 1e-3, weight_decay: float = 1e-2, betas=(0.9, 0.999),
                 eps: float = 1e-8, grad_clip: Optional[float] = None, group=None, bucket_bytes: int = 8 << 20,
                 overlap_wgrad: Optional[bool] = None, sync_always: bool = False):
        self.arena = ParamArena(model)
        if overlap_wgrad is None:
            overlap_wgrad = os.environ.get("FK_WGRAD_STREAM", "0") == "1"        # off by default: see engine.py (no speed-up with one-block-per-CU grids)
        if overlap_wgrad and self.arena.flat.is_cuda and E.wgrad_stream() is None:
            E.enable_wgrad_stream(True)
        self.m = torch.zeros_like(self.arena.flat)
        self.v = torch.zeros_like(self.arena.flat)
        self.param_groups = [dict(params=self.arena.params, lr=lr, weight_decay=weight_decay, betas=betas, eps=eps)]
        self.grad_clip = grad_clip
        self.t = 0
        # torch.optim.AdamW leaves a parameter alone whose .grad is None (after the reference's zero_grad(set_to_none=True),
        # utils/train_utils.py:134, that is a parameter the backward never reached): no decay, no moment update, its own step count.
        # The arena's gradients are never None, so "was reached" is recorded by a post-accumulate-grad hook per parameter (it also
        # fires when engine.wgrad / engine.norm_bwd wrote the gradient into the arena themselves).
        self._touched = [False] * len(self.arena.params)
        self._lag: Optional[List[int]] = None       # per parameter: optimizer steps it sat out (None: no parameter ever did)
        self.sync = GradSync(self.arena, group, bucket_bytes, always=sync_always, register=False)
        for i, p in enumerate(self.arena.params):
            p.register_post_accumulate_grad_hook(self._make_touch_hook(i))

    def _make_touch_hook(self, i: int):
        if self.sync.active:
            def hook(_param):                  # one hook per parameter serves both: "reached by the backward" and the bucket's readiness
                self._touched[i] = True
                self.sync.on_grad(i)
        else:
            def hook(_param):
                self._touched[i] = True
        return hook

    def mark_touched(self, flags=None) -> None:
        """Declare which parameters received a gradient since the last step() when the gradients were not produced by an eager
        backward (a replayed hipGraph, a hand-written arena.grad): `flags` is a sequence of booleans, None = all of them."""
        n = len(self.arena.params)
        self._touched = [True] * n if flags is None else [bool(f) for f in flags]
        assert len(self._touched) == n

    def zero_grad(self, set_to_none: bool = True):
        self.arena.grad.zero_()
        self.arena.rebind_grads()
        self._touched = [False] * len(self.arena.params)

    def _update_ranges(self):
        """[(start, end, t)] element ranges of the arena to update with bias-correction step t: one range in the usual case (every
        parameter reached by the backward, none ever skipped); otherwise maximal runs of adjacent reached parameters with equal
        step counts.  Parameters the backward did not reach are left out (torch.optim.AdamW: `if p.grad is None: continue`)."""
        a = self.arena
        if all(self._touched) and self._lag is None:
            return [(0, a.numel, self.t)]
        if self._lag is None:
            self._lag = [0] * len(a.params)
        out = []
        for i, (p, o) in enumerate(zip(a.params, a.offsets)):
            if not self._touched[i]:
                self._lag[i] += 1
                continue
            e, t = o + (p.numel() + 3) // 4 * 4, self.t - self._lag[i]
            if out and out[-1][1] == o and out[-1][2] == t:
                out[-1] = (out[-1][0], e, t)
            else:
                out.append((o, e, t))
        return out

    def step(self):
        """clip + AdamW + zero_grad over the arena: ONE launch when every parameter received a gradient since the last step (every
        model of the reference, every step); a parameter that did not is skipped like torch.optim.AdamW skips `grad is None` — no
        weight decay, no moment decay, its bias-correction count does not advance (utils/train_utils.py:117-119,134,143)."""
        g = self.param_groups[0]
        a = self.arena
        p0, p1 = a.params[0], a.params[-1]
        if p0.data_ptr() != a.flat.data_ptr() + 4 * a.offsets[0] or p1.data_ptr() != a.flat.data_ptr() + 4 * a.offsets[-1]:
            raise RuntimeError("parameters no longer live in the optimizer's arena (model.to() / .float() after constructing "
                               "FusedAdamW?): build the optimizer after the model is on its final device")
        self.arena.rebind_grads()
        scale = self.sync.finish()
        side = E.wgrad_stream()
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)      # join the weight-gradient stream before the update
        if not any(self._touched):
            # torch.optim.AdamW is a silent no-op when every .grad is None, and so is this; but here the usual way to get there is a
            # gradient that did not come from an eager backward (arena.grad written by hand, a replayed graph, gradients produced before
            # load_state_dict re-created the hooks) — say so instead of silently leaving the parameters alone.
            import warnings
            warnings.warn("FusedAdamW.step(): no parameter was reached by a backward since the last step, so nothing is updated; "
                          "if the gradients were written into arena.grad directly (or by a replayed graph), call mark_touched() first",
                          RuntimeWarning, stacklevel=2)
        self.t += 1
        for s0, s1, t in self._update_ranges():
            K.adamw_step_(a.flat[s0:s1], a.grad[s0:s1], self.m[s0:s1], self.v[s0:s1], t, g['lr'], g['betas'][0], g['betas'][1],
                          g['eps'], g['weight_decay'], clip=self.grad_clip or 0.0, grad_scale=scale, zero_grad=True)
        self._touched = [False] * len(a.params)
        E.bump_weight_epoch()
        E.refresh_shadows(self.arena.params)       # every weight shadow re-packed from the new masters in one launch

    def state_dict(self):
        return dict(t=self.t, m=self.m, v=self.v, lag=None if self._lag is None else list(self._lag),
                    param_groups=[{k: v for k, v in g.items() if k != 'params'} for g in self.param_groups])

    def load_state_dict(self, sd):
        """Resume: step count, both moment arenas and the hyper-parameters (the parameters themselves travel with the model)."""
        assert sd["m"].numel() == self.m.numel() and sd["v"].numel() == self.v.numel(), "optimizer state of another model"
        self.t = int(sd["t"])
        self._lag = None if sd.get("lag") is None else [int(x) for x in sd["lag"]]
        self.m.copy_(sd["m"].to(self.m.device))
        self.v.copy_(sd["v"].to(self.v.device))
        for g, src in zip(self.param_groups, sd.get("param_groups", [])):
            g.update({k: v for k, v in src.items() if k != 'params'})


def enable_fused_head_loss(model: torch.nn.Module, on: bool = True) -> int:
    """Training loops that only use the loss (the reference's run_train_model discards the second output of model(...),
    utils/train_utils.py:138-139) can let the vocabulary heads skip the [rows, 50257] logits: sets `fuse_head_loss` on every sub-module
    that implements it (GPT, the notebook CE BrainFormer); their forward then returns (loss, None).  Returns how many were switched."""
    n = 0
    for m in model.modules():
        if type(m).__name__ in ("GPT", "BrainFormerCE"):
            m.fuse_head_loss = on
            n += 1
    return n


# ------------------------------------------------------------------------------------------------ the step
# What grad_accum > 1 means.  The reference wraps the step in `with accelerator.accumulate(model)` but zeroes the gradients BEFORE the
# forward (utils/train_utils.py:133-134), and accelerate's optimizer wrapper executes zero_grad() and step() only on "sync" micro-steps.
# So on a sync micro-step the gradients of the earlier micro-batches are wiped before the forward, and the update uses the gradient of
# THAT micro-batch alone, scaled by 1 / grad_accum (accelerator.backward divides the loss); the other micro-batches only contribute their
# logged loss.  tests/golden/train_accum.npz pins this against the reference's own loop running under accelerate.
#   "reference": reproduce exactly that (default: a drop-in gives the reference's numbers).  The backward of a non-sync micro-step
#                cannot influence anything, so only its forward runs (the loss is logged, :147).
#   "sum":       what the construct is meant to do: gradients of all grad_accum micro-batches summed (each loss / grad_accum), one
#                update — equal to one step on the concatenated batch for a mean-reduced loss.
ACCUMULATE = "reference"


class GradAccumulation:
    """accelerate's GradientState as the reference drives it (Accelerator(gradient_accumulation_steps=k), utils/train_utils.py:98,133):
    a micro-step is a sync step when it is the k-th since the last reset, or the last batch of the loader — which also resets the
    count (accelerate's sync_with_dataloader default)."""

    def __init__(self, grad_accum: int):
        self.k, self.count = max(1, int(grad_accum)), 0

    def sync(self, end_of_loader: bool = False) -> bool:
        if end_of_loader:
            self.count = 0
            return True
        self.count += 1
        return self.count % self.k == 0


def train_step(model, batch, optimizer: FusedAdamW, step: int, cfg: TrainConfig, scheduler=None,
               micro_step: int = 0, sync: Optional[bool] = None, accumulate: Optional[str] = None):
    """One iteration of the reference's hot loop (utils/train_utils.py:128-148): lr = get_lr(step) -> zero_grad -> forward ->
    backward (-> DP gradient mean) -> clip_grad_value_ -> AdamW.step().  With cfg.grad_accum > 1 call it once per micro-batch;
    `sync` says whether this micro-step ends an accumulation window (default: micro_step == grad_accum - 1; run_train_model passes
    accelerate's rule incl. the end of the loader), `accumulate` picks the semantics (module default ACCUMULATE, see above).
    Returns the un-scaled loss of this micro-batch (what the reference logs)."""
    get_lr = scheduler or init_lr_scheduler(cfg)
    lr = get_lr(step)
    for g in optimizer.param_groups:
        g['lr'] = lr
    inputs, labels, date_info = batch
    k = max(1, cfg.grad_accum)
    mode = accumulate or ACCUMULATE
    assert mode in ("reference", "sum"), mode
    last = (micro_step == k - 1) if sync is None else bool(sync)
    optimizer.sync.enabled = last
    if k > 1 and mode == "reference":
        if not last:
            with torch.no_grad():
                loss, _ = model(inputs, labels, date_info=date_info)
            return loss.detach()
        optimizer.zero_grad()             # :134 on a sync micro-step: whatever was accumulated is dropped
    loss, _ = model(inputs, labels, date_info=date_info)
    (loss / k if k > 1 else loss).backward()
    if last:
        optimizer.step()
    return loss.detach()


class GraphedTrainStep:
    """The hot loop's forward + backward captured ONCE as a hipGraph and replayed per batch; the update (whose learning
    rate changes every step) runs eagerly after each replay.  For the small configurations (gpt2-nano, SimpleMAE B=32)
    a step is ~300 launches of a few microseconds each and the Python/ctypes launch path, not the GPU, sets the step
    time; replaying a graph removes it.  Requirements: static batch shape, one process (the gradient exchange is
    hook-driven and not captured), grad_accum == 1, a forward without host-side randomness.  With the weight-gradient side
    stream on (FusedAdamW(overlap_wgrad=True)) the dW GEMMs become a parallel branch of the graph.  Same numbers as ``train_step``: the captured kernels are the ones the eager step launches."""

    def __init__(self, model, batch, optimizer: FusedAdamW, cfg: TrainConfig, scheduler=None, warmup: int = 2):
        if _dist_info()[1] != 1:
            raise RuntimeError("GraphedTrainStep is single-process: the bucketed gradient exchange is not captured")
        if cfg.grad_accum != 1:
            raise RuntimeError("GraphedTrainStep needs grad_accum == 1")
        self.model, self.optimizer, self.cfg = model, optimizer, cfg
        self.get_lr = scheduler or init_lr_scheduler(cfg)
        self.static = tuple(t.clone() if torch.is_tensor(t) else t for t in batch)
        optimizer.sync.enabled = False
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):                  # warm-up off the default stream: workspaces, shadows, autotune
            for _ in range(warmup):
                self._fwd_bwd()
        cur.wait_stream(side)
        optimizer.zero_grad()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._fwd_bwd()
        self._touched = list(optimizer._touched)       # which parameters the captured backward reaches (hooks do not run on replay)
        optimizer.zero_grad()                          # capture recorded the launches, it did not run them

    def _fwd_bwd(self):
        inputs, labels, date_info = self.static
        loss, _ = self.model(inputs, labels, date_info=date_info)
        loss.backward()
        side = E.wgrad_stream()
        if side is not None:                            # join the weight-gradient branch: a captured graph must end on one stream
            torch.cuda.current_stream().wait_stream(side)
        return loss.detach()

    def __call__(self, batch, step: int):
        for dst, src in zip(self.static, batch):
            if torch.is_tensor(dst):
                if dst.shape != src.shape:
                    raise RuntimeError(f"GraphedTrainStep was captured for {tuple(dst.shape)}, got {tuple(src.shape)}")
                dst.copy_(src, non_blocking=True)
        lr = self.get_lr(step)
        for g in self.optimizer.param_groups:
            g['lr'] = lr
        self.graph.replay()
        self.optimizer.mark_touched(self._touched)
        self.optimizer.step()
        return self.loss


def _dist_info():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_batch(batch, rank: int, world: int):
    """accelerate ``split_batches=True`` (utils/train_utils.py:100): the loader batch is the GLOBAL batch and rank r
    gets the r-th contiguous slice."""
    if world == 1:
        return batch
    n = batch[0].shape[0]
    assert n % world == 0, f"global batch {n} not divisible by world size {world}"
    k = n // world
    return tuple(t[rank * k:(rank + 1) * k] if torch.is_tensor(t) else t for t in batch)


def save_model(model, path) -> None:
    """safetensors checkpoint with the reference's key names (utils/train_utils.py:172).  Parameters are views into the
    flat arena, which safetensors refuses to serialise as-is (shared storage), so tensors are cloned first."""
    import safetensors.torch
    sd = {k: v.detach().clone().contiguous() for k, v in model.state_dict().items() if v is not None}
    safetensors.torch.save_file(sd, str(path))


def run_train_model(model, datasets, config, project_name='transformer', save_folder=Path('logs'), logger=None):
    """Same contract as the reference: trains until max_steps, evaluates every eval_interval on the main process and
    saves the best model as safetensors.  Launch one process per GPU (torchrun) for data parallelism."""
    import safetensors.torch
    torch.manual_seed(42)
    rank, world = _dist_info()
    device = torch.device('cuda', int(os.environ.get('LOCAL_RANK', 0)))
    torch.cuda.set_device(device)
    E.set_compute_dtype('bf16' if config.mixed_precision else 'fp32')
    save_folder = Path(save_folder) / config.exp_name
    save_folder.mkdir(parents=True, exist_ok=True)
    train_dataset, val_dataset = datasets
    train_loader, val_loader = prepare_data_loaders(train_dataset, val_dataset, config)
    model.to(device).float()
    optimizer = FusedAdamW(model, lr=config.learning_rate, weight_decay=config.weight_decay, grad_clip=config.grad_clip)
    scheduler = init_lr_scheduler(config)
    to_dev = lambda b: tuple(t.to(device, non_blocking=True) if torch.is_tensor(t) else t for t in b)
    overall_step, best_val = 0, float('inf')
    accum = GradAccumulation(config.grad_accum)
    done = False
    while not done:
        n_batches = len(train_loader)
        for i_batch, batch in enumerate(train_loader):
            batch = shard_batch(to_dev(batch), rank, world)
            loss = train_step(model, batch, optimizer, overall_step, config, scheduler,
                              sync=accum.sync(end_of_loader=i_batch == n_batches - 1))
            overall_step += 1
            if rank == 0:
                print('*', end='')
                if logger is not None:
                    logger({'train/loss': float(loss), 'lr': optimizer.param_groups[0]['lr']}, overall_step)
            if overall_step % config.eval_interval == 0 and rank == 0:
                model.eval()
                vals = []
                with torch.no_grad():
                    for vb in val_loader:
                        inputs, labels, date_info = to_dev(vb)
                        vals.append(model(inputs, labels, date_info)[0].float())
                mean_val = float(torch.stack(vals).mean())
                print(f"overall_steps {overall_step}: {float(loss)}")
                print(f"val loss: {mean_val}")
                if logger is not None:
                    logger({'val/loss': mean_val}, overall_step)
                if mean_val < best_val:
                    best_val = mean_val
                    path = save_folder / f"step_{overall_step}_loss_{mean_val:.4f}.safetensors"
                    save_model(model, path)
                    print('saved model: ', path.name)
                model.train()
            if overall_step > config.max_steps:
                print('Complete training')
                done = True
                break
    return model


def simple_train_model(model, datasets, config, project_name='transformer'):
    """Single-process variant (utils/train_utils.py:189-260): no gradient clipping."""
    cfg = TrainConfig(**{**config.__dict__, 'grad_clip': 0.0})
    return run_train_model(model, datasets, cfg, project_name, Path('logs'))
