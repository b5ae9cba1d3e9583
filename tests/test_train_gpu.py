"""The step variants of the reference's hot loop (utils/train_utils.py:127-148) beyond one plain step: gradient accumulation — as
the reference really executes it under accelerate (golden produced by its own run_train_model) and as a summed accumulation — and
torch.optim.AdamW's treatment of parameters the backward did not reach."""
import numpy as np
import pytest
import torch

from oracle import ref_models as R
from tests import cases as C

pytestmark = pytest.mark.gpu


def _bf_small(cfgo):
    import frankenstein_amd as fa
    from frankenstein_amd import synth
    from frankenstein_amd.models import brainformer as bf
    fa.set_compute_dtype("fp32")
    e = cfgo.encoder
    enc = bf.MAEConfig(window_size=e.window_size, n_electrodes=e.n_electrodes, patch_size=e.patch_size, dim=e.dim, n_layers=e.n_layers,
                       head_dim=e.head_dim, hidden_dim=e.hidden_dim, n_heads=e.n_heads, n_kv_heads=e.n_kv_heads)
    cfg = bf.Config(encoder=enc, n_output_tokens=cfgo.n_output_tokens, output_dim=cfgo.output_dim, dim=cfgo.dim, n_layers=cfgo.n_layers,
                    head_dim=cfgo.head_dim, hidden_dim=cfgo.hidden_dim, n_heads=cfgo.n_heads, n_kv_heads=cfgo.n_kv_heads)
    m = bf.BrainFormer(cfg)
    st = synth.make_state({k: tuple(v.shape) for k, v in m.state_dict().items()})
    m.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()}, strict=False)
    return m.cuda()


@pytest.fixture(autouse=True)
def _restore_dtype():
    yield
    import frankenstein_amd as fa
    fa.set_compute_dtype("bf16")


def _accum_cfg(tu, **kw):
    return tu.TrainConfig(exp_name="accum", batch_size=4, grad_accum=2, learning_rate=1e-3, weight_decay=1e-5, max_steps=11,
                          eval_interval=10 ** 6, use_scheduler=True, warmup_iters=4, lr_decay_iters=20, num_workers=0,
                          pin_memory=False, grad_clip=1.0, mixed_precision=False, **kw)


def _check_against_reference_run(z, losses, model):
    np.testing.assert_allclose(losses, z["losses"], rtol=1e-4)
    for k, p in model.named_parameters():
        np.testing.assert_allclose(p.detach().float().cpu().numpy(), z["param/" + k], rtol=1e-3, atol=5e-5, err_msg=k)


def test_grad_accum_train_steps_match_the_reference_run(golden):
    """train_step driven with accelerate's sync rule (GradAccumulation) over the micro-batches the reference's loader produced:
    every logged loss and the final parameters equal the reference's own run_train_model(grad_accum=2) (tests/golden/train_accum.npz)."""
    from frankenstein_amd.utils import train_utils as tu
    z = golden("train_accum")
    cfgo, xs, ys = C.train_accum(int(z["n_items"]))
    m = _bf_small(cfgo)
    cfg = _accum_cfg(tu)
    opt = tu.FusedAdamW(m, lr=cfg.learning_rate, weight_decay=cfg.weight_decay, grad_clip=cfg.grad_clip)
    acc = tu.GradAccumulation(cfg.grad_accum)
    bpe = int(z["n_items"]) // z["order"].shape[1]
    losses, probe = [], []
    for i, o in enumerate(z["order"]):
        idx = torch.from_numpy(o)
        probe.append(float(m.encoder.transformer.h[0].attn.qw.weight.detach().double().sum()))
        l = tu.train_step(m, (xs[idx].cuda(), ys[idx].cuda(), None), opt, i, cfg, sync=acc.sync(end_of_loader=i % bpe == bpe - 1))
        losses.append(float(l))
    changed = [a != b for a, b in zip(probe[:-1], probe[1:])]
    assert changed == [bool(a != b) for a, b in zip(z["probe_sums"][:-1], z["probe_sums"][1:])]
    _check_against_reference_run(z, losses, m)


def test_run_train_model_grad_accum_matches_the_reference_run(golden, tmp_path, monkeypatch):
    """the loop itself (run_train_model: sync rule incl. the end of the loader, lr from the micro-step count) on the recorded batches"""
    from frankenstein_amd.utils import train_utils as tu
    z = golden("train_accum")
    cfgo, xs, ys = C.train_accum(int(z["n_items"]))
    m = _bf_small(cfgo)
    bpe = int(z["n_items"]) // z["order"].shape[1]
    epochs = [[(xs[torch.from_numpy(o)], ys[torch.from_numpy(o)], torch.zeros(2)) for o in z["order"][e * bpe:(e + 1) * bpe]]
              for e in range(len(z["order"]) // bpe + 1)]

    class Loader:                       # yields the reference loader's batches, epoch by epoch
        def __init__(self):
            self.epoch = 0

        def __len__(self):
            return bpe

        def __iter__(self):
            e, self.epoch = self.epoch, self.epoch + 1
            return iter(epochs[e])

    monkeypatch.setattr(tu, "prepare_data_loaders", lambda tr, va, c: (Loader(), []))
    losses = []
    tu.run_train_model(m, (None, None), _accum_cfg(tu), save_folder=tmp_path, logger=lambda d, s: losses.append(d["train/loss"]))
    assert len(losses) == 12
    _check_against_reference_run(z, losses, m)


def test_grad_accum_sum_equals_one_full_batch():
    """accumulate="sum": two half micro-batches through train_step(micro_step = 0 / 1) == one step on the whole batch (weight
    gradients accumulate straight into the arena across the micro-steps, engine.wgrad); bounds of tests/test_dp_gpu.py."""
    from frankenstein_amd.utils import train_utils as tu
    cfgo, xs, ys = C.train_accum(8)
    x, y = xs[:4].cuda(), ys[:4].cuda()
    out = {}
    for name, k in (("full", 1), ("accum", 2)):
        m = _bf_small(cfgo)
        cfg = tu.TrainConfig(mixed_precision=False, use_scheduler=False, learning_rate=1e-3, grad_accum=k)
        opt = tu.FusedAdamW(m, lr=1e-3, weight_decay=cfg.weight_decay, grad_clip=cfg.grad_clip)
        for step in range(2):
            if k == 1:
                tu.train_step(m, (x, y, None), opt, step, cfg)
            else:
                for ms in range(2):
                    tu.train_step(m, (x[2 * ms:2 * ms + 2], y[2 * ms:2 * ms + 2], None), opt, step, cfg, micro_step=ms, accumulate="sum")
        out[name] = opt.arena.flat.detach().cpu().numpy()
        assert float(opt.arena.grad.abs().max()) == 0.0
    diff = np.abs(out["full"] - out["accum"])
    assert np.quantile(diff, 0.99) < 2e-5 and diff.max() < 5e-3, (np.quantile(diff, 0.99), diff.max())


class _TwoHeads(torch.nn.Module):
    """`b` is used only when use_b is set: a parameter the backward does not always reach"""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(5)
        self.a = torch.nn.Parameter(torch.randn(7, 5, generator=g))
        self.b = torch.nn.Parameter(torch.randn(3, 5, generator=g))
        self.c = torch.nn.Parameter(torch.randn(5, generator=g))

    def forward(self, x, use_b):
        y = (x @ self.a.t()).sum() + (self.c * x).sum()
        return y + (x @ self.b.t()).pow(2).sum() if use_b else y


def test_fused_adamw_skips_parameters_without_gradient_like_torch():
    """torch.optim.AdamW (utils/train_utils.py:117-119) with zero_grad(set_to_none=True) (:134): a parameter whose grad is None is not
    decayed, its moments do not move and its bias-correction step count does not advance."""
    from frankenstein_amd.utils import train_utils as tu
    x = torch.randn(4, 5, generator=torch.Generator().manual_seed(6))
    pattern = [True, False, False, True, False, True]
    ref = _TwoHeads()
    ropt = torch.optim.AdamW(ref.parameters(), lr=1e-2, weight_decay=0.1)
    for use_b in pattern:
        ropt.zero_grad(set_to_none=True)
        ref(x, use_b).backward()
        torch.nn.utils.clip_grad_value_(ref.parameters(), 1.0)
        ropt.step()
    m = _TwoHeads().cuda()
    opt = tu.FusedAdamW(m, lr=1e-2, weight_decay=0.1, grad_clip=1.0)
    xd = x.cuda()
    for i, use_b in enumerate(pattern):
        m(xd, use_b).backward()
        opt.step()
        if i == 1:
            assert opt._lag is not None and opt._lag[1] == 1
    for (k, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().numpy(), rtol=2e-6, atol=2e-7, err_msg=k)
    sd = opt.state_dict()
    assert sd["lag"] == [0, 3, 0]


def test_fused_adamw_step_without_any_reached_parameter_warns_and_mark_touched_steps():
    """A gradient written into arena.grad directly is invisible to the post-accumulate-grad hooks: step() then leaves every parameter
    alone like torch.optim.AdamW with all-None grads — but says so (RuntimeWarning); after mark_touched() the same gradient is applied."""
    import warnings
    from frankenstein_amd.utils import train_utils as tu
    m = _TwoHeads().cuda()
    opt = tu.FusedAdamW(m, lr=1e-2, weight_decay=0.0, grad_clip=None)
    before = opt.arena.flat.clone()
    opt.arena.grad.fill_(0.5)
    with pytest.warns(RuntimeWarning, match="mark_touched"):
        opt.step()
    assert torch.equal(opt.arena.flat, before)
    opt.arena.grad.fill_(0.5)
    opt.mark_touched()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        opt.step()
    assert float((opt.arena.flat - before).abs().max()) > 1e-3
