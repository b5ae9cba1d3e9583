"""CPU tests of the host logic: reference-compatible module surface (state-dict keys), LR schedule, batch
sharding, and the data-parallel gradient exchange (GradSync) under gloo with world_size 2."""
import os
import socket

import numpy as np
import pytest
import torch

from oracle import ref_models as R
from oracle import ref_train as RT
from tests import cases as C


def test_state_dict_keys_match_reference_layout():
    from frankenstein_amd.models import brainformer as bf
    from frankenstein_amd.models import gpt2_model as g2
    from frankenstein_amd.models.notebook_models import BrainEncoder, Franky
    cfgo, _, _ = C.bf_l1_small()
    e = cfgo.encoder
    enc = bf.MAEConfig(window_size=e.window_size, n_electrodes=e.n_electrodes, patch_size=e.patch_size, dim=e.dim,
                       n_layers=e.n_layers, head_dim=e.head_dim, hidden_dim=e.hidden_dim, n_heads=e.n_heads, n_kv_heads=e.n_kv_heads)
    cfg = bf.Config(encoder=enc, n_output_tokens=8, output_dim=12, dim=64, n_layers=2, head_dim=8, hidden_dim=96, n_heads=4, n_kv_heads=4)
    m = bf.BrainFormer(cfg)
    sd = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    want = dict(R.brainformer_shapes(cfgo, "to_motion"))
    want["encoder.attn_mask"] = (128, 128)          # registered buffer of the reference (models/brainformer.py:298)
    assert sd == want
    assert m.encoder.attn_mask.dtype == torch.bool
    assert torch.equal(m.encoder.attn_mask, R.block_causal_mask(128, 16))
    gcfg = R.gpt_config(block_size=64, vocab_size=211, n_layer=2, n_head=4, n_embd=64, bias=True)
    g = g2.GPT(g2.GPTConfig(block_size=64, vocab_size=211, n_layer=2, n_head=4, n_embd=64, bias=True))
    sd = {k: tuple(v.shape) for k, v in g.state_dict().items()}
    want = dict(R.gpt_shapes(gcfg))
    want["lm_head.weight"] = want["transformer.wte.weight"]
    assert sd == want
    assert g.lm_head.weight is g.transformer.wte.weight
    fr = Franky(BrainEncoder(cfg), g)
    assert "brain_model.perceiver.to_words.weight" in fr.state_dict() and "llm_model.lm_head.weight" in fr.state_dict()
    # defaults of the config dataclasses (models/brainformer.py:17-53)
    d = bf.MAEConfig()
    assert (d.window_size, d.patch_size, d.dim, d.n_layers, d.head_dim, d.hidden_dim, d.n_heads) == (1024, 48, 256, 4, 32, 1024, 8)
    c = bf.Config(encoder=d)
    assert (c.n_output_tokens, c.output_dim, c.head_dim, c.hidden_dim, c.n_heads) == (32, 1024, 16, 512, 4)


def test_rope_cache_matches_reference_golden(golden):
    from frankenstein_amd.models.brainformer import build_complex_rope_cache
    z = golden("ops")
    c = build_complex_rope_cache(8, 16, 10000)
    assert c.dtype == torch.complex64
    np.testing.assert_array_equal(c.real.numpy(), z["rope_cache_re"])
    np.testing.assert_array_equal(c.imag.numpy(), z["rope_cache_im"])


def test_lr_schedule_and_train_config(golden):
    from frankenstein_amd.utils.train_utils import TrainConfig, init_lr_scheduler, shard_batch
    z = golden("ops")
    cfg = TrainConfig()
    assert (cfg.batch_size, cfg.learning_rate, cfg.weight_decay, cfg.grad_clip, cfg.warmup_iters, cfg.lr_decay_iters) == (256, 1e-3, 1e-5, 1.0, 2000, 50000)
    f = init_lr_scheduler(cfg)
    np.testing.assert_allclose([f(int(i)) for i in z["lr_its"]], z["lr_vals"], rtol=1e-12)
    assert [RT.get_lr(i) for i in (0, 1000, 50001)] == [f(i) for i in (0, 1000, 50001)]
    assert init_lr_scheduler(TrainConfig(use_scheduler=False))(12345) == 1e-3
    b = (torch.arange(8).view(8, 1), torch.arange(8), None)
    s = shard_batch(b, 1, 4)
    assert s[0].flatten().tolist() == [2, 3] and s[1].tolist() == [2, 3] and s[2] is None


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    import torch.distributed as dist
    from frankenstein_amd.utils.train_utils import GradSync, ParamArena
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.Tanh(), torch.nn.Linear(53, 11), torch.nn.Tanh(), torch.nn.Linear(11, 3))
    arena = ParamArena(net)
    sync = GradSync(arena, bucket_bytes=4096)        # several buckets
    assert len(sync.buckets) > 1
    x = torch.randn(8, 37, generator=torch.Generator().manual_seed(1))
    y = torch.randn(8, 3, generator=torch.Generator().manual_seed(2))
    k = 8 // world
    xs, ys = x[rank * k:(rank + 1) * k], y[rank * k:(rank + 1) * k]
    ((net(xs) - ys) ** 2).mean().backward()
    scale = sync.finish()
    g = arena.grad * scale
    # reference: full batch on one process
    ref = torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.Tanh(), torch.nn.Linear(53, 11), torch.nn.Tanh(), torch.nn.Linear(11, 3))
    ref.load_state_dict(net.state_dict())
    ((ref(x) - y) ** 2).mean().backward()
    flat = torch.cat([torch.nn.functional.pad(p.grad.flatten(), (0, (-p.numel()) % 4)) for p in ref.parameters()])
    out[rank] = float((g - flat).abs().max())
    # second iteration re-arms the buckets
    arena.grad.zero_()
    ((net(xs) - ys) ** 2).mean().backward()
    g2 = arena.grad * sync.finish()
    out[rank + world] = float((g2 - flat).abs().max())
    dist.destroy_process_group()


def test_gradsync_gloo_world2_equals_full_batch():
    """DP equivalence: 2 ranks x half batch, mean-reduced == 1 rank x full batch."""
    import torch.multiprocessing as mp
    world = 2
    out = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert len(out) == 2 * world and max(out.values()) < 1e-6, dict(out)



def _accum_worker(rank, world, port, out):
    """accumulate="sum" under DP: the exchange is switched off on the first micro-step (GradSync.enabled, what accelerate's no_sync
    does) and the buckets go out on the second, carrying the accumulated sum."""
    import torch.distributed as dist
    from frankenstein_amd.utils.train_utils import GradSync, ParamArena
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    mk = lambda: torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.Tanh(), torch.nn.Linear(53, 11), torch.nn.Tanh(), torch.nn.Linear(11, 3))
    net = mk()
    arena = ParamArena(net)
    sync = GradSync(arena, bucket_bytes=4096)
    x = torch.randn(16, 37, generator=torch.Generator().manual_seed(1))
    y = torch.randn(16, 3, generator=torch.Generator().manual_seed(2))
    k = 16 // world
    xs, ys = x[rank * k:(rank + 1) * k], y[rank * k:(rank + 1) * k]
    launched = []
    orig = sync._launch
    sync._launch = lambda b: (launched.append(b), orig(b))[1]
    for ms in range(2):                              # two micro-batches of the rank's shard, loss / grad_accum each
        sync.enabled = ms == 1
        h = k // 2
        (((net(xs[ms * h:(ms + 1) * h]) - ys[ms * h:(ms + 1) * h]) ** 2).mean() / 2).backward()
        if ms == 0:
            assert not launched
    g = arena.grad * sync.finish()
    ref = mk()
    ref.load_state_dict(net.state_dict())
    ((ref(x) - y) ** 2).mean().backward()
    flat = torch.cat([torch.nn.functional.pad(p.grad.flatten(), (0, (-p.numel()) % 4)) for p in ref.parameters()])
    out[rank] = float((g - flat).abs().max())
    out[rank + world] = len(launched) == len(sync.buckets)
    dist.destroy_process_group()


def test_gradsync_gloo_world2_grad_accum_sum():
    """2 ranks x 2 micro-batches, summed accumulation, mean-reduced == 1 rank x the whole batch; one exchange per window."""
    import torch.multiprocessing as mp
    world = 2
    out = mp.get_context("spawn").Manager().dict()
    mp.spawn(_accum_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert max(out[0], out[1]) < 1e-6 and out[2] and out[3], dict(out)


def test_grad_accumulation_sync_rule_is_accelerates():
    from frankenstein_amd.utils.train_utils import GradAccumulation
    acc = GradAccumulation(2)
    got = [acc.sync(end_of_loader=i % 5 == 4) for i in range(12)]
    assert got == RT.accum_sync_flags(12, 2, 5) == [False, True, False, True, True, False, True, False, True, True, False, True]
    one = GradAccumulation(1)
    assert all(one.sync() for _ in range(3))

def test_param_arena_views_and_padding():
    from frankenstein_amd.utils.train_utils import ParamArena
    net = torch.nn.Sequential(torch.nn.Linear(5, 3), torch.nn.Linear(3, 2))
    before = [p.detach().clone() for p in net.parameters()]
    a = ParamArena(net)
    assert a.numel % 4 == 0 and all(o % 4 == 0 for o in a.offsets)
    for p, b, o in zip(net.parameters(), before, a.offsets):
        assert torch.equal(p, b) and p.data_ptr() == a.flat.data_ptr() + 4 * o and p.grad.data_ptr() == a.grad.data_ptr() + 4 * o
    net(torch.ones(1, 5)).sum().backward()
    assert float(a.grad.abs().sum()) > 0          # autograd accumulated in place into the arena


def test_reference_import_names_resolve(monkeypatch):
    """the notebooks' `from models import brainformer` style imports, aliased onto this package."""
    import sys
    import frankenstein_amd
    from frankenstein_amd import models, utils
    for k, v in {"models": models, "models.brainformer": models.brainformer, "models.gpt2_model": models.gpt2_model,
                 "utils": utils, "utils.train_utils": utils.train_utils}.items():
        monkeypatch.setitem(sys.modules, k, v)
    from models import brainformer  # noqa: F401
    from models.brainformer import Encoder, CrossBlock, build_complex_rope_cache, Config, MAEConfig  # noqa: F401
    from models.gpt2_model import GPT, GPTConfig  # noqa: F401
    from utils.train_utils import TrainConfig, run_train_model, count_parameters, simple_train_model, train_step  # noqa: F401
    assert frankenstein_amd.compute_dtype() == torch.bfloat16


def test_post_accumulate_hook_fires_for_directly_written_grads():
    """engine.wgrad accumulates weight gradients itself and hands autograd None; GradSync's bucket readiness relies on
    the parameter's post-accumulate-grad hook still running (once, after the last use of the weight)."""
    import torch

    class Direct(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            ctx.save_for_backward(x, w)
            return x @ w.t()

        @staticmethod
        def backward(ctx, dy):
            x, w = ctx.saved_tensors
            w.grad.add_(dy.t() @ x)
            return dy @ w, None

    w = torch.nn.Parameter(torch.randn(3, 4))
    w.grad = torch.zeros_like(w)
    fired = []
    w.register_post_accumulate_grad_hook(lambda p: fired.append(p.grad.clone()))
    x = torch.randn(5, 4, requires_grad=True)
    (Direct.apply(Direct.apply(x, w)[:, :3] @ torch.randn(3, 4), w)).sum().backward()      # weight used twice
    assert len(fired) == 1
    assert torch.equal(fired[0], w.grad) and w.grad.abs().sum() > 0


def test_soundstream_parameter_count_matches_notebook():
    """notebooks_trainer/vq_brain_trainer.ipynb cell 1: SoundStream(C=256, D=64, codebook_size=1024, n_electrodes=512) has 5.61 M
    parameters (the VQ layer of the reference holds buffers only, like the one here)."""
    from frankenstein_amd.models import vq_brain as vq
    m = vq.SoundStream(C=256, D=64, codebook_size=1024, n_electrodes=512)
    assert round(sum(p.numel() for p in m.parameters()) / 1e6, 2) == 5.61
    keys = set(m.state_dict())
    assert {"encoder.layers.0.weight", "encoder.layers.2.layers.0.layers.0.weight", "encoder.layers.2.layers.6.bias",
            "decoder.layers.2.layers.0.weight", "decoder.layers.6.bias"} <= keys
    assert tuple(m.state_dict()["decoder.layers.2.layers.0.weight"].shape) == (256, 256, 4)      # ConvTranspose1d [Cin, Cout, K]


def test_public_surface_of_the_reference_modules_is_present():
    """Every top-level class / function of the reference's models/*.py and utils/*.py exists under the same name."""
    from frankenstein_amd.models import brainformer as bf, gpt2_model as g2, vq_brain as vq, simple_mae as sm
    from frankenstein_amd.utils import train_utils as tu, data_utils as du
    want = {
        bf: "MAEConfig Config build_complex_rope_cache apply_rope build_advanced_causal_mask MLP CausalSelfAttention CausalCrossAttention "
            "RMSNorm Block CrossBlock Encoder MAE BrainFormer default_generation cache_generation",
        g2: "LayerNorm CausalSelfAttention MLP Block GPTConfig GPT",
        vq: "CausalConv1d CausalConvTranspose1d ResidualUnit EncoderBlock DecoderBlock Encoder Decoder SoundStream",
        sm: "build_complex_rope_cache apply_rope build_advanced_causal_mask MLP CausalSelfAttention CausalCrossAttention RMSNorm CrossBlock "
            "SimpleEncoderConfig SimpleMAEConfig SimpleEncoder SimpleMAE",
        tu: "TrainConfig count_parameters init_lr_scheduler prepare_data_loaders run_train_model simple_train_model",
        du: "min_max_per_block_scaling z_score_per_block_scaling process_signal process_text process_file process_all_files process_string "
            "remove_punctuation save_sentences_to_txt load_sentences_from_txt find_long_samples pad_truncate_brain_list get_tokenizer "
            "pad_token_list remove_padding BrainDataset MAX_INPUT_LEN MAX_TOKENS DATE_TO_INDEX",
    }
    for mod, names in want.items():
        missing = [n for n in names.split() if not hasattr(mod, n)]
        assert not missing, (mod.__name__, missing)
    for meth in ("generate", "generate_beam_search", "beam_search", "crop_block_size", "from_pretrained", "configure_optimizers", "estimate_mfu",
                 "get_num_params"):
        assert hasattr(g2.GPT, meth), meth
    for meth in ("get_sub_att_matrix",):
        assert hasattr(bf.MAE, meth)
    assert len(du.DATE_TO_INDEX) == 24 and du.DATE_TO_INDEX["t12.2022.08.25"] == 23
    assert du.process_string("Hello, World! It's fine.") == "hello world it's fine"


def test_wer_known_answers():
    """utils.metrics.wer = total word edit distance / total reference words: the documented example of the HF `wer` metric the
    reference's Whisper notebook loads (0.5), single substitutions / deletions / insertions, and the token-id variant with the
    reference's -100 label padding dropped."""
    from frankenstein_amd.utils.metrics import wer, token_error_rate, edit_distance
    assert wer(["this is the reference", "there is another one"], ["this is the prediction", "there is an other sample"]) == 0.5
    assert wer(["hello world"], ["hello duck"]) == 0.5
    assert wer(["a b c d"], ["a b c d"]) == 0.0
    assert wer(["a b c d"], ["a c d"]) == 0.25 and wer(["a b c d"], ["a b x c d"]) == 0.25
    assert wer(["a b"], ["x y z w"]) == 2.0                      # more errors than reference words is allowed
    assert edit_distance("kitten", "sitting") == 3 and edit_distance([], [1, 2]) == 2
    assert token_error_rate([[1, 2, 3, -100, -100]], [[1, 2, 3]]) == 0.0
    assert token_error_rate([[1, 2, 3, -100]], [[1, 3]]) == pytest.approx(1 / 3)
    with pytest.raises(ValueError):
        wer(["a"], ["a", "b"])
    with pytest.raises(ValueError):
        wer([""], ["a"])


def test_gpt_init_weights_statistics():
    """GPT._init_weights + the c_proj rescale (models/gpt2_model.py:141-145,170-176 of the reference): Linear / Embedding weights
    N(0, 0.02), residual projections N(0, 0.02 / sqrt(2 L)), zero biases, lm_head tied to wte."""
    import math
    from frankenstein_amd.models import gpt2_model as g2
    torch.manual_seed(0)
    L = 3
    g = g2.GPT(g2.GPTConfig(block_size=64, vocab_size=1024, n_layer=L, n_head=4, n_embd=256, dropout=0.0, bias=True))
    assert g.lm_head.weight is g.transformer.wte.weight
    for name, p in g.named_parameters():
        if name.endswith("bias"):
            assert float(p.abs().max()) == 0.0, name
        elif "ln_" in name:
            assert float((p - 1).abs().max()) == 0.0, name
        else:
            want = 0.02 / math.sqrt(2 * L) if name.endswith("c_proj.weight") else 0.02
            n = p.numel()
            # std of n normal samples has relative standard error 1/sqrt(2n): 5 sigma bounds
            assert abs(float(p.std()) / want - 1.0) < 5.0 / math.sqrt(2 * n), (name, float(p.std()), want)
            assert abs(float(p.mean())) < 5.0 * want / math.sqrt(n), (name, float(p.mean()))


def test_bench_self_launch_dry_run_world2():
    """`python bench.py --gpus 2` without a launcher starts torch.distributed.run itself (child processes; gloo rendezvous on
    127.0.0.1), rank 0 prints the one JSON line, and without GPUs the real run refuses with a clear message instead of an assert."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["dry_run"] and out["n_gpus"] == 2 and out["world"] == 2 and out["steps"] == 3 and out["max_rank_time_s"] == 0.002
    # the data-parallel self-checks of the real run (bench.dp_report), exercised over gloo: replicas that started from different
    # seeds end with identical parameter checksums, the exchange was measured, its bucket layout is reported, RCCL's CU share capped
    dp = out["dp"]
    assert dp["param_checksum_equal"] is True and dp["world"] == 2 and dp["n_buckets"] >= 2 and len(dp["bucket_mb"]) == dp["n_buckets"]
    assert dp["exposed_comm_ms"] >= 0.0 and out["nccl_max_nchannels"] == "8"
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=120)
        assert r.returncode == 2 and "GPU(s) visible" in r.stderr and "AssertionError" not in r.stderr


STREAMS = {"dkdv": "attn_dkdv_asm.inc", "dq": "attn_dq_asm.inc", "fwd": "attn_fwd_asm.inc", "dq16": "attn_dq16_asm.inc", "dkdvw": "attn_dkdvw_asm.inc",
           "mlpb": "mlp_bwd_asm.inc"}


@pytest.mark.parametrize("gen", sorted(STREAMS))
def test_generated_streams_are_current(tmp_path, gen):
    """frankenstein_amd/csrc/attn_*_asm.inc and mlp_bwd_asm.inc are build inputs that are committed: the generators (tools/gen) reproduce them byte for byte."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    out = tmp_path / "gen.inc"
    env = {k: v for k, v in os.environ.items() if not k.startswith("FK_GEN_")}
    subprocess.run([sys.executable, str(root / "tools" / "gen" / f"gen_{gen}_asm.py"), str(out)], check=True, env=env, capture_output=True)
    assert out.read_bytes() == (root / "frankenstein_amd" / "csrc" / STREAMS[gen]).read_bytes()


def _verify_stream():
    import importlib.util
    from pathlib import Path
    spec = importlib.util.spec_from_file_location("verify_stream", Path(__file__).resolve().parents[1] / "tools" / "gen" / "verify_stream.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("gen", sorted(STREAMS))
def test_generated_streams_keep_their_hazard_rules(gen):
    """every asm block of the committed streams replayed against the rules the generators promise (counted LDS waits, VALU -> consumer
    distance, MFMA result -> VALU distance, fragment overwrite behind its MFMA, M0 -> LDS-DMA distance): tools/gen/verify_stream.py"""
    from pathlib import Path
    vs = _verify_stream()
    text = (Path(__file__).resolve().parents[1] / "frankenstein_amd" / "csrc" / STREAMS[gen]).read_text()
    blocks = list(vs.blocks(text))
    assert len(blocks) >= (1 if gen == "mlpb" else 3)
    if gen == "mlpb":
        assert sum(i.startswith("v_mfma") for i in blocks[0][1]) == 48
    errs = [e for name, ins in blocks for e in vs.check_block(name, ins)]
    assert not errs, errs[:5]


def test_stream_verifier_sees_violations():
    """the checker itself: each rule broken once in a small hand-written block"""
    vs = _verify_stream()
    ok = ["ds_read_b128 v[100:103], %[a] offset:0", "s_waitcnt lgkmcnt(0)",
          "v_mfma_f32_32x32x16_bf16 v[104:119], v[100:103], %[k], 0", "v_mfma_f32_32x32x16_bf16 v[120:135], v[100:103], %[k], 0",
          "v_mfma_f32_32x32x16_bf16 v[136:151], v[100:103], %[k], 0", "v_exp_f32_e32 v104, v104", "s_nop 0", "v_mul_f32_e32 v105, v104, v104",
          "s_add_u32 m0, %[l], 0", "s_nop 0", "global_load_lds_dwordx4 %[vo], %[kb]"]
    assert vs.check_block("ok", ok) == []
    no_wait = [ok[0]] + ok[2:]
    assert any("may still be outstanding" in e for e in vs.check_block("b", no_wait))
    early_valu = ok[:3] + ["v_exp_f32_e32 v104, v104"]
    assert any("further MFMA" in e for e in vs.check_block("b", early_valu))
    back_to_back = ok[:6] + ["v_mul_f32_e32 v105, v104, v104"]
    assert any("right in front" in e for e in vs.check_block("b", back_to_back))
    overwrite = ok[:3] + ["ds_read_b128 v[100:103], %[a] offset:64", "s_waitcnt lgkmcnt(0)"]
    assert any("operand of the MFMA issued last" in e for e in vs.check_block("b", overwrite))
    m0_late = ok[:8] + ["s_add_u32 m0, %[l], 0", "global_load_lds_dwordx4 %[vo], %[kb]"]
    assert any("M0 written" in e for e in vs.check_block("b", m0_late))
    unwaited = ok + ["ds_read_b128 v[100:103], %[a] offset:0"]
    assert any("not waited for" in e for e in vs.check_block("b", unwaited))
    # the fused MLP backward's stream also writes LDS and stores rows: an LDS write counts in lgkmcnt, neither may read a pending register
    tile = ["ds_read_b128 v[100:103], %[a] offset:0", "v_cvt_pk_bf16_f32 v110, v111, v112", "s_nop 0", "ds_write_b128 v120, v[108:111]", "s_waitcnt lgkmcnt(0)",
            "global_store_dwordx4 v121, v[100:103], %[g] nt"]
    assert vs.check_block("ok2", tile) == []
    assert any("may still be outstanding" in e for e in vs.check_block("b", tile[:4] + [tile[5], tile[4]]))
    assert any("right in front" in e for e in vs.check_block("b", tile[:2] + tile[3:]))
    assert any("not waited for" in e for e in vs.check_block("b", tile[:4]))


def test_dense_mask_tables_follow_the_reference_slice_and_broadcast():
    """Mask.from_dense on the host: the table is always [Bm, Hm, t_q, t_k] (size-1 query / key axes expanded the way SDPA broadcasts
    them, longer masks sliced from the end like mask[..., -t_q:, -t_k:], models/brainformer.py:160-162), and the strides handed to the
    kernel (c = batch stride, q_off = head stride) describe exactly that table — the key-padding form [B, 1, 1, N_k] used to keep a
    256-byte table under a batch stride of 4096."""
    from frankenstein_amd import kernels as K
    m = K.Mask.from_dense(torch.ones(4, 1, 1, 64, dtype=torch.bool), 64, 64)
    assert tuple(m.limits.shape) == (4, 1, 64, 64) and m.limits.is_contiguous() and (m.c, m.q_off) == (4096, 0)
    pad = torch.rand(3, 1, 1, 10, generator=torch.Generator().manual_seed(0)) < 0.5
    m = K.Mask.from_dense(pad, 7, 10)
    assert torch.equal(m.limits.bool(), pad.expand(3, 1, 7, 10))
    big = torch.rand(2, 3, 12, 15, generator=torch.Generator().manual_seed(1)) < 0.5
    m = K.Mask.from_dense(big, 5, 9)
    assert torch.equal(m.limits.bool(), big[..., -5:, -9:]) and (m.c, m.q_off) == (3 * 45, 45)
    m = K.Mask.from_dense(torch.ones(6, 1, dtype=torch.bool), 6, 8)            # [N_q, 1]: one column for every key
    assert tuple(m.limits.shape) == (1, 1, 6, 8) and (m.c, m.q_off) == (0, 0)
    with pytest.raises(ValueError, match="does not broadcast"):
        K.Mask.from_dense(torch.ones(2, 1, 3, 8, dtype=torch.bool), 6, 8)
    with pytest.raises(ValueError, match="does not broadcast"):
        K._check_dense(K.Mask.from_dense(torch.ones(2, 1, 6, 8, dtype=torch.bool), 6, 8), 4, 2, 6, 8)
    K._check_dense(K.Mask.from_dense(torch.ones(1, 2, 6, 8, dtype=torch.bool), 6, 8), 4, 2, 6, 8)


def test_gpt_from_pretrained_maps_a_hugging_face_state_dict(monkeypatch):
    """GPT.from_pretrained (models/gpt2_model.py:229-284) without the network: transformers' own GPT2LMHeadModel, randomly initialised from
    its default (gpt2, 124 M) config, stands in for the downloaded checkpoint — so the state dict has the real HF layout (Conv1D weights
    stored [in, out], tied lm_head) — plus the two causal-mask buffers old checkpoints carry.  Checked functionally: a Conv1D computes
    x @ W + b, the loaded nn.Linear-style weight must give the same product; embeddings / norms / biases are copied as they are."""
    transformers = pytest.importorskip("transformers")
    from frankenstein_amd.models import gpt2_model as g2
    torch.manual_seed(0)
    hf_model = transformers.GPT2LMHeadModel(transformers.GPT2Config())
    hf_sd = dict(hf_model.state_dict())
    hf_sd["transformer.h.0.attn.bias"] = torch.ones(1, 1, 1024, 1024, dtype=torch.bool).tril()        # skipped by the loader
    hf_sd["transformer.h.0.attn.masked_bias"] = torch.tensor(-1e4)

    class Fake:
        def state_dict(self):
            return hf_sd

    asked = []
    monkeypatch.setattr(transformers.GPT2LMHeadModel, "from_pretrained", classmethod(lambda cls, name: (asked.append(name), Fake())[1]))
    m = g2.GPT.from_pretrained("gpt2", dict(dropout=0.0))
    assert asked == ["gpt2"]
    c = m.config
    assert (c.n_layer, c.n_head, c.n_embd, c.block_size, c.vocab_size, c.bias, c.dropout) == (12, 12, 768, 1024, 50257, True, 0.0)
    sd = m.state_dict()
    assert m.lm_head.weight is m.transformer.wte.weight and torch.equal(sd["lm_head.weight"], hf_sd["transformer.wte.weight"])
    x = torch.randn(3, 768, generator=torch.Generator().manual_seed(1))
    for layer in (0, 11):
        for name, xin in (("attn.c_attn", x), ("attn.c_proj", x), ("mlp.c_fc", x), ("mlp.c_proj", torch.randn(3, 3072, generator=torch.Generator().manual_seed(2)))):
            k = f"transformer.h.{layer}.{name}"
            want = xin @ hf_sd[k + ".weight"] + hf_sd[k + ".bias"]                       # transformers' Conv1D
            got = torch.nn.functional.linear(xin, sd[k + ".weight"], sd[k + ".bias"])
            torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
        for name in ("ln_1.weight", "ln_1.bias", "ln_2.weight", "ln_2.bias"):
            assert torch.equal(sd[f"transformer.h.{layer}.{name}"], hf_sd[f"transformer.h.{layer}.{name}"])
    for k in ("transformer.wte.weight", "transformer.wpe.weight", "transformer.ln_f.weight", "transformer.ln_f.bias"):
        assert torch.equal(sd[k], hf_sd[k])
    with pytest.raises(AssertionError):
        g2.GPT.from_pretrained("gpt2", dict(bias=False))              # only dropout may be overridden (models/gpt2_model.py:233)
