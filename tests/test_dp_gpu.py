"""Data-parallel rehearsal on ONE GPU: two ranks share cuda:0 and exchange gradients through gloo (RCCL refuses two
ranks on one device).  Exercises the real training step — autograd hooks launching bucket all-reduces during the
backward, the fused AdamW with the 1/world scale — and checks DP equivalence: 2 ranks x half batch
== 1 rank x full batch, and identical parameters on both ranks."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build():
    import frankenstein_amd as fa
    from frankenstein_amd import synth
    from frankenstein_amd.models import brainformer as bf
    fa.set_compute_dtype("fp32")
    enc = bf.MAEConfig(window_size=32, n_electrodes=16, patch_size=4, dim=64, n_layers=2, head_dim=16, hidden_dim=128,
                       n_heads=4, n_kv_heads=4)
    cfg = bf.Config(encoder=enc, n_output_tokens=8, output_dim=12, dim=64, n_layers=1, head_dim=16, hidden_dim=96,
                    n_heads=4, n_kv_heads=4)
    m = bf.BrainFormer(cfg)
    sd = m.state_dict()
    st = synth.make_state({k: tuple(v.shape) for k, v in sd.items()})
    m.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()}, strict=False)
    return m.cuda()


def _batch(B):
    from frankenstein_amd import synth
    return torch.from_numpy(synth.make_inputs(B, 32, 16)).cuda(), torch.from_numpy(synth.make_motion_targets(B, 8, 12)).cuda()


def _worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from frankenstein_amd.utils import train_utils as tu
    m = _build()
    if rank != 0:                           # replicas that did NOT start identical: the optimizer's rank-0 broadcast has to fix it
        with torch.no_grad():
            for p_ in m.parameters():
                p_.add_(0.01 * rank)
    cfg = tu.TrainConfig(mixed_precision=False, use_scheduler=False, learning_rate=1e-3)
    opt = tu.FusedAdamW(m, lr=1e-3, weight_decay=cfg.weight_decay, grad_clip=cfg.grad_clip, bucket_bytes=64 << 10)
    assert opt.sync.world == world and len(opt.sync.buckets) > 1
    x, y = _batch(4)
    for step in range(2):
        xb, yb, _ = tu.shard_batch((x, y, None), rank, world)
        tu.train_step(m, (xb, yb, None), opt, step, cfg)
    torch.cuda.synchronize()
    out[rank] = opt.arena.flat.detach().cpu().numpy()
    dist.destroy_process_group()


def test_two_ranks_equal_one_rank_full_batch():
    import torch.multiprocessing as mp
    world = 2
    out = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert len(out) == world
    np.testing.assert_array_equal(out[0], out[1])            # replicas stay bit-identical
    # single process, full batch (L1 loss is a mean over equal-sized shards -> mean of shard gradients == full gradient)
    from frankenstein_amd.utils import train_utils as tu
    m = _build()
    cfg = tu.TrainConfig(mixed_precision=False, use_scheduler=False, learning_rate=1e-3)
    opt = tu.FusedAdamW(m, lr=1e-3, weight_decay=cfg.weight_decay, grad_clip=cfg.grad_clip)
    x, y = _batch(4)
    for step in range(2):
        tu.train_step(m, (x, y, None), opt, step, cfg)
    ref = opt.arena.flat.detach().cpu().numpy()
    # Adam turns rounding-level gradient differences of (near-)zero-gradient entries into +-lr updates: compare robustly
    diff = np.abs(out[0] - ref)
    assert np.quantile(diff, 0.99) < 2e-5 and diff.max() < 5e-3, (np.quantile(diff, 0.99), diff.max())
    import frankenstein_amd as fa
    fa.set_compute_dtype("bf16")


def _rccl_worker(rank, port, out):
    """one rank, backend "nccl" (= RCCL): communicator creation, the rank-0 broadcast of the parameter arena and the bucketed
    all-reduces issued from the autograd hooks all run through RCCL (a one-rank group, so each collective is the identity)."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from frankenstein_amd.utils import train_utils as tu
    m = _build()
    cfg = tu.TrainConfig(mixed_precision=False, use_scheduler=False, learning_rate=1e-3)
    opt = tu.FusedAdamW(m, lr=1e-3, weight_decay=cfg.weight_decay, grad_clip=cfg.grad_clip, bucket_bytes=64 << 10, sync_always=True)
    assert opt.sync.active and opt.sync.world == 1 and len(opt.sync.buckets) > 1
    launched = []
    orig = opt.sync._launch
    opt.sync._launch = lambda b: (launched.append(b), orig(b))[1]
    opt.sync.measure = True
    x, y = _batch(4)
    for step in range(2):
        tu.train_step(m, (x, y, None), opt, step, cfg)
    torch.cuda.synchronize()
    import bench                                   # the self-validation block of a multi-GPU bench line, through RCCL
    out["dp"] = bench.dp_report(opt.sync, opt.arena, torch.device("cuda", 0), 2)
    out["flat"] = opt.arena.flat.detach().cpu().numpy()
    out["launched"] = len(launched)
    out["nbuckets"] = len(opt.sync.buckets)
    out["rccl"] = ".".join(map(str, torch.cuda.nccl.version()))
    dist.destroy_process_group()


def test_rccl_one_rank_group_runs_the_collective_path():
    import torch.multiprocessing as mp
    out = mp.get_context("spawn").Manager().dict()
    mp.spawn(_rccl_worker, args=(_free_port(), out), nprocs=1, join=True)
    assert out["launched"] == 2 * out["nbuckets"] and out["rccl"]
    dp = out["dp"]
    assert dp["param_checksum_equal"] is True and dp["n_buckets"] == out["nbuckets"] and 0.0 <= dp["exposed_comm_ms"] < 50.0
    from frankenstein_amd.utils import train_utils as tu
    m = _build()
    cfg = tu.TrainConfig(mixed_precision=False, use_scheduler=False, learning_rate=1e-3)
    opt = tu.FusedAdamW(m, lr=1e-3, weight_decay=cfg.weight_decay, grad_clip=cfg.grad_clip)
    x, y = _batch(4)
    for step in range(2):
        tu.train_step(m, (x, y, None), opt, step, cfg)
    np.testing.assert_array_equal(out["flat"], opt.arena.flat.detach().cpu().numpy())     # identity collectives: same bits
    import frankenstein_amd as fa
    fa.set_compute_dtype("bf16")
