#!/usr/bin/env python3
"""bench.py — headline benchmark: brainformer-small training throughput on MI355X.

Workload (BASELINE.json configs[1]): BrainFormer (6-layer encoder d=384, 6 heads x 64,
SwiGLU 1536; window 600, patch 25 -> N=6144 tokens; 2-block perceiver with 32 queries; L1 head 128), per-GPU
batch 32 x T=600 x 256 electrodes, bf16 compute / fp32 masters.  One step = forward + backward + (DP gradient
all-reduce) + clip_grad_value_ + AdamW over one synthetic batch already resident in HBM.

    python bench.py --gpus N --steps K --warmup W

N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (RANK / LOCAL_RANK /
WORLD_SIZE in the environment), or plain `python bench.py --gpus N`, in which case this process starts that launcher itself as a
CHILD process before anything touches the GPU, relays rank 0's JSON line and exits with the child's code.  One rank per GPU, RCCL.

Prints ONE JSON line on rank 0.  `value` = total frames/s of the whole job (frames = B*T input time-steps).
`roofline` is for the dominant kernel family (measured live with HIP events on the launch stream);
`cpu_baseline` times the CPU oracle (oracle/, a port of the reference's PyTorch path) on this box's host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

MFMA_PEAK_BF16 = 2.5e15          # dense bf16 MFMA, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
FLOP_PER_FRAME = 1.8173e9        # cfg2 fwd+bwd algorithmic FLOPs per input frame (visible attention only)
FLOP_PER_FRAME_CE = 1.8211e9     # ... with the CE head (25 tokens, V = 50257): + 0.75 GF/sample forward


def cfg2_model(dtype="bf16", head="l1"):
    """head 'l1': the file-class BrainFormer (32 output tokens x 128, L1 loss) — the headline workload.  head 'ce': cfg2's CE variant
    (the notebook CE BrainFormer, 25 output tokens, V = 50257, cross entropy with -100 padding) with the fused head loss
    (train_utils.enable_fused_head_loss: the reference's training loop only uses the loss, utils/train_utils.py:138-139)."""
    import frankenstein_amd as fa
    from frankenstein_amd.models import brainformer as bf
    fa.set_compute_dtype(dtype)
    enc = bf.MAEConfig(window_size=600, n_electrodes=256, patch_size=25, dim=384, n_layers=6, head_dim=64,
                       hidden_dim=1536, n_heads=6, n_kv_heads=6)
    if head == "ce":
        from frankenstein_amd.models.notebook_models import BrainFormerCE
        from frankenstein_amd.utils import train_utils as tu
        cfg = bf.Config(encoder=enc, n_output_tokens=25, output_dim=50257, dim=384, n_layers=2, head_dim=64,
                        hidden_dim=768, n_heads=6, n_kv_heads=6)
        m = BrainFormerCE(cfg)
        tu.enable_fused_head_loss(m)
        return m, cfg
    cfg = bf.Config(encoder=enc, n_output_tokens=32, output_dim=128, dim=384, n_layers=2, head_dim=64,
                    hidden_dim=768, n_heads=6, n_kv_heads=6)
    return bf.BrainFormer(cfg), cfg


def init_weights(model, seed=42):
    """random-init weights of the architecture (no checkpoints exist offline): N(0, 1/sqrt(fan_in)) etc."""
    from frankenstein_amd import synth
    sd = model.state_dict()
    shapes = {k: tuple(v.shape) for k, v in sd.items() if v is not None}
    st = synth.make_state(shapes, seed)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()}, strict=False)


def attn_flops(B, H, Nq, Nk, D, C, bwd):
    vis = 0
    for q in range(0, Nq, C):                      # block-causal: queries of block b see (b+1)*C keys
        vis += min(C, Nq - q) * min(Nk, (q // C + 1) * C)
    return (10 if bwd else 4) * B * H * vis * D


def host_cores() -> int:
    """CPU share of this process: cgroup quota if any (the GPU box gives 16 of 256 logical CPUs), else affinity."""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(q) // int(per)))
    except Exception:
        pass
    return n


def _mem_available_gb() -> float:
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) / 2 ** 20
    except Exception:
        pass
    return 0.0


def cpu_baseline(steps=2):
    """Oracle (port of the reference CPU path) fwd+bwd+clip+AdamW at cfg2, fp32, all host threads: B=1 (1 warm-up + best of `steps`)
    and, where the host has the memory for the dense-mask math path (SURVEY 8d asks for B=1 and B=4; B=4 keeps ~60 GB of fp32 score
    tensors alive for the backward), one step at B=4.  `value` is the better per-frame rate of the two."""
    torch.set_num_threads(host_cores())
    from oracle import ref_models as R
    from oracle import ref_train as RT
    from tests import cases as C

    def run(B, n_steps, warm):
        cfg, x, tgt = C.cfg2(B)
        sd = C.state(R.brainformer_shapes(cfg, "to_motion"))
        state, cur = {}, sd
        loss_fn = lambda s: R.brainformer_l1(s, x, tgt, cfg)[0]
        times = []
        for i in range(n_steps + warm):
            t0 = time.perf_counter()
            _, _, cur = RT.train_step(loss_fn, cur, state, i + 1, 1e-3)
            times.append(time.perf_counter() - t0)
        return min(times[warm:])

    best1 = run(1, steps, 1)
    samples = [{"batch": 1, "s_per_step": round(best1, 2), "frames_per_s": round(600 / best1, 2)}]
    if _mem_available_gb() > 150.0:
        t4 = run(4, 1, 0)
        samples.append({"batch": 4, "s_per_step": round(t4, 2), "frames_per_s": round(2400 / t4, 2)})
    best = max(samples, key=lambda d: d["frames_per_s"])
    return {"value": best["frames_per_s"], "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"cfg2 brainformer-small, fp32, fwd+bwd+clip+AdamW on the host cores: B=1 (600 frames/step), 1 warm-up + best of {steps} steps"
                      + (", and one step at B=4 (2400 frames/step, threads already warm)" if len(samples) > 1 else "")
                      + f"; value = the better per-frame rate (B={best['batch']}, {best['s_per_step']} s/step)",
            "samples": samples}


def _latest_profile(suffix):
    """newest profiles/rNN_<suffix> (the committed rocprofv3 counter passes are named per round); None if there is none"""
    found = sorted((ROOT / "profiles").glob(f"r[0-9][0-9]_{suffix}"))
    return found[-1] if found else None


def mfma_util_pmc():
    """MFMA-pipe busy fraction of the whole step and of the roofline kernels from the newest COMMITTED rocprofv3 counter pass
    (profiles/rNN_pmc_mfma_util.json: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs)) — builder-run evidence, not measured
    in this run (counters need rocprofv3 around the process); `source` says which file.  None if absent."""
    try:
        path = _latest_profile("pmc_mfma_util.json")
        pm = json.load(open(path))
        ks = pm["kernels"]
        pick = lambda frag: next((round(v["mfma_util_pct"], 1) for k, v in ks.items() if frag in k), None)
        return {"whole_step_pct": round(pm["whole_trace"]["mfma_util_pct"], 1), "attn_bwd_dkdv_pct": pick("attn_bwd_dkdv"),
                "attn_bwd_dq_pct": pick("attn_bwd_dq"), "executed_mfma_tflops": round(pm["whole_trace"]["mfma_tflops"], 1),
                "source": f"profiles/{path.name}, builder-run rocprofv3 --pmc pass over one benchmark step (not measured in this run)"}
    except Exception:
        return None


def visible_gpus():
    """GPU count WITHOUT loading the HIP runtime (torch.cuda.device_count() may fall back to hipGetDeviceCount = hipInit in this
    process): FK_VISIBLE_GPUS if set, else the KFD topology nodes that have SIMDs; None when neither can be read (the ranks then
    report a missing device themselves)."""
    if os.environ.get("FK_VISIBLE_GPUS"):
        return int(os.environ["FK_VISIBLE_GPUS"])
    nodes = Path("/sys/class/kfd/kfd/topology/nodes")
    if not nodes.is_dir():
        return 0                                         # no KFD driver: no AMD GPU on this host
    try:
        n = 0
        for node in nodes.iterdir():
            props = dict(l.split() for l in (node / "properties").read_text().splitlines() if len(l.split()) == 2)
            n += int(props.get("simd_count", "0")) > 0
        return n
    except Exception:
        return None


def dp_env(env=None):
    """Environment of a data-parallel rank.  NCCL_MAX_NCHANNELS bounds the workgroups (= CUs) an RCCL collective occupies: the
    all-reduces of GradSync run on the communication stream BESIDE persistent one-block-per-CU GEMM grids, and a second occupant that
    takes CUs at random makes those grids run a second partial wave (DESIGN.md 5.1 "two large-tile grids on two streams": 65 ms steps
    became 80-160 ms).  81 MB of gradients per 50 ms step need ~3 GB/s per link, so a few channels are plenty."""
    env = os.environ if env is None else env
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("NCCL_MAX_NCHANNELS", "8")
    return env


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: run `torch.distributed.run` as a child process (fresh ranks; this process
    never loads the HIP runtime) and relay its output."""
    import socket
    import subprocess
    if not args.dry_run:
        n = visible_gpus()
        if n is not None and n < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but only {n} GPU(s) visible on this host; "
                  f"run with --gpus <= {n}, or launch one rank per GPU on a node that has {args.gpus}", file=sys.stderr)
            return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    proc = subprocess.run(cmd, env=dp_env(dict(os.environ)))
    return proc.returncode


def dry_run(args, rank, world):
    """Launcher / rendezvous / timing-reduction plumbing without a GPU (gloo): what tests/test_host_cpu.py drives at world size 2."""
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
        dist.barrier()
    t = torch.tensor([0.001 * (rank + 1)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    # the data-parallel self-checks of the real run, on a toy arena: bucketed exchange, replica checksum, exposed-wait bookkeeping
    from frankenstein_amd.utils import train_utils as tu
    torch.manual_seed(rank)                              # replicas start DIFFERENT: the arena broadcast has to align them
    net = torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.Tanh(), torch.nn.Linear(53, 3))
    arena = tu.ParamArena(net)
    gs = tu.GradSync(arena, bucket_bytes=4096, measure=True)
    x = torch.randn(8, 37, generator=torch.Generator().manual_seed(100 + rank))
    for _ in range(max(1, args.steps)):
        net(x).pow(2).mean().backward()
        scale = gs.finish()
        with torch.no_grad():
            arena.flat.add_(arena.grad, alpha=-0.1 * scale)
            arena.grad.zero_()
    dp = dp_report(gs, arena, torch.device("cpu"), args.steps)
    if rank == 0:
        print(json.dumps({"metric": "neural frames/sec (train fwd+bwd+AdamW, whole job)", "dry_run": True, "n_gpus": world,
                          "world": world, "steps": args.steps, "warmup": args.warmup, "max_rank_time_s": float(t), "dp": dp,
                          "nccl_max_nchannels": os.environ.get("NCCL_MAX_NCHANNELS")}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def dp_report(gs, arena, dev, steps):
    """Self-validation of a data-parallel run, identical on every rank after the collectives inside:
    param_checksum_equal — MIN and MAX over ranks of a checksum of the parameter arena agree (replicas did not drift apart);
    exposed_comm_ms — per step, time the compute stream spent waiting for the gradient exchange after the last backward kernel
    (HIP events around GradSync.finish; mean over the timed steps, MAX over ranks);  n_buckets / bucket_mb — the exchange's shape."""
    import torch.distributed as dist
    flat = arena.flat.detach()
    cs = torch.stack([flat.double().sum(), flat.double().abs().sum(), (flat.double() * torch.arange(1, flat.numel() + 1, device=flat.device,
                                                                     dtype=torch.float64).remainder(8191)).sum()]).to(dev)
    lo, hi = cs.clone(), cs.clone()
    exp = torch.tensor([gs.exposed_ms_mean(last=steps)], dtype=torch.float64, device=dev)
    if gs.active:
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(exp, op=dist.ReduceOp.MAX)
    return {"param_checksum_equal": bool(torch.equal(lo, hi)), "param_checksum": [float(v) for v in cs.cpu()],
            "exposed_comm_ms": round(float(exp), 4), "n_buckets": len(gs.buckets),
            "bucket_mb": [round((e - s0) * 4 / 2 ** 20, 2) for s0, e, _ in gs.buckets], "world": gs.world}


def parity_live(dev):
    """bf16-vs-reference numbers at the benchmarked shape, MEASURED IN THIS RUN (rank 0, after the timed region): cfg2 at B = 1 in the
    benchmark's precision, forward + backward, against the reference's own fp32 CPU run stored in tests/golden/cfg2_b1.npz /
    cfg2_b1_samples.npz (loss, predictions, six encoder rows, 256 evenly spaced samples of every gradient; produced by
    tests/golden/make_golden.py from /root/reference).  Same quantities and bounds as tests/test_models_gpu.py::
    test_cfg2_b1_bf16_vs_reference.  None when the fixtures are absent."""
    import numpy as np
    from frankenstein_amd import synth
    try:
        z = np.load(ROOT / "tests" / "golden" / "cfg2_b1.npz", allow_pickle=False)
        zs = np.load(ROOT / "tests" / "golden" / "cfg2_b1_samples.npz", allow_pickle=False)
    except Exception:
        return None
    m, _ = cfg2_model("bf16", "l1")
    init_weights(m)
    m.to(dev)
    x = torch.from_numpy(synth.make_inputs(1, 600, 256)).to(dev)
    tgt = torch.from_numpy(synth.make_motion_targets(1, 32, 128)).to(dev)
    loss, pred = m(x, tgt)
    with torch.no_grad():
        ctx = m.encoder(x)
    loss.backward()
    rel = abs(float(loss) - float(z["loss"])) / float(z["loss"])
    perr = float(np.abs(pred.float().detach().cpu().numpy() - z["pred"]).max())
    eerr = float(np.abs(ctx[0, [0, 1, 255, 256, 3071, 6143]].float().cpu().numpy() - z["enc_rows"]).max())
    want = {str(n): r for n, r in zip(zs["grad_names"], zs["grad_samples"])}
    cos, seen = {}, set()
    for k, p_ in m.named_parameters():
        if id(p_) in seen or k not in want:
            continue
        seen.add(id(p_))
        a = (p_.grad if p_.grad is not None else torch.zeros_like(p_)).detach().double().flatten().cpu().numpy()
        idx = np.arange(a.size) if a.size <= 256 else (np.arange(256, dtype=np.int64) * a.size) // 256
        r = np.zeros(256)
        r[: idx.size] = a[idx]
        den = float(np.linalg.norm(r) * np.linalg.norm(want[k]))
        cos[k] = float(np.dot(r, want[k]) / den) if den > 0 else 1.0
    worst = min(cos, key=cos.get)
    return {"source": "measured in this run against tests/golden/cfg2_b1*.npz (the reference's fp32 CPU run)",
            "shape": "cfg2 at B=1 (6L d=384 6x64 heads, N=6144), bf16 vs reference fp32 CPU, forward + backward",
            "loss_rel_err": rel, "pred_max_abs_err": perr, "pred_max_abs": float(np.abs(z["pred"]).max()), "enc_rows_max_abs_err": eerr,
            "enc_rows_max_abs": float(np.abs(z["enc_rows"]).max()), "grad_cosine_min": cos[worst], "grad_cosine_min_param": worst,
            "grad_cosine_median": float(np.median(list(cos.values()))), "n_params": len(cos),
            "bounds": {"loss_rel_err": 1e-2, "pred_max_abs_err": 5e-2, "grad_cosine_min": 0.99},
            "within_bounds": bool(rel < 1e-2 and perr < 5e-2 and cos[worst] >= 0.99)}


def other_configs(dev, steps=10, warmup=3):
    """The other single-GPU BASELINE.json configurations, measured live on rank 0 after the headline run (about two seconds): bf16
    training step (fwd + bwd + clip + AdamW) on synthetic inputs, ms/step, frames/s and the fraction of the dense bf16 MFMA peak from
    SURVEY 8d's algorithmic work per frame (cfg5: 57.7 MFLOP/frame; cfg1: 16.0 GFLOP/sample = 80 MFLOP/frame).  These are launch- and
    traffic-bound problems (a few hundred kernels of 5-30 us per step): the fraction says how far, not a target."""
    import frankenstein_amd as fa
    from frankenstein_amd.utils import train_utils as tu
    fa.set_compute_dtype("bf16")
    tc = tu.TrainConfig(mixed_precision=True, use_scheduler=False, learning_rate=1e-4)
    g = torch.Generator(device=dev).manual_seed(4321)

    def timeit(fn):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    out = {}
    from frankenstein_amd.models import simple_mae as sm
    ecfg = sm.SimpleEncoderConfig(block_size=600, patch_size=256, n_layers=6, dim=384, hidden_dim=1536, head_dim=64, n_heads=6, n_kv_heads=6)
    dcfg = sm.SimpleMAEConfig(n_layers=2, dim=384, hidden_dim=1536, head_dim=64, n_heads=6, n_kv_heads=6)
    m = sm.SimpleMAE(ecfg, dcfg).to(dev)
    opt = tu.FusedAdamW(m, lr=1e-4, weight_decay=1e-5, grad_clip=1.0)
    st = [0]
    for B in (32, 256):
        x = torch.randn(B, 600, 256, device=dev, generator=g)

        def step():
            tu.train_step(m, (x, None, None), opt, st[0], tc)
            st[0] += 1
        dt = timeit(step)
        ent = {"workload": f"SimpleMAE pre-training (BASELINE configs[4], SURVEY cfg5): 6+2 layers d=384, 600 frame tokens, 75 % masked, per-GPU batch {B}",
               "ms_per_step": round(dt * 1e3, 3), "frames_per_s": round(B * 600 / dt, 1),
               "mfma_frac": round(B * 600 / dt * 57.7e6 / MFMA_PEAK_BF16, 4)}
        if B == 32:          # ~300 launches of 5-30 us: the eager step is bound by the host launch path; the same step replayed from one hipGraph
            try:
                gstep = tu.GraphedTrainStep(m, (x, None, None), opt, tc)
                dtg = timeit(lambda: gstep((x, None, None), 0))
                ent["graphed_ms_per_step"] = round(dtg * 1e3, 3)
                ent["graphed_mfma_frac"] = round(B * 600 / dtg * 57.7e6 / MFMA_PEAK_BF16, 4)
                del gstep
            except Exception:
                ent["graphed_ms_per_step"] = None
                torch.cuda.synchronize()
        out[f"cfg5_simple_mae_b{B}"] = ent
    del m, opt
    from frankenstein_amd.models import brainformer as bf
    from frankenstein_amd.models.gpt2_model import GPT, GPTConfig
    from frankenstein_amd.models.notebook_models import BrainEncoder, Franky
    enc = bf.MAEConfig(window_size=200, n_electrodes=256, patch_size=25, dim=128, n_layers=2, head_dim=32, hidden_dim=512, n_heads=4, n_kv_heads=4)
    cfg = bf.Config(encoder=enc, n_output_tokens=32, output_dim=128, dim=128, n_layers=2, head_dim=32, hidden_dim=256, n_heads=4, n_kv_heads=4)
    fr = Franky(BrainEncoder(cfg), GPT(GPTConfig(block_size=1024, vocab_size=50257, n_layer=2, n_head=4, n_embd=128, dropout=0.0, bias=True))).to(dev)
    opt = tu.FusedAdamW(fr, lr=1e-4, weight_decay=1e-5, grad_clip=1.0)
    x = torch.randn(4, 200, 256, device=dev, generator=g)
    tok = torch.randint(0, 50257, (4, 25), device=dev, generator=g)
    tok[:, -3:] = -100

    def step1():
        tu.train_step(fr, (x, tok, None), opt, st[0], tc)
        st[0] += 1
    dt = timeit(step1)
    out["cfg1_franky_b4"] = {"workload": "Franky = brain encoder + gpt2-nano (BASELINE configs[0], SURVEY cfg1), B = 4, T = 200, eager step",
                             "ms_per_step": round(dt * 1e3, 3), "frames_per_s": round(800 / dt, 1), "mfma_frac": round(800 / dt * 80e6 / MFMA_PEAK_BF16, 5)}
    try:
        gstep = tu.GraphedTrainStep(fr, (x, tok, None), opt, tc)
        dt = timeit(lambda: gstep((x, tok, None), 0))
        out["cfg1_franky_b4"]["graphed_ms_per_step"] = round(dt * 1e3, 3)
    except Exception as e:
        out["cfg1_franky_b4"]["graphed_ms_per_step"] = None
        torch.cuda.synchronize()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--head", default="l1", choices=["l1", "ce"], help="l1: the headline workload; ce: cfg2's CE-head variant (25 tokens, V = 50257)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-timers", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the live timing of BASELINE configs[4] (SimpleMAE, B = 32 / 256) and configs[0] (Franky)")
    ap.add_argument("--no-parity", action="store_true", help="skip the live bf16-vs-reference check at B = 1 (tests/golden/cfg2_b1*.npz)")
    ap.add_argument("--all-timers", action="store_true", help="HIP-event timing of every kernel family (default: the roofline kernel family only)")
    ap.add_argument("--dry-run", action="store_true", help="launcher + rendezvous only, gloo on the CPU (no GPU work)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr)
        sys.exit(2)
    if args.dry_run:
        return dry_run(args, rank, world)
    if not torch.cuda.is_available() or local >= torch.cuda.device_count():
        print(f"bench.py: rank {rank} has no GPU (LOCAL_RANK={local}, visible devices: {torch.cuda.device_count()})", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        dp_env()                                 # also when an external launcher (the driver's torchrun) started the ranks
        dist.init_process_group("nccl", device_id=dev)

    from frankenstein_amd import kernels as K
    from frankenstein_amd.utils import train_utils as tu
    model, cfg = cfg2_model(args.dtype, args.head)
    init_weights(model)
    model.to(dev)
    tcfg = tu.TrainConfig(batch_size=args.batch * world, mixed_precision=(args.dtype == "bf16"), use_scheduler=False,
                          learning_rate=1e-4)
    opt = tu.FusedAdamW(model, lr=tcfg.learning_rate, weight_decay=tcfg.weight_decay, grad_clip=tcfg.grad_clip)
    opt.sync.measure = world > 1
    sched = tu.init_lr_scheduler(tcfg)

    B, T, Cn = args.batch, 600, 256
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    def labels():
        if args.head == "ce":                       # GPT-2 token ids with a random-length -100 tail
            tok = torch.randint(0, 50257, (B, 25), device=dev, generator=g)
            tail = torch.randint(1, 11, (B, 1), device=dev, generator=g)
            return tok.masked_fill(torch.arange(25, device=dev)[None, :] >= 25 - tail, -100)
        return torch.randn(B, 32, 128, device=dev, generator=g)

    pool = [(torch.randn(B, T, Cn, device=dev, generator=g), labels(), None) for _ in range(2)]   # synthetic batches resident in HBM

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step = 0
    for _ in range(args.warmup):
        tu.train_step(model, pool[step % 2], opt, step, tcfg, sched)
        step += 1
    sync()
    if not args.no_timers:
        K.TIMERS = {}
        K.TIMER_PREFIX = None if args.all_timers else "attn_"    # ~2000 event pairs per step cost ~1.2 ms; the roofline needs attention only
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = tu.train_step(model, pool[step % 2], opt, step, tcfg, sched)
        step += 1
    sync()
    dt = time.perf_counter() - t0
    timers, K.TIMERS = K.TIMERS, None
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    assert torch.isfinite(loss).item(), "loss diverged"
    dp = dp_report(opt.sync, opt.arena, dev, args.steps) if world > 1 else None

    if rank == 0:
        frames = B * T * world * args.steps
        value = frames / dt
        fpf = FLOP_PER_FRAME_CE if args.head == "ce" else FLOP_PER_FRAME
        fams = {}
        for name, evs in (timers or {}).items():
            ms = [a.elapsed_time(b) for a, b in evs]
            fams[name] = (sum(ms), len(ms))
        roof = None
        enc = {n: v for n, v in fams.items() if n.startswith("attn_") and ":%dx6x6144x6144x64" % B in n}
        detail = {}
        for n, (tot, cnt) in sorted(fams.items(), key=lambda kv: -kv[1][0])[:8]:
            detail[n] = {"ms_total_per_step": round(tot / args.steps, 3), "launches_per_step": cnt // args.steps}
        if enc:
            name, (tot, cnt) = max(enc.items(), key=lambda kv: kv[1][0])
            bwd = name.startswith("attn_bwd")
            fl = attn_flops(B, 6, 6144, 6144, 64, 256, bwd)
            avg_s = tot / cnt / 1e3
            traffic, traffic_src = None, None
            try:   # HBM bytes per call from the newest committed rocprofv3 PMC passes (FETCH_SIZE x2 corrected + WRITE_SIZE)
                tpath = _latest_profile("pmc_traffic.json")
                pm = json.load(open(tpath))
                traffic = pm["fk_attn_bwd_bytes_per_call"] if bwd else None
                traffic_src = f"profiles/{tpath.name}, builder-run rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (not measured in this run)"
            except Exception:
                pass
            roof = {"kernel": "fk_attn_bwd (attn_bwd_dq [+ delta] and attn_bwd_dkdv launches of one call)" if bwd else "fk_attn_fwd",
                    "bound": "mfma", "achieved": round(fl / avg_s / 1e12, 2), "peak": MFMA_PEAK_BF16 / 1e12,
                    "unit": "TFLOP/s", "frac": round(fl / avg_s / MFMA_PEAK_BF16, 4), "traffic": traffic,
                    "traffic_source": traffic_src,
                    "traffic_note": "HBM bytes per call; algorithmic bytes 1.21e9 with every tensor counted once; the two deterministic kernels (dQ, dK/dV) each have to read Q, K, V, dO, so their compulsory traffic is 1.82e9",
                    "flops_per_launch": fl, "avg_launch_ms": round(avg_s * 1e3, 3)}
        out = {
            "metric": "neural frames/sec (train fwd+bwd+AdamW, whole job)", "value": round(value, 1), "unit": "frames/s",
            "n_gpus": world, "world": world, "rccl": (".".join(map(str, torch.cuda.nccl.version())) if world > 1 else None),
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "brainformer-small (6L d=384 6x64 heads, N=6144 tokens) + 2-block perceiver, "
                                   + ("L1 head" if args.head == "l1" else "CE head (25 tokens, V=50257, fused head loss)") + "; "
                                   "fwd+bwd+clip+AdamW", "per_gpu_batch": B, "global_batch": B * world, "frames_T": T,
                       "electrodes": Cn, "parallelism": f"dp{world}", "weights": "random-init (seed 42)"},
            "roofline": roof,
            "step_roofline": {"bound": "mfma", "achieved": round(value / world * fpf / 1e12, 2),
                              "mfma_util_pmc": mfma_util_pmc(),
                              "peak": MFMA_PEAK_BF16 / 1e12, "unit": "TFLOP/s",
                              "frac": round(value / world * fpf / MFMA_PEAK_BF16, 4),
                              "note": f"whole step per GPU: frames/s x {fpf / 1e9:.4f} GFLOP/frame (SURVEY §8d)"},
            "dp": dp, "nccl_max_nchannels": os.environ.get("NCCL_MAX_NCHANNELS") if world > 1 else None,
            "kernel_families": detail,
            "loss": round(float(loss), 5),
        }
        # the side legs never cost the headline line: a failure is reported in its block
        if world == 1 and args.dtype == "bf16" and not args.no_parity:      # N = 1 only, like cpu_baseline: the other ranks of a DP run do not wait for it
            try:
                out["parity"] = parity_live(dev)
            except Exception as e:
                out["parity"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
        if world == 1 and not args.no_other_configs:
            del model, opt, pool
            torch.cuda.empty_cache()
            try:
                out["other_configs"] = other_configs(dev)
            except Exception as e:
                out["other_configs"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
            import frankenstein_amd as fa
            fa.set_compute_dtype(args.dtype)
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as e:
                out["cpu_baseline"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
