"""Which parameters differ between two identical bf16 training runs when the weight-gradient side stream is on (FK_WGRAD_STREAM=1)?
(The operand-capture mode of round 2 is gone: tools/dq_coresidency_probe.py replays the attention backward itself, tools/coresidency_sweep.py
every kernel, beside a concurrent GEMM.)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FK_WGRAD_STREAM", "1")
import torch
import frankenstein_amd as fa
from frankenstein_amd.models import brainformer as bf
from frankenstein_amd.utils import train_utils as tu

fa.set_compute_dtype("bf16")
enc = bf.MAEConfig(window_size=475, n_electrodes=256, patch_size=25, dim=320, n_layers=2, head_dim=64, hidden_dim=840, n_heads=5, n_kv_heads=5)
cfg = bf.Config(encoder=enc, n_output_tokens=9, output_dim=70, dim=320, n_layers=1, head_dim=64, hidden_dim=328, n_heads=5, n_kv_heads=5)
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.randn(3, 475, 256, device="cuda", generator=g)
y = torch.randn(3, 9, 70, device="cuda", generator=g)
if os.environ.get("PROBE_CFG2"):                     # the benchmark's dimensions at B = 2 (ring-buffered GEMMs, large-tile TN)
    enc = bf.MAEConfig(window_size=600, n_electrodes=256, patch_size=25, dim=384, n_layers=2, head_dim=64, hidden_dim=1536, n_heads=6, n_kv_heads=6)
    cfg = bf.Config(encoder=enc, n_output_tokens=32, output_dim=64, dim=384, n_layers=1, head_dim=64, hidden_dim=1536, n_heads=6, n_kv_heads=6)
    x = torch.randn(2, 600, 256, device="cuda", generator=g)
    y = torch.randn(2, 32, 64, device="cuda", generator=g)
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
runs = []
from frankenstein_amd import engine as E
for rep in range(3):
    torch.manual_seed(0)
    m = bf.BrainFormer(cfg).cuda()
    opt = tu.FusedAdamW(m, lr=2e-3, weight_decay=0.0, grad_clip=1.0)
    tc = tu.TrainConfig(mixed_precision=True, use_scheduler=False, learning_rate=2e-3)
    for s in range(nsteps):
        if os.environ.get("PROBE_SYNC"):
            # the step split by hand with a full device sync between backward and update
            for gq in opt.param_groups: gq['lr'] = 2e-3
            loss, _ = m(x, y, date_info=None)
            loss.backward()
            torch.cuda.synchronize()
            if os.environ.get("PROBE_GRADS"):        # compare the gradients themselves instead of the updated parameters
                runs.append({k: v.grad.detach().clone() for k, v in m.named_parameters()})
                break
            opt.step()
        else:
            tu.train_step(m, (x, y, None), opt, s, tc)
    torch.cuda.synchronize()
    if not os.environ.get("PROBE_GRADS"):
        runs.append({k: v.detach().clone() for k, v in m.named_parameters()})
for k in runs[0]:
    d = [int((runs[0][k] != runs[r][k]).sum()) for r in (1, 2)]
    if any(d):
        md = max(float((runs[0][k] - runs[r][k]).abs().max()) for r in (1, 2))
        print(f"{k:60s} differing elements vs run 0: {d}  of {runs[0][k].numel()}  max |diff| {md:.3e}  (max |value| {float(runs[0][k].abs().max()):.3e})")
print("done")
