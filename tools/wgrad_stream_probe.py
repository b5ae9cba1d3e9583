"""Which parameters differ between two identical bf16 training runs when the weight-gradient side stream is on (FK_WGRAD_STREAM=1)?"""
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
caps = []
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
            if os.environ.get("PROBE_CAPTURE"):      # keep the operands / results of every attention backward of this run
                E._CAPTURE = []
            loss, _ = m(x, y, date_info=None)
            loss.backward()
            torch.cuda.synchronize()
            if E._CAPTURE is not None:
                caps.append(E._CAPTURE)
                E._CAPTURE = None
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
if caps:
    # which operand of which attention backward differs between run 0 and the others?  (index 0 = first backward executed = last layer)
    for r in range(1, len(caps)):
        for i, (a, b) in enumerate(zip(caps[0], caps[r])):
            for key in a:
                nd = int((a[key] != b[key]).sum())
                if nd:
                    d = (a[key].float() - b[key].float()).abs()
                    idx = (a[key] != b[key]).nonzero()
                    cols = sorted(set(idx[:, -1].tolist()))[:24]
                    rows = sorted(set(idx[:, 0].tolist()))[:12] if idx.shape[1] > 1 else []
                    print(f"run {r} vs 0: attention backward #{i}: {key} differs in {nd} of {a[key].numel()} elements, max |diff| {float(d.max()):.3e}; "
                          f"last-dim indices {cols}{'...' if len(cols) == 24 else ''}; first-dim indices {rows}")
    # did an operand change WHILE the attention backward ran (clone before the launch vs clone after it, same run)?
    for r, cap in enumerate(caps):
        for i, c in enumerate(cap):
            for key in ("qkv", "o", "do", "lse"):
                nd = int((c[key] != c[key + "_pre"]).sum())
                if nd:
                    print(f"run {r}: attention backward #{i}: {key} changed during the call in {nd} elements")
if caps:
    # did an operand change WHILE the attention backward ran (clone before the launch vs clone after it, same run)?
    for r, cap in enumerate(caps):
        for i, c in enumerate(cap):
            for key in ("qkv", "o", "do", "lse"):
                nd = int((c[key] != c[key + "_pre"]).sum())
                if nd:
                    print(f"run {r}: attention backward #{i}: {key} changed during the call in {nd} elements")
print("done")
