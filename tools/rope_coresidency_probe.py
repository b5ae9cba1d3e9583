"""fk_rope (pure elementwise: two loads, eight packed multiply-adds, one store per thread) beside the small bf16 weight-gradient GEMM on a
second stream: which elements differ from the quiet run, mapped to (lane, element of the thread's 8-element chunk), and by how much
against an fp64 evaluation."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frankenstein_amd import kernels as K

dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
B, N, H, D = 3, 4864, 5, 64
d = H * D
x0 = (torch.randn(B, N, 3 * d, device=dev, generator=g) * 0.5).bfloat16()
table = torch.randn(N, D // 2, 2, device=dev, generator=g)
M = B * N
ga = (torch.randn(M, 320, device=dev, generator=g) * 0.5).bfloat16()
gb = (torch.randn(M, 840, device=dev, generator=g) * 0.5).bfloat16()
side = torch.cuda.Stream()


def run():
    t = x0.clone()
    K.rope_(t, 2 * H, D, table, 0)
    return t


ref = run()
torch.cuda.synchronize()
# fp64 evaluation
xv = x0[..., :2 * d].double().view(B, N, 2 * H, D // 2, 2)
c, s = table[None, :, None, :, 0].double(), table[None, :, None, :, 1].double()
want = torch.stack([xv[..., 0] * c - xv[..., 1] * s, xv[..., 0] * s + xv[..., 1] * c], -1).view(B, N, 2 * d)
print("quiet vs fp64: max |err| / bf16 ulp", float(((ref[..., :2 * d].double() - want).abs() / torch.exp2(torch.floor(torch.log2(want.abs().clamp_min(1e-30))) - 7)).max()))
from collections import Counter
lanes, elems, mags = Counter(), Counter(), []
ndiff = 0
for rep in range(int(os.environ.get("PROBE_REPS", "8"))):
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(10):
            K.gemm_tn(ga, gb)
    out = run()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    idx = (out != ref).nonzero()
    ndiff += idx.shape[0]
    for b_, n_, c_ in idx[:20000].tolist():
        chunk = (b_ * N + n_) * (2 * d // 8) + c_ // 8          # one thread per 8-element chunk, chunks of the rotated columns only
        lanes[chunk % 64 // 16] += 1
        elems[c_ % 8] += 1
    if idx.shape[0]:
        o_, w_ = out[..., :2 * d].double(), want
        sel = (out != ref)[..., :2 * d]
        ulp = torch.exp2(torch.floor(torch.log2(w_[sel].abs().clamp_min(1e-30))) - 7)
        mags.append(float(((o_[sel] - w_[sel]).abs() / ulp).median()))
print("differing elements over all runs:", ndiff)
print("by lane quarter (0: lanes 0-15 ... 3: lanes 48-63):", dict(sorted(lanes.items())))
print("by element of the 8-element chunk:", dict(sorted(elems.items())))
print("median |contended - fp64| / ulp per run:", [round(m, 1) for m in mags])
