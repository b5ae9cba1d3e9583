"""Isolated timing of every GEMM shape of the cfg2 step (per encoder layer), interleaved rounds in one process.
Usage: python tools/gemm_bench.py [rounds]   -> one line per shape: us, TFLOP/s, algorithmic GB/s."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frankenstein_amd import kernels as K

M, d, H = 32 * 6144, 384, 1536
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
def rnd(*s): return (torch.randn(*s, device=dev, generator=g) * 0.5).to(torch.bfloat16)

x, dy = rnd(M, d), rnd(M, d)
qkv, h13, gg, dh13 = rnd(M, 3 * d), rnd(M, 2 * H), rnd(M, H), rnd(M, 2 * H)
w_qkv, w_proj, w13, w2 = rnd(3 * d, d), rnd(d, d), rnd(2 * H, d), rnd(d, H)
w13t, w2t, w_qkvt = rnd(d, 2 * H), rnd(H, d), rnd(d, 3 * d)
table = torch.randn(6144, 32, 2, device=dev)

cases = {
    "nt qkv+rope  N=1152 K=384": (lambda: K.gemm_nt_rope(x, w_qkv, None, table, 6144, 0, 64, 2 * d), 2 * M * 1152 * 384, M * (384 + 1152) * 2),
    "nt proj+res  N=384  K=384": (lambda: K.gemm_nt(x, w_proj, None, residual=dy), 2 * M * 384 * 384, M * 384 * 3 * 2),
    "nt swiglu    N=3072 K=384": (lambda: K.gemm_nt_swiglu(x, w13), 2 * M * 3072 * 384, M * (384 + 3072 + 1536) * 2),
    "nt down+res  N=384  K=1536": (lambda: K.gemm_nt(gg, w2, None, residual=dy), 2 * M * 384 * 1536, M * (1536 + 384 * 2) * 2),
    "nt dswiglu   N=1536 K=384": (lambda: K.gemm_nt_dswiglu(dy, w2t, h13), 2 * M * 1536 * 384, M * (384 + 3072 * 2) * 2),
    "nt d_up      N=384  K=3072": (lambda: K.gemm_nt(dh13, w13t), 2 * M * 384 * 3072, M * (3072 + 384) * 2),
    "nt d_qkv     N=384  K=1152": (lambda: K.gemm_nt(qkv, w_qkvt), 2 * M * 384 * 1152, M * (1152 + 384) * 2),
    "fused dswiglu + d_up (one launch)": (lambda: K.mlp_bwd_fused(dy, w2t, h13, w13t), 2 * M * 1536 * 384 + 2 * M * 384 * 3072, M * (384 + 3072 * 2 + 384) * 2),
    "tn dW_qkv    1152x384": (lambda: K.gemm_tn(qkv, x), 2 * M * 1152 * 384, M * (1152 + 384) * 2),
    "tn dW_up     3072x384": (lambda: K.gemm_tn(dh13, x), 2 * M * 3072 * 384, M * (3072 + 384) * 2),
    "tn dW_down   384x1536": (lambda: K.gemm_tn(dy, gg), 2 * M * 384 * 1536, M * (1536 + 384) * 2),
    "tn dW_proj   384x384": (lambda: K.gemm_tn(dy, x), 2 * M * 384 * 384, M * (384 + 384) * 2),
}
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for f, _, _ in cases.values():
    f()
torch.cuda.synchronize()
times = {k: [] for k in cases}
for _ in range(rounds):
    for k, (f, _, _) in cases.items():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); f(); b.record(); torch.cuda.synchronize()
        times[k].append(a.elapsed_time(b) / 2)
tot = 0.0
for k, (_, fl, by) in cases.items():
    t = sorted(times[k])[len(times[k]) // 2]
    tot += t
    print(f"{k:30s} {t * 1e3:8.1f} us  {fl / t / 1e9:7.1f} TF/s  {by / t / 1e6:7.1f} GB/s(alg)")
print(f"sum per layer {tot:.3f} ms  -> x6 = {6 * tot:.2f} ms/step")
