"""Where a wave of fk_mlp_bwd_fused spends its cycles: -DMF_STAMP build (python tools/build_variant.py mf_stamp mlp_fused.hip -DMF_STAMP),
FRANKEN_HIP_LIB=.../lib_mf_stamp.so python tools/stamp_mlp.py.  Segment sums over all waves (s_memtime ticks; every stamp drains lgkmcnt,
so the instrumented kernel is a little slower than the product) and the clock the wave lifetimes imply."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frankenstein_amd import kernels as K, _lib
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
M, d, H = 32 * 6144, 384, 1536
def rnd(*s): return (torch.randn(*s, device=dev, generator=g) * 0.5).to(torch.bfloat16)
dy, h13, w2t, w13t = rnd(M, d), rnd(M, 2 * H), rnd(H, d), rnd(d, 2 * H)
x = rnd(M, d)
lib = _lib.lib()
buf = (ctypes.c_ulonglong * 16)()
names = ["prologue (chunk 0's first product and SwiGLU', alone)", "A: first product groups 0-4", "barrier B (W13T + h13 tile landed, W2T slot free)",
         "A: group 5 + the stream's operands", "prologue: setup, requests issued", "prologue: first wait (everything landed) + barrier", "B: the generated step (second product beside the next chunk's SwiGLU', barrier A inside)", "tail wait", "last chunk's second product", "dx epilogue"]
def run(n, between):
    K.mlp_bwd_fused(dy, w2t, h13, w13t); torch.cuda.synchronize(); lib.fk_debug_mf_stamps(buf, 1)
    ts = []
    for _ in range(n):
        if between: between()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); K.mlp_bwd_fused(dy, w2t, h13, w13t); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    lib.fk_debug_mf_stamps(buf, 1)
    us = sorted(ts)[n // 2] * 1e3
    v = [x / n for x in buf]
    waves = (M // 128) * 4
    chunks = H // 32
    print(f"  {us:.1f} us per call; wave lifetime {v[15] / waves:.0f} ticks -> {v[15] / (1024 * us):.0f} MHz if the 1024 wave slots are always occupied")
    tot = sum(v[:10])
    for i, nme in enumerate(names):
        per = v[i] / waves / (chunks if i in (1, 2, 3, 6, 7) else 1)
        print(f"    {nme:52s} {100 * v[i] / tot:5.1f} %   {per:8.0f} ticks per wave" + (" and chunk" if i in (1, 2, 3, 6, 7) else ""))
print("back to back:")
run(5, None)
print("with a memory-bound launch (norm forward over the same rows) between the calls:")
gam = torch.ones(d, device=dev)
run(5, lambda: K.norm_fwd(x, gam, gam, 1e-5, 0))
