"""Shader clock during the cfg2 attention launches: -DFK_STAMP build (tools/build_variant.sh stamp frankenstein_amd/csrc/attention.hip
-DFK_STAMP), FRANKEN_HIP_LIB=.../lib_stamp.so python tools/clock_attn.py.  Sum of wave lifetimes (s_memtime) / (wave slots x event time)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frankenstein_amd import kernels as K, _lib

B, H, N, D, Cb = 32, 6, 6144, 64, 256
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
qkv = (torch.randn(B, N, 3 * H * D, device=dev, generator=g) * 0.5).to(torch.bfloat16)
q, k, v = (qkv[..., i * H * D:(i + 1) * H * D].unflatten(-1, (H, D)) for i in range(3))
mask = K.Mask(K.MASK_BLOCK_CAUSAL, Cb)
o, lse = K.attn_fwd(q, k, v, mask, q_prescaled=True)
do = (torch.randn(B, N, H, D, device=dev, generator=g) * 0.5).to(torch.bfloat16)
dqkv = torch.empty_like(qkv)
dq, dk, dv = (dqkv[..., i * H * D:(i + 1) * H * D].unflatten(-1, (H, D)) for i in range(3))
lib = _lib.lib()
buf = (ctypes.c_ulonglong * 16)()
def timed(f, n=5):
    f(); torch.cuda.synchronize(); lib.fk_debug_stamps(buf, 1)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    lib.fk_debug_stamps(buf, 1)
    return a.elapsed_time(b) / n, [x / n for x in buf[8:12]]
slots = 256 * 8                                                   # backward: 2 workgroups x 4 waves per CU
fslots = 256 * 8                                                  # forward (attn_fwd_asm_kernel): 2 workgroups x 4 waves per CU
ms, life = timed(lambda: K.attn_fwd(q, k, v, mask, q_prescaled=True))
print(f"fwd : {ms:.3f} ms, wave lifetimes {(life[0] + life[3]) / 1e9:.3f} G ticks -> {(life[0] + life[3]) / fslots / (ms * 1e3):.0f} MHz if every slot is always occupied")
ms, life = timed(lambda: K.attn_bwd(q, k, v, o, do, lse, dq, dk, dv, mask, q_prescaled=True))
print(f"bwd : {ms:.3f} ms (dQ + dK/dV), lifetimes dQ {life[1] / 1e9:.3f} G, dK/dV {life[2] / 1e9:.3f} G ticks -> {(life[1] + life[2]) / slots / (ms * 1e3):.0f} MHz")
