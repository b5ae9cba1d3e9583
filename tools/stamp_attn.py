"""Cycle breakdown of the dK/dV attention kernel from an -DFK_STAMP build (tools/build_variant.sh stamp <src> -DFK_STAMP):
FRANKEN_HIP_LIB=frankenstein_amd/variants/lib_stamp.so python tools/stamp_attn.py"""
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
for _ in range(2):
    K.attn_bwd(q, k, v, o, do, lse, dq, dk, dv, mask, q_prescaled=True)
lib.fk_debug_stamps(buf, 1)
K.attn_bwd(q, k, v, o, do, lse, dq, dk, dv, mask, q_prescaled=True)
lib.fk_debug_stamps(buf, 1)
names = ["dma issue + stats load", "S/dP MFMA (+acc init, row reads)", "exp2 / dS VALU", "dV/dK MFMA (+cvt, tr reads)", "stats store", "wait + barrier", "prologue", "epilogue stores"]
tot = sum(buf[:8])
for n, c in zip(names, buf[:8]):
    print(f"{n:36s} {c / 1e6:10.1f} Mcycles  {100 * c / tot:5.1f} %")
print(f"total wave-cycles {tot / 1e6:.1f} M  (= {tot / (256 * 4 * 2) / 1e6:.3f} M cycles per wave slot at 2 waves/SIMD)")
