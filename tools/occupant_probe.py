"""What does a persistent NT GEMM launch lose when another kernel holds a few CUs?  (what 8 RCCL channels look like.)
An occupant of 8 workgroups, each holding LDS for ~3 ms (tools/probes/occupant.hip, built here with hipcc), is started on a second stream;
right behind it the cfg2 GEMM launches run on the main stream, timed with events: quiet time, time beside the occupant, ratio.
FK_NT_GRID_MULT (read once per process) sets the workgroups per CU-sized wave of the persistent kernels: run once per value.
usage: FK_NT_GRID_MULT=4 python tools/occupant_probe.py [occupant LDS KiB, default 48]"""
import ctypes
import os
import subprocess
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frankenstein_amd import kernels as K

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = "/tmp/libfk_occupant.so"
if not os.path.exists(so):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(root, "tools/probes/occupant.hip"), "-o", so], check=True)
occ = ctypes.CDLL(so)
occ.fk_occupy.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
lds_kib = int(sys.argv[1]) if len(sys.argv) > 1 else 48

dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
M, d, H = 32 * 6144, 384, 1536
def rnd(*s): return (torch.randn(*s, device=dev, generator=g) * 0.5).to(torch.bfloat16)
x, dy, gg, dh13 = rnd(M, d), rnd(M, d), rnd(M, H), rnd(M, 2 * H)
w13, w_proj, w2, w13t = rnd(2 * H, d), rnd(d, d), rnd(d, H), rnd(d, 2 * H)
sink = torch.zeros(4, dtype=torch.int32, device=dev)
cases = {"nt swiglu  N=3072 K=384 ": lambda: K.gemm_nt_swiglu(x, w13),
         "nt proj+res N=384 K=384  ": lambda: K.gemm_nt(x, w_proj, None, residual=dy),
         "nt down+res N=384 K=1536 ": lambda: K.gemm_nt(gg, w2, None, residual=dy),
         "nt d_up     N=384 K=3072 ": lambda: K.gemm_nt(dh13, w13t),
         "tn dW_up    3072x384     ": lambda: K.gemm_tn(dh13, x)}
side = torch.cuda.Stream()


def timed(f, occupied, reps=7):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        if occupied:
            with torch.cuda.stream(side):
                rc = occ.fk_occupy(8, lds_kib * 1024, 300_000, sink.data_ptr(), torch.cuda.current_stream().cuda_stream)      # 3 ms at 100 MHz
                assert rc == 0, rc
            torch.cuda._sleep(200_000)                      # ~0.1 ms on the main stream: the occupant is resident before the GEMM starts
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


for f in cases.values():
    f()
print(f"FK_NT_GRID_MULT={os.environ.get('FK_NT_GRID_MULT', '1')}  occupant: 8 workgroups x {lds_kib} KiB of LDS for 3 ms")
for name, f in cases.items():
    q, o = timed(f, False), timed(f, True)
    print(f"{name} quiet {q * 1e3:7.1f} us   beside the occupant {o * 1e3:7.1f} us   x{o / q:.3f}")
