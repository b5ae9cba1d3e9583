"""cfg2 (the benchmark step, B = 32): eager train_step against GraphedTrainStep (forward + backward replayed from one hipGraph, update eager):
does removing the host launch path / inter-dispatch gaps of ~400 dependent launches buy anything at this size?  Alternating timings."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from frankenstein_amd.utils import train_utils as tu

dev = torch.device("cuda", 0)
model, cfg = bench.cfg2_model("bf16", "l1")
bench.init_weights(model)
model.to(dev)
tc = tu.TrainConfig(batch_size=32, mixed_precision=True, use_scheduler=False, learning_rate=1e-4)
opt = tu.FusedAdamW(model, lr=1e-4, weight_decay=tc.weight_decay, grad_clip=tc.grad_clip)
sched = tu.init_lr_scheduler(tc)
g = torch.Generator(device=dev).manual_seed(1234)
batch = (torch.randn(32, 600, 256, device=dev, generator=g), torch.randn(32, 32, 128, device=dev, generator=g), None)
st = [0]


def eager():
    tu.train_step(model, batch, opt, st[0], tc, sched); st[0] += 1


def timeit(fn, n=10, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


print(f"eager   {timeit(eager):.3f} ms/step")
gs = tu.GraphedTrainStep(model, batch, opt, tc, sched)
graphed = lambda: gs(batch, 0)
for _ in range(3):
    print(f"graphed {timeit(graphed):.3f} ms/step")
    print(f"eager   {timeit(eager):.3f} ms/step")
