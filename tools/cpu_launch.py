import sys, os, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from frankenstein_amd.utils import train_utils as tu
from frankenstein_amd import kernels as K
model, cfg = bench.cfg2_model("bf16"); bench.init_weights(model); model.cuda()
tcfg = tu.TrainConfig(batch_size=32, mixed_precision=True, use_scheduler=False, learning_rate=1e-4)
opt = tu.FusedAdamW(model, lr=1e-4, weight_decay=tcfg.weight_decay, grad_clip=tcfg.grad_clip)
x = (torch.randn(32, 600, 256, device="cuda"), torch.randn(32, 32, 128, device="cuda"), None)
for i in range(3): tu.train_step(model, x, opt, i, tcfg)
torch.cuda.synchronize()
for timers in (False, True):
    K.TIMERS = {} if timers else None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(5): tu.train_step(model, x, opt, i, tcfg)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"timers={timers}: host enqueue {1e3*(t1-t0)/5:.1f} ms/step, wall {1e3*(t2-t0)/5:.1f} ms/step")
