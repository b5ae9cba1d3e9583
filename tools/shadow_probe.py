import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from frankenstein_amd.utils import train_utils as tu
from frankenstein_amd import engine as E
model, cfg = bench.cfg2_model("bf16"); bench.init_weights(model); model.cuda()
tcfg = tu.TrainConfig(batch_size=32, mixed_precision=True, use_scheduler=False, learning_rate=1e-4)
opt = tu.FusedAdamW(model, lr=1e-4, weight_decay=tcfg.weight_decay, grad_clip=tcfg.grad_clip)
x = (torch.randn(32, 600, 256, device="cuda"), torch.randn(32, 32, 128, device="cuda"), None)
for i in range(2): tu.train_step(model, x, opt, i, tcfg)
names = {p.data_ptr(): n for n, p in model.named_parameters()}
log = []
orig = E._run_jobs
def spy(ent):
    log.append([names.get(r().data_ptr(), "?") for r in ent.params] + [tuple(ent.tensor.shape)])
    orig(ent)
E._run_jobs = spy
tu.train_step(model, x, opt, 2, tcfg)
print(len(log), "lazy re-packs in a steady-state step")
for l in log[:40]: print(l)
