"""SimpleMAE B=32 training steps only (for rocprofv3 --kernel-trace --stats): which kernels fill a 6 ms step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import frankenstein_amd as fa
from frankenstein_amd.utils import train_utils as tu
from frankenstein_amd.models import simple_mae as sm

fa.set_compute_dtype("bf16")
B = int(os.environ.get("B", "32"))
ecfg = sm.SimpleEncoderConfig(block_size=600, patch_size=256, n_layers=6, dim=384, hidden_dim=1536, head_dim=64, n_heads=6, n_kv_heads=6)
dcfg = sm.SimpleMAEConfig(n_layers=2, dim=384, hidden_dim=1536, head_dim=64, n_heads=6, n_kv_heads=6)
m = sm.SimpleMAE(ecfg, dcfg).cuda()
opt = tu.FusedAdamW(m, lr=1e-4, weight_decay=1e-5, grad_clip=1.0)
tc = tu.TrainConfig(mixed_precision=True, use_scheduler=False, learning_rate=1e-4)
x = torch.randn(B, 600, 256, device="cuda")
for i in range(20):
    tu.train_step(m, (x, None, None), opt, i, tc)
torch.cuda.synchronize()
