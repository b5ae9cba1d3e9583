"""Throughput of the other BASELINE.json configurations on one MI355X (not the headline bench): frames/s of a bf16 training step.
cfg5 = SimpleMAE pre-training, cfg4 front end = SoundStream tokenizer training (notebook size), cfg1 = Franky (B=4, T=200)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import frankenstein_amd as fa
from frankenstein_amd.utils import train_utils as tu

fa.set_compute_dtype("bf16")
def timeit(step, n=10, w=3):
    for _ in range(w): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

# cfg5: SimpleMAE
from frankenstein_amd.models import simple_mae as sm
ecfg = sm.SimpleEncoderConfig(block_size=600, patch_size=256, n_layers=6, dim=384, hidden_dim=1536, head_dim=64, n_heads=6, n_kv_heads=6)
dcfg = sm.SimpleMAEConfig(n_layers=2, dim=384, hidden_dim=1536, head_dim=64, n_heads=6, n_kv_heads=6)
m = sm.SimpleMAE(ecfg, dcfg).cuda()
opt = tu.FusedAdamW(m, lr=1e-4, weight_decay=1e-5, grad_clip=1.0)
for B in (32, 256):
    x = torch.randn(B, 600, 256, device="cuda")
    def step():
        out = m(x); loss = out[0] if isinstance(out, tuple) else out
        loss.backward(); opt.step()
    dt = timeit(step)
    print(f"cfg5 SimpleMAE pre-training     B={B:4d}: {dt * 1e3:7.2f} ms/step  {B * 600 / dt / 1e3:9.1f} k frames/s")

# cfg4 front end: SoundStream tokenizer
from frankenstein_amd.models import vq_brain as vq
net = vq.SoundStream(C=256, D=64, codebook_size=1024, n_electrodes=512).cuda()
opt2 = tu.FusedAdamW(net, lr=1e-4, weight_decay=1e-5, grad_clip=1.0)
tc = tu.TrainConfig(mixed_precision=True, use_scheduler=False, learning_rate=1e-4)
x = torch.randn(16, 768, 512, device="cuda")
st = [0]
def step2():
    tu.train_step(net, (x, None, None), opt2, st[0], tc); st[0] += 1
dt = timeit(step2)
print(f"cfg4 SoundStream tokenizer      B=  16: {dt * 1e3:7.2f} ms/step  {16 * 768 / dt / 1e3:9.1f} k frames/s")

# cfg1: Franky (brain encoder + gpt2-nano), B=4, T=200
from frankenstein_amd.models import brainformer as bf
from frankenstein_amd.models.gpt2_model import GPT, GPTConfig
from frankenstein_amd.models.notebook_models import BrainEncoder, Franky
enc = bf.MAEConfig(window_size=200, n_electrodes=256, patch_size=25, dim=128, n_layers=2, head_dim=32, hidden_dim=512, n_heads=4, n_kv_heads=4)
cfg = bf.Config(encoder=enc, n_output_tokens=32, output_dim=128, dim=128, n_layers=2, head_dim=32, hidden_dim=256, n_heads=4, n_kv_heads=4)
fr = Franky(BrainEncoder(cfg), GPT(GPTConfig(block_size=1024, vocab_size=50257, n_layer=2, n_head=4, n_embd=128, dropout=0.0, bias=True))).cuda()
opt3 = tu.FusedAdamW(fr, lr=1e-4, weight_decay=1e-5, grad_clip=1.0)
x = torch.randn(4, 200, 256, device="cuda"); tok = torch.randint(0, 50257, (4, 25), device="cuda"); tok[:, -3:] = -100
def step3():
    tu.train_step(fr, (x, tok, None), opt3, st[0], tc); st[0] += 1
dt = timeit(step3)
print(f"cfg1 Franky (gpt2-nano decoder) B=   4: {dt * 1e3:7.2f} ms/step  {4 * 200 / dt / 1e3:9.1f} k frames/s")

# the same steps with forward + backward replayed from one hipGraph (train_utils.GraphedTrainStep)
def graphed(name, model, batch, opt, units):
    try:
        g = tu.GraphedTrainStep(model, batch, opt, tc)
        dt = timeit(lambda: g(batch, 0))
        print(f"{name:32s} graphed: {dt * 1e3:7.2f} ms/step  {units / dt / 1e3:9.1f} k frames/s")
    except Exception as e:       # a forward with a host sync cannot be captured
        print(f"{name:32s} graphed: not capturable ({type(e).__name__}: {str(e)[:120]})")
        torch.cuda.synchronize()

graphed("cfg1 Franky (gpt2-nano decoder)", fr, (x, tok, None), opt3, 4 * 200)
graphed("cfg4 SoundStream tokenizer", net, (torch.randn(16, 768, 512, device="cuda"), None, None), opt2, 16 * 768)
graphed("cfg5 SimpleMAE B=32", m, (torch.randn(32, 600, 256, device="cuda"), None, None), opt, 32 * 600)
