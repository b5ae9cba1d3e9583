"""Greedy decoding latency of the cfg1 decoder (gpt2-nano: 2 layers, d=128, V=50257, 32 brain-prefix tokens): key/value-cached
incremental steps vs the reference-style full re-forward per token."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import frankenstein_amd as fa
from frankenstein_amd.models.gpt2_model import GPT, GPTConfig

fa.set_compute_dtype("bf16")
g = GPT(GPTConfig(block_size=1024, vocab_size=50257, n_layer=2, n_head=4, n_embd=128, dropout=0.0, bias=True)).cuda().eval()
prefix = torch.randn(1, 32, 128, device="cuda")
start = torch.full((1, 1), 50256, dtype=torch.long, device="cuda")
modes = {"re-forward": dict(use_cache=False), "kv-cache": dict(use_cache=True, use_graph=False),
         "kv-cache + hipGraph": dict(use_cache=True, use_graph=True)}
for n_new in (25, 200, 900):
    for name, kw in modes.items():
        g.generate(start, 4, prefix=prefix, top_k=1, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            g.generate(start, n_new, prefix=prefix, top_k=1, **kw)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        print(f"{n_new:4d} new tokens, {name:20s}: {dt * 1e3:8.1f} ms  ({n_new / dt:7.0f} tokens/s)")
