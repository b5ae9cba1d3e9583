import torch, time
dev = torch.device("cuda")
n = 1 << 30   # 1 GiB bf16 elements -> 2 GiB
x = torch.empty(n, dtype=torch.bfloat16, device=dev); y = torch.empty_like(x)
def t(f, reps=5):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
ms = t(lambda: x.fill_(1.0)); print(f"fill  2 GiB: {ms*1e3:.0f} us  {2*n/ms/1e9:.2f} TB/s written")
ms = t(lambda: y.copy_(x)); print(f"copy  2 GiB: {ms*1e3:.0f} us  {4*n/ms/1e9:.2f} TB/s (r+w)")
ms = t(lambda: x.sum()); print(f"sum   2 GiB: {ms*1e3:.0f} us  {2*n/ms/1e9:.2f} TB/s read")
xs = x[: 1 << 28]; ys = y[: 1 << 28]
ms = t(lambda: xs.fill_(1.0)); print(f"fill  512 MiB: {ms*1e3:.0f} us  {2*(1<<28)/ms/1e9:.2f} TB/s written")
ms = t(lambda: ys.copy_(xs)); print(f"copy  512 MiB: {ms*1e3:.0f} us  {4*(1<<28)/ms/1e9:.2f} TB/s (r+w)")
