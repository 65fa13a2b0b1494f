import torch, time
x = torch.empty(1 << 28, dtype=torch.float32, device="cuda")   # 1 GiB
y = torch.empty_like(x)
x.normal_()
for name, fn, nbytes in (("copy 1 GiB (read + write)", lambda: y.copy_(x), 2 * x.numel() * 4),
                         ("fill 1 GiB (write only)", lambda: y.fill_(1.0), x.numel() * 4),
                         ("sum 1 GiB (read only)", lambda: x.sum(), x.numel() * 4),
                         ("add 2 in 1 out", lambda: torch.add(x, y, out=y), 3 * x.numel() * 4)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    print(f"{name:28s} {ms*1e3:8.1f} us  {nbytes / ms / 1e9:7.2f} TB/s")
