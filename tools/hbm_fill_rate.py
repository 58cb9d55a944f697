import torch, time
x = torch.empty(32 * 1024**3 // 8, dtype=torch.int64, device="cuda")
y = torch.empty_like(x)
for name, fn in (("fill_", lambda: x.fill_(1)), ("zero_", lambda: x.zero_()), ("copy_", lambda: y.copy_(x))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    b = x.numel() * 8 * (2 if name == "copy_" else 1)
    print(name, "%.2f ms" % (dt * 1e3), "%.2f TB/s" % (b / dt / 1e12))
