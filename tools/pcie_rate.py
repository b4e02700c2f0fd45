import torch, time
for mb in (4, 10, 39, 160):
    n = mb << 20
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    d = torch.empty(n, dtype=torch.uint8, device="cuda")
    for name, f in (("H2D", lambda: d.copy_(h, non_blocking=True)), ("D2H", lambda: h.copy_(d, non_blocking=True))):
        f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): f()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"{name} {mb:4d} MB pinned: {dt*1e3:7.3f} ms  {n/dt/1e9:6.1f} GB/s")
