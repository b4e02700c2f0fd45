"""Does a device-to-host copy of pinned memory make progress while another stream keeps the GPU busy with kernels? (torch only)"""
import time, torch
dev = torch.device("cuda", 0)
x = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
src = torch.empty(160 << 20, dtype=torch.uint8, device=dev)
dst = torch.empty(160 << 20, dtype=torch.uint8).pin_memory()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def busy(n):
    with torch.cuda.stream(sa):
        for _ in range(n):
            y = x @ x
torch.cuda.synchronize()
t0 = time.perf_counter(); busy(20); torch.cuda.synchronize(); tk = time.perf_counter() - t0
t0 = time.perf_counter()
with torch.cuda.stream(sb):
    dst.copy_(src, non_blocking=True)
sb.synchronize(); tc = time.perf_counter() - t0
print(f"20 matmuls alone {tk*1e3:.2f} ms; copy alone {tc*1e3:.2f} ms ({160/1024/tc:.1f} GiB/s)")
for order in ("copy first", "kernels first"):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if order == "copy first":
        with torch.cuda.stream(sb):
            dst.copy_(src, non_blocking=True)
        busy(20)
    else:
        busy(20)
        with torch.cuda.stream(sb):
            dst.copy_(src, non_blocking=True)
    sb.synchronize(); t_copy = time.perf_counter() - t0
    torch.cuda.synchronize(); t_all = time.perf_counter() - t0
    print(f"{order}: copy done after {t_copy*1e3:.2f} ms, everything after {t_all*1e3:.2f} ms")
