"""HBM calibration on the GPU box: achieved GB/s of fill / copy / 1-read-2-write patterns (torch elementwise kernels)."""
import torch, time, json
dev = torch.device("cuda:0")
n = 1 << 29  # 2 GiB of fp32
a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
b = torch.empty_like(a); c = torch.empty_like(a)
def t(fn, bytes_, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return bytes_ * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9
out = {}
out["fill_write_only"] = t(lambda: b.fill_(1.0), 4 * n)
out["copy_1r_1w"] = t(lambda: b.copy_(a), 8 * n)
out["add_2r_1w"] = t(lambda: torch.add(a, b, out=c), 12 * n)
out["sum_read_only"] = t(lambda: a.sum(), 4 * n)
def r1w2():
    torch.mul(a, 2.0, out=b); 
out["mul_1r_1w"] = t(r1w2, 8 * n)
print(json.dumps(out))
