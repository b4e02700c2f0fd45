"""While a 512-frame detection is in flight: how long does a 20 MB device-to-host copy take (pinned linear / pageable / strided)?"""
import sys, time, ctypes as C
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from vulkansift_amd import api
L = api.lib()
L.vksift_setLogLevel(api.VKSIFT_LOG_ERROR)
L.vksift_hip_stream_create.restype = C.c_void_p
L.vksift_hip_host_malloc.restype = C.c_void_p; L.vksift_hip_host_malloc.argtypes = [C.c_size_t]
L.vksift_hip_malloc.restype = C.c_void_p; L.vksift_hip_malloc.argtypes = [C.c_size_t]
L.vksift_hip_memcpy2d_d2h.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]
B, W, H = 512, 640, 480
gen = np.stack([api.gen_synthetic_image(0x5EED0000 + i, W, H) for i in range(64)])
d = torch.from_numpy(np.ascontiguousarray(np.concatenate([gen] * 8))).cuda()
inst = api.Instance(api.default_config(sift_buffer_count=B, input_image_max_size=W * H), batch_capacity=B)
N = 20 << 20
dev = L.vksift_hip_malloc(2 * N); pin = L.vksift_hip_host_malloc(N); page = np.empty(N, np.uint8)
s1 = L.vksift_hip_stream_create()
def timed(fn, busy):
    torch.cuda.synchronize()
    if busy:
        inst.detectFeaturesBatchDevice(d.data_ptr(), B, W, H, 0)
    t0 = time.perf_counter(); fn(); L.vksift_hip_stream_sync(s1); dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return dt * 1e3
for _ in range(2):
    inst.detectFeaturesBatchDevice(d.data_ptr(), B, W, H, 0)
torch.cuda.synchronize()
for name, fn in (("pinned linear", lambda: L.vksift_hip_memcpy_d2h(pin, dev, N, s1)),
                 ("pageable linear", lambda: L.vksift_hip_memcpy_d2h(page.ctypes.data, dev, N, s1)),
                 ("pinned strided (2-D)", lambda: L.vksift_hip_memcpy2d_d2h(pin, 4096, dev, 8192, 4096, N // 4096, s1))):
    print(f"{name:22s}: idle GPU {timed(fn, False):7.2f} ms   behind a queued detection {timed(fn, True):7.2f} ms")
