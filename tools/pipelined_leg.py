"""bench.py's pipelined host-protocol leg alone (for a kernel / copy timeline): usage pipelined_leg.py [steps]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import importlib.util
spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from vulkansift_amd import api
api.lib().vksift_setLogLevel(api.VKSIFT_LOG_ERROR)
W, H, B = 640, 480, 512
gen = np.stack([api.gen_synthetic_image(0x5EED0000 + i, W, H) for i in range(64)])
host = np.ascontiguousarray(np.concatenate([gen] * 8))
frames = [host[i] for i in range(B)]
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
print("pipelined frames/s", bench.pipelined_protocol(api, 0, frames, W, H, B, True, steps), "C client" if bench.protocol_client(api) else "python loops")
