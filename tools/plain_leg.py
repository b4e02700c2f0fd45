"""The plain-API leg of bench.py alone (reference entry points only; deferred submission): python tools/plain_leg.py [--no-match] [--budget S]
VKSIFT_DEFER=0 gives the same calls launched one by one."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from vulkansift_amd import api  # noqa: E402

api.load()
api.lib().vksift_setLogLevel(api.VKSIFT_LOG_ERROR)
W, H = 640, 480
frames = [api.gen_synthetic_image(0x5EED0000 + i, W, H) for i in range(128)]
budget = float(sys.argv[sys.argv.index("--budget") + 1]) if "--budget" in sys.argv else 6.0
print("PLAINLEG " + json.dumps(bench.plain_api_protocol(api, 0, frames, W, H, "--no-match" not in sys.argv, budget_s=budget)))
