#!/usr/bin/env python3
"""The single-image leg of bench.py alone (BASELINE config 2 read literally: vksift_detectFeatures + getFeaturesNumber + downloadFeatures
[+ match] per frame, C caller), for A/B runs: VKSIFT_LIB=<other build> python tools/single_leg.py; PROTO_TRACE=1 prints the host phases."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from vulkansift_amd import api  # noqa: E402

api.lib().vksift_setLogLevel(api.VKSIFT_LOG_ERROR)
w, h = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "640x480").split("x"))
img = api.gen_synthetic_image(0xC0FFEE, w, h)
print("SINGLE " + json.dumps(bench.single_image_latency(api, 0, img)))
