#!/usr/bin/env python3
"""BASELINE config 3 alone (64 x 1920x1080, detect only): the roofline_c3 object of bench.py. Run on the GPU box."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from vulkansift_amd import api
api.lib().vksift_setLogLevel(api.VKSIFT_LOG_ERROR)
torch.cuda.set_device(0)
r = bench.c3_roofline(api, torch, torch.device("cuda", 0), steps=int(sys.argv[1]) if len(sys.argv) > 1 else 5)
print("C3LEG " + json.dumps(r))
