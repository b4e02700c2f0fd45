#!/usr/bin/env python3
"""bench.py — headline benchmark of the vksift detect/match hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): SIFT detect+match frames/s on 640x480 frames with ~2k keypoints.
One "step" = one pass of the hot path over one batch of `--batch` (default 128) synthetic 640x480 frames that are
already resident in HBM: batched detection (default vksift_Config: 2x up-sampling, automatic octave
count = 5, 3 scales/octave) followed by the 2-NN self-match of every frame (matchFeatures(i, i) of BASELINE
config 2, issued through the batched extension vksift_ext_matchFeaturesBatch). Every step recomputes everything; nothing is cached between steps.

Multi-GPU: one process per GPU; each rank owns its own batch (weak scaling, detection is per image and
needs no collective); the timed region is bracketed by a barrier + device synchronize and the maximum
over ranks is reported. PyTorch is used for torch.distributed (RCCL) and device buffers only.

Extra objects on the JSON line:
  roofline     pyramid+DoG pass of octave 0 (k_input_blit_2x + 6 k_blur_lean launches per step, 77 % of the pyramid's
               bytes): algorithmic bytes per launch / average launch duration, measured with HIP
               events recorded on the stream those kernels are launched on (the library's instance stream), inside
               the timed region; the coarser octaves run concurrently on their own streams and share the HBM
               bandwidth, so the figure is conservative. "traffic" is the PMC-measured HBM traffic per launch from a
               separate rocprofv3 --pmc run (profiles/*pmc*.json), if present
  cpu_baseline the CPU oracle (a scalar C port of the same algorithm) timed on a bounded sample (32 frames on 16 host
               threads, ~25 core-seconds), rank 0, N=1
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=128, help="frames per step and per GPU")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--no-match", action="store_true", help="detect only (BASELINE config 3 style runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-input", action="store_true", help="hand over host images (vksift_ext_detectFeaturesBatch): PCIe-inclusive rate, not the headline value")
    ap.add_argument("--cpu-frames", type=int, default=32, help="frames of the workload given to the CPU oracle")
    ap.add_argument("--cpu-threads", type=int, default=16, help="worker threads of the CPU baseline (one frame per thread at a time)")
    return ap.parse_args()


def cpu_baseline(frames, do_match, threads):
    """Time the oracle (scalar C port of the same algorithm) on a bounded sample of the same workload: one frame per
    worker thread at a time (the ctypes calls release the GIL, so the threads run on separate host cores). The first
    frame is also timed alone for the single-thread rate."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O

    cfg = O.default_config(math_mode=0)

    def one(img):
        feats, _ = O.detect(cfg, img)
        if do_match and len(feats) >= 2:
            O.match_2nn(feats, feats)
        return len(feats)

    t0 = time.perf_counter()
    n0 = one(frames[0])
    t_single = time.perf_counter() - t0
    threads = max(1, min(threads, len(frames)))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as ex:
        nfeat = list(ex.map(one, frames))
    dt = time.perf_counter() - t0
    return {
        "value": len(frames) / dt,
        "unit": "frames/s",
        "cores": threads,
        "kind": "port",
        "single_thread_value": 1.0 / t_single,
        "sample": f"{len(frames)} of the benchmark's {frames[0].shape[1]}x{frames[0].shape[0]} frames, detect"
                  + ("+self-match" if do_match else "") + f", {int(np.mean(nfeat))} features/frame, {threads} worker threads x "
                  + f"{len(frames) // threads} frame(s) each, {dt:.1f} s wall ({dt * threads:.0f} core-seconds); one frame alone {t_single:.2f} s; "
                  + f"host has {os.cpu_count()} logical cores",
    }


def pmc_traffic(w, h, batch):
    """HBM bytes per k_blur_stream launch measured with rocprofv3 --pmc (FETCH_SIZE / WRITE_SIZE in separate passes,
    gfx950 FETCH_SIZE correction applied: see profiles/README.md). Only valid for the workload it was measured on."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json")), reverse=True):
        try:
            d = json.load(open(path))
            if (d.get("width"), d.get("height"), d.get("batch")) == (w, h, batch):
                return d["hbm_bytes_per_blur_launch"]
        except Exception:
            pass
    return None


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    from vulkansift_amd import api

    api.lib().vksift_setLogLevel(api.VKSIFT_LOG_ERROR)
    W, H, B = args.width, args.height, args.batch
    do_match = not args.no_match

    # synthetic frames (seeded per global frame index), uploaded once: inputs are HBM-resident when timing starts
    frames = [api.gen_synthetic_image(0x5EED0000 + rank * B + i, W, H) for i in range(B)]
    d_frames = torch.from_numpy(np.stack(frames)).to(dev)
    torch.cuda.synchronize()

    cfg = api.default_config(sift_buffer_count=B, gpu_device_index=dev.index, input_image_max_size=max(W * H, 1024))
    inst = api.Instance(cfg, batch_capacity=B)

    def step():
        if args.host_input:
            inst.detectFeaturesBatch(frames, 0)   # host memcpy into pinned staging + H2D inside the timed region
        else:
            inst.detectFeaturesBatchDevice(d_frames.data_ptr(), B, W, H, 0)
        if do_match:
            for i0 in range(0, B, 64):      # 2-NN self-match of every frame, batched launches of <= 64 pairs
                ids = list(range(i0, min(B, i0 + 64)))
                inst.matchFeaturesBatch(ids, ids)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    nfeat = [inst.getFeaturesNumber(i) for i in range(B)]

    inst.setProfiling(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    acc = inst.getAccumulatedDetectTimings()
    match_ms = inst.getMatchTime() if do_match else None
    inst.close()

    if rank == 0:
        frames_total = B * world * args.steps
        launches = max(acc["nb_blur_launches"], 1)
        alg_per_launch = acc["pyramid_algorithmic_bytes"] / launches
        avg_launch_s = (acc["pyramid_ms"] * 1e-3) / launches
        achieved = alg_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        out = {
            "metric": "SIFT detect+match frames/sec (640x480, ~2k kp)" if do_match else "SIFT detect frames/sec",
            "value": frames_total / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"BASELINE config 2: {W}x{H} uint8 frames, detect" + (" + 2-NN self-match" if do_match else "")
                            + ", default vksift_Config (2x up-sampling, auto octaves, 3 scales/octave), inputs resident in HBM",
                "frames_per_step_per_gpu": B,
                "octaves": 5 if (W, H) == (640, 480) else None,
                "mean_features_per_frame": float(np.mean(nfeat)),
                "parallelism": f"batch split x{world}, no collectives",
                "input": "host images, upload inside the timed region" if args.host_input else "resident in HBM",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_blur_lean (pyramid + DoG pass, octave 0: 1280x960 planes)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": pmc_traffic(W, H, B),
                "algorithmic_bytes_per_launch": alg_per_launch,
                "avg_launch_us": avg_launch_s * 1e6,
                "launches_per_step": launches / max(acc["nb_calls"], 1),
            },
            "stage_ms_per_step": {k: acc[k] / max(acc["nb_calls"], 1) for k in
                                  ("upload_ms", "pyramid_ms", "extrema_ms", "scan_ms", "orientation_ms", "descriptor_ms", "total_ms")},
            "last_match_ms": match_ms,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(frames[: max(1, min(args.cpu_frames, B))], do_match, args.cpu_threads)
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
