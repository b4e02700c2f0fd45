#!/usr/bin/env python3
"""bench.py — headline benchmark of the vksift detect/match hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): SIFT detect+match frames/s on 640x480 frames with ~2k keypoints.

One "step" = one pass of the hot path over one batch of `--sub-batches` x `--batch` (default 3 x 512 = 1536) synthetic
640x480 frames that are already resident in HBM: per 512 frames one batched detection (default vksift_Config: 2x
up-sampling, automatic octave count = 5, 3 scales/octave) followed by the 2-NN self-match of every frame
(matchFeatures(i, i) of BASELINE config 2, issued through the batched extension). Every step recomputes everything;
nothing is cached between steps. 20 steps are ~1.3 s of timed GPU work. (512 frames per call: 40 GB of scale-space for the two
pyramid buffers of an instance — sized for 288 GB of HBM; 128 per call is 7 % slower, 1024 no faster.)

Multi-GPU: one process per GPU; each rank owns its own frames (weak scaling: detection is per image and needs no
collective); the timed region is bracketed by a barrier + device synchronize and the maximum over ranks is reported.
After the timed region every rank also runs its shard of BASELINE config 4 (50k x 50k 2-NN, query rows sharded, ONE
RCCL all-gather of the reference descriptors) and rank 0 prints the CRC-32 of the concatenated records: it must be the
same number at N = 1, 2, 4, 8. PyTorch is used for torch.distributed (RCCL) and device buffers only.

Objects on the JSON line besides the contract fields:
  roofline          the WHOLE scale-space + DoG pass (round 6): every scale-space launch of EVERY octave (fused up-sampling + seed blur, the
                    two-scale launches, the 9 / 11 / 13-tap launches, the per-scale launches over the coarse octaves) — the interval
                    pyramid_all_ms, first launch of octave 0 to the last blur launch of the coarsest octave — plus the streaming extrema
                    scan (k_extrema_lean, ONE launch over all octaves: it forms the DoG values); durations from HIP events recorded on the
                    streams the kernels run on, inside the timed region. `frac` is what the hardware moved: HBM bytes of those launches
                    from the rocprofv3 --pmc capture of the same kernel sources (profiles/*pmc_traffic*.json: `all_octaves`; refused for any
                    other source hash) / those durations / 8 TB/s, with `all_launches` rows (bytes, us, fraction per launch shape) from that
                    capture. `algorithmic` = this build's own minimum (41.25 B per octave-0 pixel, 37 B per pixel of the coarser octaves,
                    24 B per scanned pixel; <= traffic by construction). `octave0_and_scan` = rounds 1-5's definition (octave 0's five
                    launches + the scan) with its own `frac`, `algorithmic` and `per_launch` rows; `survey_8d` = SURVEY.md 8(d)'s pricing of
                    the reference's schedule (72.25 + 20 B/px), kept for comparison with rounds 1-3 only.
  roofline_c3       the same for BASELINE config 3 (64 x 1920x1080, detect only), 5 steps
  plain_api         the 20 reference entry points ONLY (C caller, instance from vksift_createInstance): detect into N buffers then read them,
                    the same with two buffer sets, the two-buffer ping-pong — what deferred submission gives a caller of the reference
  value_host_input  the reference's own measurement protocol on the same frames (src/perf/wrappers/vulkansift_wrapper.cpp:
                    30-33): host images in, vksift_getFeaturesNumber + vksift_downloadFeatures (+ matches) out, strictly serial
  value_host_input_pipelined  the same inputs and outputs with two buffer sets: the next batch's detection is queued before
                    the current batch's results are fetched (the asynchronous API as vulkansift.h:43-47 intends)
  single_image_ms   BASELINE config 2 literally: ONE 640x480 image through plain vksift_detectFeatures (+ matchFeatures),
                    10 warm-up + 100 timed runs as src/perf/perf_runtime.cpp:63-81
  config5           BASELINE config 5 at --gpus N: per rank 64 x 1920x1080 (up-sampling on) detect + the 32 consecutive pairs matched in
                    both directions, weak-scaled, resident and host inputs, per-rank CRC-32 of fixed frames (must not depend on N)
  cpu_baseline      the CPU oracle (scalar C port of the same algorithm) rebuilt -O3 -march=native on the box, one frame per
                    host core on all host cores, rank 0, N=1
"""
import argparse
import ctypes as C
import glob
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import threading
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
INT8_PEAK_TOPS = 3944.0  # dense int8 MFMA peak used for the matcher


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=512, help="frames per batched detection call")
    ap.add_argument("--sub-batches", type=int, default=3, help="batched detection calls per step (frames per step = batch x sub-batches)")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--no-match", action="store_true", help="detect only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip value_host_input / single_image_ms / roofline_c3 / sharded match")
    ap.add_argument("--host-input", action="store_true", help="time the PCIe-inclusive protocol as the main loop (not the headline value)")
    ap.add_argument("--serialize-match", action="store_true",
                    help="the next detection's scale-space starts BEHIND the matching queued before it instead of beside it "
                         "(vksift_hip_tune VKSIFT_TUNE_PYR_GATE: stage times become kernel times; not the headline schedule)")
    ap.add_argument("--tune", default="", help="development A/B: 'knob=value,...' passed to vksift_hip_tune (include/vksift_hip.h) before anything runs; "
                                               "the line records it in config.tune — not the shipped configuration")
    ap.add_argument("--fp16", action="store_true", help="VKSIFT_PYRAMID_PRECISION_FLOAT16: binary16 scale-space storage (not the headline configuration)")
    ap.add_argument("--extras-timeout", type=float, default=300.0, help="seconds the legs after the timed region may take before the headline line is printed without them")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (RCCL) even at world size 1: exercises every collective branch of the multi-GPU path on one GPU")
    ap.add_argument("--match-rows", type=int, default=50000, help="rows of A and of B in the sharded 2-NN leg (BASELINE config 4)")
    ap.add_argument("--dry-launch", action="store_true", help="bring the N ranks up (gloo when there is no GPU), print who came up, run nothing: checks the launch path")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no torch.distributed environment: become the launcher. Re-executes this script under
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>` (one rank per GPU, the
    same command line the driver uses) and returns its exit code. Refuses, loudly and with a non-zero status, when the box has fewer
    than N GPUs: a line that says n_gpus 1 for --gpus 8 would be a flat fake scaling curve."""
    import socket

    if not args.dry_launch:
        import torch

        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} asked for, but this box has {have} visible GPU(s): refusing to run "
                             f"(one rank per GPU; nothing is reported for a world size that did not run)\n")
            return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    sys.stderr.write("bench.py: launching " + " ".join(cmd[1:]) + "\n")
    return subprocess.call(cmd, env=env)


def dry_launch(args, rank, local_rank, world):
    """--dry-launch: every rank joins a process group (RCCL when each rank has its GPU, gloo otherwise), the ranks' LOCAL_RANKs are
    gathered, rank 0 prints one JSON line. Nothing of the benchmark runs."""
    import torch
    import torch.distributed as dist

    use_gpu = torch.cuda.is_available() and torch.cuda.device_count() >= world
    if use_gpu:
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        t = torch.tensor([local_rank], dtype=torch.int64, device=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend="gloo")
        t = torch.tensor([local_rank], dtype=torch.int64)
    got = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(got, t)
    # with a GPU per rank: the library's OWN RCCL communicator comes up too (vksift_ext_shardGroupCreate: ncclCommInitRank on every rank)
    # and says how many ranks it sees (ncclCommCount / ncclCommUserRank through vksift_ext_shardGroupInfo); without GPUs: null
    rccl = None
    if use_gpu:
        try:
            from vulkansift_amd import multigpu
            grp = multigpu.ShardGroup(local_rank, world, rank, dist if world > 1 else None)
            info = grp.info()
            grp.close()
            r = torch.tensor([info["rccl_ranks"], info["rccl_rank"]], dtype=torch.int64, device=t.device)
            allr = [torch.zeros_like(r) for _ in range(world)]
            dist.all_gather(allr, r)
            rccl = {"rccl_ranks": [int(x[0].item()) for x in allr], "rccl_rank": [int(x[1].item()) for x in allr]}
        except Exception as e:  # noqa: BLE001
            rccl = {"error": repr(e)[:200]}
    if rank == 0:
        print(json.dumps({"dry_launch": True, "n_gpus": world, "backend": "nccl" if use_gpu else "gloo",
                          "local_ranks": [int(x.item()) for x in got], "gpus_visible": torch.cuda.device_count() if torch.cuda.is_available() else 0,
                          "rccl_ranks": rccl}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0


# ---------------------------------------------------------------------------------------------------------------------
def kernel_source_sha():
    h = hashlib.sha256()
    for p in sorted(glob.glob(os.path.join(ROOT, "vulkansift_amd", "csrc", "hip", "*.hip"))):
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(w, h, batch, fp16=False):
    """HBM bytes per step of the octave-0 blur + scan launches from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE;
    gfx950 FETCH_SIZE correction applied, see profiles/README.md). A file measured on other kernel sources — or in the other
    pyramid precision mode: binary16 planes move half the bytes — is refused."""
    sha = kernel_source_sha()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json")), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if (d.get("width"), d.get("height"), d.get("batch")) == (w, h, batch) and d.get("kernel_source_sha") == sha \
                and bool(d.get("fp16", False)) == bool(fp16):
            d["_path"] = os.path.join("profiles", os.path.basename(path))
            return d
    return None


def own_pyramid_bytes(octs, src_px, texel=4):
    """This build's algorithmic minimum for the scale-space of ONE image, all octaves (S = 3): octave 0 reads the u8 source once and
    writes its 6 planes, every later octave's plane 0 is stored by the scale-S launch of the octave in front (4 B per texel of the
    smaller octave) and it writes 5 planes; 4 plane reads per octave (the two-scale launch reads its source once for two scales)."""
    b = float(src_px)
    for o, (w, h) in enumerate(octs):
        px = float(w) * h
        b += texel * px * ((6 if o == 0 else 5) + 4)
        if o > 0:
            b += texel * px                       # its seed, stored by the previous octave's scale-S launch
    return b


def whole_pass(acc, pmc, octs, src_px, batch, texel=4):
    """Every octave's scale-space launches (first launch of octave 0 to the last blur launch of the coarsest octave: pyramid_all_ms, HIP
    events on the stream they run on) + the extrema scan. Counter bytes when the capture holds all launches, else algorithmic."""
    calls = max(acc["nb_calls"], 1)
    t = (acc.get("pyramid_all_ms", 0.0) + acc["scan_ms"]) * 1e-3
    if t <= 0 or not octs:
        return None
    alg = calls * batch * (own_pyramid_bytes(octs, src_px, texel) + sum(float(w) * h for w, h in octs) * 6 * texel)
    out = {"interval_ms_per_call": t / calls * 1e3, "pyramid_all_ms_per_call": acc.get("pyramid_all_ms", 0.0) / calls, "scan_ms_per_call": acc["scan_ms"] / calls,
           "launches_per_call": acc.get("nb_blur_launches_all", 0) / calls + 1.0,
           "algorithmic": {"bytes_per_call": alg / calls, "achieved": alg / t / 1e9, "frac": alg / t / 1e9 / HBM_PEAK_GBPS}}
    allo = (pmc or {}).get("all_octaves")
    if allo:
        phys = allo["hbm_bytes_per_call"] * calls / t / 1e9
        out.update({"traffic_per_call": allo["hbm_bytes_per_call"], "achieved": phys, "frac": phys / HBM_PEAK_GBPS,
                    "traffic_over_algorithmic": allo["hbm_bytes_per_call"] * calls / alg, "launches": allo.get("launches")})
    return out


def roofline_from(acc, pmc, label, octs=None, src_px=0, batch=0, texel=4):
    """The scale-space + DoG pass of octave 0 (blur launches) and the extrema scan over all octaves (one launch), HIP-event durations
    from inside the timed region.
      frac / achieved  what the HARDWARE moved: HBM bytes of these launches from the rocprofv3 --pmc capture of the same kernel
                       sources (profiles/*pmc_traffic*.json) / their measured duration / 8 TB/s. Without a matching capture they fall
                       back to this build's own algorithmic minimum (below) — never to the reference-schedule pricing.
      algorithmic      this build's minimum for the same result: octave 0: 1 B read (u8) + 6 planes written + 5 plane reads (4 with the
                       two-scale launch) + 1 B/px-of-octave-1 seed = 41.25 B per octave-0 pixel; scan: the 6 Gaussian planes read once =
                       24 B per pixel of every octave. By construction <= traffic (halo re-reads are the difference).
      survey_8d        SURVEY.md 8(d)'s pricing of the REFERENCE's schedule (72.25 B per octave-0 pixel incl. 20 B/px of DoG writes + 20
                       B/px DoG reads in the scan): kept for comparison with earlier rounds; this build performs neither, so the
                       figure overstates what the hardware does and is not `frac`."""
    calls = max(acc["nb_calls"], 1)
    launches = max(acc["nb_blur_launches"], 1)
    pyr_s = acc["pyramid_ms"] * 1e-3
    scan_s = acc["scan_ms"] * 1e-3
    t = pyr_s + scan_s
    alg_pyr = float(acc["pyramid_algorithmic_bytes"])                   # 72.25 B per octave-0 pixel, whole run
    alg_scan = float(acc["scan_algorithmic_bytes"])                    # + 20 B per pixel of every octave for the extrema scan (one launch)
    n_launch = launches + calls                                        # blur launches + one scan per detection call
    own_pyr = alg_pyr * (41.25 / 72.25)                                 # this build's minimum, same pixel counts
    own_scan = alg_scan * (24.0 / 20.0)
    own = (own_pyr + own_scan) / t / 1e9 if t > 0 else 0.0
    survey = (alg_pyr + alg_scan) / t / 1e9 if t > 0 else 0.0
    out = {
        "bound": "hbm",
        "kernel": label,
        "achieved": own,
        "peak": HBM_PEAK_GBPS,
        "unit": "GB/s",
        "frac": own / HBM_PEAK_GBPS,
        "basis": "algorithmic minimum of this build (no PMC capture of these kernel sources)",
        "traffic": None,
        "avg_launch_us": t / n_launch * 1e6,
        "launches_per_call": n_launch / calls,
        "algorithmic": {"bytes_per_launch": (own_pyr + own_scan) / n_launch, "achieved": own, "frac": own / HBM_PEAK_GBPS,
                        "note": "41.25 B per octave-0 pixel (blur launches) + 24 B per pixel of every octave (scan): what this build must move"},
        "survey_8d": {"bytes_per_launch": (alg_pyr + alg_scan) / n_launch, "achieved": survey, "frac": survey / HBM_PEAK_GBPS,
                      "note": "reference-schedule pricing (72.25 + 20 B/px): counts DoG writes / reads this build does not perform; not the headline fraction"},
        "blur_launches": {"avg_us": pyr_s / launches * 1e6, "per_call": launches / calls},
        "scan_launch": {"avg_us": scan_s / calls * 1e6},
    }
    if pmc is not None:
        per_call = pmc["hbm_bytes_per_call"]
        phys = per_call * calls / t / 1e9 if t > 0 else 0.0
        out["traffic"] = per_call / (n_launch / calls)                 # HBM bytes per launch (PMC)
        out["achieved"] = phys
        out["frac"] = phys / HBM_PEAK_GBPS
        out["basis"] = "HBM bytes from rocprofv3 --pmc (FETCH_SIZE x 2 + WRITE_SIZE) of the same kernel sources / HIP-event duration in this run"
        out["traffic_source"] = pmc.get("_path", "profiles/")
        out["traffic_over_algorithmic"] = per_call * calls / (own_pyr + own_scan)
        if pmc.get("per_launch"):
            out["per_launch"] = pmc["per_launch"]                       # bytes, us and fraction of each launch kind, from the capture itself
    # Round 6: the HEADLINE fraction prices the WHOLE pass — every octave's blur launches inside the interval, not
    # octave 0's alone. The octave-0 figures above move to `octave0_and_scan` (rounds 1-5's definition, kept for comparison).
    wp = whole_pass(acc, pmc, octs, src_px, batch, texel) if octs else None
    if wp is not None:
        old = {k: out[k] for k in ("achieved", "frac", "basis", "traffic", "avg_launch_us", "launches_per_call", "algorithmic") if k in out}
        if "traffic_over_algorithmic" in out:
            old["traffic_over_algorithmic"] = out.pop("traffic_over_algorithmic")
        out["octave0_and_scan"] = dict(old, kernel=label, note="rounds 1-5's `frac`: octave 0's blur launches + the scan; the coarse octaves' launches outside the interval")
        out["kernel"] = "every scale-space launch of a detection call, all octaves (seed, two-scale launches, 9 / 11 / 13 taps per octave), + k_extrema_lean over all octaves"
        n_l = max(wp["launches_per_call"], 1.0)
        out["avg_launch_us"] = wp["interval_ms_per_call"] * 1e3 / n_l
        out["launches_per_call"] = n_l
        out["algorithmic"] = dict(wp["algorithmic"], bytes_per_launch=wp["algorithmic"]["bytes_per_call"] / n_l,
                                  note="octave 0: 41.25 B/px, later octaves 37 B/px (36 for the last), + 24 B per pixel of every octave for the scan: what this build must move")
        if "frac" in wp:
            out.update({"achieved": wp["achieved"], "frac": wp["frac"], "traffic": wp["traffic_per_call"] / n_l, "traffic_over_algorithmic": wp["traffic_over_algorithmic"],
                        "basis": "HBM bytes from rocprofv3 --pmc (FETCH_SIZE x 2 + WRITE_SIZE) of EVERY scale-space launch + the scan, same kernel sources / HIP-event durations in this run",
                        "all_launches": wp.get("launches")})
        else:
            out.update({"achieved": wp["algorithmic"]["achieved"], "frac": wp["algorithmic"]["frac"], "traffic": None,
                        "basis": "algorithmic minimum of this build over all octaves (no PMC capture of these kernel sources that holds every launch)"})
        out["whole_pass_ms_per_call"] = {"pyramid_all_ms": wp["pyramid_all_ms_per_call"], "scan_ms": wp["scan_ms_per_call"]}
    return out


# ---------------------------------------------------------------------------------------------------------------------
def usable_cores():
    """host cores this process may really use: the affinity mask, capped by the cgroup CPU quota (cpu.max) — os.cpu_count() reports
    the machine, not the container"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, int(quota / period + 0.5)))
            break
        except Exception:
            continue
    return max(1, n)


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def opencv_probe():
    """the reference's CPU comparison is cv::SIFT::create() defaults + BFMatcher knnMatch(k=2)
    (src/perf/wrappers/opencv_wrapper.cpp:5,16, src/perf/perf_common.cpp:113-119). Timed only if OpenCV is importable on this box;
    otherwise said so — never fabricated."""
    try:
        import cv2  # noqa: F401
    except Exception as e:  # noqa: BLE001
        return {"available": False, "reason": f"import cv2 failed: {type(e).__name__}", "pkg_config": bool(subprocess.run(
            "pkg-config --exists opencv4", shell=True, capture_output=True).returncode == 0)}
    return {"available": True, "version": cv2.__version__}


def opencv_baseline(frames, do_match, runs=5):
    import cv2

    sift = cv2.SIFT_create(0, 3, 0.04, 10, 1.6)
    bf = cv2.BFMatcher(cv2.NORM_L2)
    sift.detectAndCompute(frames[0], None)
    t0 = time.perf_counter()
    n = 0
    for i in range(runs):
        kp, des = sift.detectAndCompute(frames[i % len(frames)], None)
        if do_match and des is not None and len(des) >= 2:
            bf.knnMatch(des, des, k=2)
        n += len(kp)
    dt = time.perf_counter() - t0
    return {"value": runs / dt, "unit": "frames/s", "threads": cv2.getNumThreads(), "features_per_frame": n / runs,
            "what": "cv::SIFT::create(0,3,0.04,10,1.6) detectAndCompute" + (" + BFMatcher(NORM_L2).knnMatch(k=2) self-match" if do_match else "")}


def _cpu_worker(job):
    """one PROCESS per core (own heap: the oracle allocates its pyramid per call, and threads of one process serialised in malloc/
    page-fault handling — round 2 measured 9.6x on 256 threads)"""
    so, frames, do_match = job
    from oracle import oracle as O

    if so:
        O.use_library(so)
    cfg = O.default_config(math_mode=0)
    n = 0
    for img in frames:
        feats, _ = O.detect(cfg, img)
        if do_match and len(feats) >= 2:
            O.match_2nn(feats, feats)
        n += len(feats)
    return n


def _opencv_leg(frames, do_match):
    probe = opencv_probe()
    if not probe["available"]:
        return dict(probe, note="OpenCV unavailable on this box: the reference's OpenCV-SIFT comparison was not run")
    try:
        return dict(probe, **opencv_baseline(frames, do_match))
    except Exception as e:  # noqa: BLE001
        return dict(probe, error=repr(e)[:200])


def cpu_baseline(frames, do_match, per_worker=2, hd_frame=False):
    """The oracle rebuilt on this box with -O3 -march=native (its fp32 operation order stays pinned: -ffp-contract=off), one worker
    PROCESS per usable host core, `per_worker` frames each. hd_frame: also one 1920x1080 frame per core, detect only (BASELINE config
    3's frames: north_star asks for both sizes beside a CPU figure)."""
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    from oracle import oracle as O

    tmp = tempfile.mkdtemp(prefix="vksift_oracle_")
    so = os.path.join(tmp, "liboracle_native.so")
    flags = ["-O3", "-march=native", "-std=c99", "-fPIC", "-ffp-contract=off", "-fno-fast-math"]
    try:
        subprocess.run(["gcc"] + flags + ["-shared", "-o", so, os.path.join(ROOT, "oracle", "sift_oracle.c"), "-lm"], check=True, capture_output=True)
        O.use_library(so)
        build = "gcc " + " ".join(flags)
    except Exception:
        so = None
        build = "committed oracle/Makefile flags (-O2 -mavx2 -mfma): the -march=native rebuild failed on this box"

    t0 = time.perf_counter()
    nf1 = _cpu_worker((so, [frames[0]], do_match))
    t_single = time.perf_counter() - t0
    cores = usable_cores()
    jobs = [(so, [frames[(w * per_worker + k) % len(frames)] for k in range(per_worker)], do_match) for w in range(cores)]
    with ProcessPoolExecutor(max_workers=cores, mp_context=mp.get_context("spawn")) as ex:
        list(ex.map(_cpu_worker, [(so, [], do_match)] * cores))            # start the workers and load the library outside the timing
        t0 = time.perf_counter()
        nfeat = list(ex.map(_cpu_worker, jobs))
        dt = time.perf_counter() - t0
        hd = None
        if hd_frame:
            from vulkansift_amd import api
            hd_imgs = [api.gen_synthetic_image(0x5EED0000 + i, 1920, 1080) for i in range(min(cores, 4))]
            t0 = time.perf_counter()
            nf_hd = list(ex.map(_cpu_worker, [(so, [hd_imgs[w % len(hd_imgs)]], False) for w in range(cores)]))
            dt_hd = time.perf_counter() - t0
            hd = {"value": cores / dt_hd, "unit": "frames/s", "cores": cores, "per_core_value": 1.0 / dt_hd,
                  "sample": f"{cores} frames of 1920x1080 (BASELINE config 3's frames, 2x up-sampling, 7 octaves), one per worker process, detect only, "
                            f"{int(sum(nf_hd) / cores)} features/frame, {dt_hd:.1f} s wall"}
    total = cores * per_worker
    return {
        "value": total / dt,
        "unit": "frames/s",
        "cores": cores,
        "kind": "port",
        "single_thread_value": 1.0 / t_single,
        "scaling_vs_one_core": (total / dt) * t_single,
        "machine_logical_cpus": os.cpu_count(),
        "cpu_model": cpu_model(),
        "opencv": _opencv_leg(frames, do_match),
        **({"frames_1920x1080": hd} if hd else {}),
        "build": build,
        "sample": f"{total} frames of the benchmark's {frames[0].shape[1]}x{frames[0].shape[0]} workload ({per_worker} per worker process, one process per usable core), detect"
                  + ("+self-match" if do_match else "") + f", {int(sum(nfeat) / total)} features/frame, {dt:.1f} s wall ({dt * cores:.0f} core-seconds); "
                  + f"one frame alone on one core {t_single:.2f} s ({nf1} features)",
    }


# ---------------------------------------------------------------------------------------------------------------------
_PROTO = None


def protocol_client(api):
    """tests/native/protocol_client.c built with gcc against the library (public API only): the host-protocol loops as a C caller
    runs them. None when gcc is unavailable (the Python loops below are the fall-back, 4-8 us of interpreter time per frame slower)."""
    global _PROTO
    if _PROTO is None:
        try:
            libdir = os.path.dirname(api.LIB_PATH)
            so = os.path.join(tempfile.mkdtemp(prefix="vksift_proto_"), "libproto.so")
            subprocess.run(["gcc", "-O2", "-std=gnu11", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"), "-o", so,
                            os.path.join(ROOT, "tests", "native", "protocol_client.c"), "-L" + libdir, "-lvulkansift", "-Wl,-rpath," + libdir],
                           check=True, capture_output=True)
            L = C.CDLL(so)
            for f in (L.proto_serial, L.proto_pipelined):
                f.restype = C.c_double
                f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]
            L.proto_plain.restype = C.c_double
            L.proto_plain.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]
            L.proto_single.restype = C.c_double
            L.proto_single.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
            _PROTO = L
        except Exception:  # noqa: BLE001
            _PROTO = False
    return _PROTO or None


def reference_protocol(api, inst, frames, W, H, B, do_match, steps):
    """src/perf/wrappers/vulkansift_wrapper.cpp:30-33 per frame = detect(host image) + getFeaturesNumber + downloadFeatures; here
    per batch of B frames, followed by the self-match and the download of its records. Strictly serial: nothing is queued while
    the host waits or copies. Returns frames/s."""
    lib = api.lib()
    cap = inst.cfg.max_nb_sift_per_buffer
    feat_buf = np.zeros(cap, api.FEATURE_DTYPE)
    match_buf = np.zeros(cap, api.MATCH_DTYPE)
    ids = list(range(B))
    ptrs = inst.imagePointerArray(frames)   # what a C caller passes: the marshalling of 128 numpy arrays is not part of the protocol
    pc = protocol_client(api)
    if pc is not None:
        dt = pc.proto_serial(inst._h, ptrs, B, W, H, int(do_match), steps, feat_buf.ctypes.data, match_buf.ctypes.data)
        return B * steps / dt

    def step():
        inst.detectFeaturesBatchPtrs(ptrs, B, W, H, 0)
        for i in range(B):
            n = lib.vksift_getFeaturesNumber(inst._h, i)
            if n:
                lib.vksift_downloadFeatures(inst._h, feat_buf.ctypes.data, i)
        if do_match:
            inst.matchFeaturesBatch(ids, ids)
            for k in range(B):
                if lib.vksift_ext_getMatchesNumberBatch(inst._h, k):
                    lib.vksift_ext_downloadMatchesBatch(inst._h, k, match_buf.ctypes.data)

    step()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = time.perf_counter() - t0
    return B * steps / dt


def pipelined_protocol(api, dev_index, frames, W, H, B, do_match, steps):
    """The same inputs and outputs per frame as reference_protocol (host images in; counts, features and match records out), with
    the asynchronous API used the way include/vulkansift/vulkansift.h:43-47 intends: two sets of B SIFT buffers, the detection of
    the next batch is queued BEFORE the results of the current one are fetched, so the host's staging and result copies run beside
    the GPU work instead of in front of and behind it. Accessors wait for the detection that filled the buffer they read, not for
    the one queued after it. Returns frames/s."""
    lib = api.lib()
    cfg = api.default_config(sift_buffer_count=2 * B, gpu_device_index=dev_index, input_image_max_size=max(W * H, 1024))
    feat_buf = np.zeros(cfg.max_nb_sift_per_buffer, api.FEATURE_DTYPE)
    match_buf = np.zeros(cfg.max_nb_sift_per_buffer, api.MATCH_DTYPE)
    with api.Instance(cfg, batch_capacity=B) as inst:
        ids = [list(range(B)), list(range(B, 2 * B))]
        ptrs = inst.imagePointerArray(frames)
        pc = protocol_client(api)
        if pc is not None:
            dt = pc.proto_pipelined(inst._h, ptrs, B, W, H, int(do_match), steps, feat_buf.ctypes.data, match_buf.ctypes.data)
            return B * steps / dt

        def collect(s):
            for i in ids[s]:
                if lib.vksift_getFeaturesNumber(inst._h, i):
                    lib.vksift_downloadFeatures(inst._h, feat_buf.ctypes.data, i)
            if do_match:
                for k in range(B):
                    if lib.vksift_ext_getMatchesNumberBatch(inst._h, k):
                        lib.vksift_ext_downloadMatchesBatch(inst._h, k, match_buf.ctypes.data)

        def run(n):
            inst.detectFeaturesBatchPtrs(ptrs, B, W, H, 0)
            if do_match:
                inst.matchFeaturesBatch(ids[0], ids[0])
            for it in range(n):
                cur, nxt = it & 1, (it & 1) ^ 1
                if it + 1 < n:
                    inst.detectFeaturesBatchPtrs(ptrs, B, W, H, nxt * B)   # queued behind the matching of `cur`; staged while the GPU works
                collect(cur)
                if it + 1 < n and do_match:
                    inst.matchFeaturesBatch(ids[nxt], ids[nxt])    # the match slots are free again once `cur`'s records are out

        run(2)
        t0 = time.perf_counter()
        run(steps)
        dt = time.perf_counter() - t0
    return B * steps / dt


def plain_api_protocol(api, dev_index, frames, W, H, do_match, budget_s=6.0):
    """The reference's 20 entry points and nothing else (tests/native/protocol_client.c: proto_plain; an instance from
    vksift_createInstance): one host image per vksift_detectFeatures call (vulkansift.c:315-344), count + features read per buffer.
    What the library does with such a caller is its own business — runs of detect calls are staged and launched as one batched
    detection (deferred submission, include/vksift_ext.h) — so this is the rate an application written against the reference gets
    without touching an extension. frames/s per calling pattern:
      detect_n_then_read[N]   a run of N detect calls into N buffers, then the N buffers are read
      two_sets[N]             the same with 2 N buffers: the next run is issued before the current set is read (vulkansift.h:43-47)
      ping_pong               two buffers, detect(frame k + 1) then read(frame k)
    with_match: the two_sets[64] pattern with every frame also self-matched by vksift_matchFeatures / vksift_downloadMatches (one
    pair per call: the reference's matching interface has a single result slot, so there is nothing to batch)."""
    pc = protocol_client(api)
    if pc is None:
        return {"available": False, "reason": "gcc unavailable: tests/native/protocol_client.c could not be built"}
    out = {"caller": "C (tests/native/protocol_client.c: proto_plain), reference entry points only, instance from vksift_createInstance",
           "detect_n_then_read": {}, "two_sets": {}}
    per_leg = budget_s / 8.0

    def leg(nbuf, n, mode, match):
        cfg = api.default_config(sift_buffer_count=nbuf, gpu_device_index=dev_index, input_image_max_size=max(W * H, 1024))
        feat_buf = np.zeros(cfg.max_nb_sift_per_buffer, api.FEATURE_DTYPE)
        match_buf = np.zeros(cfg.max_nb_sift_per_buffer, api.MATCH_DTYPE)
        with api.Instance(cfg) as inst:
            ptrs = inst.imagePointerArray(frames)
            # a first short call sizes the timed one for the leg's share of the budget
            frames_per_iter = 1 if mode == 2 else n
            probe = 12 if mode == 2 else 3
            dt = pc.proto_plain(inst._h, ptrs, len(frames), n, W, H, mode, int(match), probe, feat_buf.ctypes.data, match_buf.ctypes.data)
            steps = int(max(probe, min(2000 if mode == 2 else 400, per_leg / max(dt / probe, 1e-6))))
            dt = pc.proto_plain(inst._h, ptrs, len(frames), n, W, H, mode, int(match), steps, feat_buf.ctypes.data, match_buf.ctypes.data)
            stats = inst.getDeferredStats()
        return steps * frames_per_iter / dt, stats

    for n in (2, 8, 64):
        out["detect_n_then_read"][str(n)], _ = leg(n, n, 0, False)
        out["two_sets"][str(n)], st = leg(2 * n, n, 1, False)
    out["mean_images_per_launch_two_sets_64"] = round(st[1] / max(st[0], 1), 1)
    out["ping_pong"], st = leg(2, 1, 2, False)
    out["ping_pong_deferred_images"] = st[1]
    if do_match:
        out["two_sets_64_with_match"], _ = leg(128, 64, 1, True)
    return out


def single_image_latency(api, dev_index, img, runs=100, warm=10, detect_only=False):
    """perf_runtime.cpp:63-81: warm-up, then the mean over `runs` of detect + count + download of ONE host image (and of the
    same followed by matchFeatures(0, 0) + downloadMatches = BASELINE config 2)."""
    lib = api.lib()
    h, w = img.shape
    cfg = api.default_config(gpu_device_index=dev_index, input_image_max_size=max(w * h, 1024))
    out = {}
    with api.Instance(cfg) as inst:
        feat_buf = np.zeros(cfg.max_nb_sift_per_buffer, api.FEATURE_DTYPE)
        match_buf = np.zeros(cfg.max_nb_sift_per_buffer, api.MATCH_DTYPE)
        pc = protocol_client(api)
        img = np.ascontiguousarray(img)
        for with_match in ((False,) if detect_only else (False, True)):
            key = "detect_match_ms" if with_match else "detect_ms"
            ts = []
            for i in range(warm + runs):
                t0 = time.perf_counter()
                inst.detectFeatures(img, 0)
                n = lib.vksift_getFeaturesNumber(inst._h, 0)
                lib.vksift_downloadFeatures(inst._h, feat_buf.ctypes.data, 0)
                if with_match:
                    lib.vksift_matchFeatures(inst._h, 0, 0)
                    lib.vksift_downloadMatches(inst._h, match_buf.ctypes.data)
                if i >= warm:
                    ts.append(time.perf_counter() - t0)
            out[key + "_python_caller"] = float(np.mean(ts) * 1e3)
            out[key] = out[key + "_python_caller"]
            out["features"] = int(n)
            if pc is not None:
                nf = C.c_uint32(0)
                out[key] = 1e3 * pc.proto_single(inst._h, img.ctypes.data, w, h, int(with_match), warm, runs, feat_buf.ctypes.data, match_buf.ctypes.data, C.byref(nf))
                out["features"] = int(nf.value)
    out["caller"] = "C (tests/native/protocol_client.c: proto_single)" if pc is not None else "python (ctypes)"
    out["protocol"] = f"{warm} warm-up + {runs} timed runs, host image in, count + features (+ matches) downloaded, plain vksift_detectFeatures/matchFeatures"
    return out


def c3_roofline(api, torch, dev, steps=5):
    """BASELINE config 3: 64 x 1920x1080, detect only, inputs resident in HBM."""
    W, H, B = 1920, 1080, 64
    frames = np.stack([api.gen_synthetic_image(0x5EED0000 + i, W, H) for i in range(8)])
    d_frames = torch.from_numpy(np.concatenate([frames] * (B // 8))).to(dev)      # 8 distinct frames, repeated: generation is host time
    cfg = api.default_config(sift_buffer_count=B, gpu_device_index=dev.index, input_image_max_size=W * H)
    inst = api.Instance(cfg, batch_capacity=B)
    inst.detectFeaturesBatchDevice(d_frames.data_ptr(), B, W, H, 0)
    torch.cuda.synchronize()
    nfeat = float(np.mean([inst.getFeaturesNumber(i) for i in range(0, B, 8)]))
    inst.setProfiling(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        inst.detectFeaturesBatchDevice(d_frames.data_ptr(), B, W, H, 0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    acc = inst.getAccumulatedDetectTimings()
    octs = [inst.getScaleSpaceOctaveResolution(o) for o in range(inst.getScaleSpaceNbOctaves())]
    inst.close()
    r = roofline_from(acc, pmc_traffic(W, H, B), "the 5 blur launches of octave 0 (3840x2160 planes: seed, two-scale launch, 9 / 11 / 13 taps) + k_extrema_lean over all octaves, 64 x 1920x1080 frames",
                      octs, W * H, B)
    r.update({"workload": "BASELINE config 3: 64 x 1920x1080 uint8 frames, detect only, default vksift_Config, inputs resident in HBM",
              "steps": steps, "frames_per_s": B * steps / dt, "ms_per_step": dt / steps * 1e3, "mean_features_per_frame": nfeat,
              "stage_ms_per_step": {k: acc[k] / max(acc["nb_calls"], 1) for k in ("pyramid_ms", "pyramid_all_ms", "extrema_ms", "scan_ms", "orientation_ms", "descriptor_ms", "total_ms")}})
    return r


def c5_leg(api, torch, dist, dev, rank, world, steps=3, distinct=16, use_dist=None):
    """BASELINE config 5 (512 x 1920x1080, up-sampling on, detect + match of the consecutive pairs (2i, 2i+1) in both directions as
    src/examples/test_sift_match.cpp:67-80 does, split over 8 GPUs = 64 frames per GPU): every rank runs one GPU's share on ITS
    frames (seeds 0x5EED0000 + rank*64 + i), weak scaling, no collective in the data path. Timed like the headline (barrier +
    synchronize on both sides, maximum over ranks), resident inputs and host inputs. For comparing N = 1, 2, 4, 8 bit for bit:
    a CRC-32 per rank over the feature bytes of four fixed frames and the records of pair 0 — rank r's value must not depend on N."""
    use_dist = world > 1 if use_dist is None else use_dist
    W, H, B = 1920, 1080, 64
    # `distinct` different frames per rank, cycled (host-side generation costs 0.6 s per 1080p frame): consecutive frames differ
    base = [api.gen_synthetic_image(0x5EED0000 + rank * B + i, W, H) for i in range(distinct)]
    frames = [base[i % distinct] for i in range(B)]
    d_frames = torch.from_numpy(np.stack(frames)).to(dev)
    cfg = api.default_config(sift_buffer_count=B, gpu_device_index=dev.index, input_image_max_size=W * H)
    even, odd = list(range(0, B, 2)), list(range(1, B, 2))
    out = {}
    with api.Instance(cfg, batch_capacity=B) as inst:
        ptrs = inst.imagePointerArray(frames)

        def step(host):
            if host:
                inst.detectFeaturesBatchPtrs(ptrs, B, W, H, 0)
            else:
                inst.detectFeaturesBatchDevice(d_frames.data_ptr(), B, W, H, 0)
            inst.matchFeaturesBatch(even, odd)
            inst.matchFeaturesBatch(odd, even)

        for host in (False, True):
            step(host)
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step(host)
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            dt = time.perf_counter() - t0
            if use_dist:
                t = torch.tensor([dt], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            out["frames_per_s_host_input" if host else "frames_per_s"] = world * B * steps / dt
            out["ms_per_step_host_input" if host else "ms_per_step"] = dt / steps * 1e3
        crc = 0
        nfeat = []
        for i in (0, 5, 10, 15):
            f = inst.downloadFeatures(i)
            nfeat.append(len(f))
            crc = zlib.crc32(f.tobytes(), crc)
        crc = zlib.crc32(inst.downloadMatchesBatch(0).tobytes(), crc) & 0xFFFFFFFF      # pair (1, 0) of the last call
    crcs = [crc]
    if use_dist:
        t = torch.zeros(world, dtype=torch.int64, device=dev)
        t[rank] = crc
        dist.all_reduce(t)
        crcs = [int(x) for x in t.tolist()]
    out.update({"workload": f"BASELINE config 5, one GPU's share per rank: {B} x {W}x{H} (up-sampling on, 7 octaves) detect + the {len(even)} consecutive pairs "
                            f"matched in both directions, x{world} ranks (weak scaling: 512 frames at 8 GPUs)",
                "steps": steps, "frames_per_rank": B, "distinct_frames_per_rank": distinct, "mean_features_per_frame": float(np.mean(nfeat)),
                "crc32_by_rank": crcs})
    return out


def sharded_match(api, torch, dist, dev, rank, world, rows):
    """BASELINE config 4 through the C entry vksift_ext_matchSharded: query rows of A sharded over the ranks, the reference set B
    all-gathered once (RCCL, uint8 rows) inside the library, every rank scans all of B. Returns (ms, CRC-32 of all records)."""
    from vulkansift_amd import multigpu

    a = api.gen_synthetic_descriptors(1, rows)
    b = api.gen_synthetic_descriptors(2, rows)
    lo, hi = multigpu.shard_bounds(rows, world, rank)
    blo, bhi = multigpu.shard_bounds(rows, world, rank)
    d_a = torch.from_numpy(a[lo:hi]).to(dev)
    d_b_shard = torch.from_numpy(b[blo:bhi]).to(dev)
    ms, rec = multigpu.sharded_match_timed(d_a, lo, d_b_shard, rows, world, rank, repeats=5)
    rec_all = multigpu.gather_records(rec, rows, world, rank)
    crc = zlib.crc32(rec_all.cpu().numpy().tobytes()) & 0xFFFFFFFF if rank == 0 else None
    return ms, crc


def multigpu_mod():
    from vulkansift_amd import multigpu
    return multigpu


# ---------------------------------------------------------------------------------------------------------------------
def main():
    args = parse_args()
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))                       # this process becomes the launcher of N ranks
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if env_world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world}: the launcher's --nproc-per-node and --gpus must agree")
    if args.dry_launch:
        sys.exit(dry_launch(args, int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), env_world))
    # The contract is ONE JSON line on stdout. Libraries loaded below write banners there through C stdio (RCCL prints its version
    # block on communicator creation): everything written to fd 1 before the result line goes to stderr instead.
    sys.stdout.flush()
    fd_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    use_dist = world > 1 or args.force_dist
    if not torch.cuda.is_available() or torch.cuda.device_count() < (local_rank + 1 if world > 1 else 1):
        sys.stderr.write(f"bench.py: rank {rank} (LOCAL_RANK {local_rank}) has no GPU: {torch.cuda.device_count() if torch.cuda.is_available() else 0} "
                         f"visible, world size {world} needs one per rank\n")
        sys.exit(2)
    if use_dist:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if use_dist else 0)

    from vulkansift_amd import api

    api.lib().vksift_setLogLevel(api.VKSIFT_LOG_ERROR)
    W, H, B, NSUB = args.width, args.height, args.batch, max(1, args.sub_batches)
    do_match = not args.no_match
    if args.serialize_match:
        api.lib().vksift_hip_tune(6, 1)   # VKSIFT_TUNE_PYR_GATE
    for kv in filter(None, args.tune.split(",")):
        api.lib().vksift_hip_tune(int(kv.split("=")[0]), int(kv.split("=")[1]))

    # synthetic frames (seeded per global frame index), uploaded once: inputs are HBM-resident when timing starts.
    # Up to 128 generated frames per rank (85 ms of host time each) and, for longer batches, their three mirror images: B distinct
    # frames; the sub-batches of a step rotate through them with a different first frame.
    ngen = min(B, 128)
    gen = np.stack([api.gen_synthetic_image(0x5EED0000 + rank * 128 + i, W, H) for i in range(ngen)])
    variants = [gen, gen[:, :, ::-1], gen[:, ::-1, :], gen[:, ::-1, ::-1]]
    host = np.ascontiguousarray(np.concatenate([variants[(k // ngen) % 4][: min(ngen, B - k)] for k in range(0, B, ngen)]))
    assert host.shape[0] == B or B > 4 * ngen
    if host.shape[0] < B:                      # more than 512 frames per call: repeat
        host = np.ascontiguousarray(np.concatenate([host] * ((B + host.shape[0] - 1) // host.shape[0]))[:B])
    frames = [host[i] for i in range(B)]
    d_sub = [torch.from_numpy(np.roll(host, -k, axis=0).copy()).to(dev) for k in range(NSUB)]
    torch.cuda.synchronize()

    cfg = api.default_config(sift_buffer_count=B, gpu_device_index=dev.index, input_image_max_size=max(W * H, 1024),
                             pyramid_precision_mode=1 if args.fp16 else 0)
    inst = api.Instance(cfg, batch_capacity=B)

    all_ids = list(range(B))

    def match_all():
        inst.matchFeaturesBatch(all_ids, all_ids)      # 2-NN self-match of every frame (the library runs launches of <= 64 pairs)

    def step():
        for k in range(NSUB):
            if args.host_input:
                inst.detectFeaturesBatch(frames, 0)   # host memcpy into pinned staging + H2D inside the timed region
            else:
                inst.detectFeaturesBatchDevice(d_sub[k].data_ptr(), B, W, H, 0)
            if do_match:
                match_all()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    nfeat = [inst.getFeaturesNumber(i) for i in range(B)]

    inst.setProfiling(True)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0

    rank_elapsed = [elapsed]
    if use_dist:
        # every rank's own time of the same K steps (between the same two barriers): the line carries them all, value uses the maximum
        te = torch.zeros(world, dtype=torch.float64, device=dev)
        te[rank] = elapsed
        dist.all_reduce(te, op=dist.ReduceOp.SUM)
        rank_elapsed = [float(x) for x in te.tolist()]
        elapsed = max(rank_elapsed)

    acc = inst.getAccumulatedDetectTimings()
    placement = inst.getScaleSpacePlacement()
    match_ms = inst.getMatchTime() if do_match else None
    inst.setProfiling(False)

    # The headline object first: the extra legs below run under a watchdog that prints it if one of them stalls (a collective
    # that one rank never enters would otherwise cost the whole line).
    out = None
    if rank == 0:
        frames_total = B * NSUB * world * args.steps
        pmc = pmc_traffic(W, H, B, fp16=args.fp16)
        out = {
            "metric": "SIFT detect+match frames/sec (640x480, ~2k kp)" if do_match else "SIFT detect frames/sec",
            "value": frames_total / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            # frames/s of every rank alone (its frames / its own time between the two barriers); value = all frames / the slowest rank's time
            "per_rank_value": [B * NSUB * args.steps / max(e, 1e-9) for e in rank_elapsed],
            "torch_distributed_world": dist.get_world_size() if use_dist else 1,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 arithmetic, binary16 scale-space storage" if args.fp16 else "f32",
            "data": "synthetic",
            "config": {
                "workload": f"BASELINE config 2: {W}x{H} uint8 frames, detect" + (" + 2-NN self-match" if do_match else "")
                            + ", default vksift_Config (2x up-sampling, auto octaves, 3 scales/octave), inputs resident in HBM",
                "frames_per_step_per_gpu": B * NSUB,
                "frames_per_detection_call": B,
                "octaves": 5 if (W, H) == (640, 480) else None,
                "mean_features_per_frame": float(np.mean(nfeat)),
                "parallelism": f"batch split x{world}, no collectives",
                "input": "host images, upload inside the timed region" if args.host_input else "resident in HBM",
                **({"tune": args.tune + " (development A/B, not the shipped configuration)"} if args.tune else {}),
                "protocols": "value_host_input: the REFERENCE's own protocol and SURVEY.md 8(d)'s definition of the metric "
                             "(src/perf/wrappers/vulkansift_wrapper.cpp:30-33: host image in, count + features + matches downloaded, strictly "
                             "serial) — the figure to compare with the reference's published runtimes. value: batched detection on HBM-resident "
                             "frames, nothing downloaded (kernel-side figure, the bench contract's definition of `value`: inputs resident when the "
                             "timed region starts). value_host_input_pipelined: the same inputs and outputs with the asynchronous API (both legs run "
                             "in C: tests/native/protocol_client.c, public API only). "
                             "single_image_ms: BASELINE config 2 literally (one image per call)",
            },
            "roofline": roofline_from(acc, pmc, "the 5 blur launches of octave 0 (1280x960 planes: k_blur_lean<5,1> seed, k_blur_pair_wide<5,7>, k_blur_wide<9/11/13>): scale-space construction, + k_extrema_lean over all octaves: the scan that forms the DoG values",
                                      [inst.getScaleSpaceOctaveResolution(o) for o in range(inst.getScaleSpaceNbOctaves())], W * H, B, 2 if args.fp16 else 4),
            "stage_ms_per_call": {k: acc[k] / max(acc["nb_calls"], 1) for k in
                                  ("upload_ms", "pyramid_ms", "pyramid_all_ms", "extrema_ms", "scan_ms", "orientation_ms", "descriptor_ms", "total_ms")},
            "last_match_ms": match_ms,
            # where the scale-space buffers were put: rates (GB/s, 8 B per texel) of one whole-batch blur launch on every candidate
            # memory range the instance timed when it allocated them (allocation order), and the one(s) it kept: one per scale-space buffer
            "scale_space_placement": placement,
            # K5 / K6 carry no roofline claim (latency / LDS-atomic / VALU bound, sparse reads): time per 1000 output
            # features, the unit of the reference's own figure (0.38 ms per 1000 features for K5 + K6 on an RTX 2060)
            "keypoint_stages_ms_per_1000_features": {
                k: acc[k + "_ms"] / max(acc["nb_calls"], 1) / max(B * float(np.mean(nfeat)), 1.0) * 1000.0 for k in ("orientation", "descriptor")},
        }

    emitted = threading.Lock()

    def emit(obj):
        if rank != 0 or not emitted.acquire(blocking=False):
            return
        sys.stdout.flush()
        C.CDLL(None).fflush(None)
        os.dup2(fd_stdout, 1)
        print(json.dumps(obj), flush=True)
        os.dup2(2, 1)

    def bail():
        if out is not None:
            emit(dict(out, extras_error=f"an extra leg did not finish within {args.extras_timeout} s; headline only"))
        os._exit(0)

    watchdog = threading.Timer(args.extras_timeout, bail)
    watchdog.daemon = True
    watchdog.start()

    extras = {}
    if not args.no_extras and world == 1:
        # (before the sharded-match leg: once RCCL has created its streams, HIP maps this library's streams onto the hardware queues
        # differently and the per-stage intervals of an overlapped detection get attributed differently — same step time)
        extras["value_host_input"] = reference_protocol(api, inst, frames, W, H, B, do_match, 6)
    inst.close()
    if not args.no_extras and world == 1:
        try:
            extras["value_host_input_pipelined"] = pipelined_protocol(api, dev.index, frames, W, H, B, do_match, 12)
        except Exception as e:  # noqa: BLE001
            extras["value_host_input_pipelined"] = {"error": repr(e)[:300]}
    if not args.no_extras and world == 1:
        try:
            extras["plain_api"] = plain_api_protocol(api, dev.index, frames[:128], W, H, do_match)
        except Exception as e:  # noqa: BLE001
            extras["plain_api"] = {"error": repr(e)[:300]}
    if not args.no_extras and world == 1:
        # in a fresh process: how HIP maps an instance's dozen streams onto the four hardware queues depends on the streams the
        # process created before (the 640x480 instance above), and with it the attribution of an overlapped detection's time to
        # its stages — same step time, stage intervals up to 2x apart. A fresh process is the reproducible case.
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "c3_leg.py")], capture_output=True, text=True, timeout=300)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("C3LEG ")][-1]
            extras["roofline_c3"] = json.loads(line[6:])
        except Exception as e:  # noqa: BLE001
            extras["roofline_c3"] = {"error": repr(e)[:300]}
        extras["single_image_ms"] = single_image_latency(api, dev.index, frames[0])
        # the sizes the reference publishes single-image detection times for (docs/Performances.md:26-35: 1536x1024 and 3456x2304, RTX 2060 /
        # GTX 1050 class GPUs, real photographs) on the same synthetic image family: detection only, same protocol
        for (pw, ph) in ((1536, 1024), (3456, 2304)):
            try:
                r = single_image_latency(api, dev.index, api.gen_synthetic_image(0x5EED0000, pw, ph), runs=30, warm=5, detect_only=True)
                extras["single_image_ms"][f"{pw}x{ph}"] = {"detect_ms": r["detect_ms"], "features": r["features"]}
            except Exception as e:  # noqa: BLE001
                extras["single_image_ms"][f"{pw}x{ph}"] = {"error": repr(e)[:200]}
        # the same workload in the FP16 pyramid mode (binary16 scale-space storage, fp32 arithmetic: DESIGN.md 2.3), fresh process;
        # its roofline is priced with the same per-pixel counts at 2 bytes per pyramid texel
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--fp16", "--no-extras", "--no-cpu-baseline", "--steps", "5", "--warmup", "2",
                                "--width", str(W), "--height", str(H), "--batch", str(B)], capture_output=True, text=True, timeout=300)
            d16 = json.loads(r.stdout.strip().splitlines()[-1])
            extras["fp16_mode"] = {"value": d16["value"], "unit": d16["unit"], "dtype": d16["dtype"], "mean_features_per_frame": d16["config"]["mean_features_per_frame"],
                                   "roofline_frac": d16["roofline"]["frac"], "roofline_achieved": d16["roofline"]["achieved"], "roofline_basis": d16["roofline"]["basis"],
                                   "stage_ms_per_call": d16["stage_ms_per_call"]}
        except Exception as e:  # noqa: BLE001
            extras["fp16_mode"] = {"error": repr(e)[:300]}
    if not args.no_extras and world == 1 and do_match:
        # the same workload with the matching of a step serialised in front of the next detection's scale-space instead of running beside it
        # (fresh process): what the blur launches and the scan reach when nothing shares the chip with them. The shipped schedule overlaps
        # them because it is ~1 % faster in frames/s; its stage times — and the headline roofline.frac — therefore include the contention.
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--serialize-match", "--no-extras", "--no-cpu-baseline", "--steps", "5", "--warmup", "2",
                                "--width", str(W), "--height", str(H), "--batch", str(B)], capture_output=True, text=True, timeout=300)
            ds = json.loads(r.stdout.strip().splitlines()[-1])
            extras["matching_serialised"] = {"value": ds["value"], "unit": ds["unit"], "roofline_frac": ds["roofline"]["frac"], "roofline_achieved": ds["roofline"]["achieved"],
                                             "roofline_basis": ds["roofline"]["basis"], "stage_ms_per_call": ds["stage_ms_per_call"],
                                             "note": "not the shipped schedule: scale-space behind the previous matching (VKSIFT_TUNE_PYR_GATE = 1)"}
        except Exception as e:  # noqa: BLE001
            extras["matching_serialised"] = {"error": repr(e)[:300]}
    if not args.no_extras:
        # BASELINE config 5 (north_star's multi-GPU workload): every rank, weak scaling; its collectives are timing barriers only
        try:
            extras["config5"] = c5_leg(api, torch, dist, dev, rank, world, use_dist=use_dist)
        except Exception as e:  # noqa: BLE001
            extras["config5"] = {"error": repr(e)[:300]}
    if not args.no_extras:
        # every rank takes part (the all-gather is a collective); a failure of this leg must not cost the headline line
        try:
            ms, crc = sharded_match(api, torch, dist, dev, rank, world, args.match_rows)
            if use_dist:
                t = torch.tensor([ms], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                ms = float(t.item())
            ops = 2.0 * args.match_rows * args.match_rows * 128
            extras["sharded_match"] = {"workload": f"BASELINE config 4: 2-NN {args.match_rows} x {args.match_rows} x 128-D, query rows sharded x{world}, one RCCL all-gather of B",
                                       "ms": ms, "tops_int8": ops / (ms * 1e-3) / 1e12, "frac_of_int8_peak": ops / (ms * 1e-3) / 1e12 / INT8_PEAK_TOPS / world,
                                       "records_crc32": crc,
                                       # ncclCommCount / ncclCommUserRank of the library's OWN communicator on rank 0 (vksift_ext_shardGroupInfo):
                                       # how many ranks RCCL itself saw in the all-gather
                                       "rccl_ranks": (getattr(multigpu_mod().sharded_match_timed, "last_info", None) or {}).get("rccl_ranks"),
                                       "group": getattr(multigpu_mod().sharded_match_timed, "last_info", None)}
        except Exception as e:  # noqa: BLE001
            extras["sharded_match"] = {"error": repr(e)[:300]}

    if rank == 0:
        out.update(extras)
        # the host-protocol figures beside `value` (SURVEY.md 8(d)'s own definition of the metric has host images in and results out)
        for k in ("value_host_input", "value_host_input_pipelined"):
            if isinstance(out.get(k), (int, float)) and out["value"] > 0:
                out[k + "_over_value"] = out[k] / out["value"]
        if not args.no_cpu_baseline:
            # rank 0's host cores, after the timed region (the other ranks are idle by now). At N = 1 the full sample incl. a 1920x1080
            # frame per core; at N > 1 (where the bench contract does not ask for it) one frame per core, so the line still carries it
            out["cpu_baseline"] = cpu_baseline(frames, do_match, per_worker=2 if world == 1 else 1, hd_frame=(world == 1 and not args.no_extras))
            out["cpu_baseline"]["measured_at_n_gpus"] = world
    watchdog.cancel()
    if rank == 0:
        emit(out)

    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
