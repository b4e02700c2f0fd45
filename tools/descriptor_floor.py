#!/usr/bin/env python
"""What k_descriptor's time is made of: its ISA, priced per mnemonic with measured issue costs, times the work of one bench call —
against the hardware's own instruction count and the measured time ("move it or prove the floor").

    python tools/descriptor_floor.py collect [--frames 512] [--out gpurun_out/descriptor_floor_inputs.json]      (on the GPU box)
        the bench's frames through one batched detection with stage timing: descriptor_ms / orientation_ms; every feature record
        downloaded and, per keypoint, the kernel's own sample enumeration restated in numpy (window, analytic row spans: features.hip
        k_descriptor) -> samples, wave steps, loop iterations; runs tools/_valu_cost_table (tools/microbench/valu_cost_table.hip)
    python tools/descriptor_floor.py report --inputs <json> [--sq profiles/r05_sq_counters.json] [--out profiles/r05_descriptor_floor]   (anywhere with hipcc)
        compiles features.hip to ISA, splits k_descriptor<2, true, false> into its regions (per keypoint / row-span pass / sample loop /
        odd tail / epilogue), counts every mnemonic, prices it, writes the reconciliation (markdown + json)
"""
import argparse
import collections
import json
import math
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


# ------------------------------------------------------------------------------------------------ collect (GPU box)
def sample_counts(feats, w0, h0, upsampling=True):
    """per keypoint: N = samples enumerated by k_descriptor (the concatenated analytic row spans of its clipped window).
    float32 arithmetic in the kernel's order; the kernel's fused operations make single spans differ by one texel now and then
    (the spans are conservative by construction, the exact cells are re-derived per sample): a count, not a bit-exact restatement."""
    import numpy as np
    f32 = np.float32
    K = len(feats)
    if K == 0:
        return np.zeros(0, np.int64)
    o = feats["octave_idx"].astype(np.int64)
    oi = o + 1 if upsampling else o  # octave_idx is relative to the input resolution: -1 is the up-sampled octave
    ow = (w0 * 2 if upsampling else w0) >> oi
    oh = (h0 * 2 if upsampling else h0) >> oi
    sigma, kori = feats["sigma"].astype(f32), feats["orientation"].astype(f32)
    sf = np.exp2(o.astype(f32)).astype(f32)
    lam = f32(3.0) * (sigma / sf)
    radius = f32(math.sqrt(2.0)) * lam * f32(5.0) * f32(0.5)
    R = np.floor(radius + f32(0.5)).astype(np.int64)
    kcos = (np.cos(kori.astype(np.float64)).astype(f32) / lam)[:, None]
    ksin = (np.sin(kori.astype(np.float64)).astype(f32) / lam)[:, None]
    sx, sy = feats["scale_x"].astype(f32), feats["scale_y"].astype(f32)
    rsx, rsy = np.floor(sx + f32(0.5)), np.floor(sy + f32(0.5))  # roundf for positive values
    cxi, cyi = rsx.astype(np.int64), rsy.astype(np.int64)
    dx0, dx1 = np.maximum(-R, 1 - cxi)[:, None], np.minimum(R, ow - 2 - cxi)[:, None]
    dy0, dy1 = np.maximum(-R, 1 - cyi), np.minimum(R, oh - 2 - cyi)
    offx, offy = (rsx - sx)[:, None], (rsy - sy)[:, None]
    T = f32(2.5) + f32(0.01)
    nrow = int(max((dy1 - dy0 + 1).max(), 1))
    dy = dy0[:, None] + np.arange(nrow)[None, :]
    live = dy <= dy1[:, None]
    fy = dy.astype(f32) + offy
    lo = np.broadcast_to(dx0.astype(f32) - f32(0.5), fy.shape).copy()
    hi = np.broadcast_to(dx1.astype(f32) + f32(0.5), fy.shape).copy()
    for aa, bb in ((kcos, ksin * fy), (-ksin, kcos * fy)):
        ok = np.abs(aa) > 1e-12
        with np.errstate(divide="ignore", invalid="ignore"):
            ctr = -bb / aa - offx
            half = T / np.abs(aa)
        lo = np.where(ok, np.maximum(lo, ctr - half), lo)
        hi = np.where(ok, np.minimum(hi, ctr + half), np.where(np.abs(bb) < T, hi, lo - f32(2.0)))
    xl = np.maximum(np.ceil(lo - f32(0.01)).astype(np.int64), dx0)
    xh = np.minimum(np.floor(hi + f32(0.01)).astype(np.int64), dx1)
    out = (np.maximum(xh - xl + 1, 0) * live).sum(axis=1)
    return out


def orientation_counts(feats, w0, h0, upsampling=True):
    """k_orientation's work for one image: it runs once per KEYPOINT (the records of one keypoint's orientations share position, scale and
    sigma), one wave each, npix = (2 r + 1)^2 window texels in steps of 64; keypoints whose window touches the image border take the general loop.
    -> (keypoints, steps of the interior loop, steps of the border loop, sum of r + 1 = iterations of the weight-sum loop)"""
    import numpy as np
    f32 = np.float32
    if len(feats) == 0:
        return 0, 0, 0, 0
    key = np.stack([feats["scale_x"].view(np.uint32), feats["scale_y"].view(np.uint32), feats["sigma"].view(np.uint32),
                    feats["scale_idx"].astype(np.uint32), feats["octave_idx"].astype(np.int64).astype(np.uint32)], axis=1)
    _, first = np.unique(key, axis=0, return_index=True)
    f = feats[np.sort(first)]
    o = f["octave_idx"].astype(np.int64)
    oi = o + 1 if upsampling else o
    ow = (w0 * 2 if upsampling else w0) >> oi
    oh = (h0 * 2 if upsampling else h0) >> oi
    lam = f32(1.5) * (f["sigma"].astype(f32) / np.exp2(o.astype(f32)).astype(f32))
    r = np.floor(f32(3.0) * lam).astype(np.int64)
    npix = (2 * r + 1) ** 2
    steps = (npix + 63) // 64
    cx = np.floor(f["scale_x"].astype(f32) + f32(0.5)).astype(np.int64)
    cy = np.floor(f["scale_y"].astype(f32) + f32(0.5)).astype(np.int64)
    interior = (cx - r >= 1) & (cx + r <= ow - 2) & (cy - r >= 1) & (cy + r <= oh - 2)
    return len(f), int(steps[interior].sum()), int(steps[~interior].sum()), int((r + 1).sum())


def collect(args):
    import numpy as np
    import torch
    import vulkansift_amd.api as api
    W, H, B = args.width, args.height, args.frames
    dev = torch.device("cuda:0")
    ngen = min(B, 128)
    gen = np.stack([api.gen_synthetic_image(0x5EED0000 + i, W, H) for i in range(ngen)])
    variants = [gen, gen[:, :, ::-1], gen[:, ::-1, :], gen[:, ::-1, ::-1]]
    host = np.ascontiguousarray(np.concatenate([variants[(k // ngen) % 4][: min(ngen, B - k)] for k in range(0, B, ngen)]))
    d = torch.from_numpy(host).to(dev)
    inst = api.Instance(api.default_config(sift_buffer_count=B, input_image_max_size=W * H), batch_capacity=B)
    for _ in range(2):
        inst.detectFeaturesBatchDevice(d.data_ptr(), B, W, H, 0)
    torch.cuda.synchronize()
    inst.setProfiling(True)
    for _ in range(args.calls):
        inst.detectFeaturesBatchDevice(d.data_ptr(), B, W, H, 0)
    torch.cuda.synchronize()
    acc = inst.getAccumulatedDetectTimings()
    inst.setProfiling(False)
    stage = {k: acc[k] / max(acc["nb_calls"], 1) for k in ("pyramid_ms", "scan_ms", "orientation_ms", "descriptor_ms", "total_ms")}
    K = steps = samples = iters = odd = 0
    ori = [0, 0, 0, 0]
    hist = collections.Counter()
    per_oct = collections.Counter()
    nfr = min(B, args.count_frames)
    for b in range(nfr):
        f = inst.downloadFeatures(b)
        n = sample_counts(f, W, H)
        for k, v in enumerate(orientation_counts(f, W, H)):
            ori[k] += v
        t = (n + 63) // 64
        # two waves per keypoint (batch >= 8): the first takes ceil(t / 2) steps, the second floor(t / 2); a wave's loop does two
        # steps per iteration and one odd step in the tail
        r0, r1 = (t + 1) // 2, t // 2
        K += len(f)
        samples += int(n.sum())
        steps += int(t.sum())
        iters += int((r0 // 2).sum() + (r1 // 2).sum())
        odd += int((r0 % 2).sum() + (r1 % 2).sum())
        for o in f["octave_idx"]:
            per_oct[int(o)] += 1
        for v in t:
            hist[int(v)] += 1
    scale = B / nfr
    costs = None
    exe = os.path.join(ROOT, "tools", "_valu_cost_table")
    if os.path.exists(exe):
        costs = json.loads(subprocess.run([exe], capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1])
    out = {
        "workload": f"{B} x {W}x{H} frames in one detection call (bench.py's frames)", "frames": B, "counted_frames": nfr,
        "stage_ms_per_call": stage,
        "keypoints": K * scale, "samples": samples * scale, "wave_steps": steps * scale, "pair_iterations": iters * scale, "odd_tail_steps": odd * scale,
        "waves_per_keypoint": 2,
        "mean_samples_per_keypoint": samples / max(K, 1), "mean_steps_per_keypoint": steps / max(K, 1),
        "lane_fill": samples / max(steps * 64, 1),
        "keypoints_per_octave": dict(sorted(per_oct.items())),
        "orientation": {"keypoints": ori[0] * (B / nfr), "interior_steps": ori[1] * (B / nfr), "border_steps": ori[2] * (B / nfr), "weight_sum_iterations": ori[3] * (B / nfr)},
        "costs_ps": costs,
    }
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "costs_ps"}))


# ------------------------------------------------------------------------------------------------ report
KERNEL = "_ZN12_GLOBAL__N_112k_descriptorILi2ELb1ELb0EEEv5MultiINS_8FeatArgsEE"


ORI_KERNEL = "_ZN12_GLOBAL__N_113k_orientationILb1ELb0EEEv5MultiINS_8FeatArgsEE"


def kernel_isa(kernel=None):
    kernel = kernel or KERNEL
    src = os.path.join(ROOT, "vulkansift_amd", "csrc", "hip", "features.hip")
    out = "/tmp/_features_floor.s"
    import vulkansift_amd.build as b  # the flags the shipped kernels are compiled with
    cmd = [b.HIPCC] + [f for f in b.HIPFLAGS if f != "-fPIC"] + b._extra_flags("hip/features.hip") + b.INCLUDES + ["-S", "--cuda-device-only", "-o", out, src]
    subprocess.run(cmd, check=True, capture_output=True, cwd="/tmp")
    lines = open(out).read().split("\n")
    # (the name is matched as a prefix: further kernel parameters — the dense-row / posting struct of round 6 — extend the mangled name)
    s = next(i for i, l in enumerate(lines) if re.match(re.escape(kernel) + r"\w*:", l))
    e = next(i for i in range(s, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    return lines[s + 1:e + 1]


def blocks_of(lines):
    """[(label, depth, [instruction mnemonic, ...], [full text, ...])] in text order; depth from LLVM's loop annotations"""
    blocks = [["entry", 0, [], []]]
    pending = None
    for l in lines:
        m = re.match(r"^(\.LBB\d+_\d+):(.*)$", l)
        if m:
            blocks.append([m.group(1), None, [], []])
            pending = blocks[-1]
            d = re.search(r"Depth=(\d+)", m.group(2))
            if d:
                pending[1] = int(d.group(1))
            continue
        t = l.strip()
        if t.startswith(";"):
            d = re.search(r"Depth=(\d+)", t)
            if pending is not None and d:
                pending[1] = max(pending[1] or 0, int(d.group(1)))  # "This Inner Loop Header: Depth=n" follows the parent lines
            continue
        if not t or t.startswith("."):
            continue
        pending = None
        blocks[-1][2].append(t.split()[0])
        blocks[-1][3].append(t)
    for b in blocks:
        b[1] = b[1] or 0
    return blocks


def regions(blocks):
    """text-order split: the sample loop is the innermost-depth run of blocks holding the eight ds_add_u64 of two samples"""
    n_add = [sum(1 for m in b[2] if m == "ds_add_u64") for b in blocks]
    # main loop: the contiguous run of blocks at depth >= 3 that contains 8 atomics
    runs, cur = [], []
    for i, b in enumerate(blocks):
        if b[1] >= 3:
            cur.append(i)
        else:
            if cur:
                runs.append(cur)
            cur = []
    if cur:
        runs.append(cur)
    main = next(r for r in runs if sum(n_add[i] for i in r) == 8)
    # the slow path inside it: the block a s_cbranch_vccnz of the loop jumps to (inputs below 2^-96: never seen)
    targets = {t.split()[-1] for i in main for t in blocks[i][3] if t.startswith("s_cbranch_vccnz")}
    slow = []
    for i in main:
        if blocks[i][0] in targets:  # ... and the blocks behind it up to the unconditional branch back into the loop
            j = i
            while j in main:
                slow.append(j)
                if any(t.startswith("s_branch") for t in blocks[j][3]):
                    break
                j += 1
    after = main[-1] + 1
    epi = next(i for i in range(after, len(blocks)) if blocks[i][1] <= 1)
    tail = list(range(after, epi))
    reg = {
        "entry (once per workgroup)": [0],
        "per keypoint: constants, zeroing, row spans, prefix scan, run search": list(range(1, main[0])),
        "sample loop, two samples per lane per iteration": [i for i in main if i not in slow],
        "odd tail step": tail,
        "epilogue (wave 0: normalise, clamp, pack)": list(range(epi, len(blocks))),
    }
    return reg, slow


def klass(m):
    if m.startswith("v_"):
        return "valu"
    if m.startswith("s_"):
        return "salu"
    if m.startswith("ds_"):
        return "lds"
    if m.startswith("buffer_") or m.startswith("global_"):
        return "vmem"
    return "other"


def cost_key(m):
    """mnemonic -> key of the measured table"""
    if re.match(r"v_cmp_.*_f32(_e32)?$", m) and not m.startswith("v_cmp_class"):
        return "v_cmp_f32_e32"
    if re.match(r"v_cmp_.*_f32_e64$", m):
        return "v_cmp_f32_e64"
    if re.match(r"v_cmp_.*_[ui]32(_e32)?$", m):
        return "v_cmp_u32_e32"
    if re.match(r"v_cmp_.*_[ui]32_e64$", m):
        return "v_cmp_u32_e64"
    alias = {"v_sub_u32_e32": "v_add_u32_e32", "v_subrev_u32_e32": "v_add_u32_e32", "v_or_b32_e32": "v_and_b32_e32", "v_xor_b32_e32": "v_and_b32_e32",
             "v_add_f32_e64": "v_mul_f32_e64", "v_sub_f32_e64": "v_mul_f32_e64", "v_subrev_f32_e32": "v_sub_f32_e32", "v_lshrrev_b32_e32": "v_lshlrev_b32_e32",
             "v_ashrrev_i32_e32": "v_lshlrev_b32_e32", "v_max_f32_e32": "v_min_f32_e32", "v_max_i32_e32": "v_min_f32_e32", "v_min_i32_e32": "v_min_f32_e32",
             "v_min_u32_e32": "v_min_f32_e32", "v_max_u32_e32": "v_min_f32_e32", "v_min_f32_e64": "v_min_f32_e32", "v_max_f32_e64": "v_min_f32_e32",
             "v_med3_f32": "v_max3_f32", "v_min3_f32": "v_max3_f32", "v_mad_u32_u24": "v_lshl_add_u32", "v_add3_u32": "v_lshl_add_u32", "v_or3_b32": "v_and_or_b32",
             "v_lshl_or_b32": "v_and_or_b32", "v_ceil_f32_e32": "v_floor_f32_e32", "v_rndne_f32_e32": "v_floor_f32_e32", "v_trunc_f32_e32": "v_floor_f32_e32",
             "v_cvt_f32_u32_e32": "v_cvt_f32_i32_e32", "v_fma_f32": "v_fma_f32", "v_fmac_f32_e64": "v_fma_f32", "v_mul_lo_u32": "v_rcp_f32_e32", "v_mul_hi_u32": "v_rcp_f32_e32",
             "v_rsq_f32_e32": "v_rcp_f32_e32", "v_readfirstlane_b32": "v_mov_b32_e32", "v_mov_b32_e64": "v_mov_b32_e32", "v_accvgpr_write_b32": "v_mov_b32_e32",
             "v_mbcnt_lo_u32_b32": "v_and_b32_e32", "v_mbcnt_hi_u32_b32": "v_and_b32_e32", "v_mov_b32_dpp": "v_mov_b32_e32", "v_add_u32_dpp": "v_add_u32_e32",
             "v_bfe_u32": "v_and_or_b32", "v_lshlrev_b64": "v_pk_add_f32", "v_add_u32_e64": "v_add_u32_e32", "v_sub_u32_e64": "v_add_u32_e32",
             "v_pk_mov_b32": "v_pk_add_f32", "v_cndmask_b32_dpp": "v_cndmask_b32_e32", "v_bfi_b32": "v_and_or_b32", "v_sub_co_u32_e32": "v_add_u32_e32", "v_add_co_u32_e32": "v_add_u32_e32"}
    return alias.get(m, m)


def report(args):
    inp = json.load(open(args.inputs))
    costs = dict(inp["costs_ps"])
    # 128 back-to-back v_cndmask_b32_e32 on a VCC that an s_mov wrote long before measure 9.5 ns each — five times the e64 form on an SGPR
    # pair, and five times what the same instruction costs behind the v_cmp that wrote its VCC (the triple v_cmp + s_nop 1 + v_cndmask:
    # 3.1 ns in all). The kernel's selects all follow their compares: priced as the e64 form.
    anomaly = costs["v_cndmask_b32_e32"]
    costs["v_cndmask_b32_e32"] = costs["v_cndmask_b32_e64"]
    blocks = blocks_of(kernel_isa())
    reg, slow = regions(blocks)
    K, IT, ODD = inp["keypoints"], inp["pair_iterations"], inp["odd_tail_steps"]
    WPK = inp["waves_per_keypoint"]
    # executions per call of each region (wave-level): per-keypoint regions run in both waves of the keypoint's workgroup; the epilogue's
    # body in wave 0 only (wave 1 takes the branch around it: a handful of instructions, counted as the body's 1/2)
    execs = {
        "entry (once per workgroup)": None,
        "per keypoint: constants, zeroing, row spans, prefix scan, run search": K * WPK,
        "sample loop, two samples per lane per iteration": IT,
        "odd tail step": ODD,
        "epilogue (wave 0: normalise, clamp, pack)": K * WPK / 2,
    }
    # inner loops of the per-keypoint part that run more than once per keypoint: the binary search over the prefix array
    mean_rows = 2.0 * math.sqrt(inp["mean_samples_per_keypoint"] / 0.5) / 2.0 + 1  # window side ~ sqrt(2 N)
    search_iters = max(1.0, math.log2(max(mean_rows, 2.0)))
    rows = []
    unknown = collections.Counter()
    total = collections.Counter()
    total_ps = 0.0
    for name, idxs in reg.items():
        n = execs[name]
        if n is None:
            continue
        cnt = collections.Counter()
        ps = 0.0
        for i in idxs:
            b = blocks[i]
            mult = 1.0
            if name.startswith("per keypoint") and b[1] >= 3 and len(b[2]) <= 20 and any(t.startswith("s_cbranch_execnz") for t in b[3]) and any(m == "ds_read_b32" for m in b[2]) and not any(m.startswith("v_cvt") for m in b[2]):
                mult = search_iters  # the bisection loop (one LDS read, a compare, two selects per step)
            for m in b[2]:
                k = klass(m)
                cnt[k] += mult
                if k == "valu":
                    ck = cost_key(m)
                    if ck in costs:
                        ps += costs[ck] * mult
                    else:
                        unknown[m] += 1
                        ps += costs["v_cvt_u32_f32_e32"] * mult  # unknown mnemonics priced as a half-rate instruction
        rows.append({"region": name, "executions_per_call": n, "instructions": {k: round(v, 1) for k, v in cnt.items()}, "valu_ps_per_execution": ps})
        for k, v in cnt.items():
            total[k] += v * n
        total_ps += ps * n
    simds = 1024
    valu_ms = total_ps / simds * 1e-9
    # LDS: the 64-bit atomics of the sample loop at the measured rate for 16+ distinct addresses per wave (a CU's LDS serves all its waves)
    n_atomics = IT * 8 + ODD * 4
    lds_ms = n_atomics * costs["ds_add_u64@16"] / 256 * 1e-9
    # the sample loop's VALU instructions by cost class
    loop = next(r for r in rows if r["region"].startswith("sample loop"))
    by_key = collections.Counter()
    for i in reg["sample loop, two samples per lane per iteration"]:
        for m in blocks[i][2]:
            if klass(m) == "valu":
                by_key[cost_key(m)] += 1
    loop_table = sorted(((k, c, costs.get(k, costs["v_cvt_u32_f32_e32"])) for k, c in by_key.items()), key=lambda x: -x[1] * x[2])
    res = {
        "kernel": "k_descriptor<2, true, false> (two waves per keypoint, image index fastest, fp32 planes)",
        "inputs": {k: v for k, v in inp.items() if k != "costs_ps"},
        "regions": rows,
        "predicted_wave_instructions_per_call": {k: v for k, v in total.items()},
        "predicted_valu_issue_ms": valu_ms,
        "lds_atomic_ms_per_cu_pipe": lds_ms,
        "measured_descriptor_ms": inp["stage_ms_per_call"]["descriptor_ms"],
        "valu_issue_share_of_measured": valu_ms / inp["stage_ms_per_call"]["descriptor_ms"],
        "sample_loop_valu_by_mnemonic": [{"mnemonic": k, "per_iteration": c, "ps_each": p, "ps": c * p} for k, c, p in loop_table],
        "unpriced_mnemonics": dict(unknown),
        "v_cndmask_b32_e32_on_a_stale_scalar_vcc_ps": anomaly,
        "slow_path_blocks_excluded": [blocks[i][0] for i in slow],
    }
    if args.sq and os.path.exists(args.sq):
        sq = json.load(open(args.sq)).get("k_descriptor")
        if sq:
            d = 1.0  # the file holds per-dispatch means already (tools/make_profile_report.py)
            res["hardware_counters_per_dispatch"] = {k: sq[k] / d for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_BUSY_CU_CYCLES", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT") if k in sq}
            res["predicted_over_counted"] = {"valu": total["valu"] / (sq["SQ_INSTS_VALU"] / d), "salu": total["salu"] / (sq["SQ_INSTS_SALU"] / d),
                                             "lds": total["lds"] / (sq["SQ_INSTS_LDS"] / d), "vmem": total["vmem"] / (sq["SQ_INSTS_VMEM"] / d)}
    ori = orientation_report(inp, costs, args.sq)
    if ori:
        res["orientation"] = ori
    json.dump(res, open(args.out + ".json", "w"), indent=1)
    with open(args.out + ".md", "w") as f:
        w = f.write
        w("# k_descriptor: ISA x work x measured issue costs against the measured time\n\n")
        w(f"Workload: {inp['workload']}; {inp['keypoints']:.0f} keypoints, {inp['samples'] / 1e6:.1f} M samples "
          f"({inp['mean_samples_per_keypoint']:.0f} per keypoint, lanes {100 * inp['lane_fill']:.1f} % filled), {inp['wave_steps'] / 1e6:.2f} M wave steps = "
          f"{inp['pair_iterations'] / 1e6:.2f} M two-sample iterations + {inp['odd_tail_steps'] / 1e6:.2f} M odd tail steps.\n\n")
        w("| region | executions per call (waves) | VALU | SALU | LDS | VMEM | VALU issue time each (ns per SIMD) |\n|---|---|---|---|---|---|---|\n")
        for r in rows:
            i = r["instructions"]
            w(f"| {r['region']} | {r['executions_per_call']:.3g} | {i.get('valu', 0):.0f} | {i.get('salu', 0):.0f} | {i.get('lds', 0):.0f} | {i.get('vmem', 0):.0f} | {r['valu_ps_per_execution'] / 1e3:.1f} |\n")
        w(f"\nPredicted wave-instructions per call: VALU {total['valu'] / 1e6:.1f} M, SALU {total['salu'] / 1e6:.1f} M, LDS {total['lds'] / 1e6:.1f} M, VMEM {total['vmem'] / 1e6:.1f} M.\n")
        if "predicted_over_counted" in res:
            h = res["hardware_counters_per_dispatch"]
            p = res["predicted_over_counted"]
            w(f"Counted by the hardware per dispatch (`{os.path.basename(args.sq)}`): VALU {h['SQ_INSTS_VALU'] / 1e6:.1f} M, SALU {h['SQ_INSTS_SALU'] / 1e6:.1f} M, "
              f"LDS {h['SQ_INSTS_LDS'] / 1e6:.1f} M, VMEM {h['SQ_INSTS_VMEM'] / 1e6:.1f} M -> predicted / counted: VALU {p['valu']:.3f}, SALU {p['salu']:.3f}, LDS {p['lds']:.3f}, VMEM {p['vmem']:.3f}.\n")
        w(f"\n**VALU issue time** (every VALU instruction at its measured stand-alone cost, 1024 SIMDs): **{valu_ms:.2f} ms** of the measured "
          f"**{res['measured_descriptor_ms']:.2f} ms** = {100 * res['valu_issue_share_of_measured']:.0f} %.\n\n")
        w(f"LDS: {n_atomics / 1e6:.1f} M 64-lane `ds_add_u64` at {costs['ds_add_u64@16'] / 1e3:.2f} ns per CU (16 distinct addresses per wave; 64 distinct: "
          f"{costs['ds_add_u64@64'] / 1e3:.2f}, 8: {costs['ds_add_u64@8'] / 1e3:.2f}, one: {costs['ds_add_u64@1'] / 1e3:.1f}) = {lds_ms:.2f} ms on each CU's LDS pipe, concurrent with the VALU.\n\n")
        w(f"(`v_cndmask_b32_e32` is priced as the e64 form: back to back on a VCC written by an `s_mov` long before it measures {anomaly / 1e3:.1f} ns, behind the `v_cmp` that "
          f"wrote its VCC — every select of this kernel — the triple v_cmp + s_nop 1 + v_cndmask measures {costs['triple:v_cmp+s_nop1+v_cndmask'] / 1e3:.2f} ns in all.)\n\n")
        w("Sample loop, VALU instructions per iteration (two samples per lane) by mnemonic, most expensive first:\n\n| mnemonic | per iteration | ps each | ns |\n|---|---|---|---|\n")
        for k, c, p in loop_table:
            w(f"| `{k}` | {c} | {p:.0f} | {c * p / 1e3:.2f} |\n")
        if ori:
            w("\n# k_orientation: the same reconciliation\n\n")
            oi = ori["inputs"]
            w(f"{oi['keypoints']:.0f} keypoints (one wave each), {oi['interior_steps'] / 1e6:.2f} M steps of the interior window loop, {oi['border_steps'] / 1e6:.2f} M of the border loop, "
              f"{oi['weight_sum_iterations'] / max(oi['keypoints'], 1):.1f} iterations of the weight-sum loop per keypoint.\n\n")
            w("| region | executions per call (waves) | VALU | SALU | LDS | VMEM | VALU issue time each (ns per SIMD) |\n|---|---|---|---|---|---|---|\n")
            for r in ori["regions"]:
                i = r["instructions"]
                w(f"| {r['region']} | {r['executions_per_call']:.3g} | {i.get('valu', 0):.0f} | {i.get('salu', 0):.0f} | {i.get('lds', 0):.0f} | {i.get('vmem', 0):.0f} | {r['valu_ps_per_execution'] / 1e3:.1f} |\n")
            t = ori["predicted_wave_instructions_per_call"]
            w(f"\nPredicted wave-instructions per call: VALU {t.get('valu', 0) / 1e6:.1f} M, SALU {t.get('salu', 0) / 1e6:.1f} M")
            if "predicted_over_counted" in ori:
                w(f"; counted: VALU {ori['hardware_counters_per_dispatch']['SQ_INSTS_VALU'] / 1e6:.1f} M, SALU {ori['hardware_counters_per_dispatch']['SQ_INSTS_SALU'] / 1e6:.1f} M "
                  f"-> predicted / counted VALU {ori['predicted_over_counted']['valu']:.3f}, SALU {ori['predicted_over_counted']['salu']:.3f}")
            w(f".\n\n**VALU issue time {ori['predicted_valu_issue_ms']:.2f} ms** of the measured **{ori['measured_orientation_ms']:.2f} ms** = "
              f"{100 * ori['predicted_valu_issue_ms'] / ori['measured_orientation_ms']:.0f} %.\n")
        if unknown:
            w(f"\nMnemonics without a measured cost (priced as half-rate): {dict(unknown)}\n")
    print(json.dumps({k: res[k] for k in ("predicted_valu_issue_ms", "measured_descriptor_ms", "valu_issue_share_of_measured", "predicted_wave_instructions_per_call", "unpriced_mnemonics") if k in res}
                     | ({"predicted_over_counted": res["predicted_over_counted"]} if "predicted_over_counted" in res else {})))


def orientation_report(inp, costs, sq_path):
    """the same reconciliation for k_orientation<true, false>: one wave per keypoint; regions = per keypoint (everything outside the two window
    loops, the weight-sum loop's body times its iteration count) / interior window loop / border window loop; the general-form fallback
    blocks of both loops (v_div_scale inside) are left out like the descriptor's"""
    o = inp.get("orientation")
    if not o:
        return None
    blocks = blocks_of(kernel_isa(ORI_KERNEL))
    has = lambda b, m: any(x == m for x in b[2])  # noqa: E731
    # the window loops: contiguous runs of blocks at depth >= 2 that hold the histogram atomic
    runs, cur = [], []
    for i, b in enumerate(blocks):
        if b[1] >= 2:
            cur.append(i)
        elif cur:
            runs.append(cur)
            cur = []
    if cur:
        runs.append(cur)
    loops = [r for r in runs if any(has(blocks[i], "ds_add_u32") for i in r)]
    # a run may hold both loops back to back: split at the blocks that load the four taps
    loop_blocks = []
    for r in loops:
        heads = [i for i in r if sum(1 for m in blocks[i][2] if m == "buffer_load_dword") >= 4]
        if len(heads) <= 1:
            loop_blocks.append(r)
        else:
            for a, bnd in zip(heads, heads[1:] + [r[-1] + 1]):
                loop_blocks.append([i for i in r if a <= i < bnd])
    def priced(idxs, mult=None):
        cnt, ps = collections.Counter(), 0.0
        for i in idxs:
            b = blocks[i]
            if any(m == "v_div_scale_f32" for m in b[2]) and b[1] >= 2:
                continue  # general-form fallback of a window loop
            k = (mult or {}).get(i, 1.0)
            for m in b[2]:
                c = klass(m)
                cnt[c] += k
                if c == "valu":
                    ps += costs.get(cost_key(m), costs["v_cvt_u32_f32_e32"]) * k
        return cnt, ps
    loop_blocks.sort(key=lambda r: sum(len(blocks[i][2]) for i in r))
    in_loops = {i for r in loop_blocks for i in r}
    interior, border = loop_blocks[0], loop_blocks[-1]
    # the weight-sum loop: the depth-3 block with the lane broadcast; its iterations per keypoint come from the records
    per_kp = [i for i in range(1, len(blocks)) if i not in in_loops]
    mult = {i: o["weight_sum_iterations"] / max(o["keypoints"], 1) for i in per_kp if blocks[i][1] >= 3}
    rows, total, total_ps = [], collections.Counter(), 0.0
    for name, idxs, n, m in (("per keypoint: weight sum, window set-up, smoothing, peaks", per_kp, o["keypoints"], mult),
                             ("interior window loop, one texel per lane per step", interior, o["interior_steps"], None),
                             ("border window loop", border, o["border_steps"], None)):
        cnt, ps = priced(idxs, m)
        rows.append({"region": name, "executions_per_call": n, "instructions": {k: round(v, 1) for k, v in cnt.items()}, "valu_ps_per_execution": ps})
        for k, v in cnt.items():
            total[k] += v * n
        total_ps += ps * n
    res = {"kernel": "k_orientation<true, false> (one wave per keypoint)", "inputs": o, "regions": rows,
           "predicted_wave_instructions_per_call": dict(total), "predicted_valu_issue_ms": total_ps / 1024 * 1e-9,
           "measured_orientation_ms": inp["stage_ms_per_call"]["orientation_ms"]}
    if sq_path and os.path.exists(sq_path):
        sq = json.load(open(sq_path)).get("k_orientation<")
        if sq:
            res["hardware_counters_per_dispatch"] = {k: sq[k] for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM") if k in sq}
            res["predicted_over_counted"] = {"valu": total["valu"] / sq["SQ_INSTS_VALU"], "salu": total["salu"] / sq["SQ_INSTS_SALU"]}
    return res


def main():
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    c = sub.add_parser("collect")
    c.add_argument("--frames", type=int, default=512)
    c.add_argument("--count-frames", type=int, default=128)
    c.add_argument("--calls", type=int, default=5)
    c.add_argument("--width", type=int, default=640)
    c.add_argument("--height", type=int, default=480)
    c.add_argument("--out", default="gpurun_out/descriptor_floor_inputs.json")
    r = sub.add_parser("report")
    r.add_argument("--inputs", required=True)
    r.add_argument("--sq", default=os.path.join(ROOT, "profiles", "r05_sq_counters.json"))
    r.add_argument("--out", default=os.path.join(ROOT, "profiles", "r05_descriptor_floor"))
    a = ap.parse_args()
    (collect if a.cmd == "collect" else report)(a)


if __name__ == "__main__":
    main()
