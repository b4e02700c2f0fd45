"""Independent numpy restatement of the three sparse shaders of the reference — K4 ExtractKeypoints.comp,
K5 ComputeOrientation.comp, K6 ComputeDescriptors.comp — written from the GLSL, not from oracle/sift_oracle.c,
and on purpose in a different formulation, so that a shared misreading of a shader cannot hide
(VERDICT r01 "weak #1"). TEST INFRASTRUCTURE ONLY.

How it differs from the C oracle / the HIP kernels:

  K4  * the 26-neighbour test is "the centre is the unique maximum (minimum) of its 3x3x3 cube", evaluated with
        sliding-window reductions over the whole DoG stack at once
      * the refinement solves H * off = -g with a float64 LAPACK solve (the shader writes the adjugate out by
        hand); all candidates of an octave advance together under masks (structure-of-arrays)
      * the out-of-range DoG layer of quirk Q1 is a physical zero plane appended to the stack
  K5  * the whole (2r+1)^2 window is evaluated as arrays; exp / atan2 are computed in float64 and rounded to
        fp32 where the shader holds an fp32 value; the histogram is a weighted bincount
      * the smoothing uses np.roll; the Q3 peak formula is spelled with numpy uint32 wrap-around arithmetic
  K6  * window as arrays, the eight trilinear corners as a broadcast (.., 2, 2, 2) block, scatter by bincount;
        the floored modulo of quirk Q5 is numpy's `%` on negative integers (floored by definition)

All comparisons against the oracle therefore carry small tolerances (documented in tests/test_np_features.py):
float64-vs-fp32 intermediates differ in the last bits, and truncations uint(x) can flip by one count.
"""
import numpy as np
from numpy.lib.stride_tricks import sliding_window_view

f32 = np.float32
u32 = np.uint32
PI32 = f32(3.14159265358979323846)       # GLSL: the literal is a 32-bit float
TWO_PI32 = f32(f32(2.0) * PI32)

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("scale_x", "<f4"), ("scale_y", "<f4"), ("scale_idx", "<u4"), ("octave_idx", "<i4"),
                     ("sigma", "<f4"), ("orientation", "<f4"), ("intensity", "<f4"), ("descriptor", "u1", (128,))])


def _round_half_away(v):
    """GLSL round(): the half-way direction is implementation defined; the build fixes it to "away from zero"."""
    v = np.asarray(v, dtype=np.float64)
    return np.sign(v) * np.floor(np.abs(v) + 0.5)


# ------------------------------------------------------------------------------------------------------------------
# K4 — ExtractKeypoints.comp
# ------------------------------------------------------------------------------------------------------------------
def extract_keypoints(dog, S, octave_idx, seed_sigma=1.6, intensity_threshold=0.04, edge_threshold=10.0):
    """dog: (S+2, H, W) float32 DoG planes of one octave. Returns KP_DTYPE records in raster (s, y, x) order
    (the reference's order is whatever its atomics produce; the build defines raster order)."""
    dog = np.asarray(dog, dtype=f32)
    nl, H, W = dog.shape
    assert nl == S + 2
    thr = f32(f32(intensity_threshold) / f32(S))                         # sift_detector.c:1136
    pre = f32(thr * f32(0.8))                                            # ExtractKeypoints.comp:57

    # --- candidates: unique extremum of the 3x3x3 cube, |c| > 0.8 thr (:56-116) -----------------------------------
    cubes = sliding_window_view(dog, (3, 3, 3))                          # (S, H-2, W-2, 3, 3, 3)
    centre = dog[1:-1, 1:-1, 1:-1]
    cmax = cubes.max(axis=(3, 4, 5))
    cmin = cubes.min(axis=(3, 4, 5))
    n_eq = (cubes == centre[..., None, None, None]).sum(axis=(3, 4, 5))  # how many cube texels carry the centre's value
    is_ext = ((centre == cmax) | (centre == cmin)) & (n_eq == 1) & (np.abs(centre) > pre)
    ss, yy, xx = np.nonzero(is_ext)
    s = (ss + 1).astype(np.int64)
    y = (yy + 1).astype(np.int64)
    x = (xx + 1).astype(np.int64)
    n = len(s)
    if n == 0:
        return np.zeros(0, KP_DTYPE)

    # Q1: the refinement may sit on layer S+1 and read layer S+2, which does not exist -> robust access returns 0
    vol = np.concatenate([dog, np.zeros((1, H, W), f32)], axis=0).astype(np.float64)

    off = np.zeros((n, 3))           # (oS, oX, oY), fp32 values held in float64
    grad = np.zeros((n, 3))
    alive = np.ones(n, bool)         # refinement still valid (non-singular Hessian)
    moving = np.ones(n, bool)        # not yet converged
    for step in range(5):            # NB_REFINEMENT_STEP
        idx = np.nonzero(alive & moving)[0]
        if len(idx) == 0:
            break
        si, yi, xi = s[idx], y[idx], x[idx]
        V = lambda ds, dx, dy: vol[si + ds, yi + dy, xi + dx]
        c = V(0, 0, 0)
        g = np.stack([0.5 * (V(1, 0, 0) - V(-1, 0, 0)), 0.5 * (V(0, 1, 0) - V(0, -1, 0)), 0.5 * (V(0, 0, 1) - V(0, 0, -1))], axis=1)
        Hm = np.empty((len(idx), 3, 3))
        Hm[:, 0, 0] = V(1, 0, 0) + V(-1, 0, 0) - 2.0 * c
        Hm[:, 1, 1] = V(0, 1, 0) + V(0, -1, 0) - 2.0 * c
        Hm[:, 2, 2] = V(0, 0, 1) + V(0, 0, -1) - 2.0 * c
        Hm[:, 0, 1] = Hm[:, 1, 0] = 0.25 * (V(1, 1, 0) - V(1, -1, 0) - V(-1, 1, 0) + V(-1, -1, 0))
        Hm[:, 0, 2] = Hm[:, 2, 0] = 0.25 * (V(1, 0, 1) - V(1, 0, -1) - V(-1, 0, 1) + V(-1, 0, -1))
        Hm[:, 1, 2] = Hm[:, 2, 1] = 0.25 * (V(0, 1, 1) - V(0, 1, -1) - V(0, -1, 1) + V(0, -1, -1))
        det = np.linalg.det(Hm)
        singular = det == 0.0
        alive[idx[singular]] = False
        ok = ~singular
        sol = np.zeros((len(idx), 3))
        if ok.any():
            sol[ok] = np.linalg.solve(Hm[ok], -g[ok][..., None])[..., 0]
        sol = sol.astype(f32).astype(np.float64)
        off[idx[ok]] = sol[ok]
        grad[idx[ok]] = g[ok].astype(f32).astype(np.float64)
        conv = ok & (np.abs(sol) < 0.6).all(axis=1)
        moving[idx[conv]] = False
        if step < 4:                 # "do not increment keypoint pos in last iter"
            mv = idx[ok & ~conv]
            o = off[mv]
            x[mv] += ((o[:, 1] >= 0.6) & (x[mv] < W - 2)).astype(np.int64) - ((o[:, 1] <= -0.6) & (x[mv] > 1)).astype(np.int64)
            y[mv] += ((o[:, 2] >= 0.6) & (y[mv] < H - 2)).astype(np.int64) - ((o[:, 2] <= -0.6) & (y[mv] > 1)).astype(np.int64)
            s[mv] += ((o[:, 0] >= 0.6) & (s[mv] < S + 1)).astype(np.int64) - ((o[:, 0] <= -0.6) & (s[mv] > 1)).astype(np.int64)

    oS, oX, oY = off[:, 0], off[:, 1], off[:, 2]
    sub_x = (x + oX).astype(f32)
    sub_y = (y + oY).astype(f32)
    sub_s = (s + oS).astype(f32)
    val = (vol[s, y, x] + 0.5 * (grad[:, 1] * oX + grad[:, 2] * oY + grad[:, 0] * oS)).astype(f32)
    keep = alive & (np.abs(val) > thr) & (np.abs(oX) < 1.5) & (np.abs(oY) < 1.5) & (np.abs(oS) < 1.5)
    keep &= (sub_x >= 0) & (sub_x < W) & (sub_y >= 0) & (sub_y < H) & (sub_s >= 0) & (sub_s <= S + 1)

    # edge response on the 2-D Hessian at the final integer position (:195-201)
    c = vol[s, y, x]
    hxx = vol[s, y, x + 1] + vol[s, y, x - 1] - 2.0 * c
    hyy = vol[s, y + 1, x] + vol[s, y - 1, x] - 2.0 * c
    hxy = 0.25 * (vol[s, y + 1, x + 1] - vol[s, y - 1, x + 1] - vol[s, y + 1, x - 1] + vol[s, y - 1, x - 1])
    with np.errstate(divide="ignore", invalid="ignore"):
        edge = (hxx + hyy) ** 2 / (hxx * hyy - hxy * hxy)
    limit = (float(f32(edge_threshold)) + 1.0) ** 2 / float(f32(edge_threshold))
    keep &= (edge < limit) & (edge >= 0)

    k = np.nonzero(keep)[0]
    out = np.zeros(len(k), KP_DTYPE)
    factor = 2.0 ** octave_idx
    out["scale_x"] = sub_x[k]
    out["scale_y"] = sub_y[k]
    out["scale_idx"] = _round_half_away(sub_s[k]).astype(u32)
    out["octave_idx"] = octave_idx
    out["sigma"] = (float(f32(seed_sigma)) * np.exp2(sub_s[k].astype(np.float64) / S) * factor).astype(f32)
    out["intensity"] = val[k]
    out["x"] = (sub_x[k].astype(np.float64) * factor).astype(f32)
    out["y"] = (sub_y[k].astype(np.float64) * factor).astype(f32)
    return out


# ------------------------------------------------------------------------------------------------------------------
# shared by K5 / K6
# ------------------------------------------------------------------------------------------------------------------
def _gradients(plane, gx, gy):
    """central differences x0.5 with out-of-image loads = 0 (imageLoad robust access)"""
    H, W = plane.shape
    pad = np.zeros((H + 4, W + 4), np.float64)
    pad[2:-2, 2:-2] = plane

    def at(xx, yy):
        xx = np.clip(xx + 2, 0, W + 3)      # everything outside the image lands in the zero frame
        yy = np.clip(yy + 2, 0, H + 3)
        return pad[yy, xx]

    dx = (0.5 * (at(gx + 1, gy) - at(gx - 1, gy))).astype(f32)
    dy = (0.5 * (at(gx, gy + 1) - at(gx, gy - 1))).astype(f32)
    return dx, dy


def _angle_0_2pi(gy, gx):
    """atan(gy, gx) as an fp32 value, then the shader's wrap into [0, 2pi] (fp32 adds)"""
    a = np.arctan2(gy.astype(np.float64), gx.astype(np.float64)).astype(f32)
    a = np.where(a < 0, (a + TWO_PI32).astype(f32), np.where(a > TWO_PI32, (a - TWO_PI32).astype(f32), a))
    return a.astype(f32)


def _ceil_log2(v):
    return int(np.ceil(np.log2(float(v))))


# ------------------------------------------------------------------------------------------------------------------
# K5 — ComputeOrientation.comp
# ------------------------------------------------------------------------------------------------------------------
def orientation_histogram(plane, kp):
    """36-bin fixed-point histogram after the six smoothing passes. plane = Gaussian layer kp.scale_idx of the octave."""
    H, W = plane.shape
    sigma_oct = f32(f32(kp["sigma"]) / f32(2.0 ** int(kp["octave_idx"])))
    lam = f32(f32(1.5) * sigma_oct)
    r = int(np.floor(f32(f32(3) * lam)))
    es = f32(f32(-1.0) / f32(f32(f32(2.0) * lam) * lam))

    ii = np.arange(-r, r + 1)
    d2_int = (ii[:, None] ** 2 + ii[None, :] ** 2).astype(np.float64)
    M = float(np.sum(np.exp(float(es) * d2_int)) * np.sqrt(2.0))          # the shader sums this in fp32: only ceil(log2) is used
    fp = f32(1 << (30 - _ceil_log2(M)))

    cx = f32(_round_half_away(kp["scale_x"]))
    cy = f32(_round_half_away(kp["scale_y"]))
    dy, dx = np.meshgrid(ii, ii, indexing="ij")
    gx = int(cx) + dx
    gy = int(cy) + dy
    sdx = ((cx + dx.astype(f32)).astype(f32) - f32(kp["scale_x"])).astype(f32)
    sdy = ((cy + dy.astype(f32)).astype(f32) - f32(kp["scale_y"])).astype(f32)
    d2 = ((sdx * sdx).astype(f32) + (sdy * sdy).astype(f32)).astype(f32)
    outside_image = (gx < 1) | (gx >= W - 1) | (gy < 1) | (gy >= H - 1)
    skipped = outside_image & (d2 > f32(r * r))                          # Q2: '&&' — the circle only prunes out-of-image texels
    gX, gY = _gradients(plane, gx, gy)
    norm = np.sqrt(((gX * gX).astype(f32) + (gY * gY).astype(f32)).astype(f32)).astype(f32)
    w = np.exp((d2 * es).astype(f32).astype(np.float64)).astype(f32)
    mag = (w * norm).astype(f32)
    ang = _angle_0_2pi(gY, gX)
    b = (((ang * f32(36)).astype(f32)) / TWO_PI32).astype(f32).astype(np.int64)
    b = np.where(b < 0, b + 36, np.where(b >= 36, b - 36, b))
    contrib = (mag * fp).astype(f32).astype(np.uint64)                   # uint(mag * fp): truncation
    contrib[skipped] = 0
    hist = np.bincount(b.ravel(), weights=None, minlength=36) * 0
    hist = np.zeros(36, np.uint64)
    np.add.at(hist, b.ravel(), contrib.ravel())
    hist = (hist & 0xFFFFFFFF).astype(u32)

    for _ in range(3):
        for _ in range(2):
            tot = (np.roll(hist, 1) + hist + np.roll(hist, -1)).astype(u32)    # uint32 adds
            hist = (tot.astype(f32) / f32(3)).astype(f32).astype(u32)
    return hist


def orientations(plane, kp, max_nb_orientation=4):
    """principal orientations of one keypoint, in ascending bin order (the build's order for the unspecified arrival order)"""
    h = orientation_histogram(plane, kp)
    top = h.max()
    prev = np.roll(h, 1)
    nxt = np.roll(h, -1)
    peak = (h.astype(f32) >= (f32(0.8) * f32(top)).astype(f32)) & (h > prev) & (h > nxt)
    out = []
    for i in np.nonzero(peak)[0]:
        with np.errstate(over="ignore"):
            num = u32(prev[i] - nxt[i])                                  # Q3: uint wrap-around, then float()
            den = u32(u32(prev[i] - u32(u32(2) * h[i])) + nxt[i])
        pos = f32(f32(i) + f32(f32(0.5) * f32(f32(num) / f32(den))))
        out.append(f32(f32(f32(pos + f32(0.5)) * TWO_PI32) / f32(36)))
    if max_nb_orientation:
        out = out[:max_nb_orientation]
    return np.array(out, f32), h


# ------------------------------------------------------------------------------------------------------------------
# K6 — ComputeDescriptors.comp
# ------------------------------------------------------------------------------------------------------------------
def descriptor(plane, kp, theta, vlfeat=False):
    """(128 bytes, 128 raw uint32 accumulators) of one oriented keypoint"""
    H, W = plane.shape
    theta = f32(theta)
    sigma_oct = f32(f32(kp["sigma"]) / f32(2.0 ** int(kp["octave_idx"])))
    lam = f32(f32(3.0) * sigma_oct)
    radius = f32(f32(f32(np.sqrt(f32(2.0))) * lam) * f32(5)) * f32(0.5)
    R = int(np.floor(f32(radius + f32(0.5))))
    kc = f32(f32(np.cos(float(theta))) / lam)
    ks = f32(f32(np.sin(float(theta))) / lam)
    es = -0.125

    half = R // 2
    jj = np.arange(half)
    e = np.exp(es * (jj[:, None] ** 2 + jj[None, :] ** 2).astype(np.float64))
    # the shader walks the upper triangle with the off-diagonal doubled = the full symmetric sum
    M = float(e.sum() * np.sqrt(2.0)) if half > 0 else 0.0
    fp = f32(1 << (16 - _ceil_log2(M))) if M > 0 else f32(np.inf)

    cx = f32(_round_half_away(kp["scale_x"]))
    cy = f32(_round_half_away(kp["scale_y"]))
    ii = np.arange(-R, R + 1)
    dy, dx = np.meshgrid(ii, ii, indexing="ij")
    px = int(cx) + dx
    py = int(cy) + dy
    inside = (px >= 1) & (px < W - 1) & (py >= 1) & (py < H - 1)
    sdx = ((cx + dx.astype(f32)).astype(f32) - f32(kp["scale_x"])).astype(f32)
    sdy = ((cy + dy.astype(f32)).astype(f32) - f32(kp["scale_y"])).astype(f32)
    ox = ((kc * sdx).astype(f32) + (ks * sdy).astype(f32)).astype(f32)
    oy = ((kc * sdy).astype(f32) - (ks * sdx).astype(f32)).astype(f32)
    gX, gY = _gradients(plane, px, py)
    ang = _angle_0_2pi(gY, gX)
    rel = (ang - theta).astype(f32)
    rel = np.where(rel < 0, (rel + TWO_PI32).astype(f32), np.where(rel > TWO_PI32, (rel - TWO_PI32).astype(f32), rel)).astype(f32)
    norm = np.sqrt(((gX * gX).astype(f32) + (gY * gY).astype(f32)).astype(f32)).astype(f32)
    r2 = ((ox * ox).astype(f32) + (oy * oy).astype(f32)).astype(f32)
    mag = (np.exp((f32(es) * r2).astype(f32).astype(np.float64)).astype(f32) * norm).astype(f32)

    fx = (ox + f32(2)).astype(f32)
    fy = (oy + f32(2)).astype(f32)
    fb = ((rel * f32(8)).astype(f32) / TWO_PI32).astype(f32)
    if not vlfeat:
        fb = ((((-rel).astype(f32)) * f32(8)).astype(f32) / TWO_PI32).astype(f32)
    ix = np.floor((fx - f32(0.5)).astype(f32)).astype(np.int64)
    iy = np.floor((fy - f32(0.5)).astype(f32)).astype(np.int64)
    ib = np.floor(fb).astype(np.int64)
    rx = (fx - (ix.astype(f32) + f32(0.5)).astype(f32)).astype(f32)
    ry = (fy - (iy.astype(f32) + f32(0.5)).astype(f32)).astype(f32)
    rb = (fb - ib.astype(f32)).astype(f32)

    acc = np.zeros(128, np.uint64)
    for i in (0, 1):
        for j in (0, 1):
            for k in (0, 1):
                cxh = ix + i
                cyh = iy + j
                ok = inside & (cxh >= 0) & (cxh < 4) & (cyh >= 0) & (cyh < 4)
                bin_ = (ib + k) % 8                                       # numpy %: floored -> [0, 7] (quirk Q5 = OpSMod)
                wx = np.abs((f32(1.0 - i) - rx).astype(f32))
                wy = np.abs((f32(1.0 - j) - ry).astype(f32))
                wb = np.abs((f32(1.0 - k) - rb).astype(f32))
                v = (((wx * wy).astype(f32) * wb).astype(f32) * mag).astype(f32)
                q = (v * fp).astype(f32).astype(np.uint64)
                slot = cyh * 32 + cxh * 8 + bin_
                np.add.at(acc, slot[ok], q[ok])
    raw = (acc & 0xFFFFFFFF).astype(u32)

    work = raw.astype(np.uint64)
    n1 = np.sqrt(f32(u32(int((work * work).sum()) & 0xFFFFFFFF)))
    lim = np.uint64(int(f32(f32(n1) * f32(0.2))))
    work = np.minimum(work, lim)
    n2 = f32(np.sqrt(f32(u32(int((work * work).sum()) & 0xFFFFFFFF))))
    with np.errstate(divide="ignore", invalid="ignore"):
        scaled = (work.astype(f32) * f32(f32(512.0) / n2)).astype(f32)
    scaled = np.where(np.isnan(scaled), f32(0), scaled)
    desc = np.where(scaled > 255, 255, np.floor(scaled)).astype(np.uint8)
    return desc, raw


# ------------------------------------------------------------------------------------------------------------------
# one octave end to end
# ------------------------------------------------------------------------------------------------------------------
def detect_octave(gauss, dog, S, octave_idx, seed_sigma=1.6, intensity_threshold=0.04, edge_threshold=10.0, max_nb_orientation=4, vlfeat=False):
    """gauss: (S+3, H, W), dog: (S+2, H, W). Returns features ordered like the build: keypoints in raster order with their
    first orientation, then the extra orientations in (keypoint, bin) order."""
    kps = extract_keypoints(dog, S, octave_idx, seed_sigma, intensity_threshold, edge_threshold)
    first, extra = [], []
    for kp in kps:
        ang, _ = orientations(gauss[int(kp["scale_idx"])], kp, max_nb_orientation)
        rec = kp.copy()
        rec["orientation"] = ang[0] if len(ang) else f32(0)           # Q4: no peak -> theta stays 0, still described
        first.append(rec)
        for a in ang[1:]:
            r2 = kp.copy()
            r2["orientation"] = a
            extra.append(r2)
    feats = np.array(first + extra, dtype=KP_DTYPE) if (first or extra) else np.zeros(0, KP_DTYPE)
    for f in feats:
        d, _ = descriptor(gauss[int(f["scale_idx"])], f, f["orientation"], vlfeat)
        f["descriptor"] = d
    return feats
