"""Geometric quality check in the spirit of the reference's src/perf/perf_matching.cpp:30-79 (Oxford `H1toNp` homographies are
not available offline: a synthetic image is warped by a known homography instead).

putative matches = cross-checked + ratio-tested 2-NN matches (src/perf/perf_common.cpp:123-169);
a match is correct when the first keypoint, mapped by the ground-truth homography, lands within 2.5 px of the second
(perf_matching.cpp's precision threshold). This is evidence that does not come from the oracle: a SIFT that detects,
orients or describes wrongly does not survive a 12-degree rotation + scale + perspective warp."""
import numpy as np


def homography(w, h, angle_deg=12.0, scale=1.12, tx=6.0, ty=-4.0, persp=8e-5):
    c, s = np.cos(np.radians(angle_deg)) * scale, np.sin(np.radians(angle_deg)) * scale
    cx, cy = w / 2.0, h / 2.0
    T0 = np.array([[1, 0, -cx], [0, 1, -cy], [0, 0, 1.0]])
    R = np.array([[c, -s, 0], [s, c, 0], [persp, -persp, 1.0]])
    T1 = np.array([[1, 0, cx + tx], [0, 1, cy + ty], [0, 0, 1.0]])
    return T1 @ R @ T0


def warp(img, H):
    """dst(x, y) = bilinear(src, H^-1 (x, y)); outside -> mid grey. uint8 in, uint8 out."""
    h, w = img.shape
    Hi = np.linalg.inv(H)
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    den = Hi[2, 0] * xs + Hi[2, 1] * ys + Hi[2, 2]
    u = (Hi[0, 0] * xs + Hi[0, 1] * ys + Hi[0, 2]) / den
    v = (Hi[1, 0] * xs + Hi[1, 1] * ys + Hi[1, 2]) / den
    x0 = np.floor(u).astype(np.int64)
    y0 = np.floor(v).astype(np.int64)
    fx, fy = u - x0, v - y0
    inside = (x0 >= 0) & (y0 >= 0) & (x0 < w - 1) & (y0 < h - 1)
    x0c, y0c = np.clip(x0, 0, w - 2), np.clip(y0, 0, h - 2)
    f = img.astype(np.float64)
    val = (f[y0c, x0c] * (1 - fx) * (1 - fy) + f[y0c, x0c + 1] * fx * (1 - fy) + f[y0c + 1, x0c] * (1 - fx) * fy + f[y0c + 1, x0c + 1] * fx * fy)
    out = np.where(inside, val, 128.0)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def project(H, x, y):
    den = H[2, 0] * x + H[2, 1] * y + H[2, 2]
    return (H[0, 0] * x + H[0, 1] * y + H[0, 2]) / den, (H[1, 0] * x + H[1, 1] * y + H[1, 2]) / den


# Five warps of increasing difficulty, standing in for the image pairs 1->2 ... 1->6 of an Oxford sequence (viewpoint + zoom + rotation)
WARPS = [dict(angle_deg=4.0, scale=1.04, tx=3.0, ty=-2.0, persp=2e-5), dict(angle_deg=12.0, scale=1.12, tx=6.0, ty=-4.0, persp=8e-5),
         dict(angle_deg=25.0, scale=1.3, tx=-8.0, ty=5.0, persp=1.5e-4), dict(angle_deg=-40.0, scale=0.8, tx=4.0, ty=9.0, persp=2.5e-4),
         dict(angle_deg=70.0, scale=1.6, tx=0.0, ty=0.0, persp=4e-4)]


def score(feats1, feats2, idx_a, idx_b, H, w, h, tol=2.5):
    """The four metrics of the reference's computeMetrics() (src/perf/perf_matching.cpp:30-79):
      putative_match_ratio = matches / keypoints of image 1                                  (:69-70)
      precision            = matches within `tol` px of the homography / matches             (:52-66, :71-72)
      matching_score       = those inliers / keypoints of image 1                            (:73-74)
      repeatability        = the reference calls cv::evaluateFeatureDetector (region overlap); OpenCV is not available here, so
                             this is the point-based form: keypoints of image 1 visible in image 2 that have a detection
                             within `tol` px of their projection"""
    x1, y1 = feats1["x"].astype(np.float64), feats1["y"].astype(np.float64)
    x2, y2 = feats2["x"].astype(np.float64), feats2["y"].astype(np.float64)
    px, py = project(H, x1[idx_a], y1[idx_a])
    err = np.hypot(px - x2[idx_b], py - y2[idx_b])
    correct = int((err < tol).sum())
    # repeatability: keypoints of image 1 that fall inside image 2 and have a detection within tol there
    qx, qy = project(H, x1, y1)
    vis = (qx >= 0) & (qy >= 0) & (qx < w) & (qy < h)
    rep = 0
    if vis.any() and len(x2):
        d2 = (qx[vis, None] - x2[None, :]) ** 2 + (qy[vis, None] - y2[None, :]) ** 2
        rep = float((d2.min(axis=1) < tol * tol).mean())
    return {"keypoints_1": int(len(x1)), "keypoints_2": int(len(x2)), "matches": int(len(idx_a)), "correct": correct,
            "putative_match_ratio": len(idx_a) / max(len(x1), 1), "precision": correct / max(len(idx_a), 1),
            "matching_score": correct / max(len(x1), 1), "repeatability": rep}
