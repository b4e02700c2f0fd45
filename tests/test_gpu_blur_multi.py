"""Scales S+1 and S+2 as ONE launch per scale over several octaves (k_blur_lean_multi, vksift_hip_blur_multi: pyramid.hip; enqueue_tail:
vksift_detect.c) against one launch per octave and scale (VKSIFT_TUNE_TAIL_MULTI = 1) and the oracle. Both forms of use: the forked
scale-space of a small detection (every octave in the tail, on the side stream) and a batch (octave 0 and every octave that takes the
four-texel kernel in full, the coarser ones in the tail). Every plane and every feature has to be the same bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TUNE_TAIL_MULTI = 10

SHAPES = [
    (640, 480, {}),                                                    # five octaves, the coarsest in the LDS chain
    (1536, 1024, {}),                                                  # seven octaves: the tail holds more than four
    (1000, 600, {"use_hardware_interpolated_blur": False}),            # even tap counts (no multi-octave instantiation: per-plane fallback)
    (768, 512, {"nb_scales_per_octave": 4}),                           # 7 layers, other tap counts at S+1 / S+2
    (512, 384, {"nb_scales_per_octave": 2, "seed_scale_sigma": 1.2}),
    (640, 480, {"pyramid_precision_mode": 1}),                         # binary16 planes
    (322, 242, {}),                                                    # widths the strip march refuses on some octaves (generic tiles)
]


def _planes_and_features(vk, img, w, h, kw, batch):
    S = kw.get("nb_scales_per_octave", 3)
    n = max(batch, 1)
    with vk.Instance(vk.default_config(input_image_max_size=w * h, sift_buffer_count=n, **kw), batch_capacity=n) as inst:
        if batch:
            inst.detectFeaturesBatch([img] * batch, 0)
        else:
            inst.detectFeatures(img, 0)
        feats = [inst.downloadFeatures(i).tobytes() for i in range(n)]
        n_oct = inst.getScaleSpaceNbOctaves()
        planes = [[inst.downloadScaleSpaceImage(o, s) for s in range(S + 3)] for o in range(n_oct)]   # image 0's planes
    return feats, planes


@pytest.mark.parametrize("batch", [0, 8])
@pytest.mark.parametrize("w,h,kw", SHAPES)
def test_tail_launches_per_scale_equal_launches_per_octave(vk, oracle, w, h, kw, batch):
    img = vk.gen_synthetic_image_family(5200 + w + h, w, h, (w // 64) % 3)
    L = vk.lib()
    out = {}
    try:
        for name, knob in (("per_octave", 1), ("per_scale", 0)):
            L.vksift_hip_tune(TUNE_TAIL_MULTI, knob)
            out[name] = _planes_and_features(vk, img, w, h, kw, batch)
    finally:
        L.vksift_hip_tune(TUNE_TAIL_MULTI, 0)
    assert out["per_scale"][0] == out["per_octave"][0] and len(out["per_scale"][0][0]) > 164 * 50
    assert all(f == out["per_scale"][0][0] for f in out["per_scale"][0])            # every image of the batch
    for o, (pa, pb) in enumerate(zip(out["per_scale"][1], out["per_octave"][1])):
        for s, (a, b) in enumerate(zip(pa, pb)):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (o, s)
    okw = {k: (int(v) if isinstance(v, bool) else v) for k, v in kw.items()}
    okw["pyramid_fp16"] = okw.pop("pyramid_precision_mode", 0)
    ref, _ = oracle.detect(oracle.default_config(math_mode=1, **okw), img)
    assert out["per_scale"][0][0] == ref.tobytes()
