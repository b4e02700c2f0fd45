"""Bit-exact detection (HIP vs oracle, det math) on the image families that are not Gaussian blobs: step edges / corners / checker
patches (the edge-response rejection and the border windows on purpose) and 1/f noise (every octave busy), at 640x480 and 1920x1080,
default configuration and the no-up-sampling / VLFeat-format one."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same(feats, ref):
    assert len(feats) == len(ref) and len(ref) > 200, (len(feats), len(ref))
    assert np.array_equal(feats["descriptor"], ref["descriptor"])
    for name in ("x", "y", "sigma", "orientation"):
        assert np.array_equal(feats[name].view(np.uint32), ref[name].view(np.uint32)), name
    assert feats.tobytes() == ref.tobytes()


@pytest.mark.parametrize("family", ["edges", "fractal"])
@pytest.mark.parametrize("w,h,kw", [(640, 480, {}), (1920, 1080, {}), (640, 480, {"use_input_upsampling": False, "descriptor_format": 1})])
def test_detection_bit_exact_on_family(vk, oracle, family, w, h, kw):
    fam = vk.SYNTH_EDGES if family == "edges" else vk.SYNTH_FRACTAL
    img = vk.gen_synthetic_image_family(31 + w, w, h, fam)
    okw = {"use_input_upsampling": int(kw.get("use_input_upsampling", True)), "use_vlfeat_format": int(kw.get("descriptor_format", 0))}
    cfg = vk.default_config(input_image_max_size=w * h, max_nb_sift_per_buffer=200000, **kw)
    with vk.Instance(cfg) as inst:
        inst.detectFeatures(img, 0)
        feats = inst.downloadFeatures(0)
    ref, _ = oracle.detect(oracle.default_config(math_mode=1, max_nb_sift_per_buffer=200000, **okw), img)
    _same(feats, ref)


def test_batch_of_mixed_families_equals_single_detections(vk):
    """a batched detection (the timed path's kernels: multi-octave launches, batch grids) over images of all three families"""
    w, h = 640, 480
    imgs = [vk.gen_synthetic_image_family(50 + i, w, h, i % 3) for i in range(9)]
    cfg = vk.default_config(sift_buffer_count=9, input_image_max_size=w * h)
    with vk.Instance(cfg, batch_capacity=9) as inst:
        inst.detectFeaturesBatch(imgs, 0)
        batch = [inst.downloadFeatures(i) for i in range(9)]
    with vk.Instance(vk.default_config(input_image_max_size=w * h)) as inst:
        for i, img in enumerate(imgs):
            inst.detectFeatures(img, 0)
            assert inst.downloadFeatures(0).tobytes() == batch[i].tobytes(), i
