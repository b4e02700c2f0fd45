"""k_descriptor's two arithmetic paths (features.hip: desc_sample<INRANGE>): the short forms of sqrtf and '/' that hold for gradients that
are zero or of ordinary size, and the general forms a wave falls back to for a step in which some lane holds a tiny non-zero gradient
(squared magnitude below 2^-96, smaller component below 2^-64). Images that are black except for a few bright shapes put the tails of
the Gaussian blurs — values down to the subnormals — inside the descriptor windows of the keypoints around the shapes, so both paths run;
the records have to equal the oracle's (IEEE sqrtf and division throughout) byte for byte."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def dark_image(seed, w, h, n_shapes, floor=0):
    rng = np.random.default_rng(seed)
    img = np.full((h, w), floor, np.uint8)
    for _ in range(n_shapes):
        x, y = int(rng.integers(8, w - 24)), int(rng.integers(8, h - 24))
        sw, sh = int(rng.integers(2, 14)), int(rng.integers(2, 14))
        img[y:y + sh, x:x + sw] = int(rng.integers(90, 256))
        if rng.random() < 0.5:  # a dot inside: corners and blobs at several scales
            img[y + sh // 2, x + sw // 2] = 0
    return img


@pytest.mark.parametrize("seed,w,h,n,floor,kw", [
    (1, 320, 240, 14, 0, {}),
    (2, 640, 480, 40, 0, {}),
    (3, 320, 240, 10, 1, {}),                                   # a floor of 1/255: flat but not zero, differences of single ulps
    (4, 400, 300, 25, 0, {"use_input_upsampling": False}),
    (5, 320, 240, 24, 0, {"descriptor_format": 1}),
])
def test_black_images_with_bright_shapes_match_the_oracle(vk, oracle, seed, w, h, n, floor, kw):
    img = dark_image(seed, w, h, n, floor)
    with vk.Instance(vk.default_config(input_image_max_size=w * h, **kw)) as inst:
        inst.detectFeatures(img, 0)
        feats = inst.downloadFeatures(0)
        # the tails are there: some texel of the first octave's planes is a non-zero value below 2^-48 (its square is below 2^-96)
        p = inst.downloadScaleSpaceImage(0, 3)
    assert len(feats) > 20
    if floor == 0:
        assert ((p > 0) & (p < 2.0 ** -48)).any()
    okw = {("use_vlfeat_format" if k == "descriptor_format" else k): (int(v) if isinstance(v, bool) else v) for k, v in kw.items()}
    ref, _ = oracle.detect(oracle.default_config(math_mode=1, **okw), img)
    assert feats.tobytes() == ref.tobytes()


def test_a_batch_of_dark_frames_matches_the_single_image_results(vk):
    w, h, n = 320, 240, 16
    imgs = [dark_image(100 + i, w, h, 12 + i) for i in range(n)]
    single = []
    with vk.Instance(vk.default_config(input_image_max_size=w * h)) as inst:
        for im in imgs:
            inst.detectFeatures(im, 0)
            single.append(inst.downloadFeatures(0).tobytes())
    with vk.Instance(vk.default_config(sift_buffer_count=n, input_image_max_size=w * h), batch_capacity=n) as inst:
        inst.detectFeaturesBatch(imgs, 0)
        batch = [inst.downloadFeatures(i).tobytes() for i in range(n)]
    assert batch == single and sum(len(b) for b in batch) > 164 * 300


def test_in_range_forms_equal_the_general_ones_on_the_device(vk):
    """vksift_hip_selftest_inrange: sqrt_inrange / div_inrange / div_2pi_inrange (what the orientation and descriptor kernels evaluate when
    their wave-uniform range check passes) against sqrtf, '/' and x / (2 pi) as the compiler expands them, on 64 M pseudo-random
    operands per form, spread uniformly over the EXPONENTS of the guarded ranges (2^-96 .. 2^8 for the square root, divisors from 2^-49, dividends
    from 2^-64 or zero) plus the range ends: no result may differ in any bit"""
    import ctypes as C
    import torch
    L = vk.lib()
    L.vksift_hip_selftest_inrange.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    L.vksift_hip_selftest_inrange.restype = C.c_int
    bad = torch.full((1,), -1, dtype=torch.int32, device="cuda")
    total = 0
    for seed in (1, 0x9E3779B9, 0x51ED270B, 77777):
        assert L.vksift_hip_selftest_inrange(1 << 24, seed, bad.data_ptr(), None) == 0
        torch.cuda.synchronize()
        total += int(bad.item())
    assert total == 0
