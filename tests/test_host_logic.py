"""Host-side behaviour of the product library that does not need a GPU: it loads on a CPU-only box,
fails the way the reference documents when no device is present, and its host maths matches the oracle."""
import ctypes as C
import zlib

import numpy as np
import pytest
import torch

NO_GPU = not torch.cuda.is_available()


def test_default_config_matches_reference_defaults(vk):
    c = vk.lib().vksift_getDefaultConfig()   # vulkansift.c:47-64
    assert c.input_image_max_size == 1920 * 1080 and c.sift_buffer_count == 2 and c.max_nb_sift_per_buffer == 100000
    assert c.use_input_upsampling is True and c.nb_octaves == 0 and c.nb_scales_per_octave == 3
    assert abs(c.input_image_blur_level - 0.5) < 1e-7 and abs(c.seed_scale_sigma - 1.6) < 1e-7
    assert abs(c.intensity_threshold - 0.04) < 1e-8 and c.edge_threshold == 10.0
    assert c.max_nb_orientation_per_keypoint == 4 and c.descriptor_format == 0 and c.gpu_device_index == -1
    assert c.use_hardware_interpolated_blur is True and c.pyramid_precision_mode == 0
    assert bool(c.on_error_callback_function) and c.use_gpu_debug_functions is False


@pytest.mark.skipif(not NO_GPU, reason="CPU-only behaviour")
def test_graceful_failure_without_gpu(vk):
    L = vk.lib()
    L.vksift_setLogLevel(vk.VKSIFT_NO_LOG)
    # createInstance before loadVulkan -> VKSIFT_VULKAN_ERROR (vulkansift.c:175-179)
    h = C.c_void_p(None)
    cfg = vk.default_config()
    assert L.vksift_createInstance(C.byref(h), C.byref(cfg)) == vk.VKSIFT_VULKAN_ERROR and not h
    # no device: the documented "fail so callers can fall back" path (README.md:23)
    assert L.vksift_loadVulkan() == vk.VKSIFT_VULKAN_ERROR
    n = C.c_uint32(99)
    L.vksift_getAvailableGPUs(C.byref(n), None)
    assert n.value == 0
    with pytest.raises(vk.VksiftError):
        vk.Instance()
    L.vksift_setLogLevel(vk.VKSIFT_LOG_INFO)


def test_product_host_math_equals_oracle(vk, oracle):
    """vksift_hm_* (product) and orc_* (oracle) are separate restatements of the same reference code."""
    L = vk.lib()
    for kw in ({}, {"use_input_upsampling": False}, {"use_hardware_interpolated_blur": False}, {"nb_scales_per_octave": 4},
               {"seed_scale_sigma": 2.0, "input_image_blur_level": 0.7}):
        vcfg = vk.default_config(**kw)
        ocfg = oracle.default_config(**{k: (int(v) if isinstance(v, bool) else v) for k, v in kw.items()})
        S = vcfg.nb_scales_per_octave
        taps = np.zeros((S + 3, 20), np.float32)
        ntaps = np.zeros(S + 3, np.uint32)
        L.vksift_hm_blur_taps(C.byref(vcfg), taps.ctypes.data_as(C.c_void_p), ntaps.ctypes.data_as(C.c_void_p))
        otaps, ontaps = oracle.effective_taps(ocfg)
        assert np.array_equal(ntaps, ontaps) and np.array_equal(taps.view(np.uint32), otaps.view(np.uint32)), kw
        rounded = C.c_uint32(0)
        L.vksift_hm_max_octaves.restype = C.c_uint32
        mo = L.vksift_hm_max_octaves(C.byref(vcfg), C.byref(rounded))
        assert (mo, rounded.value) == oracle.max_nb_octaves(ocfg)
        for (w, h) in ((640, 480), (1920, 1080), (333, 777), (64, 64)):
            ow = (C.c_uint32 * 16)()
            oh = (C.c_uint32 * 16)()
            L.vksift_hm_octaves_for.restype = C.c_uint32
            n = L.vksift_hm_octaves_for(C.byref(vcfg), mo, w, h, ow, oh)
            assert [(ow[i], oh[i]) for i in range(n)] == oracle.scale_space_info(ocfg, w, h)
    for n_oct in (1, 3, 5, 7):
        caps = (C.c_uint32 * 16)()
        L.vksift_hm_section_caps(100000, n_oct, caps)
        assert [caps[i] for i in range(n_oct)] == oracle.section_caps(100000, n_oct)


def test_synthetic_generators_are_deterministic(vk):
    a = vk.gen_synthetic_image(0x5EED0000, 640, 480)
    b = vk.gen_synthetic_image(0x5EED0000, 640, 480)
    c = vk.gen_synthetic_image(0x5EED0001, 640, 480)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert a.std() > 30 and 80 < a.mean() < 180
    d = vk.gen_synthetic_descriptors(1, 1000)
    assert np.array_equal(d, vk.gen_synthetic_descriptors(1, 1000))
    norms = np.sqrt((d.astype(float) ** 2).sum(1))
    assert np.all((norms > 480) & (norms <= 513))
