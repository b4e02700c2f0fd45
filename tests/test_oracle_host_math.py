"""Pins the oracle's host maths to closed-form values derived from the reference source
(octave tables, section capacities, Gaussian kernel sizes and weights)."""
import numpy as np


def test_octave_tables(oracle):
    cfg = oracle.default_config()
    assert oracle.max_nb_octaves(cfg) == (7, 1440 * 1440)
    assert oracle.scale_space_info(cfg, 640, 480) == [(1280, 960), (640, 480), (320, 240), (160, 120), (80, 60)]
    assert oracle.scale_space_info(cfg, 1920, 1080) == [(3840, 2160), (1920, 1080), (960, 540), (480, 270), (240, 135), (120, 67), (60, 33)]
    noups = oracle.default_config(use_input_upsampling=0)
    assert oracle.max_nb_octaves(noups)[0] == 6
    assert oracle.scale_space_info(noups, 640, 480) == [(640, 480), (320, 240), (160, 120), (80, 60)]
    assert len(oracle.scale_space_info(noups, 1920, 1080)) == 6
    three = oracle.default_config(nb_octaves=3)
    assert oracle.scale_space_info(three, 640, 480) == [(1280, 960), (640, 480), (320, 240)]


def test_section_capacities(oracle):
    assert oracle.section_caps(100000, 5) == [51612, 25806, 12903, 6451, 3225]
    assert oracle.section_caps(100000, 7) == [50393, 25196, 12598, 6299, 3149, 1574, 787]
    assert oracle.section_caps(100000, 3) == [57142, 28571, 14285]
    assert oracle.section_caps(1000, 3) == [571, 285, 142]  # the worked example in sift_memory.c:57-58


KAT_WEIGHTS = {
    # sigma, one-sided normalised weights (7 d.p.) — SURVEY.md appendix A
    0: (1.2489997, [0.3194115, 0.2318214, 0.0886263, 0.0178475, 0.0018932, 0.0001058]),
    1: (1.2262733, [0.3253304, 0.2333031, 0.0860415, 0.0163188, 0.0015917, 0.0000798]),
    2: (1.5450078, [0.2582139, 0.2094164, 0.1117131, 0.0391976, 0.0090464, 0.0013733, 0.0001371, 0.0000090]),
    3: (1.9465880, [0.2049464, 0.1796113, 0.1208963, 0.0624999, 0.0248160, 0.0075678, 0.0017725, 0.0003189, 0.0000441]),
    4: (2.4525466, [0.1626672, 0.1496921, 0.1166529, 0.0769822, 0.0430213, 0.0203598, 0.0081595, 0.0027692, 0.0007959, 0.0001937, 0.0000399]),
    5: (3.0900156, [0.1291084, 0.1225215, 0.1047093, 0.0805887, 0.0558572, 0.0348657, 0.0195990, 0.0099217, 0.0045233, 0.0018571, 0.0006866,
                    0.0002286, 0.0000686, 0.0000185]),
}


def test_gaussian_kernels_direct(oracle):
    cfg = oracle.default_config(use_hardware_interpolated_blur=0)
    k, sizes, sig = oracle.gaussian_kernels(cfg)
    assert list(sizes) == [6, 6, 8, 9, 11, 14]
    for s, (sigma, w) in KAT_WEIGHTS.items():
        assert abs(sig[s] - sigma) < 2e-6
        assert np.allclose(k[s, : len(w)], w, atol=1.5e-7)
        assert np.all(k[s, len(w):] == 0)
    noups = oracle.default_config(use_hardware_interpolated_blur=0, use_input_upsampling=0)
    k2, sizes2, sig2 = oracle.gaussian_kernels(noups)
    assert list(sizes2) == [8, 6, 8, 9, 11, 14]
    assert np.allclose(k2[0, :8], [0.2624848, 0.2113981, 0.1104312, 0.0374176, 0.0082235, 0.0011723, 0.0001084, 0.0000065], atol=1.5e-7)


def test_interpolated_kernel_packing_and_dropped_tap(oracle):
    """sift_detector.c:122-136 + quirk Q9: the paired-tap variant never samples an unpaired last tap."""
    cfg = oracle.default_config()
    packed, sizes, _ = oracle.gaussian_kernels(cfg)
    direct, _, _ = oracle.gaussian_kernels(oracle.default_config(use_hardware_interpolated_blur=0))
    taps, ntaps = oracle.effective_taps(cfg)
    assert list(ntaps) == [5, 5, 7, 9, 11, 13]
    for s in range(6):
        K = int(sizes[s])
        assert packed[s, 0] == direct[s, 0] and packed[s, 1] == 0
        for d in range(1, K - 1, 2):
            ki = (d + 1) // 2
            assert np.isclose(packed[s, 2 * ki], direct[s, d] + direct[s, d + 1], rtol=1e-6)
            off = (d * direct[s, d] + (d + 1) * direct[s, d + 1]) / (direct[s, d] + direct[s, d + 1])
            assert np.isclose(packed[s, 2 * ki + 1], off, rtol=1e-6)
        n = int(ntaps[s])
        assert np.allclose(taps[s, :n], direct[s, :n], rtol=2e-6, atol=1e-9)   # same weights in exact arithmetic
        assert np.all(taps[s, n:] == 0)
    # DC gains quoted in SURVEY.md appendix A
    gains = [2 * taps[s].sum() - taps[s, 0] for s in range(6)]
    assert np.allclose(gains, [0.9997885, 0.9998403, 0.9999819, 1.0, 1.0, 0.9999631], atol=2e-7)


def test_thresholds_at_defaults(oracle):
    cfg = oracle.default_config()
    thr = np.float32(cfg.intensity_threshold) / np.float32(3)
    assert abs(thr - 0.013333) < 1e-6
    assert abs((np.float32(11) ** 2) / np.float32(10) - 12.1) < 1e-5


def test_byte_over_255_without_division_or_table(tmp_path):
    """The seed launch converts its u8 source through unit_of_byte (pyramid.hip): q = x * fl(1/255); q + fma(-255, q, x) * fl(1/255). It has
    to be the IEEE quotient x / 255.f for all 256 bytes (the plain product is not, for 126 of them). Checked in C with the build's own
    contraction rules (gcc -ffp-contract=off, explicit fmaf)."""
    import subprocess

    src = tmp_path / "q.c"
    src.write_text(r"""
#include <stdio.h>
#include <math.h>
#include <string.h>
int main(void) {
  const float r = 0x1.010102p-8f; int plain = 0, corrected = 0;
  float one = 1.0f / 255.0f; if (memcmp(&one, &r, 4)) return 2;
  for (int i = 0; i < 256; i++) {
    const float x = (float)i, ref = x / 255.0f, q = x * r, c = fmaf(fmaf(-255.0f, q, x), r, q);
    plain += memcmp(&q, &ref, 4) != 0; corrected += memcmp(&c, &ref, 4) != 0;
  }
  printf("%d %d\n", plain, corrected); return 0;
}
""")
    exe = tmp_path / "q"
    subprocess.run(["gcc", "-O1", "-ffp-contract=off", "-o", str(exe), str(src), "-lm"], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    assert out == ["126", "0"], out
