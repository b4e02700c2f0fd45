"""The two extra image families of vksift_synth.c (all parity images were Gaussian blobs + noise) do what they are for,
checked with the oracle on CPU:
  EDGES    the edge-response rejection (ExtractKeypoints.comp:193-206) is what decides a large share of its candidates, and keypoints
           sit close enough to the image border for the orientation / descriptor windows to leave the image (ComputeOrientation.comp:97-100)
  FRACTAL  extrema in every octave of the pyramid
and both are deterministic (CRC of the bytes)."""
import zlib

import numpy as np


def test_families_are_deterministic(vk):
    for fam, crc in ((vk.SYNTH_EDGES, None), (vk.SYNTH_FRACTAL, None)):
        a = vk.gen_synthetic_image_family(11, 320, 240, fam)
        b = vk.gen_synthetic_image_family(11, 320, 240, fam)
        c = vk.gen_synthetic_image_family(12, 320, 240, fam)
        assert a.tobytes() == b.tobytes() and a.tobytes() != c.tobytes()
        assert 20 < a.std() < 80 and a.min() >= 0 and a.max() <= 255
    assert vk.gen_synthetic_image_family(5, 160, 120, vk.SYNTH_BLOBS).tobytes() == vk.gen_synthetic_image(5, 160, 120).tobytes()
    # pinned bytes: the GPU tests and the fixtures of later rounds must see the same images
    assert zlib.crc32(vk.gen_synthetic_image_family(7, 640, 480, vk.SYNTH_EDGES).tobytes()) == 4028376474


def test_edges_family_exercises_the_edge_rejection_and_the_border_windows(vk, oracle):
    img = vk.gen_synthetic_image_family(21, 480, 360, vk.SYNTH_EDGES)
    strict, _ = oracle.detect(oracle.default_config(math_mode=1), img)                           # edge_threshold 10
    loose, _ = oracle.detect(oracle.default_config(math_mode=1, edge_threshold=1000.0), img)    # rejection all but off
    assert len(strict) > 300
    assert len(loose) > 1.15 * len(strict)          # the edge test decides one refined candidate in six here ...
    blobs = vk.gen_synthetic_image_family(21, 480, 360, vk.SYNTH_BLOBS)
    b_strict, _ = oracle.detect(oracle.default_config(math_mode=1), blobs)
    b_loose, _ = oracle.detect(oracle.default_config(math_mode=1, edge_threshold=1000.0), blobs)
    assert len(b_loose) < 1.03 * len(b_strict)      # ... and fewer than one in thirty on the blob images every other test uses
    # keypoints whose orientation window (radius 3 * 1.5 * sigma in octave pixels, >= 4.5 * sigma image pixels) leaves the image
    x, y, s = strict["x"], strict["y"], strict["sigma"]
    near = (x < 4.5 * s) | (y < 4.5 * s) | (x > 480 - 4.5 * s) | (y > 360 - 4.5 * s)
    assert near.sum() >= 10


def test_fractal_family_has_extrema_in_every_octave(vk, oracle):
    img = vk.gen_synthetic_image_family(22, 480, 360, vk.SYNTH_FRACTAL)
    feats, _ = oracle.detect(oracle.default_config(math_mode=1), img)
    assert len(feats) > 500
    assert list(np.unique(feats["octave_idx"])) == [-1, 0, 1, 2, 3]
