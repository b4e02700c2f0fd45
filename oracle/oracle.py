"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package vulkansift_amd never does (tests/test_layout.py enforces it).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

MAX_KERNEL = 20
MAX_OCTAVES = 16


class Config(C.Structure):
    _fields_ = [
        ("input_image_max_size", C.c_uint32),
        ("max_nb_sift_per_buffer", C.c_uint32),
        ("use_input_upsampling", C.c_int32),
        ("nb_octaves", C.c_int32),
        ("nb_scales_per_octave", C.c_int32),
        ("input_image_blur_level", C.c_float),
        ("seed_scale_sigma", C.c_float),
        ("intensity_threshold", C.c_float),
        ("edge_threshold", C.c_float),
        ("max_nb_orientation_per_keypoint", C.c_uint32),
        ("use_vlfeat_format", C.c_int32),
        ("use_hardware_interpolated_blur", C.c_int32),
        ("math_mode", C.c_int32),
        ("pyramid_fp16", C.c_int32),
        ("sampler_model", C.c_int32),
    ]


FEATURE_DTYPE = np.dtype(
    [
        ("x", "<f4"), ("y", "<f4"), ("scale_x", "<f4"), ("scale_y", "<f4"),
        ("scale_idx", "<u4"), ("octave_idx", "<i4"),
        ("sigma", "<f4"), ("orientation", "<f4"), ("intensity", "<f4"),
        ("descriptor", "u1", (128,)),
    ]
)
MATCH_DTYPE = np.dtype([("idx_a", "<u4"), ("idx_b1", "<u4"), ("idx_b2", "<u4"), ("dist_a_b1", "<f4"), ("dist_a_b2", "<f4")])
assert FEATURE_DTYPE.itemsize == 164 and MATCH_DTYPE.itemsize == 20


def build(force=False):
    """Compile liboracle.so with the committed Makefile (gcc only)."""
    src = [os.path.join(_HERE, "sift_oracle.c"), os.path.join(_HERE, "sift_oracle.h"),
           os.path.join(_HERE, "..", "vulkansift_amd", "csrc", "detmath.h")]
    if not force and os.path.exists(_LIB_PATH) and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in src if os.path.exists(s)):
        return _LIB_PATH
    subprocess.run(["make", "-C", _HERE, "-B", "liboracle.so"], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def use_library(path):
    """Bind another build of sift_oracle.c (bench.py's cpu_baseline: -O3 -march=native, compiled on the box it runs on)."""
    global _lib, _LIB_PATH
    _LIB_PATH = path
    _lib = None
    return lib()


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        u32p, f32p, u8p = C.POINTER(C.c_uint32), C.POINTER(C.c_float), C.POINTER(C.c_uint8)
        cfgp = C.POINTER(Config)
        L.orc_default_config.argtypes = [cfgp]
        L.orc_max_nb_octaves.argtypes = [cfgp, u32p]
        L.orc_max_nb_octaves.restype = C.c_uint32
        L.orc_scale_space_info.argtypes = [cfgp, C.c_uint32, C.c_uint32, u32p, u32p]
        L.orc_scale_space_info.restype = C.c_uint32
        L.orc_section_caps.argtypes = [C.c_uint32, C.c_uint32, u32p]
        L.orc_gaussian_kernels.argtypes = [cfgp, f32p, u32p, f32p]
        L.orc_effective_taps.argtypes = [cfgp, f32p, u32p]
        L.orc_pyramid_build.argtypes = [cfgp, C.c_void_p, C.c_uint32, C.c_uint32]
        L.orc_pyramid_build.restype = C.c_void_p
        L.orc_pyramid_free.argtypes = [C.c_void_p]
        L.orc_pyramid_nb_octaves.argtypes = [C.c_void_p]
        L.orc_pyramid_nb_octaves.restype = C.c_uint32
        L.orc_pyramid_resolution.argtypes = [C.c_void_p, C.c_uint32, u32p, u32p]
        L.orc_pyramid_gauss.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.orc_pyramid_gauss.restype = C.c_void_p
        L.orc_pyramid_dog.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.orc_pyramid_dog.restype = C.c_void_p
        L.orc_detect_from_pyramid.argtypes = [cfgp, C.c_void_p, C.c_void_p, C.c_uint32, u32p]
        L.orc_detect_from_pyramid.restype = C.c_uint32
        L.orc_detect.argtypes = [cfgp, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, u32p]
        L.orc_detect.restype = C.c_uint32
        L.orc_extract_keypoints.argtypes = [cfgp, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        L.orc_extract_keypoints.restype = C.c_uint32
        L.orc_orientations.argtypes = [cfgp, C.c_void_p, C.c_uint32, C.c_void_p, f32p, u32p]
        L.orc_orientations.restype = C.c_uint32
        L.orc_descriptor.argtypes = [cfgp, C.c_void_p, C.c_uint32, C.c_void_p, u32p]
        L.orc_match_2nn.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_match_2nn_desc.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_filter_matches.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_float, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_filter_matches.restype = C.c_uint32
        for fn in ("orc_dm_expf", "orc_dm_exp2f", "orc_dm_sinf", "orc_dm_cosf"):
            getattr(L, fn).argtypes = [C.c_float]
            getattr(L, fn).restype = C.c_float
        L.orc_dm_atan2f.argtypes = [C.c_float, C.c_float]
        L.orc_dm_atan2f.restype = C.c_float
        L.orc_dm_div_2pi.argtypes = [C.c_float]
        L.orc_dm_div_2pi.restype = C.c_float
        L.orc_check_div_3.argtypes = []
        L.orc_check_div_3.restype = C.c_uint32
        L.orc_dm_expf_nb.argtypes = [C.c_float]
        L.orc_dm_expf_nb.restype = C.c_float
        L.orc_dm_expf_nb_nonpos.argtypes = [C.c_float]
        L.orc_dm_expf_nb_nonpos.restype = C.c_float
        L.orc_dm_ceil_log2f.argtypes = [C.c_float]
        L.orc_dm_ceil_log2f.restype = C.c_int
        _lib = L
    return _lib


def default_config(**overrides):
    cfg = Config()
    lib().orc_default_config(C.byref(cfg))
    for k, v in overrides.items():
        if not hasattr(cfg, k):
            raise AttributeError(k)
        setattr(cfg, k, v)
    return cfg


def max_nb_octaves(cfg):
    r = C.c_uint32()
    n = lib().orc_max_nb_octaves(C.byref(cfg), C.byref(r))
    return n, r.value


def scale_space_info(cfg, w, h):
    ow = (C.c_uint32 * MAX_OCTAVES)()
    oh = (C.c_uint32 * MAX_OCTAVES)()
    n = lib().orc_scale_space_info(C.byref(cfg), w, h, ow, oh)
    return [(ow[i], oh[i]) for i in range(n)]


def section_caps(max_nb, n_oct):
    caps = (C.c_uint32 * MAX_OCTAVES)()
    lib().orc_section_caps(max_nb, n_oct, caps)
    return [caps[i] for i in range(n_oct)]


def gaussian_kernels(cfg):
    S = cfg.nb_scales_per_octave
    k = np.zeros((S + 3, MAX_KERNEL), np.float32)
    sizes = np.zeros(S + 3, np.uint32)
    sig = np.zeros(S + 3, np.float32)
    lib().orc_gaussian_kernels(C.byref(cfg), k.ctypes.data_as(C.POINTER(C.c_float)), sizes.ctypes.data_as(C.POINTER(C.c_uint32)),
                               sig.ctypes.data_as(C.POINTER(C.c_float)))
    return k, sizes, sig


def effective_taps(cfg):
    S = cfg.nb_scales_per_octave
    k = np.zeros((S + 3, MAX_KERNEL), np.float32)
    n = np.zeros(S + 3, np.uint32)
    lib().orc_effective_taps(C.byref(cfg), k.ctypes.data_as(C.POINTER(C.c_float)), n.ctypes.data_as(C.POINTER(C.c_uint32)))
    return k, n


class Pyramid:
    def __init__(self, cfg, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        assert img.ndim == 2
        self.cfg = cfg
        self.S = cfg.nb_scales_per_octave
        self._p = lib().orc_pyramid_build(C.byref(cfg), img.ctypes.data, img.shape[1], img.shape[0])
        self.nb_octaves = lib().orc_pyramid_nb_octaves(self._p)

    def resolution(self, o):
        w, h = C.c_uint32(), C.c_uint32()
        lib().orc_pyramid_resolution(self._p, o, C.byref(w), C.byref(h))
        return w.value, h.value

    def _plane(self, ptr, o):
        w, h = self.resolution(o)
        buf = (C.c_float * (w * h)).from_address(ptr)
        return np.frombuffer(buf, dtype=np.float32).reshape(h, w).copy()

    def gauss(self, o, s):
        return self._plane(lib().orc_pyramid_gauss(self._p, o, s), o)

    def dog(self, o, s):
        return self._plane(lib().orc_pyramid_dog(self._p, o, s), o)

    def detect(self, cap=None):
        cap = cap or self.cfg.max_nb_sift_per_buffer
        out = np.zeros(cap, FEATURE_DTYPE)
        counts = (C.c_uint32 * MAX_OCTAVES)()
        n = lib().orc_detect_from_pyramid(C.byref(self.cfg), self._p, out.ctypes.data, cap, counts)
        return out[:n].copy(), [counts[i] for i in range(self.nb_octaves)]

    def extract_keypoints(self, o, cap=100000):
        out = np.zeros(cap, FEATURE_DTYPE)
        n = lib().orc_extract_keypoints(C.byref(self.cfg), self._p, o, out.ctypes.data, cap)
        return out[:min(n, cap)].copy(), n

    def orientations(self, o, kp):
        kp = np.array(kp, dtype=FEATURE_DTYPE).reshape(1)
        ang = np.zeros(36, np.float32)
        hist = np.zeros(36, np.uint32)
        n = lib().orc_orientations(C.byref(self.cfg), self._p, o, kp.ctypes.data, ang.ctypes.data_as(C.POINTER(C.c_float)),
                                   hist.ctypes.data_as(C.POINTER(C.c_uint32)))
        return ang[:n].copy(), hist

    def descriptor(self, o, kp):
        kp = np.array(kp, dtype=FEATURE_DTYPE).reshape(1).copy()
        raw = np.zeros(128, np.uint32)
        lib().orc_descriptor(C.byref(self.cfg), self._p, o, kp.ctypes.data, raw.ctypes.data_as(C.POINTER(C.c_uint32)))
        return kp[0]["descriptor"].copy(), raw

    def close(self):
        if self._p:
            lib().orc_pyramid_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def detect(cfg, img, cap=None):
    p = Pyramid(cfg, img)
    try:
        return p.detect(cap)
    finally:
        p.close()


def match_2nn(a, b):
    """a, b: FEATURE_DTYPE arrays or (N,128) uint8 descriptor matrices. Requires len(b) >= 2 (quirk Q6)."""
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    assert len(b) >= 2
    out = np.zeros(len(a), MATCH_DTYPE)
    if a.dtype == FEATURE_DTYPE:
        assert b.dtype == FEATURE_DTYPE
        lib().orc_match_2nn(a.ctypes.data, len(a), b.ctypes.data, len(b), out.ctypes.data)
    else:
        assert a.dtype == np.uint8 and a.shape[1] == 128 and b.dtype == np.uint8 and b.shape[1] == 128
        lib().orc_match_2nn_desc(a.ctypes.data, len(a), b.ctypes.data, len(b), out.ctypes.data)
    return out


def filter_matches(m12, m21=None, ratio=0.75, cross_check=True):
    """Cross-check + Lowe ratio filter over 2-NN records (reference: test_sift_match.cpp:90-107). Returns (idx_a, idx_b)."""
    m12 = np.ascontiguousarray(m12)
    out_a = np.zeros(len(m12), np.uint32)
    out_b = np.zeros(len(m12), np.uint32)
    if cross_check:
        m21 = np.ascontiguousarray(m21)
        n = lib().orc_filter_matches(m12.ctypes.data, len(m12), m21.ctypes.data, len(m21), ratio, 1, out_a.ctypes.data, out_b.ctypes.data)
    else:
        n = lib().orc_filter_matches(m12.ctypes.data, len(m12), None, 0, ratio, 0, out_a.ctypes.data, out_b.ctypes.data)
    return out_a[:n], out_b[:n]
