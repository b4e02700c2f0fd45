"""ctypes binding of libvulkansift.so — the vksift_* C API served by HIP kernels on MI355X.

This module is a thin mirror of include/vulkansift/vulkansift.h (+ vksift_ext.h): same function
names, same argument meaning, same error behaviour (void functions report through the configured
error callback). It exists so that the parity tests and bench.py can drive the C-ABI from Python;
it contains no algorithmic code and never falls back to a CPU implementation — importing it
without the built shared library raises.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
# VKSIFT_LIB selects another build of the same library (A/B runs of compiler options); never a different implementation
LIB_PATH = os.environ.get("VKSIFT_LIB") or os.path.join(_PKG, "lib", "libvulkansift.so")

VKSIFT_SUCCESS, VKSIFT_INVALID_INPUT_ERROR, VKSIFT_VULKAN_ERROR = 0, 1, 2
VKSIFT_NO_LOG, VKSIFT_LOG_ERROR, VKSIFT_LOG_WARNING, VKSIFT_LOG_INFO, VKSIFT_LOG_DEBUG = range(5)
VKSIFT_DESCRIPTOR_FORMAT_UBC, VKSIFT_DESCRIPTOR_FORMAT_VLFEAT = 0, 1
VKSIFT_PYRAMID_PRECISION_FLOAT32, VKSIFT_PYRAMID_PRECISION_FLOAT16 = 0, 1

ERROR_CB = C.CFUNCTYPE(None, C.c_int)


class vksift_ExternalWindowInfo(C.Structure):
    _fields_ = [("context", C.c_void_p), ("window", C.c_void_p)]


class vksift_Config(C.Structure):
    _fields_ = [
        ("input_image_max_size", C.c_uint32),
        ("sift_buffer_count", C.c_uint32),
        ("max_nb_sift_per_buffer", C.c_uint32),
        ("use_input_upsampling", C.c_bool),
        ("nb_octaves", C.c_uint8),
        ("nb_scales_per_octave", C.c_uint8),
        ("input_image_blur_level", C.c_float),
        ("seed_scale_sigma", C.c_float),
        ("intensity_threshold", C.c_float),
        ("edge_threshold", C.c_float),
        ("max_nb_orientation_per_keypoint", C.c_uint32),
        ("descriptor_format", C.c_int),
        ("gpu_device_index", C.c_int32),
        ("use_hardware_interpolated_blur", C.c_bool),
        ("pyramid_precision_mode", C.c_int),
        ("on_error_callback_function", ERROR_CB),
        ("use_gpu_debug_functions", C.c_bool),
        ("gpu_debug_external_window_info", vksift_ExternalWindowInfo),
    ]


assert C.sizeof(vksift_Config) == 88, C.sizeof(vksift_Config)


class vksift_ext_DetectTimings(C.Structure):
    _fields_ = [
        ("upload_ms", C.c_float), ("pyramid_ms", C.c_float), ("extrema_ms", C.c_float), ("orientation_ms", C.c_float),
        ("descriptor_ms", C.c_float), ("total_ms", C.c_float), ("nb_blur_launches", C.c_uint32), ("pyramid_algorithmic_bytes", C.c_uint64),
        ("scan_ms", C.c_float), ("scan_algorithmic_bytes", C.c_uint64),
        ("pyramid_all_ms", C.c_float), ("nb_blur_launches_all", C.c_uint32),
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
FILTERED_MATCH_DTYPE = np.dtype([("idx_a", "<u4"), ("idx_b", "<u4"), ("dist_a_b1", "<f4"), ("dist_a_b2", "<f4")])
assert FEATURE_DTYPE.itemsize == 164 and MATCH_DTYPE.itemsize == 20 and FILTERED_MATCH_DTYPE.itemsize == 16

_lib = None


def lib():
    """Load libvulkansift.so (raises if it has not been built: there is no fallback path)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -m vulkansift_amd.build` (hipcc, gfx950). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    inst, u32, u8 = C.c_void_p, C.c_uint32, C.c_uint8
    L.vksift_loadVulkan.restype = C.c_int
    L.vksift_unloadVulkan.restype = None
    L.vksift_getAvailableGPUs.argtypes = [C.POINTER(u32), C.c_void_p]
    L.vksift_setLogLevel.argtypes = [C.c_int]
    L.vksift_createInstance.argtypes = [C.POINTER(inst), C.POINTER(vksift_Config)]
    L.vksift_createInstance.restype = C.c_int
    L.vksift_destroyInstance.argtypes = [C.POINTER(inst)]
    L.vksift_getDefaultConfig.restype = vksift_Config
    L.vksift_detectFeatures.argtypes = [inst, C.c_void_p, u32, u32, u32]
    L.vksift_matchFeatures.argtypes = [inst, u32, u32]
    L.vksift_getFeaturesNumber.argtypes = [inst, u32]
    L.vksift_getFeaturesNumber.restype = u32
    L.vksift_downloadFeatures.argtypes = [inst, C.c_void_p, u32]
    L.vksift_uploadFeatures.argtypes = [inst, C.c_void_p, u32, u32]
    L.vksift_getMatchesNumber.argtypes = [inst]
    L.vksift_getMatchesNumber.restype = u32
    L.vksift_downloadMatches.argtypes = [inst, C.c_void_p]
    L.vksift_isBufferAvailable.argtypes = [inst, u32]
    L.vksift_isBufferAvailable.restype = C.c_bool
    L.vksift_getScaleSpaceNbOctaves.argtypes = [inst]
    L.vksift_getScaleSpaceNbOctaves.restype = u8
    L.vksift_getScaleSpaceOctaveResolution.argtypes = [inst, u8, C.POINTER(u32), C.POINTER(u32)]
    L.vksift_downloadScaleSpaceImage.argtypes = [inst, u8, u8, C.c_void_p]
    L.vksift_downloadDoGImage.argtypes = [inst, u8, u8, C.c_void_p]
    L.vksift_presentDebugFrame.argtypes = [inst]
    # extensions
    L.vksift_ext_createInstanceBatched.argtypes = [C.POINTER(inst), C.POINTER(vksift_Config), u32]
    L.vksift_ext_createInstanceBatched.restype = C.c_int
    L.vksift_ext_detectFeaturesBatch.argtypes = [inst, C.POINTER(C.c_void_p), u32, u32, u32, u32]
    L.vksift_ext_detectFeaturesBatchDevice.argtypes = [inst, C.c_void_p, u32, u32, u32, u32]
    L.vksift_ext_matchFeaturesBatch.argtypes = [inst, u32, C.POINTER(u32), C.POINTER(u32)]
    L.vksift_ext_getMatchesNumberBatch.argtypes = [inst, u32]
    L.vksift_ext_getMatchesNumberBatch.restype = u32
    L.vksift_ext_downloadMatchesBatch.argtypes = [inst, u32, C.c_void_p]
    L.vksift_ext_matchFeaturesFiltered.argtypes = [inst, u32, C.POINTER(u32), C.POINTER(u32), C.c_float, C.c_bool]
    L.vksift_ext_getFilteredMatchesNumber.argtypes = [inst, u32]
    L.vksift_ext_getFilteredMatchesNumber.restype = u32
    L.vksift_ext_downloadFilteredMatches.argtypes = [inst, u32, C.c_void_p]
    L.vksift_ext_setProfiling.argtypes = [inst, C.c_bool]
    L.vksift_ext_getDetectTimings.argtypes = [inst, C.POINTER(vksift_ext_DetectTimings)]
    L.vksift_ext_getAccumulatedDetectTimings.argtypes = [inst, C.POINTER(vksift_ext_DetectTimings), C.POINTER(u32), C.c_bool]
    L.vksift_ext_getDetectTimingsSized.argtypes = [inst, C.POINTER(vksift_ext_DetectTimings), C.c_size_t]
    L.vksift_ext_pinHostMemory.argtypes = [C.c_void_p, C.c_size_t]
    L.vksift_ext_pinHostMemory.restype = C.c_int
    L.vksift_ext_unpinHostMemory.argtypes = [C.c_void_p]
    L.vksift_ext_unpinHostMemory.restype = C.c_int
    L.vksift_ext_getScaleSpacePlacement.argtypes = [inst, C.POINTER(C.c_float), C.POINTER(u32)]
    L.vksift_ext_getScaleSpacePlacement.restype = u32
    L.vksift_ext_getAccumulatedDetectTimingsSized.argtypes = [inst, C.POINTER(vksift_ext_DetectTimings), C.c_size_t, C.POINTER(u32), C.c_bool]
    L.vksift_ext_getDeferredStats.argtypes = [inst, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.vksift_ext_getDeferredStats.restype = None
    L.vksift_ext_getMatchTime.argtypes = [inst]
    L.vksift_ext_getMatchTime.restype = C.c_float
    L.vksift_ext_exportDescriptorsDevice.argtypes = [inst, u32, C.c_void_p]
    L.vksift_ext_exportDescriptorsDevice.restype = u32
    L.vksift_ext_genSyntheticImage.argtypes = [C.c_uint64, u32, u32, u32, C.c_void_p]
    L.vksift_ext_genSyntheticDescriptors.argtypes = [C.c_uint64, u32, C.c_void_p]
    L.vksift_ext_genSyntheticImageFamily.argtypes = [C.c_uint64, u32, u32, u32, C.c_void_p]
    L.vksift_ext_genSyntheticImageFamily.restype = None
    L.vksift_ext_shardGetUniqueId.argtypes = [C.c_void_p]
    L.vksift_ext_shardGetUniqueId.restype = C.c_int
    L.vksift_ext_shardGroupCreate.argtypes = [C.POINTER(C.c_void_p), C.c_int, u32, u32, C.c_void_p]
    L.vksift_ext_shardGroupCreate.restype = C.c_int
    L.vksift_ext_shardGroupDestroy.argtypes = [C.POINTER(C.c_void_p)]
    L.vksift_ext_matchSharded.argtypes = [C.c_void_p, C.c_void_p, u32, u32, C.c_void_p, u32, u32, C.c_void_p]
    L.vksift_ext_matchSharded.restype = C.c_int
    L.vksift_ext_shardGroupReserve.argtypes = [C.c_void_p, u32, u32]
    L.vksift_ext_shardGroupReserve.restype = C.c_int
    L.vksift_ext_shardGroupSynchronize.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L.vksift_ext_shardGroupSynchronize.restype = C.c_int
    L.vksift_ext_shardGroupCreateWithTransport.argtypes = [C.POINTER(C.c_void_p), C.c_int, u32, u32, C.c_void_p, C.c_void_p]
    L.vksift_ext_shardGroupCreateWithTransport.restype = C.c_int
    L.vksift_ext_shardGroupInfo.argtypes = [C.c_void_p, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
    L.vksift_ext_shardGroupInfo.restype = None
    L.vksift_ext_shardGroupLayout.argtypes = [u32, u32, u32, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
    L.vksift_ext_shardGroupLayout.restype = None
    # kernel-layer C-ABI (include/vksift_hip.h) entry points used directly by bench.py / tests
    for name in ("vksift_hip_memcpy_h2d", "vksift_hip_memcpy_d2h", "vksift_hip_memcpy_d2d"):
        getattr(L, name).argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        getattr(L, name).restype = C.c_int
    L.vksift_hip_stream_sync.argtypes = [C.c_void_p]
    L.vksift_hip_stream_sync.restype = C.c_int
    L.vksift_hip_match_2nn_desc.argtypes = [C.c_void_p, u32, u32, C.c_void_p, u32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.vksift_hip_match_scratch_u32.argtypes = [u32, u32]
    L.vksift_hip_match_scratch_u32.restype = C.c_size_t
    L.vksift_hip_abi_version.argtypes = []
    L.vksift_hip_abi_version.restype = u32
    L.vksift_hip_match_2nn_desc.restype = C.c_int
    L.vksift_hip_gather_descriptors.argtypes = [C.c_void_p, u32, C.c_void_p, C.c_void_p]
    L.vksift_hip_gather_descriptors.restype = C.c_int
    _lib = L
    return L


class VksiftError(RuntimeError):
    def __init__(self, code):
        super().__init__({1: "VKSIFT_INVALID_INPUT_ERROR", 2: "VKSIFT_VULKAN_ERROR"}.get(code, str(code)))
        self.code = code


_pending_error = []


@ERROR_CB
def _raising_callback(code):
    # ctypes cannot unwind a Python exception through C frames; record and re-raise on return.
    _pending_error.append(code)


def _check_pending():
    if _pending_error:
        code = _pending_error.pop()
        _pending_error.clear()
        raise VksiftError(code)


_loaded = False


def load():
    global _loaded
    if not _loaded:
        r = lib().vksift_loadVulkan()
        if r != VKSIFT_SUCCESS:
            raise VksiftError(r)
        _loaded = True


def unload():
    global _loaded
    if _loaded:
        lib().vksift_unloadVulkan()
        _loaded = False


def default_config(**overrides):
    cfg = lib().vksift_getDefaultConfig()
    cfg.on_error_callback_function = _raising_callback
    for k, v in overrides.items():
        if not hasattr(cfg, k):
            raise AttributeError(k)
        setattr(cfg, k, v)
    return cfg


def available_gpus():
    n = C.c_uint32(0)
    lib().vksift_getAvailableGPUs(C.byref(n), None)
    names = (C.c_char * 256 * max(n.value, 1))()
    lib().vksift_getAvailableGPUs(C.byref(n), names)
    return [names[i].value.decode() for i in range(n.value)]


def gen_synthetic_image(seed, width, height, nb_blobs=0):
    out = np.empty((height, width), np.uint8)
    lib().vksift_ext_genSyntheticImage(seed, width, height, nb_blobs, out.ctypes.data)
    return out


SYNTH_BLOBS, SYNTH_EDGES, SYNTH_FRACTAL = 0, 1, 2


def gen_synthetic_image_family(seed, width, height, family):
    """vksift_ext_genSyntheticImageFamily: SYNTH_EDGES (step edges, corners, checker patches) or SYNTH_FRACTAL (1/f noise)"""
    out = np.empty((height, width), np.uint8)
    lib().vksift_ext_genSyntheticImageFamily(seed, width, height, family, out.ctypes.data)
    return out


def gen_synthetic_descriptors(seed, rows):
    out = np.empty((rows, 128), np.uint8)
    lib().vksift_ext_genSyntheticDescriptors(seed, rows, out.ctypes.data)
    return out


class Instance:
    """Pythonic handle on a vksift_Instance; method names follow the C API without the prefix."""

    def __init__(self, config=None, batch_capacity=1):
        load()
        self.cfg = config if config is not None else default_config()
        self._h = C.c_void_p(None)
        if batch_capacity > 1:
            r = lib().vksift_ext_createInstanceBatched(C.byref(self._h), C.byref(self.cfg), batch_capacity)
        else:
            r = lib().vksift_createInstance(C.byref(self._h), C.byref(self.cfg))
        if r != VKSIFT_SUCCESS:
            self._h = C.c_void_p(None)
            raise VksiftError(r)
        self.batch_capacity = batch_capacity

    def close(self):
        if self._h:
            lib().vksift_destroyInstance(C.byref(self._h))
            self._h = C.c_void_p(None)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- pipelines
    def detectFeatures(self, image, gpu_buffer_id):
        image = np.ascontiguousarray(image, dtype=np.uint8)
        assert image.ndim == 2
        lib().vksift_detectFeatures(self._h, image.ctypes.data, image.shape[1], image.shape[0], gpu_buffer_id)
        _check_pending()

    def detectFeaturesRaw(self, ptr, width, height, gpu_buffer_id):
        lib().vksift_detectFeatures(self._h, ptr, width, height, gpu_buffer_id)
        _check_pending()

    def detectFeaturesBatch(self, images, first_gpu_buffer_id):
        imgs = [np.ascontiguousarray(im, dtype=np.uint8) for im in images]
        h, w = imgs[0].shape
        assert all(im.shape == (h, w) for im in imgs)
        ptrs = (C.c_void_p * len(imgs))(*[im.ctypes.data for im in imgs])
        lib().vksift_ext_detectFeaturesBatch(self._h, ptrs, len(imgs), w, h, first_gpu_buffer_id)
        _check_pending()

    @staticmethod
    def imagePointerArray(images):
        """the `const uint8_t *const *images` argument of vksift_ext_detectFeaturesBatch for a list of C-contiguous uint8 arrays
        (which must stay alive): a caller that submits the same frame objects again builds it once"""
        assert all(im.dtype == np.uint8 and im.flags["C_CONTIGUOUS"] for im in images)
        return (C.c_void_p * len(images))(*[im.ctypes.data for im in images])

    def detectFeaturesBatchPtrs(self, ptrs, count, width, height, first_gpu_buffer_id):
        lib().vksift_ext_detectFeaturesBatch(self._h, ptrs, count, width, height, first_gpu_buffer_id)
        _check_pending()

    def detectFeaturesBatchDevice(self, dev_ptr, count, width, height, first_gpu_buffer_id):
        lib().vksift_ext_detectFeaturesBatchDevice(self._h, dev_ptr, count, width, height, first_gpu_buffer_id)
        _check_pending()

    def matchFeatures(self, buf_a, buf_b):
        lib().vksift_matchFeatures(self._h, buf_a, buf_b)
        _check_pending()

    def matchFeaturesBatch(self, bufs_a, bufs_b):
        n = len(bufs_a)
        assert n == len(bufs_b)
        a = (C.c_uint32 * n)(*bufs_a)
        b = (C.c_uint32 * n)(*bufs_b)
        lib().vksift_ext_matchFeaturesBatch(self._h, n, a, b)
        _check_pending()

    def matchFeaturesFiltered(self, bufs_a, bufs_b, ratio=0.75, cross_check=True):
        """2-NN A->B (+ B->A), cross-check and Lowe ratio on the GPU (vksift_ext_matchFeaturesFiltered)."""
        n = len(bufs_a)
        assert n == len(bufs_b)
        a = (C.c_uint32 * n)(*bufs_a)
        b = (C.c_uint32 * n)(*bufs_b)
        lib().vksift_ext_matchFeaturesFiltered(self._h, n, a, b, ratio, cross_check)
        _check_pending()

    def downloadFilteredMatches(self, pair=0):
        n = lib().vksift_ext_getFilteredMatchesNumber(self._h, pair)
        _check_pending()
        out = np.zeros(n, FILTERED_MATCH_DTYPE)
        if n:
            lib().vksift_ext_downloadFilteredMatches(self._h, pair, out.ctypes.data)
            _check_pending()
        return out

    def getMatchesNumberBatch(self, pair):
        n = lib().vksift_ext_getMatchesNumberBatch(self._h, pair)
        _check_pending()
        return n

    def downloadMatchesBatch(self, pair):
        n = self.getMatchesNumberBatch(pair)
        out = np.zeros(n, MATCH_DTYPE)
        if n:
            lib().vksift_ext_downloadMatchesBatch(self._h, pair, out.ctypes.data)
            _check_pending()
        return out

    # -- transfers
    def getFeaturesNumber(self, gpu_buffer_id):
        n = lib().vksift_getFeaturesNumber(self._h, gpu_buffer_id)
        _check_pending()
        return n

    def downloadFeatures(self, gpu_buffer_id):
        n = self.getFeaturesNumber(gpu_buffer_id)
        out = np.zeros(n, FEATURE_DTYPE)
        if n:
            lib().vksift_downloadFeatures(self._h, out.ctypes.data, gpu_buffer_id)
            _check_pending()
        return out

    def uploadFeatures(self, feats, gpu_buffer_id):
        feats = np.ascontiguousarray(feats, dtype=FEATURE_DTYPE)
        lib().vksift_uploadFeatures(self._h, feats.ctypes.data, len(feats), gpu_buffer_id)
        _check_pending()

    def getMatchesNumber(self):
        return lib().vksift_getMatchesNumber(self._h)

    def downloadMatches(self):
        n = self.getMatchesNumber()
        out = np.zeros(n, MATCH_DTYPE)
        lib().vksift_downloadMatches(self._h, out.ctypes.data)
        _check_pending()
        return out

    def isBufferAvailable(self, gpu_buffer_id):
        return bool(lib().vksift_isBufferAvailable(self._h, gpu_buffer_id))

    # -- scale space
    def getScaleSpaceNbOctaves(self):
        return lib().vksift_getScaleSpaceNbOctaves(self._h)

    def getScaleSpaceOctaveResolution(self, octave):
        w, h = C.c_uint32(0), C.c_uint32(0)
        lib().vksift_getScaleSpaceOctaveResolution(self._h, octave, C.byref(w), C.byref(h))
        _check_pending()
        return w.value, h.value

    def downloadScaleSpaceImage(self, octave, scale):
        w, h = self.getScaleSpaceOctaveResolution(octave)
        out = np.zeros((h, w), np.float32)
        lib().vksift_downloadScaleSpaceImage(self._h, octave, scale, out.ctypes.data)
        _check_pending()
        return out

    def downloadDoGImage(self, octave, scale):
        w, h = self.getScaleSpaceOctaveResolution(octave)
        out = np.zeros((h, w), np.float32)
        lib().vksift_downloadDoGImage(self._h, octave, scale, out.ctypes.data)
        _check_pending()
        return out

    def presentDebugFrame(self):
        lib().vksift_presentDebugFrame(self._h)

    # -- extensions
    def setProfiling(self, enabled=True):
        lib().vksift_ext_setProfiling(self._h, enabled)

    def getDetectTimings(self):
        t = vksift_ext_DetectTimings()
        lib().vksift_ext_getDetectTimingsSized(self._h, C.byref(t), C.sizeof(t))
        return {f[0]: getattr(t, f[0]) for f in t._fields_}

    def getAccumulatedDetectTimings(self, reset=False):
        t = vksift_ext_DetectTimings()
        n = C.c_uint32(0)
        lib().vksift_ext_getAccumulatedDetectTimingsSized(self._h, C.byref(t), C.sizeof(t), C.byref(n), reset)
        d = {f[0]: getattr(t, f[0]) for f in t._fields_}
        d["nb_calls"] = n.value
        return d

    def getScaleSpacePlacement(self):
        """candidate memory ranges timed at allocation: {"gbps": [...], "chosen": [...]} (empty lists: plain allocation; one chosen range per scale-space buffer of the instance: one, or two with VKSIFT_PYR_PINGPONG=2)"""
        g = (C.c_float * 8)()
        ch = (C.c_uint32 * 2)()
        n = lib().vksift_ext_getScaleSpacePlacement(self._h, g, ch)
        return {"gbps": [round(float(g[i]), 1) for i in range(n)], "chosen": list(dict.fromkeys([int(ch[0]), int(ch[1])])) if n else []}

    def getMatchTime(self):
        return lib().vksift_ext_getMatchTime(self._h)

    def getDeferredStats(self):
        """(batches launched from staged vksift_detectFeatures calls, images in them) — include/vksift_ext.h"""
        b, n = C.c_uint64(0), C.c_uint64(0)
        lib().vksift_ext_getDeferredStats(self._h, C.byref(b), C.byref(n))
        return int(b.value), int(n.value)

    def exportDescriptorsDevice(self, gpu_buffer_id, dev_ptr):
        n = lib().vksift_ext_exportDescriptorsDevice(self._h, gpu_buffer_id, dev_ptr)
        _check_pending()
        return n
