"""ABI of the drop-in boundary: struct layouts, enum values and exported symbols (CPU only)."""
import ctypes as C
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")

# numbers measured on the reference headers — the fixture of this test
REF_LAYOUT = {
    "sizeof(vksift_Feature)": 164, "offsetof(vksift_Feature,x)": 0, "offsetof(vksift_Feature,y)": 4,
    "offsetof(vksift_Feature,scale_x)": 8, "offsetof(vksift_Feature,scale_y)": 12, "offsetof(vksift_Feature,scale_idx)": 16,
    "offsetof(vksift_Feature,octave_idx)": 20, "offsetof(vksift_Feature,sigma)": 24, "offsetof(vksift_Feature,orientation)": 28,
    "offsetof(vksift_Feature,intensity)": 32, "offsetof(vksift_Feature,descriptor)": 36,
    "sizeof(vksift_Match_2NN)": 20, "sizeof(vksift_ExternalWindowInfo)": 16, "sizeof(VKSIFT_GPU_NAME)": 256,
    "sizeof(vksift_Config)": 88, "_Alignof(vksift_Config)": 8,
    "offsetof(vksift_Config,input_image_max_size)": 0, "offsetof(vksift_Config,sift_buffer_count)": 4,
    "offsetof(vksift_Config,max_nb_sift_per_buffer)": 8, "offsetof(vksift_Config,use_input_upsampling)": 12,
    "offsetof(vksift_Config,nb_octaves)": 13, "offsetof(vksift_Config,nb_scales_per_octave)": 14,
    "offsetof(vksift_Config,input_image_blur_level)": 16, "offsetof(vksift_Config,seed_scale_sigma)": 20,
    "offsetof(vksift_Config,intensity_threshold)": 24, "offsetof(vksift_Config,edge_threshold)": 28,
    "offsetof(vksift_Config,max_nb_orientation_per_keypoint)": 32, "offsetof(vksift_Config,descriptor_format)": 36,
    "offsetof(vksift_Config,gpu_device_index)": 40, "offsetof(vksift_Config,use_hardware_interpolated_blur)": 44,
    "offsetof(vksift_Config,pyramid_precision_mode)": 48, "offsetof(vksift_Config,on_error_callback_function)": 56,
    "offsetof(vksift_Config,use_gpu_debug_functions)": 64, "offsetof(vksift_Config,gpu_debug_external_window_info)": 72,
    "sizeof(vksift_LogLevel)": 4, "sizeof(vksift_Result)": 4,
    "VKSIFT_SUCCESS": 0, "VKSIFT_INVALID_INPUT_ERROR": 1, "VKSIFT_VULKAN_ERROR": 2,
    "VKSIFT_NO_LOG": 0, "VKSIFT_LOG_ERROR": 1, "VKSIFT_LOG_WARNING": 2, "VKSIFT_LOG_INFO": 3, "VKSIFT_LOG_DEBUG": 4,
    "VKSIFT_DESCRIPTOR_FORMAT_UBC": 0, "VKSIFT_DESCRIPTOR_FORMAT_VLFEAT": 1,
    "VKSIFT_PYRAMID_PRECISION_FLOAT32": 0, "VKSIFT_PYRAMID_PRECISION_FLOAT16": 1,
    "VKSIFT_FEATURE_NB_HIST": 4, "VKSIFT_FEATURE_NB_ORI": 8,
}

API_FUNCS = [
    "vksift_loadVulkan", "vksift_unloadVulkan", "vksift_getAvailableGPUs", "vksift_setLogLevel", "vksift_createInstance",
    "vksift_destroyInstance", "vksift_getDefaultConfig", "vksift_detectFeatures", "vksift_matchFeatures", "vksift_getFeaturesNumber",
    "vksift_downloadFeatures", "vksift_uploadFeatures", "vksift_getMatchesNumber", "vksift_downloadMatches", "vksift_isBufferAvailable",
    "vksift_getScaleSpaceNbOctaves", "vksift_getScaleSpaceOctaveResolution", "vksift_downloadScaleSpaceImage", "vksift_downloadDoGImage",
    "vksift_presentDebugFrame",
]


def test_struct_layout_matches_reference_numbers():
    prog = "#include <stdio.h>\n#include <stddef.h>\n#include \"vulkansift/vulkansift.h\"\nint main(void){\n"
    for k in REF_LAYOUT:
        prog += f'  printf("%s=%ld\\n", "{k}", (long)({k}));\n'
    prog += "  return 0;\n}\n"
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "abi.c")
        open(src, "w").write(prog)
        exe = os.path.join(d, "abi")
        subprocess.run(["gcc", "-std=c11", "-I", INC, src, "-o", exe], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    got = dict(line.split("=") for line in out.strip().splitlines())
    for k, v in REF_LAYOUT.items():
        assert int(got[k]) == v, (k, got[k], v)


def test_headers_compile_as_cxx():
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "abi.cpp")
        open(src, "w").write('#include "vulkansift/vulkansift.h"\n#include "vksift_ext.h"\n#include "vksift_hip.h"\nint main(){return 0;}\n')
        subprocess.run(["g++", "-std=c++17", "-I", INC, "-c", src, "-o", os.path.join(d, "abi.o")], check=True)


def _declared_functions(header):
    txt = open(os.path.join(INC, header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vksift_[A-Za-z0-9_]+)\s*\(", txt)))


def test_reference_api_is_declared_exactly():
    assert _declared_functions("vulkansift/vulkansift.h") == sorted(API_FUNCS)


def test_library_exports_every_declared_symbol(vk):
    L = vk.lib()
    for header in ("vulkansift/vulkansift.h", "vksift_ext.h", "vksift_hip.h"):
        for fn in _declared_functions(header):
            assert hasattr(L, fn), f"{fn} declared in {header} but not exported by libvulkansift.so"


def test_ctypes_mirror_sizes(vk):
    assert C.sizeof(vk.vksift_Config) == 88
    assert vk.FEATURE_DTYPE.itemsize == 164 and vk.MATCH_DTYPE.itemsize == 20
    assert vk.FEATURE_DTYPE.fields["descriptor"][1] == 36
